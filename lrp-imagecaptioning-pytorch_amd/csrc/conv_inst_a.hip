// generated instantiation list of conv_mfma_kernel (see conv_launch.h)
#include "conv_launch.h"
namespace lrpx {
int launch_conv_224_8_1_4_9_fwd_dual(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<224, 8, 1, 4, 9, EPI_FWD_DUAL>(a, s); }
int launch_conv_224_8_2_2_9_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<224, 8, 2, 2, 9, EPI_REL>(a, s); }
int launch_conv_112_8_1_4_9_fwd_dual(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<112, 8, 1, 4, 9, EPI_FWD_DUAL>(a, s); }
}
