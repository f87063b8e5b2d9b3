// generated instantiation list of conv_mfma_kernel (see conv_launch.h)
#include "conv_launch.h"
namespace lrpx {
int launch_conv_14_32_1_4_1_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<14, 32, 1, 4, 1, EPI_REL>(a, s); }
int launch_conv_14_32_1_4_1_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<14, 32, 1, 4, 1, EPI_PLAIN>(a, s); }
int launch_conv_112_8_1_4_9_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<112, 8, 1, 4, 9, EPI_GUIDED>(a, s); }
}
