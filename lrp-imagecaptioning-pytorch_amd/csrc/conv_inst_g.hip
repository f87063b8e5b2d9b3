// generated instantiation list of conv_mfma_kernel (see conv_launch.h)
#include "conv_launch.h"
namespace lrpx {
int launch_conv_28_16_1_4_9_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<28, 16, 1, 4, 9, EPI_GUIDED>(a, s); }
int launch_conv_28_16_1_4_9_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<28, 16, 1, 4, 9, EPI_PLAIN>(a, s); }
int launch_conv_14_16_1_4_9_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<14, 16, 1, 4, 9, EPI_GUIDED>(a, s); }
}
