// generated instantiation list of conv_mfma_kernel (see conv_launch.h)
#include "conv_launch.h"
namespace lrpx {
int launch_conv_112_8_2_2_9_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<112, 8, 2, 2, 9, EPI_PLAIN>(a, s); }
int launch_conv_56_16_1_4_9_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<56, 16, 1, 4, 9, EPI_GUIDED>(a, s); }
int launch_conv_56_16_1_4_9_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_cfg<56, 16, 1, 4, 9, EPI_PLAIN>(a, s); }
}
