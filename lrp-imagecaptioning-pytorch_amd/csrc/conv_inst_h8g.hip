// f16+f8 image-gradient convolutions (guided backprop / plain gradient) with >= 256 output channels: 8-wave workgroups
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h8_56w_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 8, true, EPI_GUIDED, false, true>(a, s); }
int launch_h8_28w_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 8, true, EPI_GUIDED, false, true>(a, s); }
int launch_h8_14w_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 8, true, EPI_GUIDED, false, true>(a, s); }
}
