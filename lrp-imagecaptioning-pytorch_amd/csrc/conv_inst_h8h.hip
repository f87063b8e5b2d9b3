// f16+f8 plain transposed convolutions above a pool (image-gradient chains) with >= 256 output channels: 8-wave workgroups
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h8_28w_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 8, true, EPI_PLAIN, false, true>(a, s); }
int launch_h8_14w_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 8, true, EPI_PLAIN, false, true>(a, s); }
}
