// f16x3 transposed convolutions of the image-gradient chains: PLAIN epilogue (in front of a pool), 56/28/14
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h3_56_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 4, true, EPI_PLAIN>(a, s); }
int launch_h3_28_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 4, true, EPI_PLAIN>(a, s); }
int launch_h3_14_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 4, true, EPI_PLAIN>(a, s); }
}
