// f16x3 transposed convolutions of the image-gradient chains (guided backprop / plain gradient), GUIDED epilogue
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h3_224_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<224, 2, 2, false, EPI_GUIDED>(a, s); }
int launch_h3_112_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 1, 4, true, EPI_GUIDED>(a, s); }
int launch_h3_56_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 4, true, EPI_GUIDED>(a, s); }
}
