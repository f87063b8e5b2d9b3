// f16+f8 image-gradient convolutions (conv_f16x3.h with F8, GUIDED epilogue): guided backprop / plain gradient chains
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h8_224_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<224, 2, 2, false, EPI_GUIDED, false, true>(a, s); }
int launch_h8_112_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 1, 4, true, EPI_GUIDED, false, true>(a, s); }
int launch_h8_56_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 4, true, EPI_GUIDED, false, true>(a, s); }
}
