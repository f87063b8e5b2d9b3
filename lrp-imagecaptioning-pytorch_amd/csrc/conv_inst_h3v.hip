// f16x3 transposed convolutions of the image-gradient chains: GUIDED epilogue (28/14) and PLAIN (in front of a pool)
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h3_28_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 4, true, EPI_GUIDED>(a, s); }
int launch_h3_14_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 4, true, EPI_GUIDED>(a, s); }
int launch_h3_112n_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 2, 2, false, EPI_PLAIN>(a, s); }
}
