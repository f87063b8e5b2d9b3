// f16+f8 image-gradient convolutions (conv_f16x3.h with F8): GUIDED epilogue 28 / 14, PLAIN epilogue (convs above a pool)
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h8_28_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 4, true, EPI_GUIDED, false, true>(a, s); }
int launch_h8_14_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 4, true, EPI_GUIDED, false, true>(a, s); }
int launch_h8_112n_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 2, 2, false, EPI_PLAIN, false, true>(a, s); }
int launch_h8_56_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 4, true, EPI_PLAIN, false, true>(a, s); }
int launch_h8_28_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 4, true, EPI_PLAIN, false, true>(a, s); }
int launch_h8_14_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 4, true, EPI_PLAIN, false, true>(a, s); }
}
