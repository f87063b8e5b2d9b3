// conv mode 1 (B6: exact bf16 splits) image-gradient chains (guided backprop / plain gradient): GUIDED epilogue = the ReLU hook of the layer below
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_b6_112n_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 2, 2, false, EPI_GUIDED, false, false, true>(a, s); }
int launch_b6_56_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 4, true, EPI_GUIDED, false, false, true>(a, s); }
int launch_b6_28_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 4, true, EPI_GUIDED, false, false, true>(a, s); }
int launch_b6_14_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 4, true, EPI_GUIDED, false, false, true>(a, s); }
}
