// instantiations of the fp32-accurate bf16x6 convolution (conv_bf16x6.h): relevance passes
#include "conv_launch.h"
#include "conv_bf16x6.h"
namespace lrpx {
int launch_x6_56_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<56, 1, 4, true, EPI_REL>(a, s); }
int launch_x6_28_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<28, 1, 4, true, EPI_REL>(a, s); }
int launch_x6_14_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<14, 1, 4, true, EPI_REL>(a, s); }
}
