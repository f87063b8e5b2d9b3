// instantiations of the fp32-accurate bf16x6 convolution (conv_bf16x6.h): 112-pixel relevance passes, forward passes
#include "conv_launch.h"
#include "conv_bf16x6.h"
namespace lrpx {
int launch_x6_112_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<112, 1, 4, false, EPI_REL>(a, s); }
int launch_x6_112n_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<112, 2, 2, false, EPI_REL>(a, s); }
int launch_x6_56_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<56, 1, 4, true, EPI_FWD_DUAL>(a, s); }
}
