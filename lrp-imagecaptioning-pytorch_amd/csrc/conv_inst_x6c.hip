// instantiations of the fp32-accurate bf16x6 convolution (conv_bf16x6.h): forward passes
#include "conv_launch.h"
#include "conv_bf16x6.h"
namespace lrpx {
int launch_x6_28_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<28, 1, 4, true, EPI_FWD_DUAL>(a, s); }
int launch_x6_14_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<14, 1, 4, true, EPI_FWD_DUAL>(a, s); }
int launch_x6_224_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<224, 4, 2, false, EPI_REL>(a, s); }
int launch_x6_112_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<112, 1, 4, false, EPI_FWD_DUAL>(a, s); }
}
