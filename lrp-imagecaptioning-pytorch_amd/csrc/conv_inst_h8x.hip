// f16+f8 pooled-input relevance convolution with 8-wave workgroups (conv3_3: 256 output channels in one workgroup)
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h8_56w_pool(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 8, true, EPI_REL_MUL, true, true>(a, s); }
int launch_h8_28w_pool(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 8, true, EPI_REL_MUL, true, true>(a, s); }
}
