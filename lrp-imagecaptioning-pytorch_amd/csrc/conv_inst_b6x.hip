// conv mode 1 (B6) pooled-input relevance kernels with 8-wave workgroups: conv3_3, conv4_3
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_b6_56w_pool(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 8, true, EPI_REL_MUL, true, false, true>(a, s); }
int launch_b6_28w_pool(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 8, true, EPI_REL_MUL, true, false, true>(a, s); }
int launch_b6_112w_pool(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 2, 4, true, EPI_REL_MUL, true, false, true>(a, s); }
}
