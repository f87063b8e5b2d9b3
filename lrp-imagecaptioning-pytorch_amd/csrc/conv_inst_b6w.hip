// conv mode 1 (B6) relevance kernels with 8-wave workgroups (256 output channels per workgroup, one workgroup per CU): 56 / 28 / 14-pixel layers
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_b6_56w_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 8, true, EPI_REL_MUL, false, false, true>(a, s); }
int launch_b6_28w_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 8, true, EPI_REL_MUL, false, false, true>(a, s); }
int launch_b6_14w_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 8, true, EPI_REL_MUL, false, false, true>(a, s); }
}
