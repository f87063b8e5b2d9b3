// f16+f8 pooled-input relevance convolution with 8-wave workgroups (conv3_3: 256 output channels in one workgroup)
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h8_56w_pool(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 8, true, EPI_REL_MUL, true, true>(a, s); }
int launch_h8_28w_pool(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 8, true, EPI_REL_MUL, true, true>(a, s); }
}
#ifdef LRPX_STAMP
extern "C" int lrpx_debug_stamps_h8x(unsigned long long* out12, int reset) {
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(lrpx::g_stamp_h3), 96) != hipSuccess) return 1;
    if (reset) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(lrpx::g_stamp_h3), z, 96) != hipSuccess) return 1; }
    return 0;
}
#endif
