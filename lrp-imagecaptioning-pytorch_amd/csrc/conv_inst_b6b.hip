// conv mode 1 (exact bf16 splits, six products: conv_f16x3.h, B6) relevance kernels with the fused multiplicand: 56 / 28 / 14-pixel layers
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_b6_56_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 4, true, EPI_REL_MUL, false, false, true>(a, s); }
int launch_b6_28_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 4, true, EPI_REL_MUL, false, false, true>(a, s); }
int launch_b6_14_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 4, true, EPI_REL_MUL, false, false, true>(a, s); }
}
#ifdef LRPX_STAMP
extern "C" int lrpx_debug_stamps_b6b(unsigned long long* out12, int reset) {
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(lrpx::g_stamp_h3), 96) != hipSuccess) return 1;
    if (reset) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(lrpx::g_stamp_h3), z, 96) != hipSuccess) return 1; }
    return 0;
}
#endif
