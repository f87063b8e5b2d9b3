// conv mode 1 (B6) relevance kernels whose operand is unpooled while it is staged (conv_f16x3.h, POOL): conv1_2, conv2_2
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_b6_224_pool(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<224, 4, 2, false, EPI_REL_MUL, true, false, true>(a, s); }
int launch_b6_112_pool(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 1, 4, false, EPI_REL_MUL, true, false, true>(a, s); }
}
#ifdef LRPX_STAMP
extern "C" int lrpx_debug_stamps_b6p(unsigned long long* out12, int reset) {
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(lrpx::g_stamp_h3), 96) != hipSuccess) return 1;
    if (reset) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(lrpx::g_stamp_h3), z, 96) != hipSuccess) return 1; }
    return 0;
}
#endif
