// instantiations of the fp32-accurate bf16x6 convolution (conv_bf16x6.h): relevance passes
#include "conv_launch.h"
#include "conv_bf16x6.h"
namespace lrpx {
int launch_x6_56_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<56, 1, 4, true, EPI_REL>(a, s); }
int launch_x6_28_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<28, 1, 4, true, EPI_REL>(a, s); }
int launch_x6_14_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_bf16x6<14, 1, 4, true, EPI_REL>(a, s); }
}
#ifdef LRPX_STAMP
extern "C" int lrpx_debug_stamps(unsigned long long* out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(lrpx::g_stamp), 64) != hipSuccess) return 1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(lrpx::g_stamp), z, 64) != hipSuccess) return 1; }
    return 0;
}
#endif
