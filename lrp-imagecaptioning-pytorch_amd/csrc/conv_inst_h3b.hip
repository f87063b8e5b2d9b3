// instantiations of the fp16 split-product convolution (conv_f16x3.h): relevance passes of the 224/112-pixel layers
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h3_112_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 1, 4, true, EPI_REL_MUL>(a, s); }
int launch_h3_112n_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 2, 2, false, EPI_REL_MUL>(a, s); }
int launch_h3_224_rel(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<224, 2, 2, false, EPI_REL_MUL>(a, s); }
}
#ifdef LRPX_STAMP
extern "C" int lrpx_debug_stamps_h3b(unsigned long long* out12, int reset) {
    if (hipMemcpyFromSymbol(out12, HIP_SYMBOL(lrpx::g_stamp_h3), 96) != hipSuccess) return 1;
    if (reset) { unsigned long long z[12] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(lrpx::g_stamp_h3), z, 96) != hipSuccess) return 1; }
    return 0;
}
#endif
