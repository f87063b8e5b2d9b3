// conv mode 1 (B6) image-gradient chains: convs under a pool, 56 / 28-pixel layers (conv_f16x3.h, POOL)
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_b6_56_pool_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 4, true, EPI_GUIDED, true, false, true>(a, s); }
int launch_b6_28_pool_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 4, true, EPI_GUIDED, true, false, true>(a, s); }
}
