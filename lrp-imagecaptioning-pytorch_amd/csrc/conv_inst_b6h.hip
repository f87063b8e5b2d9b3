// conv mode 1 (B6) image-gradient chains: convs under a pool - the gradient arrives at the pool's output resolution and is routed to the
// windows' arg-max while it is staged (conv_f16x3.h, POOL)
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_b6_224_pool_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<224, 4, 2, false, EPI_GUIDED, true, false, true>(a, s); }
int launch_b6_112_pool_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 1, 4, false, EPI_GUIDED, true, false, true>(a, s); }
}
