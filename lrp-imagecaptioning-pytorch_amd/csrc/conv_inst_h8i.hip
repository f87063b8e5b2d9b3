// f16+f8 image-gradient convolutions UNDER a pool: the gradient arrives at the pool's output resolution and is routed to the
// window's arg-max while it is staged (as the relevance chain does), GUIDED epilogue (ReLU hook of the conv below)
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h8_224_pool_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<224, 2, 2, false, EPI_GUIDED, true, true>(a, s); }
int launch_h8_112n_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 2, 2, false, EPI_GUIDED, false, true>(a, s); }
int launch_h8_112_pool_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 1, 4, true, EPI_GUIDED, true, true>(a, s); }
}
