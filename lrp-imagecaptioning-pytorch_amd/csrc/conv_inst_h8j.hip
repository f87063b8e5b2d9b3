// f16+f8 image-gradient convolutions under a pool, >= 256 output channels (8-wave workgroups)
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h8_56w_pool_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 8, true, EPI_GUIDED, true, true>(a, s); }
int launch_h8_28w_pool_guided(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 8, true, EPI_GUIDED, true, true>(a, s); }
}
