// conv mode 1 (B6: exact bf16 splits) forward trace: ReLU(conv + b) and the channel-balanced Z+ in one pass (FWD_DUAL), wide layers
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_b6_224_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<224, 1, 4, false, EPI_FWD_DUAL, false, false, true>(a, s); }
int launch_b6_112_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 1, 4, false, EPI_FWD_DUAL, false, false, true>(a, s); }
int launch_b6_56_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 4, true, EPI_FWD_DUAL, false, false, true>(a, s); }
int launch_b6_56_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 4, true, EPI_PLAIN, false, false, true>(a, s); }
}
