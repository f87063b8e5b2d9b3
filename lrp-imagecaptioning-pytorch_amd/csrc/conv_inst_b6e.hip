// conv mode 1 (B6) forward trace: 28 / 14-pixel layers; the 14 x 14 layers K-split through the PLAIN epilogue (lrpx_vgg16_forward)
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_b6_28_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 4, true, EPI_FWD_DUAL, false, false, true>(a, s); }
int launch_b6_14_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 4, true, EPI_FWD_DUAL, false, false, true>(a, s); }
int launch_b6_14_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 4, true, EPI_PLAIN, false, false, true>(a, s); }
int launch_b6_28_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 4, true, EPI_PLAIN, false, false, true>(a, s); }
}
