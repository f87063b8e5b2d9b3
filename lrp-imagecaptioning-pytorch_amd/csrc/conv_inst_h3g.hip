// f16x3 forward-trace convolutions (conv_f16x3.h, FWD_DUAL epilogue): 56 / 28 / 14-pixel layers
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h3_56_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 4, true, EPI_FWD_DUAL>(a, s); }
int launch_h3_28_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 4, true, EPI_FWD_DUAL>(a, s); }
int launch_h3_14_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 4, true, EPI_FWD_DUAL>(a, s); }
}
