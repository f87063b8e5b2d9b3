// f16x3 forward-trace convolutions (conv_f16x3.h, FWD_DUAL epilogue): 224 / 112-pixel layers
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h3_224_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<224, 1, 4, false, EPI_FWD_DUAL>(a, s); }
int launch_h3_112_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 1, 4, true, EPI_FWD_DUAL>(a, s); }
}
