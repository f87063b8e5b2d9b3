// f16x3 forward-trace convolutions with 8-wave workgroups (256 output columns per workgroup, ONE workgroup per CU): the A tile
// is staged once per 256 of the 2 * cout columns instead of once per 128 (A/B: LRPX_FWD_WIDE)
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h3_112w_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 1, 8, true, EPI_FWD_DUAL>(a, s); }
int launch_h3_56w_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<56, 1, 8, true, EPI_FWD_DUAL>(a, s); }
int launch_h3_28w_fwd(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<28, 1, 8, true, EPI_FWD_DUAL>(a, s); }
int launch_h3_14w_plain(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<14, 1, 8, true, EPI_PLAIN>(a, s); }
}
