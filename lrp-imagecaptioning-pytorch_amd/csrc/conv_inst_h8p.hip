// f16+f8 relevance convolutions (conv_f16x3.h with F8: fp16 hi.hi product + two fp8 cross products): pooled-input 224 / 112-pixel layers
#include "conv_launch.h"
#include "conv_f16x3.h"
namespace lrpx {
int launch_h8_224_pool(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<224, 2, 2, false, EPI_REL_MUL, true, true>(a, s); }
int launch_h8_112_pool(const ConvArgs& a, hipStream_t s) { return launch_conv_f16x3<112, 1, 4, true, EPI_REL_MUL, true, true>(a, s); }
}
