// Launch helpers for the conv_mfma kernel family (instantiated in conv_inst_*.hip).
#pragma once
#include "common.h"
#include "conv_mfma.h"

namespace lrpx {

template <int HW, int KC, int MT, int NWN, int TAPS, int EPI>
int launch_conv_cfg(const ConvArgs& a, hipStream_t stream) {
    using C = ConvCfg<HW, KC, MT, NWN, TAPS>;
    long m_tiles;
    if (TAPS == 9) m_tiles = ceil_div((long)a.n_maps * HW, C::R);
    else m_tiles = ceil_div((long)a.n_maps * a.pix_per_map, C::PIX);
    const int n_blocks = (int)ceil_div(a.n_oc, 32 * NWN);
    const long grid = ceil_div(m_tiles, 8) * 8 * n_blocks;
    if (grid <= 0 || grid > 0x7fffffffL) {
        set_error("conv_mfma: grid %ld out of range", grid);
        return LRPX_EINVAL;
    }
    auto kern = conv_mfma_kernel<HW, KC, MT, NWN, TAPS, EPI>;
    static LdsOnce attr_once;
    LRPX_TRY(reserve_lds_once(attr_once, kern, C::LDS_BYTES, "conv_mfma"));
    const int ks = a.ksplit > 1 ? a.ksplit : 1;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, ks), dim3(C::NT), C::LDS_BYTES, stream, a, (int)m_tiles, n_blocks);
    return check_launch("conv_mfma");
}

// one function per instantiation, defined in conv_inst_*.hip
int launch_conv_224_8_1_4_9_fwd_dual(const ConvArgs& a, hipStream_t s);
int launch_conv_224_8_2_2_9_rel(const ConvArgs& a, hipStream_t s);
int launch_conv_112_8_1_4_9_fwd_dual(const ConvArgs& a, hipStream_t s);
int launch_conv_112_8_1_4_9_rel(const ConvArgs& a, hipStream_t s);
int launch_conv_112_8_2_2_9_rel(const ConvArgs& a, hipStream_t s);
int launch_conv_56_16_1_4_9_fwd_dual(const ConvArgs& a, hipStream_t s);
int launch_conv_56_16_1_4_9_rel(const ConvArgs& a, hipStream_t s);
int launch_conv_28_16_1_4_9_fwd_dual(const ConvArgs& a, hipStream_t s);
int launch_conv_28_16_1_4_9_rel(const ConvArgs& a, hipStream_t s);
int launch_conv_14_16_1_4_9_fwd_dual(const ConvArgs& a, hipStream_t s);
int launch_conv_14_16_1_4_9_rel(const ConvArgs& a, hipStream_t s);
int launch_conv_224_8_2_2_9_guided(const ConvArgs& a, hipStream_t s);
int launch_conv_14_32_1_4_1_rel(const ConvArgs& a, hipStream_t s);
int launch_conv_14_32_1_4_1_plain(const ConvArgs& a, hipStream_t s);
int launch_conv_112_8_1_4_9_guided(const ConvArgs& a, hipStream_t s);
int launch_conv_112_8_2_2_9_plain(const ConvArgs& a, hipStream_t s);
int launch_conv_56_16_1_4_9_guided(const ConvArgs& a, hipStream_t s);
int launch_conv_56_16_1_4_9_plain(const ConvArgs& a, hipStream_t s);
int launch_conv_28_16_1_4_9_guided(const ConvArgs& a, hipStream_t s);
int launch_conv_28_16_1_4_9_plain(const ConvArgs& a, hipStream_t s);
int launch_conv_14_16_1_4_9_guided(const ConvArgs& a, hipStream_t s);
int launch_conv_14_16_1_4_9_plain(const ConvArgs& a, hipStream_t s);

int launch_x6_56_rel(const ConvArgs& a, hipStream_t s);
int launch_x6_28_rel(const ConvArgs& a, hipStream_t s);
int launch_x6_14_rel(const ConvArgs& a, hipStream_t s);
int launch_x6_112_rel(const ConvArgs& a, hipStream_t s);
int launch_x6_224_rel(const ConvArgs& a, hipStream_t s);
int launch_x6_112n_rel(const ConvArgs& a, hipStream_t s);
int launch_x6_112_fwd(const ConvArgs& a, hipStream_t s);
int launch_x6_56_fwd(const ConvArgs& a, hipStream_t s);
int launch_x6_28_fwd(const ConvArgs& a, hipStream_t s);
int launch_x6_14_fwd(const ConvArgs& a, hipStream_t s);

// conv mode 1 on the conv_f16x3.h tiling (B6: exact bf16 splits, six products) with the fused multiplicand (REL_MUL) and pooled-input staging
int launch_b6_224_rel(const ConvArgs& a, hipStream_t s);
int launch_b6_112_rel(const ConvArgs& a, hipStream_t s);
int launch_b6_112n_rel(const ConvArgs& a, hipStream_t s);
int launch_b6_56_rel(const ConvArgs& a, hipStream_t s);
int launch_b6_28_rel(const ConvArgs& a, hipStream_t s);
int launch_b6_14_rel(const ConvArgs& a, hipStream_t s);
int launch_b6_224_pool(const ConvArgs& a, hipStream_t s);
int launch_b6_112_pool(const ConvArgs& a, hipStream_t s);
int launch_b6_56_pool(const ConvArgs& a, hipStream_t s);
int launch_b6_28_pool(const ConvArgs& a, hipStream_t s);
int launch_b6_56w_rel(const ConvArgs& a, hipStream_t s);       // ... 8-wave workgroups (LRPX_B6_WIDE)
int launch_b6_28w_rel(const ConvArgs& a, hipStream_t s);
int launch_b6_14w_rel(const ConvArgs& a, hipStream_t s);
int launch_b6_56w_pool(const ConvArgs& a, hipStream_t s);
int launch_b6_28w_pool(const ConvArgs& a, hipStream_t s);
int launch_b6_112w_pool(const ConvArgs& a, hipStream_t s);
int launch_b6_112n_guided(const ConvArgs& a, hipStream_t s);   // ... image-gradient chains (GUIDED: guided backprop / plain gradient)
int launch_b6_56_guided(const ConvArgs& a, hipStream_t s);
int launch_b6_28_guided(const ConvArgs& a, hipStream_t s);
int launch_b6_14_guided(const ConvArgs& a, hipStream_t s);
int launch_b6_224_pool_guided(const ConvArgs& a, hipStream_t s);
int launch_b6_112_pool_guided(const ConvArgs& a, hipStream_t s);
int launch_b6_56_pool_guided(const ConvArgs& a, hipStream_t s);
int launch_b6_28_pool_guided(const ConvArgs& a, hipStream_t s);
int launch_b6_224_fwd(const ConvArgs& a, hipStream_t s);       // ... forward trace (FWD_DUAL; 14 x 14: K-split PLAIN)
int launch_b6_112_fwd(const ConvArgs& a, hipStream_t s);
int launch_b6_56_fwd(const ConvArgs& a, hipStream_t s);
int launch_b6_28_fwd(const ConvArgs& a, hipStream_t s);
int launch_b6_14_fwd(const ConvArgs& a, hipStream_t s);
int launch_b6_14_plain(const ConvArgs& a, hipStream_t s);
int launch_b6_28_plain(const ConvArgs& a, hipStream_t s);
int launch_b6_56_plain(const ConvArgs& a, hipStream_t s);

int launch_h3_224_rel(const ConvArgs& a, hipStream_t s);
int launch_h3_112_rel(const ConvArgs& a, hipStream_t s);
int launch_h3_112n_rel(const ConvArgs& a, hipStream_t s);
int launch_h3_56_rel(const ConvArgs& a, hipStream_t s);
int launch_h3_28_rel(const ConvArgs& a, hipStream_t s);
int launch_h3_14_rel(const ConvArgs& a, hipStream_t s);
// f16+f8 relevance kernels (conv_f16x3.h, F8 = true): hi.hi on the fp16 matrix cores, the two cross products on fp8
int launch_h8_56_rel(const ConvArgs& a, hipStream_t s);
int launch_h8_28_rel(const ConvArgs& a, hipStream_t s);
int launch_h8_14_rel(const ConvArgs& a, hipStream_t s);
int launch_h8_56w_rel(const ConvArgs& a, hipStream_t s);      // 8-wave workgroups (256 output channels)
int launch_h8_28w_rel(const ConvArgs& a, hipStream_t s);
int launch_h8_14w_rel(const ConvArgs& a, hipStream_t s);
int launch_h8_56w_pool(const ConvArgs& a, hipStream_t s);
int launch_h8_28w_pool(const ConvArgs& a, hipStream_t s);
int launch_h8_112_rel(const ConvArgs& a, hipStream_t s);
int launch_h8_112n_rel(const ConvArgs& a, hipStream_t s);
int launch_h8_224_rel(const ConvArgs& a, hipStream_t s);
int launch_h8_224_pool(const ConvArgs& a, hipStream_t s);
int launch_h8_112_pool(const ConvArgs& a, hipStream_t s);
int launch_h8_56_pool(const ConvArgs& a, hipStream_t s);
int launch_h8_28_pool(const ConvArgs& a, hipStream_t s);
int launch_h8_224_guided(const ConvArgs& a, hipStream_t s);   // image-gradient chains on the f16+f8 kernels
int launch_h8_112_guided(const ConvArgs& a, hipStream_t s);
int launch_h8_56_guided(const ConvArgs& a, hipStream_t s);
int launch_h8_28_guided(const ConvArgs& a, hipStream_t s);
int launch_h8_14_guided(const ConvArgs& a, hipStream_t s);
int launch_h8_112n_plain(const ConvArgs& a, hipStream_t s);
int launch_h8_56_plain(const ConvArgs& a, hipStream_t s);
int launch_h8_28_plain(const ConvArgs& a, hipStream_t s);
int launch_h8_14_plain(const ConvArgs& a, hipStream_t s);
int launch_h8_56w_guided(const ConvArgs& a, hipStream_t s);   // ... >= 256 output channels: 8-wave workgroups
int launch_h8_28w_guided(const ConvArgs& a, hipStream_t s);
int launch_h8_14w_guided(const ConvArgs& a, hipStream_t s);
int launch_h8_28w_plain(const ConvArgs& a, hipStream_t s);
int launch_h8_14w_plain(const ConvArgs& a, hipStream_t s);
int launch_h8_112n_guided(const ConvArgs& a, hipStream_t s);
int launch_h8_224_pool_guided(const ConvArgs& a, hipStream_t s);   // ... under a pool: pooled-input staging
int launch_h8_112_pool_guided(const ConvArgs& a, hipStream_t s);
int launch_h8_56w_pool_guided(const ConvArgs& a, hipStream_t s);
int launch_h8_28w_pool_guided(const ConvArgs& a, hipStream_t s);
int launch_h3_224_fwd(const ConvArgs& a, hipStream_t s);    // forward trace (ReLU(conv+b) and Z+) on the fp16 matrix cores
int launch_h3_112_fwd(const ConvArgs& a, hipStream_t s);
int launch_h3_56_fwd(const ConvArgs& a, hipStream_t s);
int launch_h3_28_fwd(const ConvArgs& a, hipStream_t s);
int launch_h3_14_fwd(const ConvArgs& a, hipStream_t s);
int launch_h3_112w_fwd(const ConvArgs& a, hipStream_t s);   // ... 8-wave workgroups (A/B: LRPX_FWD_WIDE)
int launch_h3_56w_fwd(const ConvArgs& a, hipStream_t s);
int launch_h3_28w_fwd(const ConvArgs& a, hipStream_t s);
int launch_h3_14w_plain(const ConvArgs& a, hipStream_t s);
int launch_h3_224_guided(const ConvArgs& a, hipStream_t s);   // image-gradient chains (guided backprop / plain gradient)
int launch_h3_112_guided(const ConvArgs& a, hipStream_t s);
int launch_h3_56_guided(const ConvArgs& a, hipStream_t s);
int launch_h3_28_guided(const ConvArgs& a, hipStream_t s);
int launch_h3_14_guided(const ConvArgs& a, hipStream_t s);
int launch_h3_112n_plain(const ConvArgs& a, hipStream_t s);
int launch_h3_56_plain(const ConvArgs& a, hipStream_t s);
int launch_h3_28_plain(const ConvArgs& a, hipStream_t s);
int launch_h3_14_plain(const ConvArgs& a, hipStream_t s);
int launch_h3_224_pool(const ConvArgs& a, hipStream_t s);   // operand unpooled while staged (a.pool_am)
int launch_h3_112_pool(const ConvArgs& a, hipStream_t s);
int launch_h3_56_pool(const ConvArgs& a, hipStream_t s);
int launch_h3_28_pool(const ConvArgs& a, hipStream_t s);

// dense relevance GEMMs with many rows on the fp16 matrix cores (dense_f16x3.hip)
int launch_dense_f16x3(const ConvArgs& a, hipStream_t s);
int launch_dense_bf16x6(const ConvArgs& a, hipStream_t s);      // dense_f16x3.hip, B6: exact bf16 splits, REL epilogue
int launch_dense_small_f16x3(const ConvArgs& a, hipStream_t s);     // few rows: whole K per workgroup, row scales found in-kernel
// lock-step s of the AoA decoder relevance fused into that GEMM (dense_f16x3.hip, FUSE): rows are (image, word) pairs, row = image * T + word
struct AoaStepFuse {
    int T = 0, s = 0;
    const int* tmax = nullptr;                                   // [rows] word index t of the row if its caption has that word, else -1
    const float *q1 = nullptr, *dg = nullptr;                    // [B][T][H]: i tanh(g) / z~(c next), z~(g)  (lrpx_decoder.hip, aoa_rel_coef_kernel)
    float* A_next = nullptr;                                     // [rows][H] GEMM input of lock-step s + 1 (the OTHER buffer than `in`)
    float* r_glob = nullptr;                                     // [rows][H], accumulated
    float* wpart = nullptr;                                      // [rows][T][4] partial sums of r_words
};
int launch_dense_small_f16x3_aoa_step(const ConvArgs& a, const AoaStepFuse& fz, hipStream_t s);
int launch_dense_ks_aoa_step(const ConvArgs& a, const AoaStepFuse& fz, hipStream_t s);      // dense_small.hip: the same step on the fp32 MFMA (exact modes)

// few-row dense GEMMs (dense_small.hip)
bool dense_small_fits(const ConvArgs& a);
int launch_dense_small(const ConvArgs& a, hipStream_t s);

// host-side entry used by the C ABI and by the VGG16 / decoder chains
int conv_dispatch(const lrpx_conv_desc* d, hipStream_t stream, int f16_ksplit = 1);

}  // namespace lrpx
