// Implicit-GEMM convolution / dense-GEMM engine on the gfx950 fp32 MFMA (v_mfma_f32_32x32x2_f32).
//
// One kernel family serves every contraction on the LRP hot path:
//   * VGG16 forward  (plain conv + the Z+ = conv(X, W+) needed by the alpha1beta0 rule, one pass)
//   * VGG16 relevance (transposed conv of S = R/Z with W+, fused  "(.) * X"  and  "/ Z_below")
//   * the 1x1 / dense epsilon-rule contractions of the decoders  x * (W^T (r / z~))
//
// Data layout in HBM: activations / relevance are pixel-major NHWC fp32, i.e. a [rows][C] matrix
// whose row index is (map, y, x).  Weights are pre-packed fragment-major (pack_weights.hip) so that
// one wave-instruction reads the 1 KiB B fragment of one 32x8 k-step fully coalesced.
//
// Tiling: a workgroup owns PIX = 224*MT consecutive pixels (= R whole image rows of width W) times
// 32*NWN output channels; each wave owns 224 pixels (7 MFMA row-tiles) x 32 channels, i.e. 7
// accumulator tiles of 32x32 (112 VGPRs).  The A operand (input pixels incl. the 3x3 halo rows) is
// staged through LDS per K-chunk of KC input channels and re-used by the 9 taps; B fragments stream
// from L2 straight into registers (every wave of the chip with the same channel block reads the
// same bytes, so they stay cache resident).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "build_guard.h"

#ifndef LRPXH_NT_STORE
#define LRPXH_NT_STORE 3      // bit 0: float4 (wide) REL_MUL epilogue, bit 1: dword epilogue: streaming stores
#endif

#ifndef LRPX_EPI_EXP
#define LRPX_EPI_EXP 0      // timing experiments on the epilogues (wrong results): 1 = no multiplicand loads, 2 = no stores
#endif
namespace lrpx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Epilogue : int {
    EPI_FWD_DUAL = 0,   // oc <  split: out0 = relu(acc + bias)   oc >= split: out1 = acc        (Z+)
    EPI_REL = 1,        // r = X * (acc + U);  out0 = r;  out1 = r / stab(Zdiv)
    EPI_FIRST = 2,      // first VGG layer: r = X+ * acc[c] + X- * acc[c+3]  -> NCHW out0 (X stored split x+|x-)
    EPI_PLAIN = 3,      // out0 = acc (+ bias) (optionally relu)
    EPI_GUIDED = 4,     // guided backprop: out0 = max(acc,0) * [Y > 0]   (relu == 2: plain gradient, out0 = acc * [Y > 0])
    EPI_REL_MUL = 5,    // r = X * acc -> out0 and/or out1 (max|r| per map -> out1_amax).  The relevance rule with its
                        // following division folded into the multiplicand: X = x for R, X = x / safe(Z_below) for S_next
};

enum Stab : int { STAB_NONE = 0, STAB_SAFE = 1, STAB_EPS = 2, STAB_SAFE0 = 3 /* internal: r / z, 0 where z == 0 (lrpx_core.hip, div_safe0) */ };

struct ConvArgs {
    const float* in;        // [n_maps*P][cin]  A operand (already S = R/Z for relevance passes)
    long in_chunk_stride;   // 0: pixel-major NHWC.  >0: channel-chunked [cin/KC][n_maps*P][KC], stride between chunks
                            // in floats (every staged chunk row is then one contiguous stream: no partial lines)
    const float* wp;        // packed weights  [n_ocb][nchunk][taps][KC/8][64][4]
    int n_maps;             // maps (images) in `in`
    int cin;                // input channels (multiple of KC)
    int n_oc;               // output channels incl. padding (multiple of 32)
    int pix_per_map;        // H*W (conv) or rows per map (dense)
    int epi;
    int stab;               // how out1 divides by Zdiv
    int oc_split;           // EPI_FWD_DUAL: number of plain channels; else: real output channels
    int relu;               // EPI_PLAIN: apply relu
    int ksplit;             // >1 (dense GEMMs with few rows): blockIdx.y splits the K-chunks, partial results are
                            // atomically added into pre-zeroed outputs (REL out0 / PLAIN out0 only)
    const float* bias;      // [oc]                      (FWD_DUAL / PLAIN, may be null)
    const float* X;         // [n_img*P][oc_split]       multiplicand (REL / FIRST / GUIDED mask)
    const float* U;         // [n_maps][oc_split]        per-map addend inside the bracket (REL, may be null)
    const float* Zdiv;      // [n_img*P][oc_split]       denominator for out1 (REL, may be null)
    const int* map2img;     // [n_maps] image index of every map for X/Zdiv (null: identity)
    float* out0;
    float* out1;
    const unsigned* in_amax;   // f16x3 kernels: [n_maps] float bits of max|in| per map (operand scale), else unused
    unsigned* out1_amax;       // f16x3 kernels, REL: [n_maps] max|out1| per map is atomically max-ed into it (may be null)
    int out_chunk;             // REL_MUL: > 0 = the output is written channel-chunked [C/out_chunk][pixels][out_chunk]
    unsigned* out0_amax;       // f16x3 kernels, FWD_DUAL: [n_maps] max of out0 (activations) per map (may be null)
    const unsigned char* pool_am;  // f16x3 POOL kernels: [n_img][H/2*W/2][cin] window position of each 2x2 maximum; `in`
                                   // is then the low-resolution tensor [n_maps][H/2*W/2][cin]
    int tile_group;                // f16x3 map-aligned kernels: > 1 = maps come in groups of this many per image (hint)
};

__device__ __forceinline__ float stab_safe(float z) { return z + 1e-7f * (z == 0.f ? 1.f : 0.f); }
__device__ __forceinline__ float stab_eps(float z) {
    float s = (z > 0.f) ? 1.f : ((z < 0.f) ? -1.f : 0.f);
    float zt = 0.01f * s + z;
    return zt == 0.f ? 0.01f : zt;
}

template <int HW, int KC, int MT, int NWN, int TAPS>
struct ConvCfg {
    static constexpr int W = HW, H = HW;
    static constexpr int PIX = 224 * MT;                 // pixels per workgroup tile
    static constexpr int R = (TAPS == 9) ? PIX / W : 0;  // whole image rows per tile
    static constexpr int WP = W + 2;                     // LDS row width incl. zero columns
    // rows of the tile + a halo row above/below + one zero row per map boundary that can fall inside the tile
    // (none when tiles are aligned to maps, H % R == 0)
    static constexpr int NSLOT = (TAPS == 9) ? R + 2 + ((H % (R ? R : 1) == 0) ? 0 : (R - 1 + H - 1) / H) : 0;
    // floats per LDS pixel: padded by 4 (conflict-free ds_read_b128, tap shifts are immediate offsets).  SWZ (the
    // 2-row tile of the 224^2 relevance layer): un-padded 32-byte pixels, the two 16-byte halves XOR-swizzled with bit 3
    // of the pixel index — equally conflict-free and 1/3 less LDS (two workgroups per CU), at ~5 VALU per (tile, tap)
    static constexpr bool SWZ = (KC == 8) && (HW == 224) && (MT >= 2);   // only where the padded image would not fit twice
    static constexpr int STRIDE = SWZ ? 8 : KC + 4;
    static constexpr int LDS_PIX = (TAPS == 9) ? NSLOT * WP : PIX;
    static constexpr int LDS_BYTES = 2 * LDS_PIX * STRIDE * 4;   // double-buffered A tile
    static constexpr int NT = 64 * MT * NWN;
    static constexpr int KSTEPS = KC / 8;
    static_assert(TAPS == 1 || PIX % W == 0, "tile must be whole rows");
};

struct EpiCtx {
    int oc, lane, q0, g0;
    long pix0, total_pix;
    long xi_base;   // aligned tiles: index into X / Zdiv of (pixel q0, channel oc); later pixels are + dq * ncol
};

// r / z with one v_rcp_f32 + multiply (~1.5 ulp) instead of the ~10-instruction IEEE sequence; falls back to the exact
// division where 1/z would overflow (z in the denormal range)
__device__ __forceinline__ float fast_div(float r, float z) {
    return fabsf(z) > 1e-30f ? r * __builtin_amdgcn_rcpf(z) : r / z;
}

// Epilogue, per 32x32 accumulator tile j: accj[e] is pixel q = q0 + 32*j + (e&3) + 8*(e>>2), channel cx.oc.
// Split in two so the kernel can software-pipeline it: `epi_gather` does the index math and ALL global loads of the
// 16 elements of a tile (issued one tile ahead), `epi_finish` the arithmetic and the stores.  Pointers are
// __restrict__: without that every load waits for the previous element's store (possible aliasing) and the
// epilogue degenerates into dependent L2 round trips.
struct EpiRegs {
    float xv[16], zv[16], x3[16];      // x3 / nn / pp are only live in the FIRST epilogue (dead otherwise)
    unsigned nn[16], pp[16];
};

// ALIGNED (conv tiles that never straddle two maps, H % R == 0): every pixel of the workgroup tile belongs to one map,
// so X / Zdiv / out indices are linear in the tile pixel index — one multiply-add per element instead of the
// divide/modulo chain (the epilogue's VALU count rivals the MFMA count on the K = 576 layer otherwise).
template <int EPI, int HW, int TAPS, bool ALIGNED>
__device__ __forceinline__ void epi_gather(const ConvArgs& a, const EpiCtx& cx, const int j, EpiRegs& r) {
    const float* __restrict__ X = a.X;
    const float* __restrict__ Zd = a.Zdiv;
    const int* __restrict__ m2i = a.map2img;
    const int oc = cx.oc;
    const int ncol = a.oc_split;
    const unsigned P = (unsigned)a.pix_per_map;
    constexpr bool NEEDS_X = (EPI == EPI_REL || EPI == EPI_GUIDED || EPI == EPI_FIRST || EPI == EPI_REL_MUL);
    if constexpr (ALIGNED && EPI != EPI_FIRST) {
        const long xb = cx.xi_base + (long)(32 * j) * ncol;     // one 64-bit base per tile, 32-bit offsets per element
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            r.xv[e] = 0.f; r.zv[e] = 1.f;
            if (NEEDS_X && oc < ncol) {
                const int xo = ((e & 3) + 8 * (e >> 2)) * ncol;
#if LRPX_EPI_EXP & 1
                r.xv[e] = 1.f;
#else
                r.xv[e] = (X + xb)[xo];
#endif
                if (EPI == EPI_REL && Zd && a.out1) r.zv[e] = (Zd + xb)[xo];
            }
        }
        return;
    }
    if constexpr (TAPS == 9 && EPI != EPI_FIRST) {
        // conv tiles that may straddle a map boundary (28x28, 14x14): a 32-pixel accumulator tile is shorter than a
        // map (P >= 196), so it holds at most ONE boundary: one divide/modulo per tile, then linear indices on
        // either side of it
        const unsigned q0t = (unsigned)(cx.q0 + 32 * j);
        const unsigned rr = q0t / (unsigned)HW, c0 = q0t - rr * HW;
        const unsigned g = (unsigned)cx.g0 + rr;
        const unsigned n0 = g / (unsigned)HW;
        const int p0 = (int)((g - n0 * HW) * HW + c0);
        const int nmax = a.n_maps - 1;
        const long img0 = m2i ? m2i[min((int)n0, nmax)] : n0;
        const long img1 = m2i ? m2i[min((int)n0 + 1, nmax)] : n0 + 1;
        const long b0 = (img0 * P + p0) * ncol + oc, b1 = (img1 * P + p0 - (long)P) * ncol + oc;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int dq = (e & 3) + 8 * (e >> 2);
            r.xv[e] = 0.f; r.zv[e] = 1.f;
            if (NEEDS_X && oc < ncol && cx.pix0 + q0t + dq < cx.total_pix) {
                const long xi = (p0 + dq < (int)P ? b0 : b1) + dq * ncol;
                r.xv[e] = X[xi];
                if (EPI == EPI_REL && Zd && a.out1) r.zv[e] = Zd[xi];
            }
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int q = cx.q0 + 32 * j + (e & 3) + 8 * (e >> 2);
        const long gp = cx.pix0 + q;
        unsigned n, p;   // map index and pixel-in-map (32-bit; constant divisors for the conv case)
        if (TAPS == 9) {
            const unsigned rr = (unsigned)q / (unsigned)HW, c = (unsigned)q - rr * HW;
            const unsigned g = (unsigned)cx.g0 + rr;
            n = g / (unsigned)HW;
            p = (g - n * HW) * HW + c;
        } else {
            n = (unsigned)gp / P;
            p = (unsigned)gp - n * P;
        }
        r.nn[e] = n; r.pp[e] = p;
        r.xv[e] = 0.f; r.zv[e] = 1.f; r.x3[e] = 0.f;
        if (NEEDS_X && gp < cx.total_pix && oc < ncol) {
            const long img = m2i ? m2i[n] : n;
            const long xi = (img * P + p) * ncol + oc;
            r.xv[e] = X[xi];
            if (EPI == EPI_REL && Zd && a.out1) r.zv[e] = Zd[xi];
            if (EPI == EPI_FIRST && oc < 3) r.x3[e] = X[xi + 3];
        }
    }
}

// mx (optional, REL with out1): running max|out1| of this tile, m0 for pixels of the tile's first map, m1 for pixels
// past a map boundary inside the tile (only tiles of the 28x28 / 14x14 layers have one)
struct EpiMax { float m0, m1; };
template <int EPI, int HW, int TAPS, bool ALIGNED>
__device__ __forceinline__ void epi_finish(const ConvArgs& a, const EpiCtx& cx, const int j, const f32x16 accj,
                                           const EpiRegs& r, EpiMax* mx = nullptr) {
    float* __restrict__ o0 = a.out0;
    float* __restrict__ o1 = a.out1;
    const float* __restrict__ Uu = a.U;     // per-(map, channel) addend: only the small decoder GEMMs use it
    const int oc = cx.oc;
    const int ncol = a.oc_split;   // real channels of X / Zdiv / out (REL...), plain channels (FWD_DUAL)
    const unsigned P = (unsigned)a.pix_per_map;
    float bias = 0.f;
    if (EPI == EPI_FWD_DUAL || EPI == EPI_PLAIN) bias = (a.bias && oc < ncol) ? a.bias[oc] : 0.f;
    if constexpr (EPI == EPI_REL_MUL || EPI == EPI_GUIDED) {
        // one output (out1 if given, else out0), one base pointer per tile, compile-time pixel offsets: ~3 VALU per
        // element instead of a 64-bit multiply-add and two uniform branches.  GUIDED (the ReLU hook of the layer below,
        // out = max(g,0) * [y > 0]; a.relu == 2: the plain autograd mask) moves the same data: one multiplicand, one output
        if (oc >= ncol) return;
        // pixel stride / channel base of the output: NHWC, or channel-chunked [C/ch][pixels][ch] (whole 64-byte runs per
        // pixel for a consumer that walks the channels chunk by chunk: the first-layer kernel)
        const int ch = a.out_chunk;
        const int ostr = ch > 0 ? ch : ncol;
        const long obase = ch > 0 ? (long)(oc / ch) * cx.total_pix * ch + (oc % ch) : (long)oc;
#if LRPX_EPI_EXP & 4        // (same store instructions into a 1 MB window: no HBM write traffic)
        float* __restrict__ ob = (o1 ? o1 : o0) + ((((cx.pix0 + cx.q0 + 32 * j) * (long)ostr + obase) & 0x3ffffL) & ~31L) + (oc & 31);
#else
        float* __restrict__ ob = (o1 ? o1 : o0) + (cx.pix0 + cx.q0 + 32 * j) * (long)ostr + obase;
#endif
        int p0t = 0;
        if (mx && !ALIGNED && TAPS == 9) {
            const unsigned q0t = (unsigned)(cx.q0 - 4 * (cx.lane >> 5) + 32 * j);
            const unsigned rr = q0t / (unsigned)HW, c0 = q0t - rr * HW;
            const unsigned g = (unsigned)cx.g0 + rr;
            p0t = (int)((g - (g / (unsigned)HW) * HW) * HW + c0) + 4 * (cx.lane >> 5);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int dq = (e & 3) + 8 * (e >> 2);
            if (!ALIGNED && cx.pix0 + cx.q0 + 32 * j + dq >= cx.total_pix) continue;
            const float rel = EPI == EPI_GUIDED ? ((r.xv[e] > 0.f && (a.relu == 2 || accj[e] > 0.f)) ? accj[e] : 0.f)
                                                : r.xv[e] * accj[e];
            // streaming store: the S tensors (0.5 - 4 GB) are read back a whole kernel later, keeping them out of the way
            // of the weights and multiplicands in L2 is worth 2 % on the wide layers (chain 21.23 -> 21.06 ms)
#if LRPX_EPI_EXP & 2
            if (rel == 1.2345e-30f) ob[dq * ostr] = rel;
#elif LRPXH_NT_STORE & 2
            __builtin_nontemporal_store(rel, &ob[dq * ostr]);
#else
            ob[dq * ostr] = rel;
#endif
            if (mx) {
                const bool past = !ALIGNED && TAPS == 9 && p0t + dq >= (int)P;
                mx->m0 = fmaxf(mx->m0, past ? 0.f : fabsf(rel));
                mx->m1 = fmaxf(mx->m1, past ? fabsf(rel) : 0.f);
            }
        }
        return;
    }
    // pixel-in-map of this lane's first pixel, counted from the map of the TILE's first pixel (lanes 32-63 start 4
    // pixels later): only needed to attribute mx across a map boundary
    int p0_tile = 0;
    if (mx && !ALIGNED && TAPS == 9) {
        const unsigned q0t = (unsigned)(cx.q0 - 4 * (cx.lane >> 5) + 32 * j);
        const unsigned rr = q0t / (unsigned)HW, c0 = q0t - rr * HW;
        const unsigned g = (unsigned)cx.g0 + rr;
        p0_tile = (int)((g - (g / (unsigned)HW) * HW) * HW + c0) + 4 * (cx.lane >> 5);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int q = cx.q0 + 32 * j + (e & 3) + 8 * (e >> 2);
        const long gp = cx.pix0 + q;
        const bool ok = ALIGNED || gp < cx.total_pix;      // aligned tiles are never partial
        float v = accj[e];
        if (EPI == EPI_FIRST) {
            // channels 0..2 carry convT(S, W+), 3..5 convT(S, W-); combine across lanes of the same half
            const int li = cx.lane & 31;
            const float other = __shfl(v, (cx.lane & 32) + ((li + 3) & 31), 64);
            if (ok && oc < 3) o0[((long)r.nn[e] * 3 + oc) * P + r.pp[e]] = r.xv[e] * v + r.x3[e] * other;   // NCHW
            continue;
        }
        if (!ok) continue;
        if (EPI == EPI_FWD_DUAL) {
            if (oc < ncol) {
                v += bias;
                v = v > 0.f ? v : 0.f;
                o0[gp * ncol + oc] = v;
                if (mx) {     // activations are the next layer's operand: per-image maximum for its fp16 scale
                    const bool past = !ALIGNED && TAPS == 9 && p0_tile + (e & 3) + 8 * (e >> 2) >= (int)P;
                    mx->m0 = fmaxf(mx->m0, past ? 0.f : v);
                    mx->m1 = fmaxf(mx->m1, past ? v : 0.f);
                }
            } else if (oc < 2 * ncol) {
                o1[gp * ncol + (oc - ncol)] = v;
            }
        } else if (EPI == EPI_REL) {
            if (oc < ncol) {
                if (Uu) v += Uu[(TAPS == 9 ? (long)(((unsigned)cx.g0 + (unsigned)q / (unsigned)HW) / (unsigned)HW)
                                           : (long)((unsigned)gp / P)) * ncol + oc];
                if (Uu && a.ksplit > 1 && blockIdx.y != 0) v = accj[e];    // the addend joins exactly one K-split
                const float rel = r.xv[e] * v;
                if (o0) {
                    if (a.ksplit > 1) atomicAdd(&o0[gp * ncol + oc], rel);
                    else o0[gp * ncol + oc] = rel;
                }
                if (o1) {
                    float z = r.zv[e];
                    z = (a.stab == STAB_SAFE) ? stab_safe(z) : ((a.stab == STAB_EPS) ? stab_eps(z) : z);
                    const float sv = fast_div(rel, z);
                    o1[gp * ncol + oc] = sv;
                    if (mx) {
                        const bool past = !ALIGNED && TAPS == 9 && p0_tile + (e & 3) + 8 * (e >> 2) >= (int)P;
                        mx->m0 = fmaxf(mx->m0, past ? 0.f : fabsf(sv));
                        mx->m1 = fmaxf(mx->m1, past ? fabsf(sv) : 0.f);
                    }
                }
            }
        } else if (EPI == EPI_PLAIN) {
            if (oc < ncol) {
                if (a.ksplit > 1) {
                    atomicAdd(&o0[gp * ncol + oc], blockIdx.y == 0 ? v + bias : v);   // (no relu with split-K)
                } else {
                    v += bias;
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    o0[gp * ncol + oc] = v;
                }
            }
        } else {   // EPI_GUIDED: ReLU hook of the layer below, out = max(g,0) * [y > 0]
            static_assert(EPI == EPI_FWD_DUAL || EPI == EPI_REL || EPI == EPI_FIRST || EPI == EPI_PLAIN || EPI == EPI_GUIDED || EPI == EPI_REL_MUL, "unknown epilogue");
            // a.relu == 2: plain autograd ReLU backward (mask only), else the guided rule (mask and clamp)
            if (oc < ncol) {
                const float g = (r.xv[e] > 0.f && (a.relu == 2 || v > 0.f)) ? v : 0.f;
                o0[gp * ncol + oc] = g;
                if (mx) {     // the gradient is the next layer's operand: per-map maximum for its fp16 scale
                    const bool past = !ALIGNED && TAPS == 9 && p0_tile + (e & 3) + 8 * (e >> 2) >= (int)P;
                    mx->m0 = fmaxf(mx->m0, past ? 0.f : fabsf(g));
                    mx->m1 = fmaxf(mx->m1, past ? fabsf(g) : 0.f);
                }
            }
        }
    }
}

template <int HW, int KC, int MT, int NWN, int TAPS, int EPI>
__global__ __launch_bounds__(64 * MT * NWN, 2) void conv_mfma_kernel(ConvArgs a, int m_tiles, int n_blocks) {
    using C = ConvCfg<HW, KC, MT, NWN, TAPS>;
    constexpr int W = C::W, H = C::H, WP = C::WP, STRIDE = C::STRIDE, NT = C::NT;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;

    // XCD-aware block mapping: blocks b and b+8 share an XCD (round-robin dispatch), so the n-blocks of one
    // pixel tile get consecutive slots on ONE XCD and re-use the tile's A rows from that XCD's L2 (speed only).
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int mtile = (idx / n_blocks) * 8 + xcd;   // neighbouring pixel tiles run at the same time on the 8 XCDs
    const int nblk = idx % n_blocks;                // (measured: giving each XCD a contiguous tile range is 25 % slower on conv1_2)
    if (mtile >= m_tiles) return;   // whole workgroup exits together

    const int ocb = nblk * NWN + wn;                         // 32-channel block of this wave
    const bool wave_active = ocb * 32 < a.n_oc;
    const int nchunk = a.cin / KC;
    const int ks_n = a.ksplit > 1 ? a.ksplit : 1;
    const int ks_i = ks_n > 1 ? (int)blockIdx.y : 0;
    const int chunk_lo = (int)((long)nchunk * ks_i / ks_n), chunk_hi = (int)((long)nchunk * (ks_i + 1) / ks_n);
    const long total_pix = (long)a.n_maps * a.pix_per_map;

    // ---- per-lane A addresses (floats into lds) ----
    const int li = lane & 31, lh = lane >> 5;
    int abase[7];
    long g0 = 0, v0 = 0;
    if constexpr (TAPS == 9) {
        g0 = (long)mtile * C::R;              // first global image row (map*H + y) of the tile
        v0 = g0 + g0 / H;                     // its index in the virtual sequence with one zero row per map
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            int q = wm * 224 + 32 * j + li;
            int r = q / W, c = q % W;
            long g = g0 + r;
            int slot = (int)(g + g / H - v0) + 1;           // LDS row holding this pixel's own image row
            abase[j] = C::SWZ ? ((slot - 1) * WP + c)                       // pixel index of the window corner
                              : ((slot - 1) * WP + c) * STRIDE + lh * 4;    // float offset of the window corner
        }
    } else {
#pragma unroll
        for (int j = 0; j < 7; ++j) abase[j] = (wm * 224 + 32 * j + li) * STRIDE + lh * 4;
    }

    // ---- staging descriptors: every thread moves the same U 16-byte items of every K-chunk ----
    constexpr int SEG = KC / 4;
    constexpr int ROWPIX = (TAPS == 9) ? W : C::PIX;
    constexpr int NROW = (TAPS == 9) ? C::NSLOT : 1;
    constexpr int NITEM = NROW * ROWPIX * SEG;
    constexpr int U = (NITEM + NT - 1) / NT;
    constexpr int BUF = C::LDS_PIX * STRIDE;          // floats per LDS buffer (two buffers)
    constexpr int STEPS = TAPS * C::KSTEPS;           // k-steps (of 8 channels) per chunk
    constexpr int NB = 6;                             // B-fragment register queue: prefetch depth NB-1 k-steps
    int sdst[U], sgp[U];   // LDS float offset (-1: never written, stays zero) / global pixel (-1: store zeros)
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int it = tid + u * NT;
        sdst[u] = -1; sgp[u] = -1;
        if (it < NITEM) {
            const int s = it / (ROWPIX * SEG);
            const int rem = it - s * (ROWPIX * SEG);
            const int px = rem / SEG, seg = rem - px * SEG;
            if constexpr (TAPS == 9) {
                const long v_ = v0 - 1 + s;
                const long n = v_ / (H + 1);
                const int y = (int)(v_ - n * (H + 1));
                if ((v_ >= 0) && (y < H) && (n < a.n_maps)) {
                    const int lp = s * WP + px + 1;
                    sdst[u] = C::SWZ ? lp * 8 + ((seg ^ ((lp >> 3) & 1)) << 2) : lp * STRIDE + seg * 4;
                    sgp[u] = (int)((n * H + y) * W + px);
                }
            } else {
                const long gp = (long)mtile * C::PIX + px;
                sdst[u] = px * STRIDE + seg * 4;        // rows past the end are (re)written as zeros
                if (gp < total_pix) sgp[u] = (int)gp;
            }
            if (sdst[u] >= 0) sdst[u] |= (seg << 28);   // keep the 16-B segment index in the top bits
        }
    }
    f32x4 sv[U];
    // global -> registers for one chunk (issued a whole compute phase before the data is needed)
#define LRPX_STAGE_ISSUE(CHUNK)                                                                              \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                          \
        sv[u] = f32x4{0, 0, 0, 0};                                                                           \
        if (sgp[u] >= 0)                                                                                     \
            sv[u] = *reinterpret_cast<const f32x4*>(                                                         \
                a.in_chunk_stride ? a.in + (CHUNK) * a.in_chunk_stride + (long)sgp[u] * KC + ((sdst[u] >> 28) & 7) * 4 \
                                  : a.in + (long)sgp[u] * a.cin + (CHUNK) * KC + ((sdst[u] >> 28) & 7) * 4);  \
    }
#define LRPX_STAGE_COMMIT(BUFIDX)                                                                            \
    _Pragma("unroll") for (int u = 0; u < U; ++u)                                                            \
        if (sdst[u] >= 0) *reinterpret_cast<f32x4*>(lds + (BUFIDX) * BUF + (sdst[u] & 0x0fffffff)) = sv[u];

    LRPX_STAGE_ISSUE(chunk_lo)
    // zero both LDS buffers once: halo columns and inter-map zero rows are never written afterwards
    for (int i = tid; i < 2 * BUF / 4; i += NT) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0, 0, 0, 0};
    __syncthreads();
    LRPX_STAGE_COMMIT(0)

    f32x16 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // B fragments: one contiguous stream of (nchunk * STEPS) 1-KiB wave-loads per channel block
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wp) + (long)ocb * nchunk * (STEPS * 64) + lane;
    const int last_step = chunk_hi * STEPS - 1;
    f32x4 bq[NB];     // bq[0] = fragment of the current k-step, bq[i] = i steps ahead (L2 latency under load ~2 us)
#pragma unroll
    for (int i = 0; i < NB; ++i) bq[i] = f32x4{0, 0, 0, 0};
    if (wave_active) {
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) bq[i] = wp[(long)min(chunk_lo * STEPS + i, last_step) * 64];
    }
    __syncthreads();

    for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
        const bool more = chunk + 1 < chunk_hi;
        if (more) { LRPX_STAGE_ISSUE(chunk + 1) }
        if (wave_active) {
            const float* abuf = lds + ((chunk - chunk_lo) & 1) * BUF;
            const int g0step = chunk * STEPS;
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int tappix = (TAPS == 9) ? (tap / 3) * WP + (tap % 3) : 0;   // tap shift in pixels
                const int tapoff = tappix * STRIDE;
#pragma unroll
                for (int ks = 0; ks < C::KSTEPS; ++ks) {
                    const int step = tap * C::KSTEPS + ks;
                    bq[NB - 1] = wp[(long)min(g0step + step + NB - 1, last_step) * 64];   // NB-1 steps ahead
                    f32x4 av[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        if constexpr (C::SWZ) {
                            int corner = abase[j];
                            asm volatile("" : "+v"(corner));   // keep the 63 (tile, tap) addresses out of registers:
                            const int lp = corner + tappix;    // they are loop-invariant and would be hoisted + spilled
                            av[j] = *reinterpret_cast<const f32x4*>(abuf + lp * 8 + ((((lp >> 3) & 1) ^ lh) << 2));
                        } else {
                            av[j] = *reinterpret_cast<const f32x4*>(abuf + abase[j] + tapoff + ks * 8);
                        }
                    }
#pragma unroll
                    for (int j = 0; j < 7; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][e], bq[0][e], acc[j], 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < NB - 1; ++i) bq[i] = bq[i + 1];   // renamed away inside the unrolled chunk
                }
            }
        }
        // the other buffer was last read one iteration ago and every wave has passed a barrier since
        if (more) { LRPX_STAGE_COMMIT((chunk + 1 - chunk_lo) & 1) }
        __syncthreads();
    }
#undef LRPX_STAGE_ISSUE
#undef LRPX_STAGE_COMMIT
    if (!wave_active) return;

    // ---- epilogue: lane holds channel oc for 16 pixels of each of its 7 row-tiles ----
    EpiCtx cx;
    cx.oc = ocb * 32 + li;
    cx.lane = lane;
    cx.q0 = wm * 224 + 4 * lh;
    cx.g0 = (int)g0;
    cx.pix0 = (TAPS == 9) ? g0 * W : (long)mtile * C::PIX;
    cx.total_pix = total_pix;
    cx.xi_base = 0;
    // written out (a rolled loop would index acc[] dynamically -> accumulators in scratch) and software-pipelined:
    // the loads of tile j+1 are in flight while tile j is finished
    constexpr bool AL = (TAPS == 9) && (H % (C::R ? C::R : 1) == 0);
    if constexpr (AL) {
        const unsigned rr = (unsigned)cx.q0 / (unsigned)HW, cc = (unsigned)cx.q0 - rr * HW;
        const unsigned g = (unsigned)cx.g0 + rr;
        const unsigned n = g / (unsigned)HW;
        const long img = a.map2img ? a.map2img[n] : n;
        cx.xi_base = (img * a.pix_per_map + (long)((g - n * HW) * HW + cc)) * a.oc_split + cx.oc;
    }
    // FWD_DUAL on map-aligned tiles (conv1_1 of the forward trace): the activations' maximum per image for the fp16 scale of
    // the next layer, recorded here instead of by a streaming read of the 205 MB tensor afterwards
    constexpr bool AMX = (EPI == EPI_FWD_DUAL) && AL;
    EpiMax mx = {0.f, 0.f};
    EpiMax* mxp = (AMX && a.out0_amax) ? &mx : nullptr;
    EpiRegs ra, rb;
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 0, ra);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 1, rb);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 0, acc[0], ra, mxp);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 2, ra);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 1, acc[1], rb, mxp);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 3, rb);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 2, acc[2], ra, mxp);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 4, ra);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 3, acc[3], rb, mxp);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 5, rb);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 4, acc[4], ra, mxp);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 6, ra);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 5, acc[5], rb, mxp);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 6, acc[6], ra, mxp);
    if constexpr (AMX) {
        if (mxp) {
            float m = mx.m0;
#pragma unroll
            for (int o = 32; o; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
            const unsigned n = (unsigned)g0 / (unsigned)H;
            if (lane == 0 && m > 0.f && (int)n < a.n_maps) {
                const unsigned b = __builtin_bit_cast(unsigned, m);
                if (b > *reinterpret_cast<const volatile unsigned*>(&a.out0_amax[n])) atomicMax(&a.out0_amax[n], b);
            }
        }
    }
}

}  // namespace lrpx
