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

namespace lrpx {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

enum Epilogue : int {
    EPI_FWD_DUAL = 0,   // oc <  split: out0 = relu(acc + bias)   oc >= split: out1 = acc        (Z+)
    EPI_REL = 1,        // r = X * (acc + U);  out0 = r;  out1 = r / stab(Zdiv)
    EPI_FIRST = 2,      // first VGG layer: r = X+ * acc[c] + X- * acc[c+3]  -> NCHW out0 (X stored split x+|x-)
    EPI_PLAIN = 3,      // out0 = acc (+ bias) (optionally relu)
    EPI_GUIDED = 4,     // guided backprop: out0 = max(acc,0) * [Y > 0]
};

enum Stab : int { STAB_NONE = 0, STAB_SAFE = 1, STAB_EPS = 2 };

struct ConvArgs {
    const float* in;        // [n_maps*P][cin]  A operand (already S = R/Z for relevance passes)
    const float* wp;        // packed weights  [n_ocb][nchunk][taps][KC/8][64][4]
    int n_maps;             // maps (images) in `in`
    int cin;                // input channels (multiple of KC)
    int n_oc;               // output channels incl. padding (multiple of 32)
    int pix_per_map;        // H*W (conv) or rows per map (dense)
    int epi;
    int stab;               // how out1 divides by Zdiv
    int oc_split;           // EPI_FWD_DUAL: number of plain channels; else: real output channels
    int relu;               // EPI_PLAIN: apply relu
    const float* bias;      // [oc]                      (FWD_DUAL / PLAIN, may be null)
    const float* X;         // [n_img*P][oc_split]       multiplicand (REL / FIRST / GUIDED mask)
    const float* U;         // [n_maps][oc_split]        per-map addend inside the bracket (REL, may be null)
    const float* Zdiv;      // [n_img*P][oc_split]       denominator for out1 (REL, may be null)
    const int* map2img;     // [n_maps] image index of every map for X/Zdiv (null: identity)
    float* out0;
    float* out1;
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
    static constexpr int NSLOT = (TAPS == 9) ? R + 2 + (R - 1 + H - 1) / H : 0;
    static constexpr int STRIDE = KC + 4;                // floats per LDS pixel (pad: conflict-free b128)
    static constexpr int LDS_PIX = (TAPS == 9) ? NSLOT * WP : PIX;
    static constexpr int LDS_BYTES = LDS_PIX * STRIDE * 4;
    static constexpr int NT = 64 * MT * NWN;
    static constexpr int KSTEPS = KC / 8;
    static_assert(TAPS == 1 || PIX % W == 0, "tile must be whole rows");
};

struct EpiCtx {
    int oc, lane, q0, g0;
    long pix0, total_pix;
};

// Per-element epilogue.  acc[j][e] is pixel q = q0 + 32*j + (e&3) + 8*(e>>2) of the tile, channel cx.oc.
template <int EPI, int HW, int TAPS>
__device__ __forceinline__ void epilogue_tile(const ConvArgs& a, const EpiCtx& cx, const int j, const f32x16 accj) {
    const int oc = cx.oc;
    const int ncol = a.oc_split;   // real channels of X / Zdiv / out (REL...), plain channels (FWD_DUAL)
    const unsigned P = (unsigned)a.pix_per_map;
    float bias = 0.f;
    if (EPI == EPI_FWD_DUAL || EPI == EPI_PLAIN) bias = (a.bias && oc < ncol) ? a.bias[oc] : 0.f;
    {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int q = cx.q0 + 32 * j + (e & 3) + 8 * (e >> 2);
            const long gp = cx.pix0 + q;
            if (gp >= cx.total_pix) continue;
            float v = accj[e];
            // map index n and pixel-in-map p (32-bit; constant divisors for the conv case)
            unsigned n, p;
            if (TAPS == 9) {
                const unsigned r = (unsigned)q / (unsigned)HW, c = (unsigned)q - r * HW;
                const unsigned g = (unsigned)cx.g0 + r;
                n = g / (unsigned)HW;
                p = (g - n * HW) * HW + c;
            } else {
                n = (unsigned)gp / P;
                p = (unsigned)gp - n * P;
            }
            if (EPI == EPI_FWD_DUAL) {
                if (oc < ncol) {
                    v += bias;
                    a.out0[gp * ncol + oc] = v > 0.f ? v : 0.f;
                } else if (oc < 2 * ncol) {
                    a.out1[gp * ncol + (oc - ncol)] = v;
                }
            } else if (EPI == EPI_REL) {
                if (oc < ncol) {
                    const long img = a.map2img ? a.map2img[n] : n;
                    const long xi = (img * P + p) * ncol + oc;
                    if (a.U) v += a.U[(long)n * ncol + oc];
                    const float r = a.X[xi] * v;
                    if (a.out0) a.out0[gp * ncol + oc] = r;
                    if (a.out1) {
                        float z = a.Zdiv[xi];
                        z = (a.stab == STAB_SAFE) ? stab_safe(z) : ((a.stab == STAB_EPS) ? stab_eps(z) : z);
                        a.out1[gp * ncol + oc] = r / z;
                    }
                }
            } else if (EPI == EPI_FIRST) {
                // channels 0..2 carry convT(S, W+), 3..5 convT(S, W-); combine across lanes of the same half
                const int li = cx.lane & 31;
                const float other = __shfl(v, (cx.lane & 32) + ((li + 3) & 31), 64);
                if (oc < 3) {
                    const long img = a.map2img ? a.map2img[n] : n;
                    const float* xp = a.X + (img * P + p) * ncol + oc;   // image kept NHWC split: [x+ (3) | x- (3) | 0 0]
                    const float r = xp[0] * v + xp[3] * other;
                    a.out0[((long)n * 3 + oc) * P + p] = r;           // NCHW
                }
            } else if (EPI == EPI_PLAIN) {
                if (oc < ncol) {
                    v += bias;
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    a.out0[gp * ncol + oc] = v;
                }
            } else {   // EPI_GUIDED
                if (oc < ncol) {
                    const long img = a.map2img ? a.map2img[n] : n;
                    const float y = a.X[(img * P + p) * ncol + oc];
                    a.out0[gp * ncol + oc] = (y > 0.f && v > 0.f) ? v : 0.f;
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
    // pixel tile get consecutive slots on ONE XCD and re-use the tile's A rows from that XCD's L2.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int mtile = (idx / n_blocks) * 8 + xcd;
    const int nblk = idx % n_blocks;
    if (mtile >= m_tiles) return;   // whole workgroup exits together

    const int ocb = nblk * NWN + wn;                         // 32-channel block of this wave
    const bool wave_active = ocb * 32 < a.n_oc;
    const int nchunk = a.cin / KC;
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
            abase[j] = ((slot - 1) * WP + c) * STRIDE + lh * 4;   // top-left corner of its 3x3 window
        }
    } else {
#pragma unroll
        for (int j = 0; j < 7; ++j) abase[j] = (wm * 224 + 32 * j + li) * STRIDE + lh * 4;
    }

    // ---- zero the LDS once: halo columns and inter-map zero rows are never written afterwards ----
    for (int i = tid; i < C::LDS_PIX * STRIDE / 4; i += NT) reinterpret_cast<f32x4*>(lds)[i] = f32x4{0, 0, 0, 0};

    f32x16 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    const f32x4* wbase = reinterpret_cast<const f32x4*>(a.wp) + (long)ocb * nchunk * (TAPS * C::KSTEPS * 64) + lane;

    for (int chunk = 0; chunk < nchunk; ++chunk) {
        __syncthreads();   // previous chunk's LDS reads are done (also orders the initial zero fill)
        // ---- stage A: global (NHWC, 16-B segments) -> registers -> LDS ----
        {
            constexpr int SEG = KC / 4;
            constexpr int ROWPIX = (TAPS == 9) ? W : C::PIX;
            constexpr int NROW = (TAPS == 9) ? C::NSLOT : 1;
            constexpr int NITEM = NROW * ROWPIX * SEG;
            constexpr int BATCH = 8;
            const float* src0 = a.in + (long)chunk * KC;
            for (int it0 = tid; it0 < NITEM; it0 += NT * BATCH) {
                f32x4 v[BATCH];
                int dst[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    int it = it0 + u * NT;
                    dst[u] = -1;
                    if (it < NITEM) {
                        int s = it / (ROWPIX * SEG);
                        int rem = it - s * (ROWPIX * SEG);
                        int px = rem / SEG, seg = rem - px * SEG;
                        long gp;   // global pixel index
                        bool ok;
                        if constexpr (TAPS == 9) {
                            long v_ = v0 - 1 + s;
                            long n = v_ / (H + 1);
                            int y = (int)(v_ - n * (H + 1));
                            ok = (v_ >= 0) && (y < H) && (n < a.n_maps);
                            gp = (n * H + y) * W + px;
                            if (ok) dst[u] = (s * WP + px + 1) * STRIDE + seg * 4;
                        } else {
                            gp = (long)mtile * C::PIX + px;
                            ok = gp < total_pix;
                            dst[u] = px * STRIDE + seg * 4;   // rows past the end are written as zeros
                        }
                        v[u] = ok ? *reinterpret_cast<const f32x4*>(src0 + gp * a.cin + seg * 4) : f32x4{0, 0, 0, 0};
                    }
                }
#pragma unroll
                for (int u = 0; u < BATCH; ++u)
                    if (dst[u] >= 0) *reinterpret_cast<f32x4*>(lds + dst[u]) = v[u];
            }
        }
        __syncthreads();
        if (wave_active) {
            const f32x4* wp = wbase + (long)chunk * (TAPS * C::KSTEPS * 64);
            f32x4 bcur = wp[0];
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int tapoff = (TAPS == 9) ? ((tap / 3) * WP + (tap % 3)) * STRIDE : 0;
#pragma unroll
                for (int ks = 0; ks < C::KSTEPS; ++ks) {
                    constexpr int LAST = TAPS * C::KSTEPS - 1;
                    const int step = tap * C::KSTEPS + ks;
                    f32x4 bnext = bcur;
                    if (step < LAST) bnext = wp[(step + 1) * 64];
                    f32x4 av[7];
#pragma unroll
                    for (int j = 0; j < 7; ++j)
                        av[j] = *reinterpret_cast<const f32x4*>(lds + abase[j] + tapoff + ks * 8);
#pragma unroll
                    for (int j = 0; j < 7; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j][e], bcur[e], acc[j], 0, 0, 0);
                    bcur = bnext;
                }
            }
        }
    }
    if (!wave_active) return;

    // ---- epilogue: lane holds channel oc for 16 pixels of each of its 7 row-tiles ----
    EpiCtx cx;
    cx.oc = ocb * 32 + li;
    cx.lane = lane;
    cx.q0 = wm * 224 + 4 * lh;
    cx.g0 = (int)g0;
    cx.pix0 = (TAPS == 9) ? g0 * W : (long)mtile * C::PIX;
    cx.total_pix = total_pix;
    // written out: the pragma-unroll budget refuses a 7 x 16-element body, and a rolled loop would
    // index acc[] dynamically (accumulators in scratch)
    epilogue_tile<EPI, HW, TAPS>(a, cx, 0, acc[0]);
    epilogue_tile<EPI, HW, TAPS>(a, cx, 1, acc[1]);
    epilogue_tile<EPI, HW, TAPS>(a, cx, 2, acc[2]);
    epilogue_tile<EPI, HW, TAPS>(a, cx, 3, acc[3]);
    epilogue_tile<EPI, HW, TAPS>(a, cx, 4, acc[4]);
    epilogue_tile<EPI, HW, TAPS>(a, cx, 5, acc[5]);
    epilogue_tile<EPI, HW, TAPS>(a, cx, 6, acc[6]);
}

}  // namespace lrpx
