// Relevance rule of the FIRST VGG16 conv (3 signed input channels <- 64 channels at 224x224):
//   R[n][c][y][x] = x+[c] * convT(S, W+)[c] + x-[c] * convT(S, W-)[c]        (LRPtools/lrp_modules.py:124-150)
// Only 6 output values per pixel (K = 64*9 = 576 each): a 32-wide MFMA tile would waste 26/32 of its work, so
// this layer is a direct VALU convolution: one thread per output pixel, the S halo tile staged through LDS per
// 16-channel chunk, the 3456 weights read through the scalar cache (wave-uniform addresses).
// HBM: reads S once (12.8 MB per map) — this kernel sits at the HBM/VALU balance point.
#include "common.h"
#include "conv_f16x3.h"

namespace lrpx {

constexpr int FL_TH = 8, FL_TW = 32, FL_KC = 16, FL_STRIDE = FL_KC + 4;
constexpr int FL_HP = FL_TH + 2, FL_WP = FL_TW + 2;

// w6: [chunk 4][tap 9][ch 16][6] = {W+ for c=0..2, W- for c=0..2} with the kernel flipped (transposed conv)
__global__ void pack_first_layer_kernel(const float* __restrict__ w, float* __restrict__ w6, int cout, int plain) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;     // over cout*9*6
    if (idx >= cout * 9 * 6) return;
    const int o = idx % 6;
    int r = idx / 6;
    const int ch = r % FL_KC; r /= FL_KC;
    const int tap = r % 9;
    const int chunk = r / 9;
    const int co = chunk * FL_KC + ch, c = o % 3;
    const float x = w[((long)co * 3 + c) * 9 + (8 - tap)];
    if (plain) w6[idx] = o < 3 ? x : 0.f;      // plain transposed conv (guided backprop): second triple unused
    else w6[idx] = o < 3 ? fmaxf(x, 0.f) : fminf(x, 0.f);
}

__global__ __launch_bounds__(256) void first_layer_rel_kernel(const float* __restrict__ S, const float* __restrict__ w6,
                                                              const float* __restrict__ X8,
                                                              const int* __restrict__ map2img, float* __restrict__ out,
                                                              int cin, int plain, long chunk_stride) {
    constexpr int HW = 224;
    __shared__ __attribute__((aligned(16))) float lds[FL_HP * FL_WP * FL_STRIDE];
    const int n = blockIdx.z, ty0 = blockIdx.y * FL_TH, tx0 = blockIdx.x * FL_TW;
    const int tid = threadIdx.x, ly = tid >> 5, lx = tid & 31;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nchunk = cin / FL_KC;
    // staging items of this thread: (halo pixel, 16-byte segment) -> LDS offset and global element offset (without the
    // chunk term), fixed for all chunks.  The loads of chunk c+1 are in flight while chunk c is computed: one load at a
    // time with a wait after each (the previous version) left the kernel bound by global-load latency.
    constexpr int NIT = (FL_HP * FL_WP * (FL_KC / 4) + 255) / 256;
    int lds_off[NIT];
    long g_off[NIT];            // < 0: outside the image (stays zero) or no item
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int it = tid + k * 256;
        const int seg = it & 3, p = it >> 2;
        const int py = p / FL_WP, px = p - py * FL_WP;
        const int gy = ty0 + py - 1, gx = tx0 + px - 1;
        lds_off[k] = p * FL_STRIDE + seg * 4;
        const bool in_tile = it < FL_HP * FL_WP * (FL_KC / 4);
        const bool in_img = gy >= 0 && gy < HW && gx >= 0 && gx < HW;
        const long pix = ((long)n * HW + gy) * HW + gx;
        // channel-chunked S [cin/16][n_maps*HW*HW][16]: contiguous 64-byte runs per pixel; else NHWC
        g_off[k] = !in_tile ? -2 : (!in_img ? -1 : (chunk_stride ? pix * FL_KC + seg * 4 : pix * cin + seg * 4));
    }
    const long cstep = chunk_stride ? chunk_stride : FL_KC;
    f32x4 stage[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        stage[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g_off[k] >= 0) stage[k] = *reinterpret_cast<const f32x4*>(S + g_off[k]);
    }
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        __syncthreads();                              // the previous chunk has been consumed
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (g_off[k] >= -1) *reinterpret_cast<f32x4*>(lds + lds_off[k]) = stage[k];
        __syncthreads();
        if (chunk + 1 < nchunk) {
#pragma unroll
            for (int k = 0; k < NIT; ++k)
                if (g_off[k] >= 0) stage[k] = *reinterpret_cast<const f32x4*>(S + g_off[k] + (long)(chunk + 1) * cstep);
        }
        const float* wc = w6 + chunk * 9 * FL_KC * 6;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* sp = lds + ((ly + tap / 3) * FL_WP + lx + tap % 3) * FL_STRIDE;
#pragma unroll
            for (int c4 = 0; c4 < FL_KC / 4; ++c4) {
                const f32x4 s = *reinterpret_cast<const f32x4*>(sp + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* wk = wc + (tap * FL_KC + c4 * 4 + e) * 6;   // wave-uniform -> scalar loads
#pragma unroll
                    for (int o = 0; o < 6; ++o) acc[o] += s[e] * wk[o];
                }
            }
        }
    }
    const int y = ty0 + ly, x = tx0 + lx;
    const long img = map2img ? map2img[n] : n;
    const long p = (long)y * HW + x;
    if (plain) {     // image gradient: no input multiplication
#pragma unroll
        for (int c = 0; c < 3; ++c) out[((long)n * 3 + c) * HW * HW + p] = acc[c];
        return;
    }
    const float* xp = X8 + (img * HW * HW + p) * 8;          // [x+ (3) | x- (3) | 0 0]
    const f32x4 xa = *reinterpret_cast<const f32x4*>(xp), xb = *reinterpret_cast<const f32x4*>(xp + 4);
    const float xpos[3] = {xa[0], xa[1], xa[2]}, xneg[3] = {xa[3], xb[0], xb[1]};
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((long)n * 3 + c) * HW * HW + p] = xpos[c] * acc[c] + xneg[c] * acc[3 + c];
}

// ---------------------------------------------------------------------------------------------------------------
// The same rule on the matrix cores (split-product modes of the chain: S comes with its per-map maximum).
// GEMM view: M = 16 "rows" = {hi, lo} halves of W+ and W- for the 3 image channels (12 used), N = 16 pixels, K = 32
// channels of one tap: v_mfma_f32_16x16x32_f16.  S is split s = s_hi + s_lo (fp16 pair after the per-map power-of-two
// scale, as conv_f16x3.h), W likewise; one MFMA with B = s_hi gives s_hi*w_hi (rows 4c, 4c+2) and s_hi*w_lo (rows
// 4c+1, 4c+3), one with B = s_lo gives s_lo*w_hi (and the 2^-22 term s_lo*w_lo): 2 MFMAs per (16 pixels, tap, 32
// channels) = 36 per 16 pixels and K-chunk... 0.24 ms of matrix time per 320 maps against 0.74 ms of packed VALU FMAs,
// so the kernel is bound by its 4.1 GB read of S.  Row order: the four rows of image channel c sit in ONE lane group
// of the result layout (row = 4 * (lane >> 4) + reg), so R = x+ * (acc0 + acc1) + x- * (acc2 + acc3) needs no
// cross-lane traffic.
// Workgroup = one band of 8 image rows of one map, walked in 7 tiles of 32 columns x 2 chunks of 32 channels; the halo
// tile (10 x 34 pixels) is converted once into LDS (144 B per pixel: 32 hi | 32 lo | pad - conflict-free b128 reads of 16
// consecutive pixels), the next item's S is in flight in registers while the current one is multiplied.  Bands of a
// map run on ONE XCD (workgroup ids go round-robin over the 8 XCDs), so the halo rows are re-read from that L2.
constexpr int FM_TH = 8, FM_TW = 32, FM_HP = FM_TH + 2, FM_WP = FM_TW + 2, FM_PIXB = 144;
constexpr int FM_NPIX = FM_HP * FM_WP;                       // 340 halo pixels
constexpr int FM_NIT = (FM_NPIX * 8 + 255) / 256;            // float4 items per thread and K-chunk (11)
constexpr int FM_BANDS = 224 / FM_TH, FM_TILES = 224 / FM_TW;


typedef _Float16 fm_f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned fm_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned fm_u32x2 __attribute__((ext_vector_type(2)));

__global__ void pack_first_layer_mfma_kernel(const float* __restrict__ w, float* __restrict__ header, int plain) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;     // over 2 * 9 * 64 * 8 halves
    if (idx >= 2 * 9 * 64 * 8) return;
    const int kw = f16_scale_exp(reinterpret_cast<const unsigned*>(header)[1]);
    if (idx == 0) header[0] = exp2i(-kw);
    const int j = idx & 7, lane = (idx >> 3) & 63;
    const int tap = (idx >> 9) % 9, chunk = idx / (512 * 9);
    const int row = lane & 15, ch = chunk * 32 + 8 * (lane >> 4) + j;     // ch: channel of S = output channel of conv1_1
    const int c = row >> 2, kind = row & 3;                                // kind: hi(W+), lo(W+), hi(W-), lo(W-)
    float v = 0.f;
    if (row < 12) {
        const float x = w[((long)ch * 3 + c) * 9 + (8 - tap)];            // flipped kernel (transposed conv)
        v = kind < 2 ? (plain ? x : fmaxf(x, 0.f)) : (plain ? 0.f : fminf(x, 0.f));
    }
    _Float16 hi, lo;
    split2(v * exp2i(kw), hi, lo);
    reinterpret_cast<unsigned short*>(header + F16X3_HEADER_FLOATS)[idx] =
        __builtin_bit_cast(unsigned short, (kind & 1) ? lo : hi);
}

__global__ __launch_bounds__(256, 2) void first_layer_mfma_kernel(const float* __restrict__ S, const float* __restrict__ wpk,
                                                                  const float* __restrict__ X8,
                                                                  const int* __restrict__ map2img,
                                                                  const unsigned* __restrict__ s_amax,
                                                                  float* __restrict__ out, int n_maps, int plain,
                                                                  long chunk_stride) {
    constexpr int HW = 224;
    __shared__ __attribute__((aligned(16))) char lds[FM_NPIX * FM_PIXB];
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int n = (idx / FM_BANDS) * 8 + xcd, band = idx % FM_BANDS;
    if (n >= n_maps) return;
    const int ty0 = band * FM_TH;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // weight fragments of both K-chunks stay in registers for the whole band
    const fm_u32x4* wf = reinterpret_cast<const fm_u32x4*>(wpk + F16X3_HEADER_FLOATS) + lane;
    fm_f16x8 wa[2][9];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < 9; ++t) wa[c][t] = __builtin_bit_cast(fm_f16x8, wf[(c * 9 + t) * 64]);
    const int ka = f16_scale_exp(s_amax[n]);
    const float sc = exp2i(ka), inv = wpk[0] * exp2i(-ka);

    // staging items (halo pixel p, 16-byte segment seg of the 32-channel slice)
    int l_off[FM_NIT], pyx[FM_NIT];
#pragma unroll
    for (int k = 0; k < FM_NIT; ++k) {
        const int it = tid + k * 256;
        const int seg = it & 7, p = it >> 3;
        const int py = p / FM_WP, px = p - py * FM_WP;
        l_off[k] = p < FM_NPIX ? p * FM_PIXB + seg * 8 : -1;
        pyx[k] = (py << 16) | (px << 8) | seg;
    }
    const long pstr = chunk_stride ? 32 : 64, cstep = chunk_stride ? chunk_stride : 32;
    const float* __restrict__ Sn = S + (long)n * HW * HW * pstr;
    f32x4 stage[FM_NIT];
    auto in_image = [&](const int k, const int tx0) {
        const int gy = ty0 + (pyx[k] >> 16) - 1, gx = tx0 + ((pyx[k] >> 8) & 0xff) - 1;
        return l_off[k] >= 0 && (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
    };
    // branch-free: an item outside the image (or past the tile) reads element 0 of the map and is zeroed at commit
    auto issue = [&](const int tx0, const int c) {
#pragma unroll
        for (int k = 0; k < FM_NIT; ++k) {
            const int gy = ty0 + (pyx[k] >> 16) - 1, gx = tx0 + ((pyx[k] >> 8) & 0xff) - 1;
            const long off = in_image(k, tx0) ? (long)(gy * HW + gx) * pstr + (pyx[k] & 0xff) * 4 + c * cstep : 0;
            stage[k] = *reinterpret_cast<const f32x4*>(Sn + off);
        }
    };
    auto commit = [&](const int tx0) {
#pragma unroll
        for (int k = 0; k < FM_NIT; ++k) {
            const bool ok = in_image(k, tx0);
            // (packed conversions: v_cvt_pk_f16_f32, bit-identical to four scalar splits)
            unsigned h01, h23, l01, l23;
            f32x2_ hf;
            const f32x2_ z2 = {0.f, 0.f};
            split2_pk(ok ? f32x2_{stage[k][0], stage[k][1]} * f32x2_{sc, sc} : z2, h01, l01, hf);
            split2_pk(ok ? f32x2_{stage[k][2], stage[k][3]} * f32x2_{sc, sc} : z2, h23, l23, hf);
            char* d = lds + (l_off[k] >= 0 ? l_off[k] : 128);       // no item: the pad bytes of pixel 0
            *reinterpret_cast<fm_u32x2*>(d) = fm_u32x2{h01, h23};
            *reinterpret_cast<fm_u32x2*>(d + (l_off[k] >= 0 ? 64 : 8)) = fm_u32x2{l01, l23};
        }
    };
    // wave w owns image rows 2w, 2w+1 of the band, two 16-pixel tiles each
    int bbase[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)
        bbase[q] = ((2 * wave + (q >> 1)) * FM_WP + (q & 1) * 16 + (lane & 15)) * FM_PIXB + (lane >> 4) * 16;
    f32x4 acc[4];
    // the 36 (tap, tile) steps of a chunk as one sequence; the B fragments of step k + D are read before the MFMAs of step
    // k are issued (left alone the compiler emits read -> wait -> MFMA per step: 72 exposed LDS round trips per chunk)
    constexpr int D = 4;
    auto compute = [&](const int c) {
        fm_f16x8 rh[D], rl[D];
        auto rd = [&](const int k, fm_f16x8& h, fm_f16x8& l) {
            const int t = k >> 2, q = k & 3;
            const char* bp = lds + bbase[q] + ((t / 3) * FM_WP + (t % 3)) * FM_PIXB;
            h = *reinterpret_cast<const fm_f16x8*>(bp);
            l = *reinterpret_cast<const fm_f16x8*>(bp + 64);
        };
#pragma unroll
        for (int d = 0; d < D; ++d) rd(d, rh[d], rl[d]);
#pragma unroll
        for (int k = 0; k < 36; ++k) {
            const fm_f16x8 ch = rh[k % D], cl = rl[k % D];
            if (k + D < 36) {
                rd(k + D, rh[k % D], rl[k % D]);
                __builtin_amdgcn_sched_barrier(0);
            }
            acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c][k >> 2], cl, acc[k & 3], 0, 0, 0);   // small terms first
            acc[k & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wa[c][k >> 2], ch, acc[k & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const long img = map2img ? map2img[n] : n;
    const int g = lane >> 4, col = lane & 15;

    issue(0, 0);
    for (int tx = 0; tx < FM_TILES; ++tx) {
        const int tx0 = tx * FM_TW;
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        // the multiplicands of this tile's outputs, ahead of the next staging loads (the in-order counter would make the
        // epilogue wait for those otherwise)
        float xpv[4], xnv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const long p = (long)(ty0 + 2 * wave + (q >> 1)) * HW + tx0 + (q & 1) * 16 + col;
            const float* xp = X8 + (img * HW * HW + p) * 8;          // [x+ (3) | x- (3) | 0 0]
            xpv[q] = (g < 3 && !plain) ? xp[g] : 1.f;
            xnv[q] = (g < 3 && !plain) ? xp[3 + g] : 0.f;
        }
        __syncthreads();                 // the previous item has been consumed
        commit(tx0);
        __syncthreads();
        issue(tx0, 1);
        compute(0);
        __syncthreads();
        commit(tx0);
        __syncthreads();
        if (tx + 1 < FM_TILES) issue(tx0 + FM_TW, 0);
        compute(1);
        if (g < 3) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int y = ty0 + 2 * wave + (q >> 1), x = tx0 + (q & 1) * 16 + col;
                const long p = (long)y * HW + x;
                const float pos = (acc[q][0] + acc[q][1]) * inv, neg = (acc[q][2] + acc[q][3]) * inv;
                out[((long)n * 3 + g) * HW * HW + p] = plain ? pos : xpv[q] * pos + xnv[q] * neg;
            }
        }
    }
}

int first_layer_pack(const float* w, float* w6, int cout, int plain, hipStream_t s) {
    hipLaunchKernelGGL(pack_first_layer_kernel, dim3((cout * 54 + 255) / 256), dim3(256), 0, s, w, w6, cout, plain);
    return check_launch("pack_first_layer");
}

int first_layer_relevance(const float* S, const float* w6, const float* X8, const int* map2img, float* out, int n_maps,
                          int cin, int plain, int s_chunked, hipStream_t s) {
    const long chunk_stride = s_chunked ? (long)n_maps * 224 * 224 * FL_KC : 0;
    hipLaunchKernelGGL(first_layer_rel_kernel, dim3(224 / FL_TW, 224 / FL_TH, n_maps), dim3(256), 0, s, S, w6, X8,
                       map2img, out, cin, plain, chunk_stride);
    return check_launch("first_layer_relevance");
}

int first_layer_pack_mfma(const float* w, float* packed, int plain, hipStream_t s) {
    if (hipMemsetAsync(packed, 0, F16X3_HEADER_FLOATS * sizeof(float), s) != hipSuccess) {
        set_error("first_layer_pack_mfma: memset failed");
        return LRPX_ELAUNCH;
    }
    LRPX_TRY(amax_flat(w, 64 * 3 * 9, reinterpret_cast<unsigned*>(packed) + 1, s));
    hipLaunchKernelGGL(pack_first_layer_mfma_kernel, dim3((2 * 9 * 64 * 8 + 255) / 256), dim3(256), 0, s, w, packed, plain);
    return check_launch("pack_first_layer_mfma");
}

// S: NHWC [n_maps][224*224][64] (chunked32 = 0) or [2][n_maps*224*224][32]; s_amax[n] = float bits of max|S| of map n
int first_layer_relevance_mfma(const float* S, const float* packed, const float* X8, const int* map2img,
                               const unsigned* s_amax, float* out, int n_maps, int plain, int chunked32, hipStream_t s) {
    const long chunk_stride = chunked32 ? (long)n_maps * 224 * 224 * 32 : 0;
    const unsigned grid = (unsigned)((n_maps + 7) / 8) * 8 * FM_BANDS;
    hipLaunchKernelGGL(first_layer_mfma_kernel, dim3(grid), dim3(256), 0, s, S, packed, X8, map2img, s_amax, out, n_maps,
                       plain, chunk_stride);
    return check_launch("first_layer_relevance_mfma");
}

}  // namespace lrpx
