// Dense epsilon-rule / plain GEMM for FEW rows (the lock-step decoder relevance: rows = images x words, a few hundred)
// on the fp32 MFMA.  The big-tile engine (conv_mfma.h, 224 rows x 128 columns per workgroup) leaves most of the chip
// idle on a 320-row problem and pays its pipeline fill/drain plus split-K atomics: ~110 us per GEMM, 42 GEMMs per
// step.  Here one workgroup = 32 rows x 128 columns (4 waves, one 32x32 accumulator tile each, the whole K in one pass):
//   * the 32 x K slab of A is staged ONCE into LDS (row stride K+4 floats: conflict-free ds_read_b128; K > 1024 in
//     slabs of 1024 columns),
//   * B fragments stream from the fragment-major packed weights (lrpx_pack_weights, kc = 32: one contiguous 1-KiB
//     k-step of 8 per channel block) through an 8-deep register queue,
//   * 10 x N/128 workgroups for 320 rows - 120..160 per GEMM, no split-K, deterministic.
// Replaces torch.matmul / sum inside `lrp_linear_eps` (models/gridTDmodel.py:744-765) for the per-step gate rules.
#include "conv_launch.h"

namespace lrpx {

// K <= 1024: the whole 32 x K slab of A in LDS, one pass
template <int EPI>
__global__ __launch_bounds__(256, 2) void dense_small_kernel(ConvArgs a, int m_tiles, int n_blocks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mtile = blockIdx.x / n_blocks, nblk = blockIdx.x % n_blocks;
    const int K = a.cin;                         // multiple of 32
    const int stride = K + 4;
    const long rows = (long)a.n_maps * a.pix_per_map;
    const long row0 = (long)mtile * 32;
    const int ocb = nblk * 4 + wave;
    const bool wave_active = ocb * 32 < a.n_oc;
    const int li = lane & 31, lh = lane >> 5;

    // B queue first (independent of LDS): k-steps of 8, 1 KiB each, contiguous per channel block
    constexpr int NB = 8;
    const int nsteps = K / 8;
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wp) + (long)ocb * nsteps * 64 + lane;
    f32x4 bq[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) bq[i] = f32x4{0, 0, 0, 0};
    if (wave_active) {
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) bq[i] = wp[(long)min(i, nsteps - 1) * 64];
    }

    // stage the A slab: 32 rows x K floats, float4 items, rows past the end are zeros
    const int k4 = K / 4;
    for (int it = tid; it < 32 * k4; it += 256) {
        const int r = it / k4, c4 = it - r * k4;
        f32x4 v = f32x4{0, 0, 0, 0};
        if (row0 + r < rows) v = *reinterpret_cast<const f32x4*>(a.in + (row0 + r) * K + c4 * 4);
        *reinterpret_cast<f32x4*>(lds + r * stride + c4 * 4) = v;
    }
    __syncthreads();
    if (!wave_active) return;

    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* ap = lds + li * stride + lh * 4;
    for (int ks = 0; ks < nsteps; ks += NB) {         // nsteps = K/8 is a multiple of 4; handle tails by clamping
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int step = ks + u;
            bq[NB - 1] = wp[(long)min(step + NB - 1, nsteps - 1) * 64];
            if (step < nsteps) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(ap + step * 8);
                const f32x4 bv = bq[0];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < NB - 1; ++i) bq[i] = bq[i + 1];
        }
    }

    // epilogue: acc[e] is row (e&3) + 8*(e>>2) + 4*lh, column oc
    const int oc = ocb * 32 + li;
    const int ncol = a.oc_split;
    if (oc >= ncol) return;
    const unsigned P = (unsigned)a.pix_per_map;
    float bias = 0.f;
    if (EPI == EPI_PLAIN && a.bias) bias = a.bias[oc];
    float xv[16];
    long nn[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const long row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        xv[e] = 0.f; nn[e] = 0;
        if (EPI == EPI_REL && row < rows) {
            const long n = row / P, p = row - n * P;
            const long img = a.map2img ? a.map2img[n] : n;
            nn[e] = n;
            xv[e] = a.X[(img * P + p) * ncol + oc];
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const long row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row >= rows) continue;
        float v = acc[e];
        if (EPI == EPI_REL) {
            if (a.U) v += a.U[nn[e] * ncol + oc];
            a.out0[row * ncol + oc] = xv[e] * v;
        } else {
            v += bias;
            if (a.relu) v = v > 0.f ? v : 0.f;
            a.out0[row * ncol + oc] = v;
        }
    }
}

// K > 1024: the same with A staged in slabs of 1024 columns
template <int EPI>
__global__ __launch_bounds__(256, 2) void dense_slab_kernel(ConvArgs a, int m_tiles, int n_blocks) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mtile = blockIdx.x / n_blocks, nblk = blockIdx.x % n_blocks;
    const int K = a.cin;                         // multiple of 32
    constexpr int KSLAB = 1024;                  // LDS slab of A: 32 rows x min(K, 1024) floats (+4 pad per row)
    const int kslab = K < KSLAB ? K : KSLAB;
    const int stride = kslab + 4;
    const long rows = (long)a.n_maps * a.pix_per_map;
    const long row0 = (long)mtile * 32;
    const int ocb = nblk * 4 + wave;
    const bool wave_active = ocb * 32 < a.n_oc;
    const int li = lane & 31, lh = lane >> 5;

    // B queue first (independent of LDS): k-steps of 8, 1 KiB each, contiguous per channel block
    constexpr int NB = 8;
    const int nsteps = K / 8;
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wp) + (long)ocb * nsteps * 64 + lane;
    f32x4 bq[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) bq[i] = f32x4{0, 0, 0, 0};
    if (wave_active) {
#pragma unroll
        for (int i = 0; i < NB - 1; ++i) bq[i] = wp[(long)min(i, nsteps - 1) * 64];
    }
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* ap = lds + li * stride + lh * 4;

    for (int k0 = 0; k0 < K; k0 += KSLAB) {
        const int kw = min(KSLAB, K - k0);           // width of this slab (multiple of 32)
        if (k0) __syncthreads();                     // everybody is done with the previous slab
        // stage the A slab: 32 rows x kw floats, float4 items, rows past the end are zeros
        const int k4 = kw / 4;
        for (int it = tid; it < 32 * k4; it += 256) {
            const int r = it / k4, c4 = it - r * k4;
            f32x4 v = f32x4{0, 0, 0, 0};
            if (row0 + r < rows) v = *reinterpret_cast<const f32x4*>(a.in + (row0 + r) * K + k0 + c4 * 4);
            *reinterpret_cast<f32x4*>(lds + r * stride + c4 * 4) = v;
        }
        __syncthreads();
        if (wave_active) {
            const int s0 = k0 / 8, sn = kw / 8;      // global k-step range of the slab
            for (int ks = 0; ks < sn; ks += NB) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int step = ks + u;
                    if (step < sn) {
                        bq[NB - 1] = wp[(long)min(s0 + step + NB - 1, nsteps - 1) * 64];
                        const f32x4 av = *reinterpret_cast<const f32x4*>(ap + step * 8);
                        const f32x4 bv = bq[0];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], acc, 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NB - 1; ++i) bq[i] = bq[i + 1];
                    }
                }
            }
        }
    }
    if (!wave_active) return;

    // epilogue: acc[e] is row (e&3) + 8*(e>>2) + 4*lh, column oc
    const int oc = ocb * 32 + li;
    const int ncol = a.oc_split;
    if (oc >= ncol) return;
    const unsigned P = (unsigned)a.pix_per_map;
    float bias = 0.f;
    if (EPI == EPI_PLAIN && a.bias) bias = a.bias[oc];
    float xv[16];
    long nn[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const long row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        xv[e] = 0.f; nn[e] = 0;
        if (EPI == EPI_REL && row < rows) {
            const long n = row / P, p = row - n * P;
            const long img = a.map2img ? a.map2img[n] : n;
            nn[e] = n;
            xv[e] = a.X[(img * P + p) * ncol + oc];
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const long row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        if (row >= rows) continue;
        float v = acc[e];
        if (EPI == EPI_REL) {
            if (a.U) v += a.U[nn[e] * ncol + oc];
            a.out0[row * ncol + oc] = xv[e] * v;
        } else {
            v += bias;
            if (a.relu) v = v > 0.f ? v : 0.f;
            a.out0[row * ncol + oc] = v;
        }
    }
}

// The plain GEMM with the K loop of a tile split over four waves (the decoders' image-gradient BPTT: 640 rows x K = 2048:
// 137 us as 80 x 4 waves with 1024 dependent fp32 MFMAs each).  Workgroup = 16 waves = 4 column blocks x 4 K quarters over
// the same 32 x K slab of A; the four partial tiles of a column block meet in LDS and are summed in K order
// (deterministic).  Round 6: also the epsilon rule of the decoders' lock-steps in the exact modes (REL epilogue; LRPX_DENSE_KS_REL=0:
// the one-wave kernel above) - 20 -> 12 us per lock-step GEMM, the one-image drop-in's decoder relevance 1.14 -> 0.97 ms.  (Rounds 3 - 5
// kept the rule on the one-wave kernel because this order moved ONE T = 20 word relevance of the LRP goldens above its 1e-5 bound; with
// the round-6 K loop of the trace linears - lrpx_decoder.hip, linear_mfma_core - the worst row sits at 9.2e-6 in this order, 9.9e-6 in
// the other: the row is carried by the encoder features' own 1.8e-6, tests/diag_t20_words.py.)
// FUSE (round 6): lock-step s of the AoA decoder relevance (models/aoamodel.py:1114-1134) in this kernel's epilogue - what FUSE of
// dense_small_f16x3_kernel (dense_f16x3.hip) is for the fp16 split products, here in the exact arithmetic of the default mode: the product
// r_xh = xh * (W_g^T A) is consumed where it is formed, by column range [emb | glob | h] (E = H = K = 512, 128-column workgroups):
// emb -> one partial sum of r_words per (row, workgroup) (wpart, added up in a fixed order afterwards), glob -> r_glob +=, h -> the next
// lock-step's A = (q1 r_h) / dg from the per-trace tables (the expressions and their order as in aoa_rel_ca_kernel / aoa_rel_a_kernel:
// r_feat bit-identical to the two-launch step).  One launch per lock-step instead of two.
template <int EPI, bool FUSE = false>
__global__ __launch_bounds__(1024, 1) void dense_ks_kernel(ConvArgs a, int m_tiles, int n_blocks, AoaStepFuse fz) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ float wsum[16][4];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wc = wave & 3, wk = wave >> 2;
    const int mtile = blockIdx.x / n_blocks, nblk = blockIdx.x % n_blocks;
    const int K = a.cin;                         // multiple of 32
    constexpr int KSLAB = 1024;
    const int kslab = K < KSLAB ? K : KSLAB;
    const int stride = kslab + 4;
    const long rows = (long)a.n_maps * a.pix_per_map;
    const long row0 = (long)mtile * 32;
    const int ocb = nblk * 4 + wc;
    const bool wave_active = ocb * 32 < a.n_oc;
    const int li = lane & 31, lh = lane >> 5;
    constexpr int NB = 8;
    const int nsteps = K / 8;
    const f32x4* wp = reinterpret_cast<const f32x4*>(a.wp) + (long)ocb * nsteps * 64 + lane;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const float* ap = lds + li * stride + lh * 4;

    for (int k0 = 0; k0 < K; k0 += KSLAB) {
        const int kw = min(KSLAB, K - k0);           // width of this slab (multiple of 32)
        if (k0) __syncthreads();                     // everybody is done with the previous slab
        const int sn = kw / 8, sq = sn / 4;          // k-steps of the slab (multiple of 4), per K quarter
        const int s_lo = k0 / 8 + wk * sq, s_hi = s_lo + sq;      // this wave's global k-steps
        f32x4 bq[NB];
#pragma unroll
        for (int i = 0; i < NB; ++i) bq[i] = f32x4{0, 0, 0, 0};
        if (wave_active) {                           // B queue first (independent of LDS)
#pragma unroll
            for (int i = 0; i < NB - 1; ++i) bq[i] = wp[(long)min(s_lo + i, s_hi - 1) * 64];
        }
        const int k4 = kw / 4;
        for (int it = tid; it < 32 * k4; it += 1024) {
            const int r = it / k4, c4 = it - r * k4;
            f32x4 v = f32x4{0, 0, 0, 0};
            if (row0 + r < rows) v = *reinterpret_cast<const f32x4*>(a.in + (row0 + r) * K + k0 + c4 * 4);
            *reinterpret_cast<f32x4*>(lds + r * stride + c4 * 4) = v;
        }
        __syncthreads();
        if (wave_active) {
            const float* aq = ap + wk * sq * 8;
            for (int ks = 0; ks < sq; ks += NB) {
#pragma unroll
                for (int u = 0; u < NB; ++u) {
                    const int step = ks + u;
                    if (step < sq) {
                        bq[NB - 1] = wp[(long)min(s_lo + step + NB - 1, s_hi - 1) * 64];
                        const f32x4 av = *reinterpret_cast<const f32x4*>(aq + step * 8);
                        const f32x4 bv = bq[0];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv[0], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv[1], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[2], bv[2], acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[3], bv[3], acc, 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < NB - 1; ++i) bq[i] = bq[i + 1];
                    }
                }
            }
        }
    }
    __syncthreads();                                 // the slab is free: partial tiles [wk][wc][32 rows][33]
    float* red = lds;
#pragma unroll
    for (int e = 0; e < 16; ++e)
        red[((wk * 4 + wc) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh) * 33 + li] = acc[e];
    __syncthreads();
    // epilogue: thread = (column of the 128, row slice): rows (tid >> 7) + 8 i
    const int col = tid & 127, cb = col >> 5, cl = col & 31;
    const int oc = nblk * 128 + col;
    const int ncol = a.oc_split;
    if constexpr (FUSE) {                                 // (EPI_REL, P = 1, no U, n_oc = oc_split = 3 H: host-checked - every column exists)
        const int Hh = a.cin, T = fz.T, sNow = fz.s;
        const int part = (nblk * 128) / Hh;               // 0 emb, 1 glob, 2 h (workgroup-uniform)
        float rx[4];
        int tm[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = (tid >> 7) + 8 * i;
            const long row = row0 + r;
            tm[i] = row < rows ? fz.tmax[row] : -2;       // (-2: no such row; -1: the caption does not have the word)
            rx[i] = 0.f;
            if (row < rows) {
                const float v = (red[((0 * 4 + cb) * 32 + r) * 33 + cl] + red[((1 * 4 + cb) * 32 + r) * 33 + cl]) +
                                (red[((2 * 4 + cb) * 32 + r) * 33 + cl] + red[((3 * 4 + cb) * 32 + r) * 33 + cl]);
                rx[i] = a.X[(long)a.map2img[row] * ncol + oc] * v;
            }
        }
        if (part == 0) {
            // per row the sum over the workgroup's 128 columns: the 64 lanes of a wave, then the two waves that hold the row
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float v = rx[i];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (lane == 0) wsum[wave][i] = v;
            }
            __syncthreads();
            if (tid < 32) {
                const long row = row0 + tid;
                if (row < rows) {
                    const int t = fz.tmax[row], gq = tid & 7, i = tid >> 3;
                    if (sNow <= t) fz.wpart[(row * T + (t - sNow)) * 4 + (nblk & 3)] = wsum[2 * gq][i] + wsum[2 * gq + 1][i];
                }
            }
        } else if (part == 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long row = row0 + (tid >> 7) + 8 * i;
                if (sNow <= tm[i]) fz.r_glob[row * Hh + (oc - Hh)] += rx[i];
            }
        } else if (sNow + 1 < T) {
            const int ch = oc - 2 * Hh;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const long row = row0 + (tid >> 7) + 8 * i;
                if (tm[i] == -2) continue;
                float q1 = 0.f, dg = 1.f;                 // (a row that is not active at s + 1: A = 0)
                if (sNow + 1 <= tm[i]) { const long ti = (row - sNow - 1) * Hh + ch; q1 = fz.q1[ti]; dg = fz.dg[ti]; }
                fz.A_next[row * Hh + ch] = (q1 * rx[i]) / dg;
            }
        }
        return;
    }
    if (oc >= ncol || oc >= a.n_oc) return;
    float bias = 0.f;
    if (EPI == EPI_PLAIN && a.bias) bias = a.bias[oc];
    const long P = a.pix_per_map;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (tid >> 7) + 8 * i;
        const long row = row0 + r;
        if (row >= rows) continue;
        float v = (red[((0 * 4 + cb) * 32 + r) * 33 + cl] + red[((1 * 4 + cb) * 32 + r) * 33 + cl]) +
                  (red[((2 * 4 + cb) * 32 + r) * 33 + cl] + red[((3 * 4 + cb) * 32 + r) * 33 + cl]);
        if constexpr (EPI == EPI_REL) {               // (the epilogue of dense_small_kernel)
            const long n = row / P, p = row - n * P;
            const long img = a.map2img ? a.map2img[n] : n;
            if (a.U) v += a.U[n * ncol + oc];
            a.out0[row * ncol + oc] = a.X[(img * P + p) * ncol + oc] * v;
        } else {
            v += bias;
            if (a.relu) v = v > 0.f ? v : 0.f;
            a.out0[row * ncol + oc] = v;
        }
    }
}

template <int EPI>
static int launch_dense_small_t(const ConvArgs& a, hipStream_t stream) {
    const long rows = (long)a.n_maps * a.pix_per_map;
    const int m_tiles = (int)ceil_div(rows, 32);
    const int n_blocks = (int)ceil_div(a.n_oc, 128);
    const int lds = 32 * ((a.cin < 1024 ? a.cin : 1024) + 4) * (int)sizeof(float);
    const int dense_ks = switches().dense_1wave ? 0 : 1;      // (A/B switch)
    {
        if (dense_ks && a.cin >= 512 && (EPI == EPI_PLAIN || switches().dense_ks_rel)) {
            // K split over four waves per column block (16-wave workgroups)
            constexpr int RED = 16 * 32 * 33 * (int)sizeof(float);
            const int lds_ks = lds > RED ? lds : RED;
            constexpr int LDS_KS_MAX = 32 * (1024 + 4) * (int)sizeof(float);
            static LdsOnce once_ks;
            LRPX_TRY(reserve_lds_once(once_ks, dense_ks_kernel<EPI>, LDS_KS_MAX, "dense_ks"));
            hipLaunchKernelGGL(dense_ks_kernel<EPI>, dim3((unsigned)(m_tiles * n_blocks)), dim3(1024), lds_ks, stream, a, m_tiles,
                               n_blocks, AoaStepFuse{});
            return check_launch("dense_ks");
        }
    }
    const bool slabs = a.cin > 1024;
    auto kern = slabs ? dense_slab_kernel<EPI> : dense_small_kernel<EPI>;
    // the largest LDS image either kernel ever asks for (32 rows x (1024 + 4) floats), reserved once per kernel
    constexpr int LDS_MAX = 32 * (1024 + 4) * (int)sizeof(float);
    static LdsOnce once[2];
    LRPX_TRY(reserve_lds_once(once[slabs], kern, LDS_MAX, "dense_small"));
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_tiles * n_blocks)), dim3(256), lds, stream, a, m_tiles, n_blocks);
    return check_launch("dense_small");
}

// rows <= 4096, K in LDS slabs of 1024 (32 x 1028 floats), REL (out0 only) or PLAIN
bool dense_small_fits(const ConvArgs& a) {
    return a.pix_per_map > 0 && (long)a.n_maps * a.pix_per_map <= 4096 && a.cin % 32 == 0 && a.cin <= 8192 && !a.out1 && a.out0 &&
           (a.epi == EPI_REL || a.epi == EPI_PLAIN);
}

// lock-step fz.s of the AoA relevance in the exact arithmetic: the gate rule's GEMM (K split over four waves) with the step's point-wise
// code in its epilogue (FUSE above); weights from lrpx_pack_weights(PACK_DENSE_T, kc = 32)
int launch_dense_ks_aoa_step(const ConvArgs& a, const AoaStepFuse& fz, hipStream_t stream) {
    const long rows = (long)a.n_maps * a.pix_per_map;
    LRPX_REQUIRE(a.cin == 512 && a.n_oc == 3 * a.cin && a.oc_split == a.n_oc && a.pix_per_map == 1 && !a.U && a.X && a.map2img,
                 "dense_ks (AoA lock-step): built for E = H = K = 512, N = 1536, one row per map, x and map2img given");
    LRPX_REQUIRE(rows > 0 && rows < 0x7fffffffL && fz.T > 0 && rows % fz.T == 0 && fz.s >= 0 && fz.s < fz.T && fz.q1 && fz.dg && fz.tmax &&
                     fz.r_glob && fz.wpart && (fz.A_next || fz.s + 1 >= fz.T), "dense_ks (AoA lock-step): bad step arguments");
    const int m_tiles = (int)ceil_div(rows, 32), n_blocks = a.n_oc / 128;
    constexpr int RED = 16 * 32 * 33 * (int)sizeof(float);
    const int lds = 32 * (a.cin + 4) * (int)sizeof(float);
    constexpr int LDS_KS_MAX = 32 * (1024 + 4) * (int)sizeof(float);
    auto kern = dense_ks_kernel<EPI_REL, true>;
    static LdsOnce once;
    LRPX_TRY(reserve_lds_once(once, kern, LDS_KS_MAX, "dense_ks (AoA lock-step)"));
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_tiles * n_blocks)), dim3(1024), lds > RED ? lds : RED, stream, a, m_tiles, n_blocks, fz);
    return check_launch("dense_ks (AoA lock-step)");
}

int launch_dense_small(const ConvArgs& a, hipStream_t stream) {
    return a.epi == EPI_REL ? launch_dense_small_t<EPI_REL>(a, stream) : launch_dense_small_t<EPI_PLAIN>(a, stream);
}

}  // namespace lrpx
