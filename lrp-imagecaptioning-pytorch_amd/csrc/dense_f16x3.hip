// Dense relevance GEMM on the fp16 matrix cores with fp32-grade results ("f16x3", see conv_f16x3.h for the arithmetic):
// the epsilon rules of the decoders that run over every (word, pixel) row -
//   gridTD  R_feat[k] = F[k] * (W_proj^T (R_proj[k] / z~) + U)                       models/gridTDmodel.py:1125-1128
//   AoA     R_proj[k] = Vp[k] * (W_v^T (R_val[k] / z~) + U),  R_feat[k] likewise     models/aoamodel.py:1135-1148
// - are GEMMs of B*T*P rows (62 720 for 16 images x 20 words x 196 pixels; 23 040 x 2048 columns for the 36 x 2048 bottom-up
// features) that ran on the fp32 MFMA (v_mfma_f32_32x32x2_f32, 1/16 of the fp16 rate): 0.58 ms of a config-2 step and 30 %
// of the GPU time of config 5.  Here: A (fp32, any magnitude) is scaled PER MAP (the P rows of one (image, word)) by 2^kA
// from `in_amax` into [2^14, 2^15), split a = a0 + a1 while staged into LDS; the weights come pre-split and pre-scaled
// (lrpx_pack_weights_f16x2, taps = 1); a0b1 + a1b0 + a0b0 accumulate in fp32 on v_mfma_f32_32x32x16_f16.
//
// Tile: 128 rows x 128 columns per workgroup of 4 waves (2 x 2, a wave owns 2 x 2 accumulator tiles of 32 x 32), K in
// chunks of 64, the A chunk double-buffered in LDS with register-staged prefetch (row = 4 k-steps x {hi, lo} x 2 lane
// groups x 16 B = 256 B + 16 B pad: conflict-free ds_read_b128), B fragments straight from L2 one k-step ahead.  Every A
// fragment feeds 4 MFMAs and every B fragment 4 (the 3x3 conv kernels: 3 per A fragment).  Workgroup ids walk the (row
// tile, column block) grid in contiguous ranges per XCD, column blocks of a row tile next to each other: the A rows are
// fetched from HBM once and re-read from that XCD's L2.
//
// Epilogue = EPI_REL of conv_mfma.h: r = X * (acc + U); out0 = r; out1 = r / stab(Zdiv) (+ max|out1| per map).
#include "conv_launch.h"
#include "conv_f16x3.h"

namespace lrpx {

constexpr int DH_BN = 128;
#ifndef LRPXB_SCHED
#define LRPXB_SCHED 1       // 1: fences around the A-fragment prefetch and the k-steps; 0: the compiler's own schedule; 2: commit spread over the k-steps
#endif
#ifndef LRPXB_EXP
#define LRPXB_EXP 0       // timing experiments (wrong results): 1 no epilogue, 2 no MFMAs, 4 no staging loads / commits, 8 no B loads
#endif
// WM wave rows x 2 wave columns; a wave owns 64 rows x 64 columns.  WM = 2: 128-row tiles, K chunks of 64, two workgroups per CU.
// WM = 4 (many rows): 256-row tiles, 8 waves, K chunks of 32 (the two A buffers stay at 74 KB), one workgroup per CU - every B
// fragment streamed from L2 now serves 256 rows: the 62 720 x 512 x 512 rule moved 1.0 GB of weights from L2 to the CUs with
// 128-row tiles (1960 workgroups x 512 KB), the dominant term of its 186 us.
template <int WM>
struct DenseCfg {
    static constexpr int NT = WM * 128, BM = WM * 64, KC = WM == 2 ? 64 : 32, KS = KC / 16, SEGS = KC / 4;
    static constexpr int RP = NT / SEGS, NU = BM / RP;              // rows per staging pass, float4 items per thread and chunk
    static constexpr int ROWB = KS * 64 + 16;                       // LDS bytes per A row: [k-step][hi | lo][lane group 2][16 B] + pad
    static constexpr int BUF = BM * ROWB, LDS = 2 * BUF;
    static constexpr int CPI = 4 / KS;                              // chunks per loop iteration (4 k-steps: static B ring indices)
};

template <int EPI, int WM>      // EPI_REL, or EPI_PLAIN: out0 = acc + bias (optional ReLU) - the (T,V) scores of a trace
__global__ __launch_bounds__(WM * 128, WM == 2 ? 2 : 1) void dense_f16x3_kernel(ConvArgs a, int m_tiles, int n_blocks) {
    using C = DenseCfg<WM>;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // contiguous range of the (row tile, column block) grid per XCD (workgroup ids go round-robin over the 8 XCDs)
    const long total = (long)m_tiles * n_blocks;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const long t0 = (xcd * total) >> 3, t1 = ((xcd + 1) * total) >> 3;
    const long lin = t0 + idx;
    if (lin >= t1) return;
    const int mtile = (int)(lin / n_blocks), nblk = (int)(lin - (long)mtile * n_blocks);
    const long M = (long)a.n_maps * a.pix_per_map;
    const long row0 = (long)mtile * C::BM;
    const int K = a.cin, nchunk = K / C::KC;
    const unsigned P = (unsigned)a.pix_per_map;
    const unsigned* __restrict__ in_amax = a.in_amax;

    // ---- staging: thread -> NU items (row = tid / SEGS + RP u, 16-byte segment tid % SEGS of the chunk row)
    const int s_row = tid / C::SEGS, s_seg = tid % C::SEGS;
    float ssc[C::NU];
    long srow[C::NU];
#pragma unroll
    for (int u = 0; u < C::NU; ++u) {
        const long r = row0 + s_row + C::RP * u;
        const long rc = r < M ? r : M - 1;               // rows past the end re-read the last row (results dropped)
        srow[u] = rc * K;
        ssc[u] = exp2i(f16_scale_exp(in_amax[(unsigned)rc / P]));
    }
    // LDS offset of the item inside a row: k-step = seg / 4, lane group = (seg / 2) & 1, half = seg & 1 (8 bytes)
    const int s_off = (s_seg >> 2) * 64 + ((s_seg >> 1) & 1) * 16 + (s_seg & 1) * 8;
    f32x4 sv[C::NU];
    const float* __restrict__ A = a.in;
#define DH_ISSUE(CHUNK) if constexpr (!(LRPXB_EXP & 4)) { _Pragma("unroll") for (int u = 0; u < C::NU; ++u) sv[u] = *reinterpret_cast<const f32x4*>(A + srow[u] + (CHUNK) * C::KC + s_seg * 4); }
#define DH_COMMIT(BUFI)                                                                                    \
    _Pragma("unroll") for (int u = 0; u < C::NU; ++u) {                                                    \
        unsigned h0_, h1_, l0_, l1_;                                                                       \
        f32x2_ f0_, f1_;                                                                                   \
        split2_pk(f32x2_{sv[u][0], sv[u][1]} * f32x2_{ssc[u], ssc[u]}, h0_, l0_, f0_);                     \
        split2_pk(f32x2_{sv[u][2], sv[u][3]} * f32x2_{ssc[u], ssc[u]}, h1_, l1_, f1_);                     \
        char* d_ = ldsb + (BUFI) * C::BUF + (s_row + C::RP * u) * C::ROWB + s_off;                         \
        *reinterpret_cast<u32x2_*>(d_) = u32x2_{h0_, h1_};                                                 \
        *reinterpret_cast<u32x2_*>(d_ + 32) = u32x2_{l0_, l1_};                                            \
    }
#define DH_COMMIT_PART(BUFI, U0, NUM)                                                                     \
    _Pragma("unroll") for (int u = (U0); u < (U0) + (NUM); ++u) {                                          \
        unsigned h0_, h1_, l0_, l1_;                                                                       \
        f32x2_ f0_, f1_;                                                                                   \
        split2_pk(f32x2_{sv[u][0], sv[u][1]} * f32x2_{ssc[u], ssc[u]}, h0_, l0_, f0_);                     \
        split2_pk(f32x2_{sv[u][2], sv[u][3]} * f32x2_{ssc[u], ssc[u]}, h1_, l1_, f1_);                     \
        char* d_ = ldsb + (BUFI) * C::BUF + (s_row + C::RP * u) * C::ROWB + s_off;                         \
        *reinterpret_cast<u32x2_*>(d_) = u32x2_{h0_, h1_};                                                 \
        *reinterpret_cast<u32x2_*>(d_ + 32) = u32x2_{l0_, l1_};                                            \
    }
    DH_ISSUE(0)

    // ---- B fragments: [ocb][chunk16][plane 2][lane 64][16 B]; a wave without a column block of its own multiplies the
    // last valid one again and drops the result (one code path: no branch around memory instructions in the K loop)
    const int ocb0 = nblk * 4 + wn * 2;
    const int ocb_last = (a.n_oc - 1) / 32;
    const int nks = K / 16;
    const u32x4_* wp0 = reinterpret_cast<const u32x4_*>(a.wp + F16X3_HEADER_FLOATS) + (long)min(ocb0, ocb_last) * nks * 128 + lane;
    const u32x4_* wp1 = reinterpret_cast<const u32x4_*>(a.wp + F16X3_HEADER_FLOATS) + (long)min(ocb0 + 1, ocb_last) * nks * 128 + lane;
    const float inv_w = a.wp[0];
    constexpr int DH_NBQ = 4;        // B queue: k-step s + 3 is loaded while k-step s multiplies (12 MFMAs = 384 matrix cycles per
    u32x4_ bq[DH_NBQ][4];            // k-step against an L2 round trip of ~1000: one step ahead left the pipe waiting - 278 us)
    auto load_b = [&](const int ks, u32x4_ (&b)[4]) {        // [tile 0 hi, tile 0 lo, tile 1 hi, tile 1 lo]
        const int k = min(ks, nks - 1);
        b[0] = wp0[(long)k * 128]; b[1] = wp0[(long)k * 128 + 64];
        b[2] = wp1[(long)k * 128]; b[3] = wp1[(long)k * 128 + 64];
    };
#pragma unroll
    for (int i = 0; i < DH_NBQ - 1; ++i) load_b(i, bq[i]);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    DH_COMMIT(0)
    __syncthreads();
    const int a_off = (wm * 64 + li) * C::ROWB + lh * 16;
    for (int c0 = 0; c0 < nchunk; c0 += C::CPI) {           // (nchunk % CPI == 0: K % 64 == 0, host-checked)
#pragma unroll
        for (int cc = 0; cc < C::CPI; ++cc) {
            const int chunk = c0 + cc;
            {
                const int cn = min(chunk + 1, nchunk - 1);     // (past the last chunk: re-read it, nobody commits it)
                DH_ISSUE(cn)
                // keep the loads HERE: left alone the scheduler sinks them to their use (the commit at the end of the chunk)
                // and the whole global-load latency sits between the matrix phases of two chunks
                __builtin_amdgcn_sched_barrier(0);
            }
            const int bufi = C::CPI == 1 ? (chunk & 1) : cc;   // (two chunks per iteration: the buffer index is the position)
            const char* abuf = ldsb + bufi * C::BUF + a_off;
            // A fragments one k-step ahead of their MFMAs (two register sets; the compiler's own schedule reads a fragment,
            // waits for it - lgkmcnt(0) - and multiplies, one LDS round trip per pair of MFMAs)
            f16x8 af[2][4];
            auto read_a = [&](const int s_, f16x8 (&f)[4]) {
                f[0] = *reinterpret_cast<const f16x8*>(abuf + s_ * 64);
                f[1] = *reinterpret_cast<const f16x8*>(abuf + s_ * 64 + 32);
                f[2] = *reinterpret_cast<const f16x8*>(abuf + 32 * C::ROWB + s_ * 64);
                f[3] = *reinterpret_cast<const f16x8*>(abuf + 32 * C::ROWB + s_ * 64 + 32);
            };
            read_a(0, af[0]);
            if constexpr (LRPXB_SCHED == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < C::KS; ++s) {
                const int q = cc * C::KS + s;                  // position in the iteration's 4 k-steps: static ring indices
                if constexpr (!(LRPXB_EXP & 8)) load_b(chunk * C::KS + s + DH_NBQ - 1, bq[(q + DH_NBQ - 1) % DH_NBQ]);
                if (s + 1 < C::KS) read_a(s + 1, af[(s + 1) & 1]);
                if constexpr (LRPXB_SCHED == 1) __builtin_amdgcn_sched_barrier(0);
                const f16x8 a0h = af[s & 1][0], a0l = af[s & 1][1], a1h = af[s & 1][2], a1l = af[s & 1][3];
                const u32x4_(&b)[4] = bq[q % DH_NBQ];
                const f16x8 b0h = __builtin_bit_cast(f16x8, b[0]), b0l = __builtin_bit_cast(f16x8, b[1]);
                const f16x8 b1h = __builtin_bit_cast(f16x8, b[2]), b1l = __builtin_bit_cast(f16x8, b[3]);
                // small terms first
                if constexpr (LRPXB_EXP & 2) { acc[0][0][0] += a0h[0] + a1l[1] + b0l[0] + b1h[2] + a0l[0] + a1h[0] + b0h[0] + b1l[0]; continue; }
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, b0h, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, b1h, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b0h, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b1h, acc[1][1], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b0l, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b1l, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b0l, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b1l, acc[1][1], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b0h, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b1h, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b0h, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b1h, acc[1][1], 0, 0, 0);
                if constexpr (LRPXB_SCHED == 1) __builtin_amdgcn_sched_barrier(0);
                if constexpr (LRPXB_SCHED == 2 && !(LRPXB_EXP & 4)) {
                    // a quarter (half) of the next chunk's commit behind every k-step: conversions and LDS writes in the shadow of
                    // the 12 MFMAs just issued (the last chunk commits into the idle buffer: harmless, no branch)
                    constexpr int PER = C::NU / C::KS;
                    DH_COMMIT_PART(C::CPI == 1 ? ((chunk + 1) & 1) : (1 - cc), s * PER, PER)
                }
            }
            if constexpr (LRPXB_SCHED != 2) { if (chunk + 1 < nchunk && !(LRPXB_EXP & 4)) { DH_COMMIT(C::CPI == 1 ? ((chunk + 1) & 1) : (1 - cc)) } }
            __syncthreads();
        }
    }
#undef DH_ISSUE
#undef DH_COMMIT
#undef DH_COMMIT_PART

    if constexpr (LRPXB_EXP & 1) { if (acc[0][0][0] + acc[0][1][1] + acc[1][0][2] + acc[1][1][3] == 12345.f) a.out0[0] = 1.f; return; }
    // ---- epilogue (EPI_REL of conv_mfma.h): element e of tile (i, j): row = row0 + wm*64 + 32 i + (e&3) + 8 (e>>2) + 4 lh,
    // column = 32 (ocb0 + j) + li
    const float* __restrict__ X = a.X;
    const float* __restrict__ Uu = a.U;
    const float* __restrict__ Zd = a.Zdiv;
    const int* __restrict__ m2i = a.map2img;
    float* __restrict__ o0 = a.out0;
    float* __restrict__ o1 = a.out1;
    unsigned* __restrict__ oamax = o1 ? a.out1_amax : nullptr;
    const int ncol = a.oc_split;
    const int nmax = a.n_maps - 1;
    // (the per-map maxima of both row tiles are published behind the last store: amax_update reads the word first, and the wait for
    // that read is a wait for every store issued before it - cf. epi_rel_mul_wide of conv_f16x3.h)
    float mres[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    unsigned nres[2] = {0u, 0u};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        // max|out1| per map: a 32-row accumulator tile holds at most ONE map boundary (P >= 32, host-checked): m0 for the
        // rows in the map of the tile's first row, m1 for the rows past the boundary
        const long rt = row0 + wm * 64 + 32 * i;
        const unsigned nt0 = (unsigned)(rt < M ? rt : M - 1) / P;
        float m0 = 0.f, m1 = 0.f;
        // Three phases so that the loads of a phase are issued back to back (one dependent chain per element - row -> map ->
        // image -> multiplicand - made the epilogue the longest part of the kernel: 278 us for 62 720 x 512 x 512):
        // (1) per row: map, image, operand scale; (2) all multiplicands / addends / denominators; (3) arithmetic and stores.
        unsigned nn[16], xb[16];
        float sc[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const long r = rt + (e & 3) + 8 * (e >> 2) + 4 * lh;
            const long rc = r < M ? r : M - 1;
            const unsigned n = (unsigned)rc / P, p = (unsigned)rc - n * P;
            nn[e] = n;
            const unsigned img = EPI == EPI_PLAIN ? n : (unsigned)m2i[n];      // (REL: map2img is required - no branch around the load)
            xb[e] = (img * P + p) * (unsigned)ncol;               // (< 2^31: host-checked)
            sc[e] = exp2i(-f16_scale_exp(in_amax[n])) * inv_w;
        }
        if constexpr (EPI == EPI_PLAIN) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int oc = (ocb0 + j) * 32 + li;
                if (oc >= ncol) continue;
                const float bv = a.bias ? a.bias[oc] : 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const long r = rt + (e & 3) + 8 * (e >> 2) + 4 * lh;
                    if (r >= M) continue;
                    float v = acc[i][j][e] * sc[e] + bv;
                    if (a.relu) v = v > 0.f ? v : 0.f;
                    o0[r * ncol + oc] = v;
                }
            }
            continue;
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int oc = (ocb0 + j) * 32 + li;
            if (oc >= ncol) continue;
            float xv[16], uv[16], zv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                xv[e] = X[xb[e] + oc];
                uv[e] = Uu ? Uu[(long)nn[e] * ncol + oc] : 0.f;
                zv[e] = (o1 && Zd) ? Zd[xb[e] + oc] : 1.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long r = rt + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (r >= M) continue;
                const float rel = xv[e] * (acc[i][j][e] * sc[e] + uv[e]);
                if (o0) o0[r * ncol + oc] = rel;
                if (o1) {
                    float z = zv[e];
                    z = (a.stab == STAB_SAFE) ? stab_safe(z) : ((a.stab == STAB_EPS) ? stab_eps(z) : z);
                    const float sv_ = fast_div(rel, z);
                    o1[r * ncol + oc] = sv_;
                    if (nn[e] == nt0) m0 = fmaxf(m0, fabsf(sv_)); else m1 = fmaxf(m1, fabsf(sv_));
                }
            }
        }
        mres[i][0] = m0; mres[i][1] = m1; nres[i] = nt0;
    }
    if (oamax) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const long rt = row0 + wm * 64 + 32 * i;
            const float m0 = wave_max(mres[i][0]), m1 = wave_max(mres[i][1]);
            if (lane == 0 && rt < M) {
                amax_update(&oamax[nres[i]], m0);
                if ((int)nres[i] + 1 <= nmax) amax_update(&oamax[nres[i] + 1], m1);
            }
        }
    }
}

// ---- many rows, wide column blocks --------------------------------------------------------------------------------------
// 128 rows x 256 columns per workgroup, the four waves SIDE BY SIDE: a wave owns all 128 rows x 64 columns (4 x 2 accumulator
// tiles, 24 MFMAs per k-step).  Against the 2 x 2 arrangement above: a B fragment streamed from L2 is fetched by ONE wave of the
// workgroup and feeds 4 row tiles (2 x 2: fetched by two waves, 2 row tiles each), and the A rows are fetched, converted and
// written to LDS once per 256 columns instead of once per 128 - 24 KB from L2 per 768 matrix cycles of a SIMD instead of per 384
// (which is the 64 bytes per clock a CU can take in: the 2 x 2 kernel's K loop ran at the speed of its loads, not of its MFMAs).
// Same products in the same order per accumulator as dense_f16x3_kernel: bit-identical results.
// The per-row constants of the epilogue (map, multiplicand row, result scale) are computed ONCE by the first 128 threads at the
// start and parked in LDS: the epilogue's loads no longer hang on a chain row -> map -> image -> operand maximum.
#ifndef LRPXB_KC
#define LRPXB_KC 32        // K chunk of the 128 x 256 kernel (one barrier and one commit per chunk)
#endif
constexpr int DN_BM = 128, DN_BN = 256, DN_KC = LRPXB_KC, DN_KS = DN_KC / 16;
// LDS row of the A chunk: per k-step [hi | lo] (f16x3) or [p0 | p1 | p2] (B6: exact bf16 split) x 2 lane groups x 16 B, + 16 B pad (9 / 13
// 16-byte units per row: odd, conflict-free ds_read_b128)
template <bool B6> struct DnCfg {
    static constexpr int PL = B6 ? 3 : 2, ROWB = DN_KS * 32 * PL + 16, BUF = DN_BM * ROWB, LDS = 2 * BUF + DN_BM * 16;
};
#ifdef LRPX_STAMP
static __device__ unsigned long long g_stamp_dn[16384 * 10];      // per wave: start, loop start, loop end, stores issued, stores drained, sum issue..MFMAs, sum commit, sum barrier, HW_ID, XCC_ID
#endif
#ifndef LRPXB_SCR_PITCH
#define LRPXB_SCR_PITCH 32   // epilogue scratch of the 128 x 256 kernel, floats per row: 32 = no padding - the dword writes of a half wave and the
                            // 16-byte reads of two rows each cover 32 / 64 distinct banks (36: the reads of rows r, r + 1 overlap in 4 banks)
#endif
#ifndef LRPXB_SHADOW
#define LRPXB_SHADOW 1     // the 128 x 256 kernel converts / commits the next chunk between the MFMAs of the current one (0: a commit phase per chunk)
#endif
#ifndef LRPXB_NBQ
#define LRPXB_NBQ 2        // B ring of the 128 x 256 kernel: k-steps in registers (2: one ahead; 3: two ahead - measured +-0, 16 registers more)
#endif
// B6 (round 6): the same tile on the bf16 matrix cores with EXACT operands - a = a0 + a1 + a2 (three bf16 parts, split while staged; weights
// from lrpx_pack_weights_bf16x3, taps = 1), the six products with i + j <= 2 on v_mfma_f32_32x32x16_bf16, small terms first
// (conv_bf16x6.h): no operand scales, no in_amax, fp32 range.  This is what the decoders' (word, pixel) rules run on in the default
// conv mode 1 (nothing on the path narrower than fp32); per (row tile, k-step) one step of 12 MFMAs (3 A x 6 B fragments).
template <int EPI, bool HAS_U, bool HAS_O1, int RT, bool B6 = false>      // REL: with the addend U; writing out1 = r / stab(Zdiv) (+ its per-map maxima) instead of
                                                         // out0 = r; RT row tiles per wave: 4 (128-row tiles) or 3 (96 rows: the launcher picks the
                                                         // tile height that fills the 512 resident slots in fewer (rounds x rows))
__global__ __launch_bounds__(256, 2) void dense_f16x3_n256_kernel(ConvArgs a, int m_tiles, int n_blocks) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    constexpr int DN_ROWB = DnCfg<B6>::ROWB, DN_BUF = DnCfg<B6>::BUF, PL = DnCfg<B6>::PL;
    unsigned* ri_n = reinterpret_cast<unsigned*>(ldsb + 2 * DN_BUF);        // [128] map of the row
    unsigned* ri_xb = ri_n + DN_BM;                                         // [128] offset of its multiplicand row
    float* ri_sc = reinterpret_cast<float*>(ri_n + 2 * DN_BM);              // [128] result scale 2^-kA 2^-kW
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    LRPXH_T(ts0);
#ifdef LRPX_STAMP
    unsigned long long s_mfma = 0, s_commit = 0, s_barrier = 0;
#endif
    const long total = (long)m_tiles * n_blocks;
    const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    const long t0 = (xcd * total) >> 3, t1 = ((xcd + 1) * total) >> 3;
    const long lin = t0 + idx;
    if (lin >= t1) return;
    const int mtile = (int)(lin / n_blocks), nblk = (int)(lin - (long)mtile * n_blocks);
    const long M = (long)a.n_maps * a.pix_per_map;
    constexpr int BM = 32 * RT;                          // rows of the workgroup tile
    const long row0 = (long)mtile * BM;
    const int K = a.cin, nchunk = K / DN_KC;
    const unsigned P = (unsigned)a.pix_per_map;
    const unsigned* __restrict__ in_amax = a.in_amax;
    const int ncol = a.oc_split;
    const float inv_w = B6 ? 1.f : a.wp[0];

    if (tid < BM) {
        const long r = row0 + tid;
        const unsigned rc = (unsigned)(r < M ? r : M - 1);
        const unsigned n = rc / P, p = rc - n * P;
        const unsigned img = EPI == EPI_PLAIN ? n : (unsigned)a.map2img[n];          // (REL: map2img is required)
        float sc = 1.f;
        if constexpr (!B6) sc = exp2i(-f16_scale_exp(in_amax[n])) * inv_w;
        ri_n[tid] = n; ri_xb[tid] = (img * P + p) * (unsigned)ncol; ri_sc[tid] = sc;     // (< 2^31: host-checked)
    }

    // ---- staging: thread -> NU items (row = tid / SEGS + RP u, 16-byte segment tid % SEGS of the chunk row)
    constexpr int SEGS = DN_KC / 4, RP = 256 / SEGS, NU = BM / RP;
    const int s_row = tid / SEGS, s_seg = tid % SEGS;
    float ssc[NU];
    unsigned soff[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const long r = row0 + s_row + RP * u;
        const unsigned rc = (unsigned)(r < M ? r : M - 1);        // rows past the end re-read the last row (results dropped)
        soff[u] = rc * (unsigned)K + s_seg * 4;                    // (M * K < 2^31: host-checked)
        if constexpr (!B6) ssc[u] = exp2i(f16_scale_exp(in_amax[rc / P]));
        else ssc[u] = 1.f;
    }
    const int s_off = (s_seg >> 2) * (32 * PL) + ((s_seg >> 1) & 1) * 16 + (s_seg & 1) * 8;
    constexpr bool SHADOW = LRPXB_SHADOW != 0;      // the next chunk's conversion + LDS writes between the MFMAs of this one (loads two chunks ahead)
    f32x4 sv[SHADOW ? 2 : 1][NU];
    const float* __restrict__ A = a.in;
    auto issue = [&](const int chunk, const int set) {
        const float* __restrict__ Ac = A + chunk * DN_KC;
#pragma unroll
        for (int u = 0; u < NU; ++u) sv[set][u] = *reinterpret_cast<const f32x4*>(Ac + soff[u]);
    };
    auto commit_item = [&](const int bufi, const int set, const int u) {
        if constexpr (B6) {
            unsigned pa_[3], pb_[3];
            split3_pk(f32x2_{sv[set][u][0], sv[set][u][1]}, pa_[0], pa_[1], pa_[2]);
            split3_pk(f32x2_{sv[set][u][2], sv[set][u][3]}, pb_[0], pb_[1], pb_[2]);
            char* d_ = ldsb + bufi * DN_BUF + (s_row + RP * u) * DN_ROWB + s_off;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x2_*>(d_ + 32 * pl) = u32x2_{pa_[pl], pb_[pl]};
            return;
        }
        unsigned h0_, h1_, l0_, l1_;
        f32x2_ f0_, f1_;
        split2_pk(f32x2_{sv[set][u][0], sv[set][u][1]} * f32x2_{ssc[u], ssc[u]}, h0_, l0_, f0_);
        split2_pk(f32x2_{sv[set][u][2], sv[set][u][3]} * f32x2_{ssc[u], ssc[u]}, h1_, l1_, f1_);
        char* d_ = ldsb + bufi * DN_BUF + (s_row + RP * u) * DN_ROWB + s_off;
        *reinterpret_cast<u32x2_*>(d_) = u32x2_{h0_, h1_};
        *reinterpret_cast<u32x2_*>(d_ + 32) = u32x2_{l0_, l1_};
    };
    auto commit = [&](const int bufi, const int set) {
#pragma unroll
        for (int u = 0; u < NU; ++u) commit_item(bufi, set, u);
    };
    issue(0, 0);

    // ---- B fragments of the wave's two column tiles: [ocb][k-step][plane 2][lane 64][16 B], one k-step ahead in registers
    const int ocb0 = nblk * 8 + wave * 2;
    const int ocb_last = (a.n_oc - 1) / 32;
    const int nks = K / 16;
    const u32x4_* wp0 = reinterpret_cast<const u32x4_*>(a.wp + (B6 ? 0 : F16X3_HEADER_FLOATS)) + (long)min(ocb0, ocb_last) * nks * (64 * PL) + lane;
    const u32x4_* wp1 = reinterpret_cast<const u32x4_*>(a.wp + (B6 ? 0 : F16X3_HEADER_FLOATS)) + (long)min(ocb0 + 1, ocb_last) * nks * (64 * PL) + lane;
    // B ring: k-step s + NBQ - 1 is loaded while k-step s multiplies (24 MFMAs = 768 matrix cycles per k-step)
    constexpr int NBQ = LRPXB_NBQ;
    u32x4_ bq[NBQ][2 * PL];          // [tile 0 hi, lo, tile 1 hi, lo]; B6: [tile 0 p0, p1, p2, tile 1 p0, p1, p2]
    auto load_b = [&](const int ks, u32x4_ (&b)[2 * PL]) {
        const int k = min(ks, nks - 1);
#pragma unroll
        for (int pl = 0; pl < PL; ++pl) { b[pl] = wp0[(long)k * (64 * PL) + 64 * pl]; b[PL + pl] = wp1[(long)k * (64 * PL) + 64 * pl]; }
    };
#pragma unroll
    for (int i = 0; i < NBQ - 1; ++i) load_b(i, bq[i]);

    f32x16 acc[RT][2];
#pragma unroll
    for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    commit(0, 0);
    if constexpr (SHADOW) issue(min(1, nchunk - 1), 1);
    __syncthreads();
    LRPXH_T(ts1);
    const int a_off = li * DN_ROWB + lh * 16;
    static_assert(!SHADOW || (NU <= 2 * DN_KS && NBQ == 2), "at most one staging item per (k-step, pair) step; two chunks per loop trip");
    static_assert(RT == 4 || RT == 3, "two pairs of row tiles, or a pair and a single one");
    // (NBQ chunks per loop iteration: the ring positions are compile-time constants - 2 NBQ k-steps, a multiple of the ring)
    if constexpr (B6) {
        static_assert(SHADOW && NBQ == 2 && NU <= RT * DN_KS, "B6: shadow commits, a two-deep B ring");
        for (int c0 = 0; c0 < nchunk; c0 += 2) {
#pragma unroll
            for (int cc = 0; cc < 2; ++cc) {
                const int chunk = c0 + cc;
                if (chunk >= nchunk) break;                      // (uniform)
                issue(min(chunk + 2, nchunk - 1), cc & 1);
                __builtin_amdgcn_sched_barrier(0);
                const char* abuf = ldsb + (chunk & 1) * DN_BUF + a_off;
                bf16x8 af[2][3];                                  // the three planes of ONE row tile, a step ahead of its 12 MFMAs
                auto read_a = [&](const int t, bf16x8 (&f)[3]) {
                    const char* q = abuf + (32 * (t % RT)) * DN_ROWB + (t / RT) * 96;
#pragma unroll
                    for (int pl = 0; pl < 3; ++pl) f[pl] = *reinterpret_cast<const bf16x8*>(q + 32 * pl);
                };
                read_a(0, af[0]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < RT * DN_KS; ++t) {          // t = RT * k-step + row tile
                    const int s = t / RT, i = t % RT;
                    const int q = cc * DN_KS + s;
                    if (i == 0) load_b(chunk * DN_KS + s + 1, bq[(q + 1) % 2]);
                    if (t + 1 < RT * DN_KS) read_a(t + 1, af[(t + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (t < NU) commit_item((chunk + 1) & 1, (cc + 1) & 1, t);      // (shadow commit of the next chunk, as in the f16x3 loop)
                    const bf16x8 a0 = af[t & 1][0], a1 = af[t & 1][1], a2 = af[t & 1][2];
                    const u32x4_(&b)[2 * PL] = bq[q % 2];
                    f32x16 &c0_ = acc[i][0], &c1_ = acc[i][1];
#define LRPXB6_MM(A_, PLB_)                                                                                              \
    c0_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, __builtin_bit_cast(bf16x8, b[PLB_]), c0_, 0, 0, 0);               \
    c1_ = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A_, __builtin_bit_cast(bf16x8, b[PL + PLB_]), c1_, 0, 0, 0)
                    LRPXB6_MM(a2, 0); LRPXB6_MM(a1, 1); LRPXB6_MM(a0, 2);      // small terms first (conv_bf16x6.h)
                    LRPXB6_MM(a1, 0); LRPXB6_MM(a0, 1);
                    LRPXB6_MM(a0, 0);
#undef LRPXB6_MM
                    if (t < NU) {
#pragma unroll
                        for (int g = 0; g < 10; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);      // three vector instructions of the split
                        }
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);          // the LDS writes
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                __syncthreads();
            }
        }
    } else
    for (int c0 = 0; c0 < nchunk; c0 += NBQ) {
#pragma unroll
        for (int cc = 0; cc < NBQ; ++cc) {
            const int chunk = c0 + cc;
            if (chunk >= nchunk) break;                      // (uniform)
            LRPXH_T(ta);
            // SHADOW: chunk + 2 into the register set chunk's data left at the end of the previous trip (set index = cc: c0 is even)
            if constexpr (SHADOW) issue(min(chunk + 2, nchunk - 1), cc & 1);
            else issue(min(chunk + 1, nchunk - 1), 0);       // (past the last chunk: re-read it, nobody commits it)
            __builtin_amdgcn_sched_barrier(0);               // keep the loads here (cf. dense_f16x3_kernel)
            const char* abuf = ldsb + (chunk & 1) * DN_BUF + a_off;
            // A fragments of a PAIR of row tiles one step ahead of their 12 MFMAs: [tile 2p hi, lo, tile 2p + 1 hi, lo]
            f16x8 af[2][4];
            auto read_pair = [&](const int t, f16x8 (&f)[4]) {
                const char* q = abuf + (64 * (t & 1)) * DN_ROWB + (t >> 1) * 64;
                f[0] = *reinterpret_cast<const f16x8*>(q);
                f[1] = *reinterpret_cast<const f16x8*>(q + 32);
                if (RT == 4 || (t & 1) == 0) {               // (RT = 3: the second step of a k-step holds row tile 2 alone)
                    f[2] = *reinterpret_cast<const f16x8*>(q + 32 * DN_ROWB);
                    f[3] = *reinterpret_cast<const f16x8*>(q + 32 * DN_ROWB + 32);
                }
            };
            read_pair(0, af[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 2 * DN_KS; ++t) {           // t = 2 * k-step + pair
                const int s = t >> 1, pr = t & 1;
                const int q = cc * DN_KS + s;                // position of the k-step in the iteration: static ring indices
                if (pr == 0) load_b(chunk * DN_KS + s + NBQ - 1, bq[(q + NBQ - 1) % NBQ]);
                if (t + 1 < 2 * DN_KS) read_pair(t + 1, af[(t + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
                // SHADOW: item t of the NEXT chunk (loaded a chunk ago) is scaled, split and written to the other buffer under these
                // 12 MFMAs - two vector instructions behind each (the vector issue is free for most of an MFMA's 32 cycles)
                // (no branch around it - the scheduler mixes within a basic block only: the last chunk writes a copy of itself into the idle buffer)
                if constexpr (SHADOW) { if (t < NU) commit_item((chunk + 1) & 1, (cc + 1) & 1, t); }
                const f16x8 a0h = af[t & 1][0], a0l = af[t & 1][1], a1h = af[t & 1][2], a1l = af[t & 1][3];
                const u32x4_(&b)[2 * PL] = bq[q % NBQ];
                const f16x8 b0h = __builtin_bit_cast(f16x8, b[0]), b0l = __builtin_bit_cast(f16x8, b[1]);
                const f16x8 b1h = __builtin_bit_cast(f16x8, b[2]), b1l = __builtin_bit_cast(f16x8, b[3]);
                const bool two = RT == 4 || pr == 0;           // (compile-time per unrolled step)
                f32x16 &c00 = acc[2 * pr][0], &c01 = acc[2 * pr][1], &c10 = acc[two ? 2 * pr + 1 : 0][0], &c11 = acc[two ? 2 * pr + 1 : 0][1];
                // small terms first
                c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, b0h, c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, b1h, c01, 0, 0, 0);
                if (two) {
                    c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b0h, c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, b1h, c11, 0, 0, 0);
                }
                c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b0l, c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b1l, c01, 0, 0, 0);
                if (two) {
                    c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b0l, c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b1l, c11, 0, 0, 0);
                }
                c00 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b0h, c00, 0, 0, 0);
                c01 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, b1h, c01, 0, 0, 0);
                if (two) {
                    c10 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b0h, c10, 0, 0, 0);
                    c11 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, b1h, c11, 0, 0, 0);
                }
                if constexpr (SHADOW) {
                    if (two) {
#pragma unroll
                        for (int g = 0; g < 10; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // one MFMA
                            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);      // two vector instructions
                        }
                    } else {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        }
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);          // the LDS writes
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            LRPXH_T(tb);
            if constexpr (!SHADOW) { if (chunk + 1 < nchunk) commit((chunk + 1) & 1, 0); }
            LRPXH_T(tc);
            __syncthreads();
#ifdef LRPX_STAMP
            { LRPXH_T(td); s_mfma += tb - ta; s_commit += tc - tb; s_barrier += td - tc; }
#endif
        }
    }
    LRPXH_T(ts2);

    // ---- epilogue (EPI_REL of conv_mfma.h / EPI_PLAIN): element e of tile (i, j): row = row0 + 32 i + (e&3) + 8 (e>>2) + 4 lh,
    // column = 32 (ocb0 + j) + li.  STRAIGHT-LINE code: which operands exist is a template parameter (a branch on `U != null`
    // around a load makes the compiler wait for every load in flight at the merge), columns past the end load the last column,
    // rows past the end the last row; only the stores are predicated.
    const float* __restrict__ X = a.X;
    const float* __restrict__ Uu = a.U;
    const float* __restrict__ Zd = a.Zdiv;
    float* __restrict__ out = HAS_O1 ? a.out1 : a.out0;
    unsigned* __restrict__ oamax = HAS_O1 ? a.out1_amax : nullptr;
    const int nmax = a.n_maps - 1;
    float mres[RT][2];
    unsigned nres[RT];
    const int stab = a.stab;
    // Units of one accumulator tile (i = u / 2, j = u % 2).  The tile goes through a per-wave LDS scratch (the A buffers are free after
    // the last barrier) into ROW-MAJOR lanes - lane l holds rows l / 8 + 8 q, columns 4 (l % 8) .. + 3 - so that the multiplicand, the
    // addend, the denominator and the result move as 16 bytes per lane: 4 + 4 memory instructions per tile instead of 16 + 16, every
    // wave instruction 8 rows x 128 bytes.  (With dword accesses the epilogue ran at 3.8 TB/s of stores when it was all the kernel
    // did - K = 64 - against 6.9 for a fill of the same bytes.)  The loads of unit u + 1 are in flight while unit u is finished.
    constexpr int SP = LRPXB_SCR_PITCH;                                        // floats per scratch row
    float* scr = reinterpret_cast<float*>(ldsb) + wave * (32 * SP);            // [32 rows][SP]
    const int tr = lane >> 3, tc = (lane & 7) * 4;
    constexpr int DEPTH = 2;
    f32x4 xr[DEPTH][4], ur[HAS_U ? DEPTH : 1][4], zr[HAS_O1 ? DEPTH : 1][4];
    auto issue_u = [&](const int u) {
        if constexpr (EPI != EPI_PLAIN) {
            const int i = u >> 1, j = u & 1;
            const unsigned oc = (unsigned)min((ocb0 + j) * 32 + tc, ncol - 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int row = 32 * i + tr + 8 * q;
                const unsigned xb = ri_xb[row];
                xr[u % DEPTH][q] = *reinterpret_cast<const f32x4*>(X + xb + oc);
                if constexpr (HAS_U) ur[u % DEPTH][q] = *reinterpret_cast<const f32x4*>(Uu + ri_n[row] * (unsigned)ncol + oc);   // (n_maps * ncol < 2^31: host-checked)
                if constexpr (HAS_O1) zr[u % DEPTH][q] = *reinterpret_cast<const f32x4*>(Zd + xb + oc);
            }
        }
    };
    auto finish_u = [&](const int u) {
        const int i = u >> 1, j = u & 1;
        const long rt = row0 + 32 * i;
        float* __restrict__ orow = out + rt * ncol;           // one scalar base per row tile + a 32-bit offset (32 rows x ncol < 2^31)
        const int mrem = (int)(M - rt < 32 ? M - rt : 32);
        const int oc = (ocb0 + j) * 32 + tc;
        const unsigned nt0 = ri_n[32 * i];
        if (j == 0) { mres[i][0] = 0.f; mres[i][1] = 0.f; nres[i] = nt0; }
        // (the previous unit's reads of this wave's scratch are complete before it is overwritten, and the reads below see the writes
        // above: LDS instructions of one wave execute in order; the fences pin that order for the COMPILER too - ADVICE r4 - and cost nothing)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int e = 0; e < 16; ++e) scr[((e & 3) + 8 * (e >> 2) + 4 * lh) * SP + li] = acc[i][j][e];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        f32x4 bv4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (EPI == EPI_PLAIN) { if (a.bias) bv4 = *reinterpret_cast<const f32x4*>(a.bias + min(oc, ncol - 4)); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int rl = tr + 8 * q;
            const bool ok = rl < mrem && oc < ncol;
            const f32x4 av = *reinterpret_cast<const f32x4*>(scr + rl * SP + tc);
            const float sc = ri_sc[32 * i + rl];
            f32x4 res;
            if constexpr (EPI == EPI_PLAIN) {
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    res[c] = __builtin_fmaf(av[c], sc, bv4[c]);
                    if (a.relu) res[c] = res[c] > 0.f ? res[c] : 0.f;
                }
            } else {
                f32x4 rel;
#pragma unroll
                for (int c = 0; c < 4; ++c)          // (the same two roundings as the dword epilogue: fma, then the product)
                    rel[c] = xr[u % DEPTH][q][c] * __builtin_fmaf(av[c], sc, HAS_U ? ur[u % DEPTH][q][c] : 0.f);
                if constexpr (!HAS_O1) {
                    res = rel;
                } else {
                    float mx = 0.f;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        float z = zr[u % DEPTH][q][c];
                        z = (stab == STAB_SAFE) ? stab_safe(z) : ((stab == STAB_EPS) ? stab_eps(z) : z);
                        res[c] = fast_div(rel[c], z);
                        mx = fmaxf(mx, fabsf(res[c]));
                    }
                    if (ok) { if (ri_n[32 * i + rl] == nt0) mres[i][0] = fmaxf(mres[i][0], mx); else mres[i][1] = fmaxf(mres[i][1], mx); }
                }
            }
            if (ok) *reinterpret_cast<f32x4*>(orow + (unsigned)(rl * ncol + oc)) = res;
        }
    };
    issue_u(0);
#pragma unroll
    for (int u = 0; u < 2 * RT; ++u) {
        if (u + 1 < 2 * RT) issue_u(u + 1);
        finish_u(u);
    }
#ifdef LRPX_STAMP
    {
        LRPXH_T(ts3);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the stores acknowledged)
        LRPXH_T(ts4);
        unsigned hwid, xccid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xccid));
        const unsigned w = blockIdx.x * 4 + wave;
        if (lane == 0 && w < 16384) {
            unsigned long long* r = g_stamp_dn + (unsigned long)w * 10;
            r[0] = ts0; r[1] = ts1; r[2] = ts2; r[3] = ts3; r[4] = ts4; r[5] = s_mfma; r[6] = s_commit; r[7] = s_barrier; r[8] = hwid; r[9] = (xccid & 15) | ((unsigned long long)blockIdx.x << 8);
        }
    }
#endif
    if constexpr (HAS_O1) {
        if (oamax) {
#pragma unroll
            for (int i = 0; i < RT; ++i) {
                const long rt = row0 + 32 * i;
                const float m0 = wave_max(mres[i][0]), m1 = wave_max(mres[i][1]);
                if (lane == 0 && rt < M) {
                    amax_update(&oamax[nres[i]], m0);
                    if ((int)nres[i] + 1 <= nmax) amax_update(&oamax[nres[i] + 1], m1);
                }
            }
        }
    }
}

// ---- few rows (the lock-step gate rules of the decoders: rows = images x words, 320 .. 1280; K = 512) -------------------
// dense_small.hip on the fp16 matrix cores: one workgroup = 32 rows x 128 columns, the whole 32 x K slab of A staged ONCE
// into LDS - scaled per ROW by 2^kA from the row's own maximum (found while staging: the slab passes through registers) and
// split a = a0 + a1 - then 3 MFMAs (32 cycles each) per 16 of K instead of 8 fp32 MFMAs (64 cycles each): the serial chain
// of one wave drops from 16 k to 3 k matrix cycles.  B fragments stream from the f16x2 pack through a register queue.
// Epilogue REL (out0 only): out = X[src(row)] * (acc + U).  K <= 1024, K % 16 == 0.
// B queue: 11 k-steps ahead (a k-step is 3 MFMAs = 96 matrix cycles, an L2 round trip ~1000: 5 ahead left the K loop bound by
// the load latency - 28 us per GEMM against 25 for the fp32 kernel it replaces)
constexpr int DS_NB = 12;
#ifndef LRPXD_EARLY
#define LRPXD_EARLY 1     // few-row kernel, K = 512: the epilogue's multiplicand / addend loads issued before the K loop (0: in the epilogue)
#endif
#ifndef LRPXD_EXP
#define LRPXD_EXP 0       // timing experiments (wrong results): 1 no multiplicand / addend loads in the epilogue, 2 no K loop, 4 no A staging, 8 no stores
#endif
// FUSE (round 5): lock-step s of the AoA decoder relevance (models/aoamodel.py:1114-1134) inside the gate rule's GEMM.  The product
// r_xh = xh * (W_g^T A) (:1125-1128) is consumed right where it is formed - the three column ranges of r_xh = [emb | glob | h] (:1129-1133):
//   emb  -> r_words[row][i] = sum over the E embedding columns: per 128-column workgroup one partial sum (wpart), added up in a fixed order
//           by rel_words_norm_parts_kernel (deterministic);
//   glob -> r_glob += r_xh[E : E + H];
//   h    -> r_h, and from it the NEXT lock-step's GEMM input A = (r_h * i tanh(g) / z~(c)) / z~(g) (:1116-1120, aoa_rel_a_kernel's
//           expression and operation order) into the other A buffer.
// r_xh itself is never stored and the separate point-wise launch between two GEMMs (aoa_rel_ca_kernel) is gone: 2 launches per
// lock-step -> 1.  E = H, both multiples of 128 (host-checked).
template <int MAXU, bool EXACT, bool FUSE = false>   // float4 items of the A slab per thread: 16 for K = 512, 32 for K = 1024 (EXACT: no tail); else guarded
__global__ __launch_bounds__(256, 2) void dense_small_f16x3_kernel(ConvArgs a, int m_tiles, int n_blocks, AoaStepFuse fz) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    __shared__ float wsum[4][32];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mtile = blockIdx.x / n_blocks, nblk = blockIdx.x % n_blocks;
    const int K = EXACT ? MAXU * 32 : a.cin;      // (EXACT: a compile-time constant - the item -> (row, segment) maps become shifts)
    const int nks = K / 16, k4 = K / 4;
    const int pitch = nks * 64 + 16;
    unsigned* rowmax = reinterpret_cast<unsigned*>(ldsb + 32 * pitch);        // [32] float bits of max|A[row]|
    const long rows = (long)a.n_maps * a.pix_per_map;
    const long row0 = (long)mtile * 32;
    const int ocb = nblk * 4 + wave;
    const int ocb_last = (a.n_oc - 1) / 32;
    const bool wave_active = ocb <= ocb_last;
    const u32x4_* wp = reinterpret_cast<const u32x4_*>(a.wp + F16X3_HEADER_FLOATS) + (long)min(ocb, ocb_last) * nks * 128 + lane;
    u32x4_ bq[DS_NB][2];
#pragma unroll
    for (int i = 0; i < DS_NB - 1; ++i) { bq[i][0] = wp[(long)min(i, nks - 1) * 128]; bq[i][1] = wp[(long)min(i, nks - 1) * 128 + 64]; }
    // EARLY (K = 512, where the registers allow it): the epilogue's operands do not depend on the product - the row -> source-row
    // lookups are issued HERE, the multiplicands and addends right after the slab has gone to LDS, and both arrive under the K loop:
    // three dependent round trips (~3.5 of the kernel's 17 us) leave the epilogue.  Same arithmetic, bit-identical.
    constexpr bool EARLY = (LRPXD_EARLY != 0) && MAXU == 16;
    const int* __restrict__ m2i = a.map2img;
    const unsigned P = (unsigned)a.pix_per_map;
    const unsigned last = (unsigned)(rows - 1);
    int img_e[16];
    if constexpr (EARLY) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned row = min((unsigned)row0 + (unsigned)((e & 3) + 8 * (e >> 2) + 4 * lh), last);
            const unsigned n = P == 1u ? row : row / P;
            img_e[e] = (m2i && !(LRPXD_EXP & 1)) ? m2i[n] : (int)n;
        }
    }
    if (tid < 32) rowmax[tid] = 0u;
    __syncthreads();
    // pass 1: the slab through registers, row maxima into LDS.  BRANCH-FREE: every load is issued unconditionally (rows past the
    // end re-read the last row and are zeroed by a select) - with `if (row < rows) load` the compiler waits for each of the 16
    // loads before it issues the next (s_waitcnt vmcnt(0) at every control-flow merge): 7 of the kernel's 25 us
    f32x4 sv[MAXU];
    const int nu = EXACT ? MAXU : (32 * k4 + 255) / 256;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        const int it = EXACT ? tid + 256 * u : min(tid + 256 * u, 32 * k4 - 1);
        const int r = it / k4, c4 = it - r * k4;
        const long rr = min(row0 + r, rows - 1);
        sv[u] = *reinterpret_cast<const f32x4*>(a.in + rr * K + c4 * 4);
        if ((LRPXD_EXP & 4) || row0 + r >= rows || (!EXACT && (u >= nu || tid + 256 * u >= 32 * k4))) sv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        if (EXACT || u < nu) {
            const int it = tid + 256 * u;
            const int r = min(it / k4, 31);
            float m = fmaxf(fmaxf(fabsf(sv[u][0]), fabsf(sv[u][1])), fmaxf(fabsf(sv[u][2]), fabsf(sv[u][3])));
            if ((k4 & 63) == 0) {                  // a wave's 64 items lie in one row: one LDS atomic per wave
                m = wave_max(m);
                if (lane == 0) atomicMax(&rowmax[r], __builtin_bit_cast(unsigned, m));
            } else if (it < 32 * k4) {
                atomicMax(&rowmax[r], __builtin_bit_cast(unsigned, m));
            }
        }
    }
    __syncthreads();
    // pass 2: scale, split, LDS.  Row layout: [k-step][hi: lane group 0, 1 | lo: lane group 0, 1] x 16 B
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        if (EXACT || u < nu) {
            const int it = tid + 256 * u;
            if (EXACT || it < 32 * k4) {
                const int r = it / k4, c4 = it - r * k4;
                const float sc = exp2i(f16_scale_exp(rowmax[r]));
                unsigned h0_, h1_, l0_, l1_;
                f32x2_ f0_, f1_;
                split2_pk(f32x2_{sv[u][0], sv[u][1]} * f32x2_{sc, sc}, h0_, l0_, f0_);
                split2_pk(f32x2_{sv[u][2], sv[u][3]} * f32x2_{sc, sc}, h1_, l1_, f1_);
                char* d_ = ldsb + r * pitch + (c4 >> 2) * 64 + ((c4 >> 1) & 1) * 16 + (c4 & 1) * 8;
                *reinterpret_cast<u32x2_*>(d_) = u32x2_{h0_, h1_};
                *reinterpret_cast<u32x2_*>(d_ + 32) = u32x2_{l0_, l1_};
            }
        }
    }
    const int ncol = a.oc_split;
    const float* __restrict__ X = a.X;
    const float* __restrict__ Uu = a.U;
    float xv[16], uv[16];
    if constexpr (EARLY) {
        const int occ = min(ocb * 32 + li, ncol - 1);          // (a lane past the last column loads the last one and stores nothing)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned row = min((unsigned)row0 + (unsigned)((e & 3) + 8 * (e >> 2) + 4 * lh), last);
            const unsigned n = P == 1u ? row : row / P, p = P == 1u ? 0u : row - n * P;
            xv[e] = (LRPXD_EXP & 1) ? 1.f : X[((long)img_e[e] * P + p) * ncol + occ];
            uv[e] = (Uu && !(LRPXD_EXP & 1)) ? Uu[(long)n * ncol + occ] : 0.f;
        }
    }
    // FUSE: per row tmax = its word index t if the caption has that word, else -1 (row = image * T + t is active at lock-step s iff
    // s <= tmax); h columns: the coefficients of the next lock-step's A - Q1 = i tanh(g) / z~(c), DG = z~(g) at time index t - s - 1,
    // i.e. at flat index row - s - 1 of the [B][T][H] tables (lrpx_aoa_rel_coef) - issued here, they arrive under the K loop
    float q1[16], dg[16];
    unsigned m_now = 0u, m_next = 0u;                      // bit e: the lane's e-th row is active at lock-step s / s + 1
    if constexpr (FUSE) {
        const int Hh = a.cin;                              // E = H = K
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rl = (e & 3) + 8 * (e >> 2) + 4 * lh;
            const int tm = row0 + rl < rows ? fz.tmax[row0 + rl] : -1;
            m_now |= (fz.s <= tm ? 1u : 0u) << e;
            m_next |= (fz.s + 1 <= tm ? 1u : 0u) << e;
        }
        if (nblk * 128 >= 2 * Hh && fz.s + 1 < fz.T) {
            const int ch = min(ocb * 32 + li, a.oc_split - 1) - 2 * Hh;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned row = min((unsigned)row0 + (unsigned)((e & 3) + 8 * (e >> 2) + 4 * lh), last);
                const bool act = (m_next >> e) & 1u;
                const long ti = (long)(act ? (int)row - fz.s - 1 : 0) * Hh + ch;
                q1[e] = fz.q1[ti];
                dg[e] = fz.dg[ti];
                if (!act) { q1[e] = 0.f; dg[e] = 1.f; }
            }
        }
    }
    __syncthreads();
    if (!wave_active) return;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const char* ap = ldsb + li * pitch + lh * 16;
    for (int ks = 0; ks < ((LRPXD_EXP & 2) ? 0 : nks); ks += DS_NB) {
#pragma unroll
        for (int u = 0; u < DS_NB; ++u) {
            const int step = ks + u;
            const long nxt = (long)min(step + DS_NB - 1, nks - 1) * 128;
            bq[DS_NB - 1][0] = wp[nxt]; bq[DS_NB - 1][1] = wp[nxt + 64];
            if (step < nks) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(ap + step * 64);
                const f16x8 al = *reinterpret_cast<const f16x8*>(ap + step * 64 + 32);
                const f16x8 bh = __builtin_bit_cast(f16x8, bq[0][0]), bl = __builtin_bit_cast(f16x8, bq[0][1]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc, 0, 0, 0);      // small terms first
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < DS_NB - 1; ++i) { bq[i][0] = bq[i + 1][0]; bq[i][1] = bq[i + 1][1]; }
        }
    }
    const int oc = ocb * 32 + li;
    if (oc >= ncol) return;
    const float inv_w = a.wp[0];
    if constexpr (!EARLY) {
        // (branch-free as the staging: all 16 row -> source-row lookups, then all 16 multiplicands / addends, then the stores;
        // 32-bit index arithmetic - rows < 2^31 host-checked - and no division at all for the lock-step rules' pix_per_map = 1)
        long src[16];
        unsigned nn[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned row = min((unsigned)row0 + (unsigned)((e & 3) + 8 * (e >> 2) + 4 * lh), last);
            const unsigned n = P == 1u ? row : row / P, p = P == 1u ? 0u : row - n * P;
            nn[e] = n;
            const long img = (m2i && !(LRPXD_EXP & 1)) ? (long)m2i[n] : (long)n;
            src[e] = (img * P + p) * ncol + oc;
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            xv[e] = (LRPXD_EXP & 1) ? 1.f : X[src[e]];
            uv[e] = (Uu && !(LRPXD_EXP & 1)) ? Uu[(long)nn[e] * ncol + oc] : 0.f;
        }
    }
    if constexpr (FUSE) {
        const int Hh = a.cin, T = fz.T, sNow = fz.s;
        const int part = (nblk * 128) / Hh;              // 0 emb, 1 glob, 2 h (workgroup-uniform)
        float rx[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int rl = (e & 3) + 8 * (e >> 2) + 4 * lh;
            rx[e] = xv[e] * (acc[e] * (exp2i(-f16_scale_exp(rowmax[rl])) * inv_w) + uv[e]);
        }
        if (part == 0) {
            // per row the sum over this wave's 32 columns (the 32 lanes of a half-wave), then over the 4 waves in wave order
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float v = rx[e];
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
                if (li == 0) wsum[wave][(e & 3) + 8 * (e >> 2) + 4 * lh] = v;
            }
            __syncthreads();
            if (tid < 32) {
                const long row = row0 + tid;
                if (row < rows) {
                    const int t = fz.tmax[row];
                    if (sNow <= t)
                        fz.wpart[(row * T + (t - sNow)) * 4 + (nblk & 3)] = (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);
                }
            }
        } else if (part == 1) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if ((m_now >> e) & 1u) fz.r_glob[row * Hh + (oc - Hh)] += rx[e];
            }
        } else if (sNow + 1 < T) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
                if (row < rows) fz.A_next[row * Hh + (oc - 2 * Hh)] = (q1[e] * rx[e]) / dg[e];      // q1 = 0 for a row that is not active at s + 1
            }
        }
        return;
    }
    float* __restrict__ O0 = a.out0;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int rl = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const long row = row0 + rl;
        if (row >= rows) continue;
        const float v = acc[e] * (exp2i(-f16_scale_exp(rowmax[rl])) * inv_w) + uv[e];
        if (!(LRPXD_EXP & 8) || v == 12345.f) O0[row * ncol + oc] = xv[e] * v;
    }
}

int launch_dense_small_f16x3(const ConvArgs& a, hipStream_t stream) {
    const long rows = (long)a.n_maps * a.pix_per_map;
    LRPX_REQUIRE(a.cin % 16 == 0 && a.cin >= 16 && a.cin <= 1024, "dense_small_f16x3: K = %d (a multiple of 16, <= 1024)", a.cin);
    LRPX_REQUIRE(a.X && a.out0 && !a.out1 && rows > 0 && rows < 0x7fffffffL, "dense_small_f16x3: REL epilogue with x and out0 only");
    const int m_tiles = (int)ceil_div(rows, 32);
    const int n_blocks = (int)ceil_div(a.n_oc, 128);
    const int lds = 32 * ((a.cin / 16) * 64 + 16) + 128;
    constexpr int LDS_MAX = 32 * (64 * 64 + 16) + 128;
    // instantiations: K = 512 and K = 1024 exactly (no tail item, no guard in the staging loops), any other K guarded
    const int which = a.cin == 512 ? 0 : (a.cin == 1024 ? 1 : (a.cin < 512 ? 2 : 3));
    auto kern = which == 0 ? dense_small_f16x3_kernel<16, true> : (which == 1 ? dense_small_f16x3_kernel<32, true>
                : (which == 2 ? dense_small_f16x3_kernel<16, false> : dense_small_f16x3_kernel<32, false>));
    static LdsOnce once[4];
    LRPX_TRY(reserve_lds_once(once[which], kern, LDS_MAX, "dense_small_f16x3"));
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_tiles * n_blocks)), dim3(256), lds, stream, a, m_tiles, n_blocks, AoaStepFuse{});
    return check_launch("dense_small_f16x3");
}

// lock-step fz.s of the AoA relevance: the gate rule's GEMM with the step's point-wise code in its epilogue (FUSE above)
int launch_dense_small_f16x3_aoa_step(const ConvArgs& a, const AoaStepFuse& fz, hipStream_t stream) {
    const long rows = (long)a.n_maps * a.pix_per_map;
    LRPX_REQUIRE(a.cin == 512 && a.n_oc == 3 * a.cin && a.oc_split == a.n_oc && a.pix_per_map == 1 && !a.U && a.X && a.map2img,
                 "dense_small_f16x3 (AoA lock-step): built for E = H = K = 512, N = 1536, one row per map, x and map2img given");
    LRPX_REQUIRE(rows > 0 && rows < 0x7fffffffL && fz.T > 0 && rows % fz.T == 0 && fz.s >= 0 && fz.s < fz.T && fz.q1 && fz.dg && fz.tmax &&
                     fz.r_glob && fz.wpart && (fz.A_next || fz.s + 1 >= fz.T), "dense_small_f16x3 (AoA lock-step): bad step arguments");
    const int m_tiles = (int)ceil_div(rows, 32), n_blocks = a.n_oc / 128;
    const int lds = 32 * ((a.cin / 16) * 64 + 16) + 128;
    constexpr int LDS_MAX = 32 * (64 * 64 + 16) + 128;
    auto kern = dense_small_f16x3_kernel<16, true, true>;
    static LdsOnce once;
    LRPX_TRY(reserve_lds_once(once, kern, LDS_MAX, "dense_small_f16x3 (AoA lock-step)"));
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_tiles * n_blocks)), dim3(256), lds, stream, a, m_tiles, n_blocks, fz);
    return check_launch("dense_small_f16x3 (AoA lock-step)");
}

int launch_dense_f16x3(const ConvArgs& a, hipStream_t stream) {
    const long M = (long)a.n_maps * a.pix_per_map;
    LRPX_REQUIRE(a.cin % 64 == 0 && a.cin >= 64, "dense_f16x3: K = %d is not a multiple of 64", a.cin);
    const bool plain = a.epi == EPI_PLAIN;
    LRPX_REQUIRE(a.in_amax && (plain ? (a.out0 && !a.out1) : (a.X && a.map2img && (a.out0 || a.out1))),
                 "dense_f16x3: needs in_amax, an output (and x, map2img for REL)");
    LRPX_REQUIRE(M > 0 && M < 0x7fffffffL, "dense_f16x3: %ld rows out of range", M);
    LRPX_REQUIRE(plain || (long)a.n_maps * a.pix_per_map * a.oc_split < 0x7fffffffL, "dense_f16x3: too many multiplicand elements for 32-bit offsets");
    LRPX_REQUIRE(!(a.out1 && a.out1_amax) || a.pix_per_map >= 32, "dense_f16x3: out1_amax needs at least 32 rows per map");
    // 128 x 256 tiles, waves side by side: the default for many rows and at least 256 columns with ONE output (REL: out0, or out1
    // with its denominator); LRPX_DENSE_N256=0 or anything else: the 128 x 128 kernel
    const bool one_out = plain || (a.out1 ? (!a.out0 && a.Zdiv) : true);
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };      // (its epilogue moves 16 bytes per lane)
    const bool aligned = al16(a.X) && al16(a.U) && al16(a.Zdiv) && al16(a.out0) && al16(a.out1) && al16(a.bias);
    if (switches().dense_n256 && one_out && aligned && M >= 4096 && a.n_oc >= 256 && a.oc_split % 4 == 0 && a.cin % DN_KC == 0 && M * a.cin < 0x7fffffffL
        && (long)a.n_maps * a.oc_split < 0x7fffffffL && (!a.out1 || a.pix_per_map >= 32)) {
        const int n_blocks = (int)ceil_div(a.n_oc, DN_BN);
        // tile height: 128 rows, or 96 where that fills the 512 resident workgroup slots in fewer (rounds x rows) - e.g. 23 040 x 512
        // columns: 360 tiles of 128 rows leave 152 slots empty while the CUs that hold two run at half speed; 480 tiles of 96 rows do not
        auto cost = [&](int rows) { return ceil_div(ceil_div(M, rows) * n_blocks, 512) * rows; };
        const int rt = switches().dense_rt == 3 || switches().dense_rt == 4 ? switches().dense_rt : (cost(96) < cost(128) ? 3 : 4);
        const long m_tiles = ceil_div(M, 32 * rt);
        const long grid = ceil_div(m_tiles * n_blocks, 8) * 8;
        LRPX_REQUIRE(grid > 0 && grid <= 0x7fffffffL, "dense_f16x3: grid %ld out of range", grid);
        static LdsOnce once_n[10];
        const int v = plain ? 4 : (a.U ? 1 : 0) + (a.out1 ? 2 : 0);
        using Kern = void (*)(ConvArgs, int, int);
        static const Kern kerns[2][5] = {
            {dense_f16x3_n256_kernel<EPI_REL, false, false, 4>, dense_f16x3_n256_kernel<EPI_REL, true, false, 4>,
             dense_f16x3_n256_kernel<EPI_REL, false, true, 4>, dense_f16x3_n256_kernel<EPI_REL, true, true, 4>,
             dense_f16x3_n256_kernel<EPI_PLAIN, false, false, 4>},
            {dense_f16x3_n256_kernel<EPI_REL, false, false, 3>, dense_f16x3_n256_kernel<EPI_REL, true, false, 3>,
             dense_f16x3_n256_kernel<EPI_REL, false, true, 3>, dense_f16x3_n256_kernel<EPI_REL, true, true, 3>,
             dense_f16x3_n256_kernel<EPI_PLAIN, false, false, 3>}};
        Kern kern = kerns[rt == 3][v];
        LRPX_TRY(reserve_lds_once(once_n[(rt == 3 ? 5 : 0) + v], kern, DnCfg<false>::LDS, "dense_f16x3_n256"));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), DnCfg<false>::LDS, stream, a, (int)m_tiles, n_blocks);
        return check_launch("dense_f16x3_n256");
    }
    const bool wide = switches().dense_wide && M >= 8192;      // 256-row tiles (8 waves) for many rows: measured slower, off (A/B)
    const long m_tiles = ceil_div(M, wide ? 256 : 128);
    const int n_blocks = (int)ceil_div(a.n_oc, DH_BN);
    const long total = m_tiles * n_blocks;
    const long grid = ceil_div(total, 8) * 8;
    LRPX_REQUIRE(grid > 0 && grid <= 0x7fffffffL, "dense_f16x3: grid %ld out of range", grid);
    static LdsOnce attr_once[4];
    const int which = (plain ? 2 : 0) + (wide ? 1 : 0);
    if (wide) {
        auto kern = plain ? dense_f16x3_kernel<EPI_PLAIN, 4> : dense_f16x3_kernel<EPI_REL, 4>;
        LRPX_TRY(reserve_lds_once(attr_once[which], kern, DenseCfg<4>::LDS, "dense_f16x3"));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), DenseCfg<4>::LDS, stream, a, (int)m_tiles, n_blocks);
    } else {
        auto kern = plain ? dense_f16x3_kernel<EPI_PLAIN, 2> : dense_f16x3_kernel<EPI_REL, 2>;
        LRPX_TRY(reserve_lds_once(attr_once[which], kern, DenseCfg<2>::LDS, "dense_f16x3"));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), DenseCfg<2>::LDS, stream, a, (int)m_tiles, n_blocks);
    }
    return check_launch("dense_f16x3");
}

// The (word, pixel) rules with EXACT operand splits on the bf16 matrix cores (B6 above): any number of rows - ONE kernel, so that a
// row's sum does not depend on the batch it sits in; weights from lrpx_pack_weights_bf16x3 (taps = 1).
int launch_dense_bf16x6(const ConvArgs& a, hipStream_t stream) {
    const long M = (long)a.n_maps * a.pix_per_map;
    LRPX_REQUIRE(a.epi == EPI_REL && a.X && a.map2img && (a.out0 || a.out1) && !(a.out0 && a.out1), "dense_bf16x6: the REL epilogue with x, map2img and ONE output");
    LRPX_REQUIRE(!a.out1 || a.Zdiv, "dense_bf16x6: out1 needs zdiv");
    LRPX_REQUIRE(a.cin % DN_KC == 0 && a.cin >= DN_KC, "dense_bf16x6: K = %d is not a multiple of %d", a.cin, DN_KC);
    LRPX_REQUIRE(a.oc_split % 4 == 0 && a.oc_split >= 4 && a.n_oc >= a.oc_split, "dense_bf16x6: %d columns (a multiple of 4)", a.oc_split);
    LRPX_REQUIRE(M > 0 && M * a.cin < 0x7fffffffL && M * a.oc_split < 0x7fffffffL && (long)a.n_maps * a.oc_split < 0x7fffffffL,
                 "dense_bf16x6: %ld rows out of range for 32-bit offsets", M);
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    LRPX_REQUIRE(al16(a.in) && al16(a.X) && al16(a.U) && al16(a.Zdiv) && al16(a.out0) && al16(a.out1), "dense_bf16x6: operands must be 16-byte aligned");
    LRPX_REQUIRE(!a.out1_amax || a.pix_per_map >= 32, "dense_bf16x6: out1_amax needs at least 32 rows per map");
    const int n_blocks = (int)ceil_div(a.n_oc, DN_BN);
    auto cost = [&](int rows) { return ceil_div(ceil_div(M, rows) * n_blocks, 512) * rows; };
    const int rt = switches().dense_rt == 3 || switches().dense_rt == 4 ? switches().dense_rt : (cost(96) < cost(128) ? 3 : 4);
    const long m_tiles = ceil_div(M, 32 * rt);
    const long grid = ceil_div(m_tiles * n_blocks, 8) * 8;
    LRPX_REQUIRE(grid > 0 && grid <= 0x7fffffffL, "dense_bf16x6: grid %ld out of range", grid);
    static LdsOnce once_n[8];
    const int v = (a.U ? 1 : 0) + (a.out1 ? 2 : 0);
    using Kern = void (*)(ConvArgs, int, int);
    static const Kern kerns[2][4] = {
        {dense_f16x3_n256_kernel<EPI_REL, false, false, 4, true>, dense_f16x3_n256_kernel<EPI_REL, true, false, 4, true>,
         dense_f16x3_n256_kernel<EPI_REL, false, true, 4, true>, dense_f16x3_n256_kernel<EPI_REL, true, true, 4, true>},
        {dense_f16x3_n256_kernel<EPI_REL, false, false, 3, true>, dense_f16x3_n256_kernel<EPI_REL, true, false, 3, true>,
         dense_f16x3_n256_kernel<EPI_REL, false, true, 3, true>, dense_f16x3_n256_kernel<EPI_REL, true, true, 3, true>}};
    Kern kern = kerns[rt == 3][v];
    LRPX_TRY(reserve_lds_once(once_n[(rt == 3 ? 4 : 0) + v], kern, DnCfg<true>::LDS, "dense_bf16x6"));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), DnCfg<true>::LDS, stream, a, (int)m_tiles, n_blocks);
    return check_launch("dense_bf16x6");
}

}  // namespace lrpx
#ifdef LRPX_STAMP
extern "C" int lrpx_debug_stamps_dn(unsigned long long* out, int n_waves, int reset) {
    if (n_waves > 16384) return 2;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(lrpx::g_stamp_dn), (size_t)n_waves * 80) != hipSuccess) return 1;
    if (reset) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(lrpx::g_stamp_dn)) != hipSuccess || hipMemset(p, 0, (size_t)16384 * 80) != hipSuccess) return 1;
    }
    return 0;
}
#endif
