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

constexpr int DH_BM = 128, DH_BN = 128, DH_KC = 64;
constexpr int DH_ROWB = 4 * 2 * 2 * 16 + 16;        // LDS bytes per A row: [k-step 4][plane 2][lane group 2][16 B] + pad
constexpr int DH_BUF = DH_BM * DH_ROWB;
constexpr int DH_LDS = 2 * DH_BUF;

__global__ __launch_bounds__(256, 2) void dense_f16x3_kernel(ConvArgs a, int m_tiles, int n_blocks) {
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
    const long row0 = (long)mtile * DH_BM;
    const int K = a.cin, nchunk = K / DH_KC;
    const unsigned P = (unsigned)a.pix_per_map;
    const unsigned* __restrict__ in_amax = a.in_amax;

    // ---- staging: thread -> 8 items (row = tid / 16 + 16 u, 16-byte segment tid % 16 of the 64-float chunk row)
    const int s_row = tid >> 4, s_seg = tid & 15;
    float ssc[8];
    long srow[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const long r = row0 + s_row + 16 * u;
        const long rc = r < M ? r : M - 1;               // rows past the end re-read the last row (results dropped)
        srow[u] = rc * K;
        ssc[u] = exp2i(f16_scale_exp(in_amax[(unsigned)rc / P]));
    }
    // LDS offset of the item inside a row: k-step = seg / 4, lane group = (seg / 2) & 1, half = seg & 1 (8 bytes)
    const int s_off = (s_seg >> 2) * 64 + ((s_seg >> 1) & 1) * 16 + (s_seg & 1) * 8;
    f32x4 sv[8];
    const float* __restrict__ A = a.in;
#define DH_ISSUE(CHUNK) _Pragma("unroll") for (int u = 0; u < 8; ++u) sv[u] = *reinterpret_cast<const f32x4*>(A + srow[u] + (CHUNK) * DH_KC + s_seg * 4);
#define DH_COMMIT(BUF)                                                                                     \
    _Pragma("unroll") for (int u = 0; u < 8; ++u) {                                                        \
        _Float16 h[4], l[4];                                                                               \
        _Pragma("unroll") for (int e = 0; e < 4; ++e) split2(sv[u][e] * ssc[u], h[e], l[e]);                \
        char* d_ = ldsb + (BUF) * DH_BUF + (s_row + 16 * u) * DH_ROWB + s_off;                             \
        *reinterpret_cast<u32x2_*>(d_) = u32x2_{pack_f16(h[0], h[1]), pack_f16(h[2], h[3])};               \
        *reinterpret_cast<u32x2_*>(d_ + 32) = u32x2_{pack_f16(l[0], l[1]), pack_f16(l[2], l[3])};          \
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
    const int a_off = (wm * 64 + li) * DH_ROWB + lh * 16;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        {
            const int cn = min(chunk + 1, nchunk - 1);     // (past the last chunk: re-read it, nobody commits it)
            DH_ISSUE(cn)
        }
        const char* abuf = ldsb + (chunk & 1) * DH_BUF + a_off;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            load_b(chunk * 4 + s + DH_NBQ - 1, bq[(s + DH_NBQ - 1) % DH_NBQ]);
            const f16x8 a0h = *reinterpret_cast<const f16x8*>(abuf + s * 64);
            const f16x8 a0l = *reinterpret_cast<const f16x8*>(abuf + s * 64 + 32);
            const f16x8 a1h = *reinterpret_cast<const f16x8*>(abuf + 32 * DH_ROWB + s * 64);
            const f16x8 a1l = *reinterpret_cast<const f16x8*>(abuf + 32 * DH_ROWB + s * 64 + 32);
            const u32x4_(&b)[4] = bq[s % DH_NBQ];
            const f16x8 b0h = __builtin_bit_cast(f16x8, b[0]), b0l = __builtin_bit_cast(f16x8, b[1]);
            const f16x8 b1h = __builtin_bit_cast(f16x8, b[2]), b1l = __builtin_bit_cast(f16x8, b[3]);
            // small terms first
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
        }
        if (chunk + 1 < nchunk) { DH_COMMIT((chunk + 1) & 1) }
        __syncthreads();
    }
#undef DH_ISSUE
#undef DH_COMMIT

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
            const unsigned img = m2i ? (unsigned)m2i[n] : n;
            xb[e] = (img * P + p) * (unsigned)ncol;               // (< 2^31: host-checked)
            sc[e] = exp2i(-f16_scale_exp(in_amax[n])) * inv_w;
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
        if (oamax) {
            m0 = wave_max(m0); m1 = wave_max(m1);
            if (lane == 0 && rt < M) {
                amax_update(&oamax[nt0], m0);
                if ((int)nt0 + 1 <= nmax) amax_update(&oamax[nt0 + 1], m1);
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
template <int MAXU>        // float4 items of the A slab per thread: 16 for K <= 512, 32 for K <= 1024
__global__ __launch_bounds__(256, 2) void dense_small_f16x3_kernel(ConvArgs a, int m_tiles, int n_blocks) {
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int mtile = blockIdx.x / n_blocks, nblk = blockIdx.x % n_blocks;
    const int K = a.cin, nks = K / 16, k4 = K / 4;
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
    if (tid < 32) rowmax[tid] = 0u;
    __syncthreads();
    // pass 1: the slab through registers (<= 32 float4 per thread), row maxima into LDS
    f32x4 sv[MAXU];
    const int nu = (32 * k4 + 255) / 256;
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        sv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (u < nu) {
            const int it = tid + 256 * u;
            const int r = it / k4, c4 = it - r * k4;
            if (it < 32 * k4 && row0 + r < rows) sv[u] = *reinterpret_cast<const f32x4*>(a.in + (row0 + r) * K + c4 * 4);
        }
    }
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        if (u < nu) {
            const int it = tid + 256 * u;
            const int r = it / k4;
            float m = fmaxf(fmaxf(fabsf(sv[u][0]), fabsf(sv[u][1])), fmaxf(fabsf(sv[u][2]), fabsf(sv[u][3])));
            if ((k4 & 63) == 0) {                  // a wave's 64 items lie in one row: one LDS atomic per wave
                m = wave_max(m);
                if (lane == 0 && it < 32 * k4) atomicMax(&rowmax[r], __builtin_bit_cast(unsigned, m));
            } else if (it < 32 * k4) {
                atomicMax(&rowmax[r], __builtin_bit_cast(unsigned, m));
            }
        }
    }
    __syncthreads();
    // pass 2: scale, split, LDS.  Row layout: [k-step][hi: lane group 0, 1 | lo: lane group 0, 1] x 16 B
#pragma unroll
    for (int u = 0; u < MAXU; ++u) {
        if (u < nu) {
            const int it = tid + 256 * u;
            if (it < 32 * k4) {
                const int r = it / k4, c4 = it - r * k4;
                const float sc = exp2i(f16_scale_exp(rowmax[r]));
                _Float16 h[4], l[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) split2(sv[u][e] * sc, h[e], l[e]);
                char* d_ = ldsb + r * pitch + (c4 >> 2) * 64 + ((c4 >> 1) & 1) * 16 + (c4 & 1) * 8;
                *reinterpret_cast<u32x2_*>(d_) = u32x2_{pack_f16(h[0], h[1]), pack_f16(h[2], h[3])};
                *reinterpret_cast<u32x2_*>(d_ + 32) = u32x2_{pack_f16(l[0], l[1]), pack_f16(l[2], l[3])};
            }
        }
    }
    __syncthreads();
    if (!wave_active) return;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const char* ap = ldsb + li * pitch + lh * 16;
    for (int ks = 0; ks < nks; ks += DS_NB) {
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
    const int ncol = a.oc_split;
    if (oc >= ncol) return;
    const unsigned P = (unsigned)a.pix_per_map;
    const float inv_w = a.wp[0];
    float xv[16];
    long nn[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const long row = row0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        xv[e] = 0.f; nn[e] = 0;
        if (row < rows) {
            const long n = row / P, p = row - n * P;
            const long img = a.map2img ? a.map2img[n] : n;
            nn[e] = n;
            xv[e] = a.X[(img * P + p) * ncol + oc];
        }
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int rl = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const long row = row0 + rl;
        if (row >= rows) continue;
        float v = acc[e] * (exp2i(-f16_scale_exp(rowmax[rl])) * inv_w);
        if (a.U) v += a.U[nn[e] * ncol + oc];
        a.out0[row * ncol + oc] = xv[e] * v;
    }
}

int launch_dense_small_f16x3(const ConvArgs& a, hipStream_t stream) {
    const long rows = (long)a.n_maps * a.pix_per_map;
    LRPX_REQUIRE(a.cin % 16 == 0 && a.cin >= 16 && a.cin <= 1024, "dense_small_f16x3: K = %d (a multiple of 16, <= 1024)", a.cin);
    LRPX_REQUIRE(a.X && a.out0 && !a.out1 && rows > 0, "dense_small_f16x3: REL epilogue with x and out0 only");
    const int m_tiles = (int)ceil_div(rows, 32);
    const int n_blocks = (int)ceil_div(a.n_oc, 128);
    const int lds = 32 * ((a.cin / 16) * 64 + 16) + 128;
    constexpr int LDS_MAX = 32 * (64 * 64 + 16) + 128;
    const bool big = a.cin > 512;
    auto kern = big ? dense_small_f16x3_kernel<32> : dense_small_f16x3_kernel<16>;
    static LdsOnce once[2];
    LRPX_TRY(reserve_lds_once(once[big], kern, LDS_MAX, "dense_small_f16x3"));
    hipLaunchKernelGGL(kern, dim3((unsigned)(m_tiles * n_blocks)), dim3(256), lds, stream, a, m_tiles, n_blocks);
    return check_launch("dense_small_f16x3");
}

int launch_dense_f16x3(const ConvArgs& a, hipStream_t stream) {
    const long M = (long)a.n_maps * a.pix_per_map;
    LRPX_REQUIRE(a.cin % DH_KC == 0 && a.cin >= DH_KC, "dense_f16x3: K = %d is not a multiple of %d", a.cin, DH_KC);
    LRPX_REQUIRE(a.in_amax && a.X && (a.out0 || a.out1), "dense_f16x3: needs in_amax, x and an output");
    LRPX_REQUIRE(M > 0 && M < 0x7fffffffL, "dense_f16x3: %ld rows out of range", M);
    LRPX_REQUIRE((long)a.n_maps * a.pix_per_map * a.oc_split < 0x7fffffffL, "dense_f16x3: too many multiplicand elements for 32-bit offsets");
    LRPX_REQUIRE(!(a.out1 && a.out1_amax) || a.pix_per_map >= 32, "dense_f16x3: out1_amax needs at least 32 rows per map");
    const long m_tiles = ceil_div(M, DH_BM);
    const int n_blocks = (int)ceil_div(a.n_oc, DH_BN);
    const long total = m_tiles * n_blocks;
    const long grid = ceil_div(total, 8) * 8;
    LRPX_REQUIRE(grid > 0 && grid <= 0x7fffffffL, "dense_f16x3: grid %ld out of range", grid);
    static LdsOnce attr_once;
    LRPX_TRY(reserve_lds_once(attr_once, dense_f16x3_kernel, DH_LDS, "dense_f16x3"));
    hipLaunchKernelGGL(dense_f16x3_kernel, dim3((unsigned)grid), dim3(256), DH_LDS, stream, a, (int)m_tiles, n_blocks);
    return check_launch("dense_f16x3");
}

}  // namespace lrpx
