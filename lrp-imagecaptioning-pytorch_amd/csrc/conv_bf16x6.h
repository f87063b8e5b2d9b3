// fp32-accurate 3x3 convolution on the bf16 matrix cores ("bf16x6"): every fp32 operand is split EXACTLY into three
// bf16 parts a = a0 + a1 + a2 (8 + 8 + 8 mantissa bits) and the six partial products with i + j <= 2
//        a*b ~= a0b0 + (a0b1 + a1b0) + (a0b2 + a1b1 + a2b0)
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped products are <= 3 * 2^-24 relative, i.e. the result
// has fp32 accuracy, while 6 bf16 MFMAs of K=16 cost 6*32 = 192 SIMD-cycles against 8*64 = 512 for the same K on the
// fp32 MFMA (v_mfma_f32_32x32x2_f32 runs at 1/16 of the bf16 rate): 2.67x less matrix-pipe time.
//
// Same tiling, LDS double buffering, B-fragment queue and epilogues as conv_mfma.h (224 pixels x 32 channels per
// wave, 4 waves per workgroup, K-chunk of 16 channels = ONE bf16 k-step per tap).  Differences:
//   * A is split while it is staged: LDS pixel = 3 planes x 16 bf16 (32 B each) + 16 B pad = 112 B (conflict-free
//     ds_read_b128: 28 dwords stride);
//   * weights are split at pack time: per k-step three 1-KiB fragments (lrpx_pack_weights_bf16x3).
// Used for the relevance pass of every VGG16 layer with >= 64 input channels (conv1_2 .. conv5_3) and for the forward
// trace of the 112^2 .. 14^2 layers; lrpx_set_bf16x6(0) switches back to the fp32-MFMA kernels of conv_mfma.h.
#pragma once
#include "conv_mfma.h"

#ifndef LRPX6_NBQ
#define LRPX6_NBQ ((HW <= 56) ? 5 : 4)   // B-fragment register queue depth (k-steps in flight: NBQ-1); 5 spills at 112/224
#endif
#ifndef LRPX6_APIPE
#define LRPX6_APIPE 0    // 1: A fragments read one accumulator tile ahead (sched_group_barrier pinned)
#endif

namespace lrpx {

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned short f32_to_bf16_rn(float x) {
    __bf16 b = (__bf16)x;     // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
    return __builtin_bit_cast(unsigned short, b);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }

// exact three-way split: x == p0 + p1 + p2 (each a bf16), barring underflow of the smallest part
__device__ __forceinline__ void split3(float x, unsigned short& p0, unsigned short& p1, unsigned short& p2) {
    p0 = f32_to_bf16_rn(x);
    const float r1 = x - bf16_to_f32(p0);
    p1 = f32_to_bf16_rn(r1);
    const float r2 = r1 - bf16_to_f32(p1);
    p2 = f32_to_bf16_rn(r2);
}

#ifdef LRPX_STAMP
// profiling build only (make STAMP=1): per-phase shader-clock totals over all waves, read by lrpx_debug_stamps()
static __device__ unsigned long long g_stamp[8];
#define LRPX_T(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); \
                  __builtin_amdgcn_sched_barrier(0)
#else
#define LRPX_T(v)
#endif

// DB: double-buffered A tile (one barrier per chunk).  DB = false (tiles whose two buffers would not leave room for two
// workgroups per CU: the 112-pixel layers): one buffer, the prefetched registers are committed between two barriers.
template <int HW, int MT, int NWN, bool DB, int EPI>
__global__ __launch_bounds__(64 * MT * NWN, 2) void conv_bf16x6_kernel(ConvArgs a, int m_tiles, int n_blocks) {
    constexpr int KC = 16, TAPS = 9;
    using C = ConvCfg<HW, KC, MT, NWN, TAPS>;
    constexpr int W = C::W, H = C::H, NT = C::NT;
    constexpr int PSTRIDE = 112;                       // bytes per LDS pixel
    // LDS row pitch: (W+2) pixels rounded up so that pitch/16 == 7*W (mod 16).  The 16-byte slot of pixel q = r*W + c of
    // a tile is then 7*q + const (mod 16) even when the 32 pixels of an MFMA row-tile wrap onto the next image row, so
    // every 16-lane group of a ds_read_b128 hits 16 distinct slots (with pitch = (W+2)*112 the wrapped lanes collided:
    // 30-50 % extra LDS cycles on the 56/28/14 layers)
    constexpr int PITCH = W * PSTRIDE + 256;
    constexpr int BUFB = C::NSLOT * PITCH;             // bytes per LDS buffer
    constexpr int NBUF = DB ? 2 : 1;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];

    LRPX_T(t_start);
#ifdef LRPX_STAMP
    unsigned long long s_issue = 0, s_mfma = 0, s_commit = 0, s_barrier = 0;
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int mtile = (idx / n_blocks) * 8 + xcd;
    const int nblk = idx % n_blocks;
    if (mtile >= m_tiles) return;

    const int ocb = nblk * NWN + wn;
    const bool wave_active = ocb * 32 < a.n_oc;
    const int nchunk = a.cin / KC;
    const long total_pix = (long)a.n_maps * a.pix_per_map;
    const int li = lane & 31, lh = lane >> 5;

    const long g0 = (long)mtile * C::R;
    const long v0 = g0 + g0 / H;
    int abase[7];      // byte offset of the 3x3 window corner of this lane's pixel, k-half lh
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int q = wm * 224 + 32 * j + li;
        const int r = q / W, c = q % W;
        const long g = g0 + r;
        const int slot = (int)(g + g / H - v0) + 1;
        abase[j] = (slot - 1) * PITCH + c * PSTRIDE + lh * 16;
    }

    // ---- staging descriptors (16 channels = 4 float4 segments per pixel) ----
    constexpr int SEG = KC / 4;
    constexpr int NITEM = C::NSLOT * W * SEG;
    constexpr int U = (NITEM + NT - 1) / NT;
    int sdst[U], sgp[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int it = tid + u * NT;
        sdst[u] = -1; sgp[u] = -1;
        if (it < NITEM) {
            const int s = it / (W * SEG);
            const int rem = it - s * (W * SEG);
            const int px = rem / SEG, seg = rem - px * SEG;
            const long v_ = v0 - 1 + s;
            const long n = v_ / (H + 1);
            const int y = (int)(v_ - n * (H + 1));
            if ((v_ >= 0) && (y < H) && (n < a.n_maps)) {
                sdst[u] = (s * PITCH + (px + 1) * PSTRIDE + seg * 8) | (seg << 28);   // byte offset of plane 0, 4 bf16
                sgp[u] = (int)((n * H + y) * W + px);
            }
        }
    }
    f32x4 sv[U];
#define LRPX6_ISSUE(CHUNK)                                                                                   \
    _Pragma("unroll") for (int u = 0; u < U; ++u) {                                                          \
        sv[u] = f32x4{0, 0, 0, 0};                                                                           \
        if (sgp[u] >= 0)                                                                                     \
            sv[u] = *reinterpret_cast<const f32x4*>(                                                         \
                a.in_chunk_stride ? a.in + (CHUNK) * a.in_chunk_stride + (long)sgp[u] * KC + ((sdst[u] >> 28) & 7) * 4 \
                                  : a.in + (long)sgp[u] * a.cin + (CHUNK) * KC + ((sdst[u] >> 28) & 7) * 4);  \
    }
#define LRPX6_COMMIT(BUFIDX)                                                                                 \
    _Pragma("unroll") for (int u = 0; u < U; ++u)                                                            \
        if (sdst[u] >= 0) {                                                                                  \
            unsigned short p0[4], p1[4], p2[4];                                                              \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) split3(sv[u][e], p0[e], p1[e], p2[e]);             \
            char* d = ldsb + (BUFIDX) * BUFB + (sdst[u] & 0x0fffffff);                                       \
            *reinterpret_cast<u32x2*>(d) = u32x2{p0[0] | ((unsigned)p0[1] << 16), p0[2] | ((unsigned)p0[3] << 16)};      \
            *reinterpret_cast<u32x2*>(d + 32) = u32x2{p1[0] | ((unsigned)p1[1] << 16), p1[2] | ((unsigned)p1[3] << 16)}; \
            *reinterpret_cast<u32x2*>(d + 64) = u32x2{p2[0] | ((unsigned)p2[1] << 16), p2[2] | ((unsigned)p2[3] << 16)}; \
        }

    LRPX6_ISSUE(0)
    for (int i = tid; i < NBUF * BUFB / 16; i += NT) reinterpret_cast<u32x4*>(ldsb)[i] = u32x4{0, 0, 0, 0};
    __syncthreads();
    LRPX6_COMMIT(0)

    f32x16 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // B fragments: per k-step three planes of 64 lanes x 16 B, one contiguous stream per channel block
    constexpr int NBQ = LRPX6_NBQ;
    const u32x4* wp = reinterpret_cast<const u32x4*>(a.wp) + (long)ocb * nchunk * (TAPS * 3 * 64) + lane;
    const int last_step = nchunk * TAPS - 1;
    u32x4 bq[NBQ][3];
#pragma unroll
    for (int i = 0; i < NBQ; ++i)
#pragma unroll
        for (int p = 0; p < 3; ++p) bq[i][p] = u32x4{0, 0, 0, 0};
    if (wave_active) {
#pragma unroll
        for (int i = 0; i < NBQ - 1; ++i)
#pragma unroll
            for (int p = 0; p < 3; ++p) bq[i][p] = wp[((long)min(i, last_step) * 3 + p) * 64];
    }
    __syncthreads();

    LRPX_T(t_loop);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const bool more = chunk + 1 < nchunk;
        LRPX_T(ta);
        if (more) { LRPX6_ISSUE(chunk + 1) }
        LRPX_T(tb);
        if (wave_active) {
            const char* abuf = ldsb + (DB ? (chunk & 1) : 0) * BUFB;
            // A fragments run one accumulator tile ahead of the MFMAs that consume them (the LDS latency of a read
            // issued right before its MFMA is otherwise exposed 63 times per chunk)
#if LRPX6_APIPE
            bf16x8 n0 = *reinterpret_cast<const bf16x8*>(abuf + abase[0]);
            bf16x8 n1 = *reinterpret_cast<const bf16x8*>(abuf + abase[0] + 32);
            bf16x8 n2 = *reinterpret_cast<const bf16x8*>(abuf + abase[0] + 64);
            __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);           // (tile 0 of tap 0)
#endif
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const long nxt = (long)min(chunk * TAPS + tap + NBQ - 1, last_step) * 3;
#pragma unroll
                for (int p = 0; p < 3; ++p) bq[NBQ - 1][p] = wp[(nxt + p) * 64];
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, bq[0][0]);
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, bq[0][1]);
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, bq[0][2]);
#pragma unroll
                for (int j = 0; j < 7; ++j) {
#if LRPX6_APIPE
                    const bf16x8 a0 = n0, a1 = n1, a2 = n2;
                    if (!(tap == TAPS - 1 && j == 6)) {
                        const int jn = (j + 1) % 7, tn = tap + (j == 6 ? 1 : 0);
                        const char* ap = abuf + abase[jn] + (tn / 3) * PITCH + (tn % 3) * PSTRIDE;
                        n2 = *reinterpret_cast<const bf16x8*>(ap + 64);
                        n1 = *reinterpret_cast<const bf16x8*>(ap + 32);
                        n0 = *reinterpret_cast<const bf16x8*>(ap);
                    }
#else
                    const char* ap = abuf + abase[j] + (tap / 3) * PITCH + (tap % 3) * PSTRIDE;
                    const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(ap);
                    const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(ap + 32);
                    const bf16x8 a2 = *reinterpret_cast<const bf16x8*>(ap + 64);
#endif
                    // smallest terms first
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
#if LRPX6_APIPE
                    __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // the 3 reads of the NEXT tile ...
                    __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // ... then the 6 MFMAs of this one
#endif
                }
#pragma unroll
                for (int i = 0; i < NBQ - 1; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bq[i][p] = bq[i + 1][p];
            }
        }
        LRPX_T(tc);
        if constexpr (DB) {
            if (more) { LRPX6_COMMIT((chunk + 1) & 1) }
            LRPX_T(td);
            __syncthreads();
            LRPX_T(te);
#ifdef LRPX_STAMP
            s_issue += tb - ta; s_mfma += tc - tb; s_commit += td - tc; s_barrier += te - td;
#endif
        } else {
            __syncthreads();                       // every wave is done reading the single buffer
            if (more) { LRPX6_COMMIT(0) }
            __syncthreads();
        }
    }
#undef LRPX6_ISSUE
#undef LRPX6_COMMIT
    if (!wave_active) return;
    LRPX_T(t_epi);

    EpiCtx cx;
    cx.oc = ocb * 32 + li;
    cx.lane = lane;
    cx.q0 = wm * 224 + 4 * lh;
    cx.g0 = (int)g0;
    cx.pix0 = g0 * W;
    cx.total_pix = total_pix;
    cx.xi_base = 0;
    constexpr bool AL = (H % C::R == 0);
    if constexpr (AL) {
        const unsigned rr = (unsigned)cx.q0 / (unsigned)HW, cc = (unsigned)cx.q0 - rr * HW;
        const unsigned g = (unsigned)cx.g0 + rr;
        const unsigned n = g / (unsigned)HW;
        const long img = a.map2img ? a.map2img[n] : n;
        cx.xi_base = (img * a.pix_per_map + (long)((g - n * HW) * HW + cc)) * a.oc_split + cx.oc;
    }
    EpiRegs ra, rb;
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 0, ra);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 1, rb);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 0, acc[0], ra);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 2, ra);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 1, acc[1], rb);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 3, rb);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 2, acc[2], ra);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 4, ra);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 3, acc[3], rb);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 5, rb);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 4, acc[4], ra);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 6, ra);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 5, acc[5], rb);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 6, acc[6], ra);
#ifdef LRPX_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LRPX_T(t_end);
    if (lane == 0) {
        atomicAdd(&g_stamp[0], t_loop - t_start);
        atomicAdd(&g_stamp[1], s_issue);
        atomicAdd(&g_stamp[2], s_mfma);
        atomicAdd(&g_stamp[3], s_commit);
        atomicAdd(&g_stamp[4], s_barrier);
        atomicAdd(&g_stamp[5], t_end - t_epi);
        atomicAdd(&g_stamp[6], t_end - t_start);
        atomicAdd(&g_stamp[7], 1ull);
    }
#endif
}

template <int HW, int MT, int NWN, bool DB, int EPI>
int launch_conv_bf16x6(const ConvArgs& a, hipStream_t stream) {
    using C = ConvCfg<HW, 16, MT, NWN, 9>;
    constexpr int LDS = (DB ? 2 : 1) * C::NSLOT * (HW * 112 + 256);
    const long m_tiles = ceil_div((long)a.n_maps * HW, C::R);
    const int n_blocks = (int)ceil_div(a.n_oc, 32 * NWN);
    const long grid = ceil_div(m_tiles, 8) * 8 * n_blocks;
    auto kern = conv_bf16x6_kernel<HW, MT, NWN, DB, EPI>;
    static LdsOnce attr_once;
    LRPX_TRY(reserve_lds_once(attr_once, kern, LDS, "conv_bf16x6"));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(64 * MT * NWN), LDS, stream, a, (int)m_tiles, n_blocks);
    return check_launch("conv_bf16x6");
}

}  // namespace lrpx
