// conv1_2's relevance step in mode 3 (LRPtools/lrp_modules.py:124-150 under the Pool2d rule :182-195) as a STRIP-PERSISTENT kernel.
//
// The generic kernel (conv_f16x3.h <224, 2, 2, false, REL_MUL, POOL, F8>) spends a 2-row tile's time outside the matrix pipe: K is only
// 64 channels x 9 taps = 4 K-chunks, so per tile it stages 3 pooled rows for 1 pooled row of output (the halo), four times (once per chunk,
// barrier - commit - barrier), streams the whole weight set from L2 and stores 114 KB (pipe busy 0.36).  This layer's weights are
// small enough to LIVE IN REGISTERS, and its whole channel depth fits LDS for a narrow strip, so here
//   * a workgroup (4 waves, ONE per SIMD, up to 512 VGPRs each) owns a strip of 32 columns of one map and walks it top to bottom in steps of 4
//     output rows: wave (och, pg) computes rows 2 pg, 2 pg + 1 of the step for the 32 output channels of half `och` over the WHOLE K;
//   * the B fragments of its 32 channels (36 fp16 k-steps + 20 fp6 MFMA operands, read once from the pack of pack_weights_f16f8_kernel)
//     stay in registers for all 57 steps: no weight stream, no B queue;
//   * LDS holds a ring of 10 input rows x 34 pixels x all 4 K-chunks (4 planes of the generic kernel's 80-byte pixel: 16 fp16 hi | 32 fp6 |
//     block scale): a step needs rows r - 1 .. r + 4, the next step only 4 NEW rows = 2 pooled rows - every pooled pixel is loaded, split and
//     unpooled once per strip (the generic kernel: 3x, the halo rows of every tile), no halo rows, 2 halo columns per 32;
//   * one barrier per step; the staging loads and the multiplicands of a step are issued before its 112 MFMAs.
// Arithmetic per output element is the generic kernel's (same operand split, same block scales, same products, fp32 accumulation over
// chunk-major k: the summation order inside an accumulator is identical), results agree to rounding of nothing: bit-identical S1.
#include "conv_launch.h"
#include "conv_f16x3.h"

namespace lrpx {
namespace s12 {
constexpr int HW = 224, HO = 112, SW = 32, NSTRIP = HW / SW, NSTEP = HW / 4 + 1, RING = 10;
constexpr int PXB = 80, PITCH = (SW + 2) * PXB, PLANE = RING * PITCH;
constexpr int LDS_PIX = 4 * PLANE;              // 108 800 bytes
constexpr int DUMMY = LDS_PIX;                  // 256 bytes: writes of positions outside the strip's LDS columns
constexpr int LDS_BYTES = LDS_PIX + 256;
constexpr int NOPS = 112;                       // MFMAs per wave and step: 4 chunks x (10 + 10 + 8)

// op k -> chunk, tap row g, kind m (0, 1: fp6 MFMA 2g + m; 2..4: fp16 dx = m - 2), tile t
__host__ __device__ constexpr int op_chunk(int k) { return k / 28; }
__host__ __device__ constexpr int op_g(int k) { return (k % 28) < 10 ? 0 : ((k % 28) < 20 ? 1 : 2); }
__host__ __device__ constexpr int op_t(int k) { return ((k % 28) - 10 * op_g(k)) & 1; }
__host__ __device__ constexpr int op_m(int k) {
    const int r = ((k % 28) - 10 * op_g(k)) >> 1;
    return op_g(k) < 2 ? r : (r == 0 ? 0 : r + 1);
}
}  // namespace s12

// one window position of a pooled 16-channel slice: channels whose maximum sat elsewhere are zeroed, fp16 hi | fp6 | scale -> 64 bytes at d
__device__ __forceinline__ void s12_commit_pos(const unsigned (&hwu)[8], const unsigned (&rwu)[8], const float bs, const unsigned sb,
                                               const u32x4_ am, const unsigned pos, char* d) {
    unsigned hm[8], rm[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const unsigned x = am[q] ^ (0x01010101u * pos);                  // zero byte <=> winner == pos
        const unsigned eq = ((x | (x >> 1)) & 0x01010101u) ^ 0x01010101u;
        const unsigned sel = eq + 0x0c0c0c0cu;                           // v_perm_b32 selectors 0x0c / 0x0d: constants 0x00 / 0xff
        const unsigned m01 = __builtin_amdgcn_perm(0u, 0u, __builtin_amdgcn_perm(sel, sel, 0x01010000u));
        const unsigned m23 = __builtin_amdgcn_perm(0u, 0u, __builtin_amdgcn_perm(sel, sel, 0x03030202u));
        hm[2 * q] = hwu[2 * q] & m01; hm[2 * q + 1] = hwu[2 * q + 1] & m23;
        rm[2 * q] = rwu[2 * q] & m01; rm[2 * q + 1] = rwu[2 * q + 1] & m23;
    }
    const u32x6_ q6 = x6_pack(hm, rm, bs);
    *reinterpret_cast<u32x4_*>(d) = u32x4_{hm[0], hm[1], hm[2], hm[3]};
    *reinterpret_cast<u32x4_*>(d + 16) = u32x4_{hm[4], hm[5], hm[6], hm[7]};
    *reinterpret_cast<u32x4_*>(d + 32) = u32x4_{q6[0], q6[1], q6[2], q6[3]};
    *reinterpret_cast<u32x4_*>(d + 48) = u32x4_{q6[4], q6[5], 0u, sb};
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void conv12_strip_kernel(ConvArgs a, int n_groups, int tg, int dbg) {
    using namespace s12;
    extern __shared__ __attribute__((aligned(16))) char ldsb[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;
    const int och = wave & 1, pg = wave >> 1;
    // workgroup -> (map, strip): the `tg` maps of one image (the words of its caption) walk the same strip back to back on ONE XCD
    // (their multiplicand rows and winner bytes then come from that L2), an XCD owns a contiguous range of (image, strip) groups
    const int bid = blockIdx.x, xcd = bid & 7, idx = bid >> 3;
    const int g0x = (int)(((long)xcd * n_groups) >> 3), g1x = (int)(((long)(xcd + 1) * n_groups) >> 3);
    const int gl = idx / tg, wi = idx - gl * tg;
    const int G = g0x + gl;
    if (G >= g1x) return;
    const int gi = G / NSTRIP, s = G - gi * NSTRIP;
    const int n = gi * tg + wi;
    if (n >= a.n_maps) return;
    const long img = a.map2img ? a.map2img[n] : n;
    const int kA = split_scale_exp<true>(a.in_amax[n]);
    const float ssc = exp2i(kA);
    const float f = exp2i(-kA) * a.wp[0];
    constexpr int P = HW * HW, PO = HO * HO;
    const long total_pix = (long)a.n_maps * P;

    // ---- resident B fragments: [chunk][g][dx] fp16 (4 registers each), [chunk][mm] fp6 + block scale (dwords 0-5, 6) ----
    f16x8 bh[4][3][3];
    i32x8_ bm[4][5];
    {
        const u32x4_* wp = reinterpret_cast<const u32x4_*>(a.wp + F16X3_HEADER_FLOATS) + lane;
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const u32x4_* e = wp + (long)(((och * 4 + c) * 3 + g) * 7) * 64;
#pragma unroll
                for (int p = 0; p < 3; ++p) bh[c][g][p] = __builtin_bit_cast(f16x8, e[p * 64]);
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    if (2 * g + mm > 4) continue;
                    const u32x4_ lo = e[(3 + 2 * mm) * 64];
                    const u32x3_ hi = *reinterpret_cast<const u32x3_*>(e + (4 + 2 * mm) * 64);
                    bm[c][2 * g + mm] = i32x8_{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], 0};
                }
            }
    }

    // ---- staging roles (fixed per thread).  Main: pooled pixel pcm = 16 s + (tid & 15) of pooled row 2 jj + pr, chunk = wave, window row dy:
    //      both window columns.  Halo (lanes 0-7): the pooled pixel left (16 s - 1) / right (16 s + 16) of the strip, one window column.
    const int pcI = tid & 15, dy = (tid >> 4) & 1, pr = (tid >> 5) & 1, chs = wave;
    const int h_side = lane & 1, h_dy = (lane >> 1) & 1, h_pr = (lane >> 2) & 1;
    const bool h_lane = lane < 8;
    const int h_pc = h_side ? 16 * s + 16 : 16 * s - 1;
    const bool h_ok = h_pc >= 0 && h_pc < HO;
    const long CS = blk_chunk_stride((long)a.n_maps * PO);
    const float* __restrict__ in_c = a.in + (long)chs * CS;
    const unsigned char* __restrict__ am_c = a.pool_am + img * PO * 64 + chs * 16;
    const int gp_m = n * PO + pr * HO + 16 * s + pcI;                 // + 2 jj * HO
    const int gp_h = n * PO + h_pr * HO + (h_ok ? h_pc : 16 * s);
    const int ap_m = pr * HO + 16 * s + pcI, ap_h = h_pr * HO + (h_ok ? h_pc : 16 * s);
    // LDS destinations: plane of the chunk + column; the ring slot of the row is added per stage
    const int ld_m = chs * PLANE + (2 * pcI + 1) * PXB;               // window column dx = 0 (dx = 1: + PXB)
    const int ld_h = chs * PLANE + (h_side ? 33 : 0) * PXB;

    f32x4 sv[4], svh[4];
    u32x4_ amv, amh;
    auto issue = [&](const int jj) {          // loads of stage jj (pooled rows 2 jj, 2 jj + 1); jj <= 55
        const float* sp = in_c + blk_pix_off32(gp_m + 2 * jj * HO);
        const float* sh = in_c + blk_pix_off32(gp_h + 2 * jj * HO);
#pragma unroll
        for (int k = 0; k < 4; ++k) { sv[k] = reinterpret_cast<const f32x4*>(sp)[k * 32]; svh[k] = reinterpret_cast<const f32x4*>(sh)[k * 32]; }
        amv = *reinterpret_cast<const u32x4_*>(am_c + (long)(ap_m + 2 * jj * HO) * 64);
        amh = *reinterpret_cast<const u32x4_*>(am_c + (long)(ap_h + 2 * jj * HO) * 64);
    };
    auto slot_of = [](const int row) { return ((row + 2) % RING) * PITCH; };      // rows >= -2
    auto commit = [&](const int jj, const bool live) {      // rows 4 jj .. 4 jj + 3 (live = false: beyond the map, zeros)
        unsigned hwu[8], rwu[8], sb;
        float bs;
        {
            x6_split(sv, live ? ssc : 0.f, hwu, rwu, bs, sb);
            char* d = ldsb + ld_m + slot_of(4 * jj + 2 * pr + dy);
            s12_commit_pos(hwu, rwu, bs, sb, amv, 2u * dy, d);
            s12_commit_pos(hwu, rwu, bs, sb, amv, 2u * dy + 1u, d + PXB);
        }
        {
            x6_split(svh, (live && h_ok) ? ssc : 0.f, hwu, rwu, bs, sb);
            char* d = h_lane ? ldsb + ld_h + slot_of(4 * jj + 2 * h_pr + h_dy) : ldsb + DUMMY;
            s12_commit_pos(hwu, rwu, bs, sb, amh, 2u * h_dy + (h_side ? 0u : 1u), d);
        }
    };

    // ---- prologue: rows -2, -1 are zeros; stage 0 ----
    issue(0);
    for (int i = tid; i < 4 * 2 * PITCH / 16; i += 256) {
        const int pl = i / (2 * PITCH / 16), r = i - pl * (2 * PITCH / 16);
        reinterpret_cast<u32x4_*>(ldsb + pl * PLANE)[r] = u32x4_{0, 0, 0, 0};
    }
    commit(0, true);
    __syncthreads();

    f32x16 acc[2];
    const int lane_h = li * PXB + lh * 16, lane_6 = li * PXB + 32;
    // epilogue addressing: lane li holds channel perm_row_channel(li) of its half, pixels (e & 3) + 8 (e >> 2) + 4 lh of the tile's row segment
    const int chl = perm_row_channel(li);
    const char* __restrict__ Xb = reinterpret_cast<const char*>(a.X) + ((img * P + 32 * s + 4 * lh) * 64 + och * 32 + chl) * 4;
    char* __restrict__ Ob = reinterpret_cast<char*>(a.out1 ? a.out1 : a.out0) +
                            (((long)och * total_pix + (long)n * P + 32 * s + 4 * lh) * 32 + chl) * 4;
    float mx = 0.f;
    int base = 2 * pg;                                  // ring slot of input row (4 j - 2) + 2 pg, in rows: (4 j + 2 pg) % RING

    for (int j = 0; j < NSTEP; ++j) {
        const int r0 = 4 * j - 1 + 2 * pg;              // output row of tile 0 (tile 1: + 1)
        const bool v0 = r0 >= 0 && r0 < HW, v1 = r0 + 1 < HW;       // (wave-uniform)
        // multiplicands of the step's two tiles and the staging loads of the next stage: in flight during the MFMAs
        float xv[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int rr = min(max(r0 + t, 0), HW - 1);
            const char* xr = Xb + (long)rr * (HW * 64 * 4);
#pragma unroll
            for (int e = 0; e < 16; ++e) xv[t][e] = *reinterpret_cast<const float*>(xr + ((e & 3) + 8 * (e >> 2)) * 64 * 4);
        }
        const bool more = j + 1 < NSTEP - 1;            // stage j + 1 has pooled rows inside the map (j + 1 <= 55)
        if (!(dbg & 8)) issue(more ? j + 1 : 0);
        int sl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int b = base + k; sl[k] = __builtin_amdgcn_readfirstlane((b >= RING ? b - RING : b) * PITCH); }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

        auto rd = [&](const int k) -> i32x8_ {
            const int c = op_chunk(k), g = op_g(k), t = op_t(k), m = op_m(k);
            if (m < 2) {
                const int mm = 2 * g + m, ta = 2 * mm, tb = ta + 1 > 8 ? 8 : ta + 1;
                const int oa = sl[t + ta / 3] + (ta % 3) * PXB, ob = sl[t + tb / 3] + (tb % 3) * PXB;
                const char* p = ldsb + c * PLANE + lane_6 + (lh ? ob : oa);
                const u32x4_ x0 = *reinterpret_cast<const u32x4_*>(p);
                const u32x4_ x1 = *reinterpret_cast<const u32x4_*>(p + 16);
                return i32x8_{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[3], (int)x1[2]};
            }
            const char* p = ldsb + c * PLANE + lane_h + sl[t + g] + (m - 2) * PXB;
            const u32x4_ x0 = *reinterpret_cast<const u32x4_*>(p);
            return i32x8_{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], 0, 0, 0, 0};
        };
        constexpr int D = 3;
        i32x8_ ring[D];
#pragma unroll
        for (int d = 0; d < D; ++d) ring[d] = rd(d);
        if (!(dbg & 2))
#pragma unroll
        for (int k = 0; k < NOPS; ++k) {
            const int c = op_chunk(k), g = op_g(k), t = op_t(k), m = op_m(k);
            const i32x8_ cur = ring[k % D];
            if (k + D < NOPS) {
                ring[k % D] = rd(k + D);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (m < 2) {
                const i32x8_ b = bm[c][2 * g + m];
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cur, b, acc[t], 2, 2, 0, cur[6], 0, b[6]);
                const int keep = cur[7];
                asm volatile("" : : "v"(keep));
            } else {
                const u32x4_ c4 = {(unsigned)cur[0], (unsigned)cur[1], (unsigned)cur[2], (unsigned)cur[3]};
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, c4), bh[c][g][m - 2], acc[t], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }

        // ---- epilogue of the step's two tiles: out = x * (acc * 2^-kA 2^-kW), per-map maximum of what is stored ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if ((t == 0 ? v0 : v1) && !(dbg & 4)) {
                char* orow = Ob + (long)(r0 + t) * (HW * 32 * 4);
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float r = xv[t][e] * (acc[t][e] * f);
                    *reinterpret_cast<float*>(orow + ((e & 3) + 8 * (e >> 2)) * 32 * 4) = r;
                    mx = fmaxf(mx, fabsf(r));
                }
            }
        }
        // ---- rows 4 (j + 1) .. 4 (j + 1) + 3 into the slots whose rows the step before this one read last ----
        if (j + 1 < NSTEP && !(dbg & 1)) commit(j + 1, more);
        __syncthreads();
        base += 4;
        if (base >= RING) base -= RING;
    }
    unsigned* __restrict__ oamax = a.out1 ? a.out1_amax : nullptr;
    if (oamax) {
        mx = wave_max(mx);
        if (lane == 0) amax_update(&oamax[n], mx);
    }
}

int launch_strip12_224_pool(const ConvArgs& a, hipStream_t stream) {
    using namespace s12;
    LRPX_REQUIRE(a.cin == 64 && a.n_oc == 64 && a.oc_split == 64 && a.out_chunk == 32 && a.pool_am && a.in_amax && a.X && !a.in_chunk_stride &&
                     a.epi == EPI_REL_MUL && (a.out1 || a.out0),
                 "conv12_strip: built for conv1_2's relevance step (64 -> 64 channels, pooled blocked input, 32-channel-chunk output)");
    LRPX_REQUIRE((long)a.n_maps * HO * HO < (1L << 27), "conv12_strip: too many pooled pixels for 32-bit blocked offsets");
    const int tg = (a.tile_group > 1 && a.n_maps % a.tile_group == 0) ? a.tile_group : 1;
    const int n_groups = (a.n_maps / tg) * NSTRIP;
    int per_xcd = 0;
    for (int x = 0; x < 8; ++x) per_xcd = std::max(per_xcd, (int)((((long)(x + 1) * n_groups) >> 3) - (((long)x * n_groups) >> 3)));
    const long grid = (long)per_xcd * tg * 8;
    static LdsOnce attr_once;
    LRPX_TRY(reserve_lds_once(attr_once, conv12_strip_kernel, LDS_BYTES, "conv12_strip"));
    static const int dbg = getenv("LRPX_S12_DBG") ? atoi(getenv("LRPX_S12_DBG")) : 0;
    hipLaunchKernelGGL(conv12_strip_kernel, dim3((unsigned)grid), dim3(256), LDS_BYTES, stream, a, n_groups, tg, dbg);
    return check_launch("conv12_strip");
}

}  // namespace lrpx
