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
constexpr int LDS_BYTES = LDS_PIX + 256 + 4 * 8192;     // + the waves' epilogue patches (2 tiles x 32 x 32 floats each)
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

template <int D>
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
    // loads: wave-uniform 64-bit bases (the chunk's plane of S, the image's winner bytes of that chunk) + 32-bit lane offsets; a stage is two
    // pooled rows = 224 pooled pixels = 7 blocks of the blocked layout further on: + 14 336 bytes in either tensor
    const char* __restrict__ in_c = reinterpret_cast<const char*>(a.in + (long)chs * CS);
    const char* __restrict__ am_c = reinterpret_cast<const char*>(a.pool_am) + img * PO * 64 + chs * 16;
    constexpr unsigned STAGE_B = 2 * HO * 64;
    static_assert((2 * HO) % 32 == 0 && (2 * HO / 32) * 512 * 4 == (int)STAGE_B, "a stage advances both tensors by the same whole number of blocks");
    const int hp = h_ok ? h_pc : 16 * s;
    const unsigned om0 = blk_pix_off32(n * PO + pr * HO + 16 * s + pcI) * 4u, oh0 = blk_pix_off32(n * PO + h_pr * HO + hp) * 4u;
    const unsigned am0 = (unsigned)(pr * HO + 16 * s + pcI) * 64u, ah0 = (unsigned)(h_pr * HO + hp) * 64u;
    // LDS destinations: plane of the chunk + column; the ring slot of the row is added per stage
    const int ld_m = chs * PLANE + (2 * pcI + 1) * PXB;               // window column dx = 0 (dx = 1: + PXB)
    const int ld_h = chs * PLANE + (h_side ? 33 : 0) * PXB;

    // one register set for both tasks of a thread: the main slice of a stage is loaded one step ahead; the halo slice follows into the same
    // registers once the main one has been split (sv) / its masks are done (amv)
    f32x4 sv[4];
    u32x4_ amv;
    auto load_main = [&](const int i, const int jj) {       // i = 0..3: the slice's four parts, 4: its winner bytes; stage jj <= 55
        if (i < 4) sv[i] = *reinterpret_cast<const f32x4*>(in_c + (om0 + (unsigned)jj * STAGE_B + (unsigned)i * 512u));
        else amv = *reinterpret_cast<const u32x4_*>(am_c + (am0 + (unsigned)jj * STAGE_B));
    };
    auto load_halo = [&](const int i, const int jj) {
        if (i < 4) sv[i] = *reinterpret_cast<const f32x4*>(in_c + (oh0 + (unsigned)jj * STAGE_B + (unsigned)i * 512u));
        else amv = *reinterpret_cast<const u32x4_*>(am_c + (ah0 + (unsigned)jj * STAGE_B));
    };
    auto slot_of = [](const int row) { return ((row + 2) % RING) * PITCH; };      // rows >= -2

    // ---- the commit of a stage as 51 slices of a few instructions each (placed between the MFMAs of a step) ----
    unsigned c_hw[8], c_rw[8], c_hm[8], c_rm[8], c_sb = 0;
    float c_m = 0.f, c_bs = 0.f, c_sc = 0.f;
    u32x6_ c_q6 = {0, 0, 0, 0, 0, 0};
    char* c_d = ldsb + DUMMY;
    auto split_pair = [&](const f32x4 (&v)[4], const int c2) {           // x6_split, one channel pair
        const int c = 2 * c2;
        const f32x2_ x2 = f32x2_{v[c >> 2][c & 3], v[c >> 2][(c & 3) + 1]} * f32x2_{c_sc, c_sc};
        const f16x2_ h = __builtin_convertvector(x2, f16x2_);
        const f32x2_ hf = __builtin_convertvector(h, f32x2_);
        const f16x2_ r = __builtin_convertvector((x2 - hf) * f32x2_{2048.f, 2048.f}, f16x2_);
        c_hw[c2] = __builtin_bit_cast(unsigned, h);
        c_rw[c2] = __builtin_bit_cast(unsigned, r);
        c_m = fmaxf(c_m, fmaxf(fabsf(x2[0]), fabsf(x2[1])));
    };
    auto split_fin = [&]() {
        c_bs = c_m * (16.f / 15.f * 0.25f);
        const int e = (int)((__builtin_bit_cast(unsigned, c_bs) >> 23) & 0xffu);
        c_sb = (unsigned)max(e - 11, 0);
    };
    auto mask_q = [&](const u32x4_ am, const unsigned pos, const int q) {
        const unsigned x = am[q] ^ (0x01010101u * pos);
        const unsigned eq = ((x | (x >> 1)) & 0x01010101u) ^ 0x01010101u;
        const unsigned sel = eq + 0x0c0c0c0cu;
        const unsigned m01 = __builtin_amdgcn_perm(0u, 0u, __builtin_amdgcn_perm(sel, sel, 0x01010000u));
        const unsigned m23 = __builtin_amdgcn_perm(0u, 0u, __builtin_amdgcn_perm(sel, sel, 0x03030202u));
        c_hm[2 * q] = c_hw[2 * q] & m01; c_hm[2 * q + 1] = c_hw[2 * q + 1] & m23;
        c_rm[2 * q] = c_rw[2 * q] & m01; c_rm[2 * q + 1] = c_rw[2 * q + 1] & m23;
    };
    auto write_half = [&](const int h) {
        if (h == 0) {
            *reinterpret_cast<u32x4_*>(c_d) = u32x4_{c_hm[0], c_hm[1], c_hm[2], c_hm[3]};
            *reinterpret_cast<u32x4_*>(c_d + 16) = u32x4_{c_hm[4], c_hm[5], c_hm[6], c_hm[7]};
        } else {
            *reinterpret_cast<u32x4_*>(c_d + 32) = u32x4_{c_q6[0], c_q6[1], c_q6[2], c_q6[3]};
            *reinterpret_cast<u32x4_*>(c_d + 48) = u32x4_{c_q6[4], c_q6[5], 0u, c_sb};
        }
    };
    constexpr int NSLICE = 51;
    // slice i of the commit of stage jj (live = false: beyond the map, zeros)
    auto commit_slice = [&](const int i, const int jj, const bool live) {
        if (i == 0) { c_m = 0.f; c_sc = live ? ssc : 0.f; }
        if (i < 8) { split_pair(sv, i); return; }
        if (i == 8) { split_fin(); c_d = ldsb + ld_m + slot_of(4 * jj + 2 * pr + dy); return; }
        if (i < 31) {                          // the two window columns of the main task: 9..19, 20..30
            const int p = (i - 9) / 11, r = (i - 9) % 11;
            if (r < 4) { mask_q(amv, 2u * dy + (unsigned)p, r); if (r == 3) c_q6 = x6_pack(c_hm, c_rm, c_bs); return; }
            if (r < 8) return;                 // (room: the conversion above is a multi-pass instruction)
            if (r == 8) { write_half(0); return; }
            if (r == 9) { write_half(1); return; }
            c_d += PXB;
            return;
        }
        if (i == 31) { c_m = 0.f; c_sc = (live && h_ok) ? ssc : 0.f; }
        if (i < 39) { split_pair(sv, i - 31); return; }
        if (i == 39) { split_fin(); c_d = h_lane ? ldsb + ld_h + slot_of(4 * jj + 2 * h_pr + h_dy) : ldsb + DUMMY; return; }
        if (i < 44) { mask_q(amv, 2u * h_dy + (h_side ? 0u : 1u), i - 40); if (i == 43) c_q6 = x6_pack(c_hm, c_rm, c_bs); return; }
        if (i == 49) { write_half(0); return; }
        if (i == 50) { write_half(1); return; }
    };
    // ---- prologue: rows -2, -1 are zeros; stage 0; the main loads of stage 1 ----
#pragma unroll
    for (int i = 0; i < 5; ++i) load_main(i, 0);
    for (int i = tid; i < 4 * 2 * PITCH / 16; i += 256) {
        const int pl = i / (2 * PITCH / 16), r = i - pl * (2 * PITCH / 16);
        reinterpret_cast<u32x4_*>(ldsb + pl * PLANE)[r] = u32x4_{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < 31; ++i) commit_slice(i, 0, true);
#pragma unroll
    for (int i = 0; i < 5; ++i) load_halo(i, 0);
#pragma unroll
    for (int i = 31; i < NSLICE; ++i) commit_slice(i, 0, true);
#pragma unroll
    for (int i = 0; i < 5; ++i) load_main(i, 1);
    __syncthreads();

    f32x16 acc[2];
    const int lane_h = li * PXB + lh * 16, lane_6 = li * PXB + 32;
    // Epilogue through a wave-private LDS patch per tile (32 pixels x 32 channels, unpadded rows: conflict-free for the dword writes of
    // the MFMA layout - lane li owns channel perm_row_channel(li), pixels (e & 3) + 8 (e >> 2) + 4 lh - and for the float4 reads):
    // read back, a lane owns channels 4 (lane % 8) .. + 3 of pixels lane / 8 + 8 q, q = 0..3 - every global access of the epilogue is then
    // 1 KB contiguous per wave instruction (8 whole lines): 8 loads + 8 stores per wave and step instead of 32 + 32 dword accesses, which
    // cost this kernel a fifth of its time in address processing.  The read-back half runs between the MFMAs of the NEXT step.
    char* scr = ldsb + LDS_PIX + 256 + wave * 8192;
    const int chl = perm_row_channel(li);
    const int scr_w = (4 * lh * 32 + chl) * 4;                     // + ((e & 3) + 8 (e >> 2)) * 128 + t * 4096
    const int scr_r = ((lane >> 3) * 32 + 4 * (lane & 7)) * 4;     // + q * 8 * 128 + t * 4096
    const char* __restrict__ Xb = reinterpret_cast<const char*>(a.X) + ((img * P + 32 * s + (lane >> 3)) * 64 + och * 32 + 4 * (lane & 7)) * 4;
    char* __restrict__ Ob = reinterpret_cast<char*>(a.out1 ? a.out1 : a.out0) +
                            (((long)och * total_pix + (long)n * P + 32 * s + (lane >> 3)) * 32 + 4 * (lane & 7)) * 4;
    f32x4 xv[2][4];
    f32x4 ev[2];
    float mx = 0.f;
    int base = 2 * pg;                                  // ring slot of input row (4 j - 2) + 2 pg, in rows: (4 j + 2 pg) % RING
    auto xload = [&](const int u, const int r0) {       // multiplicand of unit u = (tile t, pixel group q) of the step whose tile 0 is row r0
        const int t = u >> 2, q = u & 3;
        const int rr = min(max(r0 + t, 0), HW - 1);
        xv[t][q] = *reinterpret_cast<const f32x4*>(Xb + (long)rr * (HW * 64 * 4) + q * (8 * 64 * 4));
    };
    auto epi_read = [&](const int u) { ev[u % 2] = *reinterpret_cast<const f32x4*>(scr + (u >> 2) * 4096 + scr_r + (u & 3) * (8 * 128)); };
    auto epi_fin = [&](const int u, const int r0) {     // out = x * (acc * 2^-kA 2^-kW), per-map maximum of what is stored
        const int t = u >> 2, q = u & 3;
        const int row = r0 + t;
        const f32x4 r = xv[t][q] * (ev[u % 2] * f32x4{f, f, f, f});
        if (row >= 0 && row < HW) {
            *reinterpret_cast<f32x4*>(Ob + (long)row * (HW * 32 * 4) + q * (8 * 32 * 4)) = r;
            mx = fmaxf(fmaxf(mx, fmaxf(fabsf(r[0]), fabsf(r[1]))), fmaxf(fabsf(r[2]), fabsf(r[3])));
        }
    };

    for (int j = 0; j < NSTEP; ++j) {
        const int r0 = 4 * j - 1 + 2 * pg;              // output row of tile 0 (tile 1: + 1)
        const bool live = j + 1 < NSTEP - 1;            // stage j + 1 has pooled rows inside the map (j + 1 <= 55)
        const int jl = j + 2 < NSTEP - 1 ? j + 2 : 0;           // the stage whose main loads this step issues (beyond the map: any)
        const int jh = live ? j + 1 : 0;                        // ... whose halo loads
        int sl[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int b = base + k; sl[k] = __builtin_amdgcn_readfirstlane((b >= RING ? b - RING : b) * PITCH); }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

        // the operand of op k: 16 bytes (fp16 k-step) or 32 bytes (fp6 MFMA: 24 bytes of fields, a clear dword, the block-scale dword) of one
        // LDS pixel per lane, kept in the order they are read in (the scale is handed to the MFMA from where it lands: no register moves)
        auto rd = [&](const int k) -> i32x8_ {
            const int c = op_chunk(k), g = op_g(k), t = op_t(k), m = op_m(k);
            if (m < 2) {
                const int mm = 2 * g + m, ta = 2 * mm, tb = ta + 1 > 8 ? 8 : ta + 1;
                const int oa = sl[t + ta / 3] + (ta % 3) * PXB, ob = sl[t + tb / 3] + (tb % 3) * PXB;
                const char* p = ldsb + c * PLANE + lane_6 + (lh ? ob : oa);
                const u32x4_ x0 = *reinterpret_cast<const u32x4_*>(p);
                const u32x4_ x1 = *reinterpret_cast<const u32x4_*>(p + 16);
                return i32x8_{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
            }
            const char* p = ldsb + c * PLANE + lane_h + sl[t + g] + (m - 2) * PXB;
            const u32x4_ x0 = *reinterpret_cast<const u32x4_*>(p);
            return i32x8_{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], 0, 0, 0, 0};
        };
        // what runs in the gap behind MFMA k (compile-time k): the read-back half of the previous step's epilogue, this step's
        // multiplicand loads, the commit of the next stage, the loads of the stage after it
        auto gap = [&](const int k) {
            // (nothing here is conditional on the step: a value that is only sometimes redefined stays live around the whole loop.  Step 0
            // reads back an unwritten patch for rows < 0 - never stored -, the last step commits a stage nobody reads.)
            // the multiplicands of the PREVIOUS step's tiles arrive in gaps 4..11 and are used in gaps 41..48: their 32 registers are dead
            // while the commit slices hold theirs (rows shared by the words of an image: L2 hits, ~30 MFMAs of lead)
            if (k >= 4 && k < 12) xload(k - 4, r0 - 4);
            if (k >= 40 && k < 48) epi_read(k - 40);
            if (k >= 41 && k < 49) epi_fin(k - 41, r0 - 4);
            // commit of stage j + 1: main slices 0..30 in gaps 49..79; its halo slice is loaded into the same registers as soon as they are
            // free (gaps 57 / 58: data, 73: winner bytes) and committed in gaps 88..107 (slices 31..50)
            if (k >= 49 && k < 80) commit_slice(k - 49, j + 1, live);
            if (k == 57) { load_halo(0, jh); load_halo(1, jh); }
            if (k == 58) { load_halo(2, jh); load_halo(3, jh); }
            if (k == 73) load_halo(4, jh);
            if (k >= 88 && k < 108) commit_slice(k - 88 + 31, j + 1, live);
            // behind the last slice: the main loads of stage j + 2
            if (k >= 108) { load_main(k - 108, jl); if (k == 111) load_main(4, jl); }
        };
        i32x8_ ring[D];
#pragma unroll
        for (int d = 0; d < D; ++d) ring[d] = rd(d);
#pragma unroll
        for (int k = 0; k < NOPS; ++k) {
            const int c = op_chunk(k), g = op_g(k), t = op_t(k), m = op_m(k);
            const i32x8_ cur = ring[k % D];
            if (k + D < NOPS) {
                ring[k % D] = rd(k + D);
                __builtin_amdgcn_sched_barrier(0);
            }
            {
            if (m < 2) {
                const i32x8_ b = bm[c][2 * g + m];
                acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cur, b, acc[t], 2, 2, 0, cur[7], 0, b[6]);
                const int keep = cur[6];
                asm volatile("" : : "v"(keep));
            } else {
                const u32x4_ c4 = {(unsigned)cur[0], (unsigned)cur[1], (unsigned)cur[2], (unsigned)cur[3]};
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, c4), bh[c][g][m - 2], acc[t], 0, 0, 0);
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            gap(k);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the step's two tiles into the wave's LDS patches (their read-back: gaps 0..9 of the next step) ----
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                *reinterpret_cast<float*>(scr + t * 4096 + scr_w + ((e & 3) + 8 * (e >> 2)) * 128) = acc[t][e];
        __syncthreads();
        base += 4;
        if (base >= RING) base -= RING;
    }
    {       // epilogue of the last step
        const int r0 = 4 * (NSTEP - 1) - 1 + 2 * pg;
#pragma unroll
        for (int u = 0; u < 8; ++u) xload(u, r0);
#pragma unroll
        for (int u = 0; u < 8; ++u) { epi_read(u); epi_fin(u, r0); }
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
    static const int dbg = getenv("LRPX_S12_DBG") ? atoi(getenv("LRPX_S12_DBG")) : 0;
    static const int depth = getenv("LRPX_S12_D") ? atoi(getenv("LRPX_S12_D")) : 4;
#define S12_LAUNCH(DD)                                                                                                   \
    {                                                                                                                    \
        static LdsOnce attr_once;                                                                                        \
        LRPX_TRY(reserve_lds_once(attr_once, conv12_strip_kernel<DD>, LDS_BYTES, "conv12_strip"));                       \
        hipLaunchKernelGGL(conv12_strip_kernel<DD>, dim3((unsigned)grid), dim3(256), LDS_BYTES, stream, a, n_groups, tg, dbg); \
    }
    if (depth <= 2) S12_LAUNCH(2) else if (depth <= 4) S12_LAUNCH(4) else S12_LAUNCH(6)
#undef S12_LAUNCH
    return check_launch("conv12_strip");
}

}  // namespace lrpx
