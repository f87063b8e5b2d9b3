// fp32-grade 3x3 convolution on the fp16 matrix cores ("f16x3"): both operands are scaled by a power of two into the
// fp16 range and split into two halves a = a0 + a1 (11 + 11 significand bits, |a - a0 - a1| <= 2^-22 |a|); the three
// products
//        a*b ~= a0b0 + (a0b1 + a1b0)
// are accumulated in fp32 by v_mfma_f32_32x32x16_f16.  The dropped a1b1 term and the split residue are <= 2^-21
// relative PER PRODUCT with random sign - an order of magnitude below the rounding of the fp32 accumulation over
// K = 576..4608 terms that every fp32 implementation (the reference's included) carries - while 3 fp16 MFMAs of K=16
// cost 96 SIMD-cycles against 192 for bf16x6 and 512 on the fp32 MFMA.
//
// Range: fp16 has 5 exponent bits, so the A operand (S = R / Z, any magnitude) is scaled PER MAP by 2^kA with
// max|S_map| * 2^kA in [2^14, 2^15); the producer of S records max|S| per map (`amax`, float bits, atomicMax) and this
// kernel records it for the S it writes (out1).  Weights are scaled per layer at pack time (2^kW, header of the packed
// blob).  Entries more than ~2^29 below their map's maximum flush to zero: an absolute error of 2^-29 max|S|, far
// inside the 1e-4 * max|R| contract of the path.  The epilogue multiplies the accumulators by 2^-kA * 2^-kW (exact).
// An amax that is too small (stale) overflows to inf and is caught by lrpx_check - never silently wrong.
//
// Tiling, LDS double buffering, B-fragment queue and epilogues as conv_bf16x6.h; LDS pixel = 2 planes x 16 fp16 (32 B
// each) + 16 B pad = 80 B; row pitch 80*W + 256 so that the 16-byte slot of tile pixel q is 5q + const (mod 16) across
// image-row wraps (conflict-free ds_read_b128).
#pragma once
#include "conv_mfma.h"
#include "conv_bf16x6.h"
#include "blocked.h"

namespace lrpx {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
typedef float f32x2_ __attribute__((ext_vector_type(2)));

constexpr int F16X3_HEADER_FLOATS = 16;   // packed weights: [0] = 2^-kW, [1] = bits of max|W| (diagnostic)

// exponent k with amax * 2^k in [2^14, 2^15); 0 for an all-zero / denormal-only / non-finite tensor
__host__ __device__ __forceinline__ int f16_scale_exp(unsigned amax_bits) {
    const int e = (int)((amax_bits >> 23) & 0xff);
    if (e == 0 || e == 255) return 0;
    const int k = 14 - (e - 127);
    return k > 120 ? 120 : k;
}
// Which narrow format carries the two cross products of the F8 kernels: 1 = block-scaled fp6 e2m3 (v_mfma_scale_f32_32x32x64_f8f6f4
// at 32 cycles per K = 64, as fast as one fp16 MFMA of K = 16), 0 = fp8 e4m3 (64 cycles; rounds 1-2 of this kernel)
#ifndef LRPXH_XP6
#define LRPXH_XP6 1
#endif
// F8 variant (cross products on the fp8 matrix cores, see below): operands are scaled into [2^11, 2^12) instead, so
// that x * 2^-4 fits fp8 e4m3 (max 448) and the fp16 residual * 2^4 does too.  (fp6: every 16-channel slice carries its own
// block exponent, the operands keep the fp16 range.)
template <bool F8>
__host__ __device__ __forceinline__ int split_scale_exp(unsigned amax_bits) {
    if constexpr (!F8 || (LRPXH_XP6 != 0)) return f16_scale_exp(amax_bits);
    const int e = (int)((amax_bits >> 23) & 0xff);
    if (e == 0 || e == 255) return 0;
    const int k = 11 - (e - 127);
    return k > 120 ? 120 : k;
}
__host__ __device__ __forceinline__ float exp2i(int k) {   // 2^k, |k| <= 126
    const unsigned b = (unsigned)(k + 127) << 23;
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_bit_cast(float, b);
#else
    float f; __builtin_memcpy(&f, &b, 4); return f;
#endif
}

__device__ __forceinline__ unsigned pack_f16(_Float16 a, _Float16 b) {
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}
// two-way split of an (already scaled) value: x ~= hi + lo
__device__ __forceinline__ void split2(float x, _Float16& hi, _Float16& lo) {
    hi = (_Float16)x;                 // v_cvt_f16_f32, round to nearest even; inf / nan stay inf / nan
    lo = (_Float16)(x - (float)hi);   // exact difference in fp32, rounded once
}

// the same for two values at once, results as packed fp16 pairs: v_cvt_pk_f16_f32 (gfx950) converts and packs in one instruction
// (round to nearest even, bit-identical to two scalar conversions: 5 VALU instructions per pair instead of 10)
typedef _Float16 f16x2_ __attribute__((ext_vector_type(2)));
#ifndef LRPXH_PK_CVT
#define LRPXH_PK_CVT 1
#endif
__device__ __forceinline__ void split2_pk(const f32x2_ x, unsigned& hi, unsigned& lo, f32x2_& hi_f) {
    if constexpr (LRPXH_PK_CVT != 0) {
        const f16x2_ h = __builtin_convertvector(x, f16x2_);
        hi_f = __builtin_convertvector(h, f32x2_);
        const f16x2_ l = __builtin_convertvector(x - hi_f, f16x2_);
        hi = __builtin_bit_cast(unsigned, h);
        lo = __builtin_bit_cast(unsigned, l);
    } else {
        _Float16 h0, h1, l0, l1;
        split2(x[0], h0, l0); split2(x[1], h1, l1);
        hi_f = f32x2_{(float)h0, (float)h1};
        hi = pack_f16(h0, h1); lo = pack_f16(l0, l1);
    }
}

// B6 (conv mode 1, "bf16x6" in this kernel's tiling): EXACT three-way split of two values at once, x == p0 + p1 + p2 with every part a
// bf16 (8 + 8 + 8 significand bits, fp32's exponent range: no operand scale, no per-map maximum), packed pairs; the same roundings as
// split3 of conv_bf16x6.h (v_cvt_pk_bf16_f32, round to nearest even), 10 VALU instructions per pair
typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2_ bf16pair_to_f32(const unsigned p) {
    return f32x2_{__builtin_bit_cast(float, p << 16), __builtin_bit_cast(float, p & 0xffff0000u)};
}
__device__ __forceinline__ void split3_pk(const f32x2_ x, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_));
    const f32x2_ r1 = x - bf16pair_to_f32(p0);
    p1 = __builtin_bit_cast(unsigned, __builtin_convertvector(r1, bf16x2_));
    const f32x2_ r2 = r1 - bf16pair_to_f32(p1);
    p2 = __builtin_bit_cast(unsigned, __builtin_convertvector(r2, bf16x2_));
}

// two / four floats -> fp8 e4m3 (OCP, round to nearest even), packed
__device__ __forceinline__ unsigned pack_fp8x4(float x0, float x1, float x2, float x3) {
    int w = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, 0, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(x2, x3, w, true);
    return (unsigned)w;
}
typedef int i32x8_ __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// amax[.] = max(amax[.], m) on float bits.  Up to a million callers update a few hundred words: the plain cached read
// first turns all but the first few updates per word into an L1/L2 hit - a stale (smaller) value only costs a
// redundant atomic, never a wrong maximum.
__device__ __forceinline__ void amax_update(unsigned* p, float m) {
    const unsigned b = __builtin_bit_cast(unsigned, m);
    if (m > 0.f && b > *reinterpret_cast<const volatile unsigned*>(p)) atomicMax(p, b);
}

#ifdef LRPX_STAMP
// profiling build only (make STAMP=1): per-phase shader-clock totals of the kernels with HW == LRPX_STAMP_HW
#ifndef LRPX_STAMP_HW
#define LRPX_STAMP_HW 224
#endif
static __device__ unsigned long long g_stamp_h3[12];
#define LRPXH_T(v) unsigned long long v; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory"); \
                   __builtin_amdgcn_sched_barrier(0)
#else
#define LRPXH_T(v)
#endif

// F8 slot map (shared with pack_weights_f16f8_kernel).  The 18 cross-product (tap, 16 channels) K-slices of a K-chunk -
// "L": (x - hi) * W and "S": x * (W - hi_W), taps 0-8 each - fill five K = 64 fp8 MFMAs of four slices: MFMA m covers
// taps 2m and 2m+1 (tap 9 does not exist: zero weights), lanes 0-31 (k = 0..31) hold the L slices of both taps, lanes
// 32-63 (k = 32..63) the S slices.  The two lane halves then read the same pixels, 16 bytes apart (planes 48 / 32).
__host__ __device__ constexpr int f8_slot_tap(int m, int i) { return 2 * m + i; }        // > 8: zero weights

// REL_MUL epilogue with 16-byte accesses ("wide"): out = x * acc for the wave's 7 accumulator tiles of 32 pixels x 32
// channels.  In the MFMA result layout a lane owns ONE channel of 16 pixels, so the plain epilogue issues 112 dword loads
// and 112 dword stores per lane, each touching two 128-byte runs - measured 21 % of a workgroup tile's time on the 28x28
// layers (s_memtime stamps), not bandwidth: the memory pipeline of the CU moves 256 bytes per instruction.  Here every
// tile goes through a wave-private LDS patch (32 rows x 36 floats; the staging buffers are free after the K loop) and
// comes back as float4 along the channels: lane L owns the channel quad L % 8 of pixel rows L / 8 + 8k, k = 0..3 - 4
// loads + 4 stores of 16 bytes per lane and tile, 4x fewer memory instructions, whole 128-byte runs per 8 lanes.
// Arithmetic, results and the per-map maxima are the same as in epi_gather / epi_finish (bit-identical outputs).
template <int HW, bool AL, int EPI>
__device__ __forceinline__ void epi_rel_mul_wide(const ConvArgs& a, f32x16 (&acc)[7], float* __restrict__ scr, const int wm,
                                                 const int ocb, const int lane, const long g0, const long total_pix,
                                                 unsigned* __restrict__ oamax, const int* __restrict__ tab) {
    constexpr int PITCH_F = 36;
    const int li = lane & 31, lh = lane >> 5;
    const int qd = lane & 7, r0 = lane >> 3;
    const int ncol = a.oc_split;
    const int oc4 = ocb * 32 + 4 * qd;
    const bool col_ok = oc4 < ncol;
    const float* __restrict__ X = a.X;
    const int ch = a.out_chunk;
    const int ostr = ch > 0 ? ch : ncol;
    // stores: one wave-uniform base + (uniform element offset + the lane's offset), 32 bits (cf. epi_rel_mul_al)
    const int ocu = ocb * 32, ocl4 = 4 * (lane & 7);
    char* __restrict__ Ou = reinterpret_cast<char*>(a.out1 ? a.out1 : a.out0) +
                            ((g0 * HW + wm * 224) * (long)ostr + (ch > 0 ? (long)(ocu / ch) * total_pix * ch + ocu % ch : (long)ocu)) * 4;
    const unsigned ooff = (unsigned)(((long)(lane >> 3) * ostr + (ch > 0 ? (long)(ocl4 / ch) * total_pix * ch + ocl4 % ch : (long)ocl4)) * 4);
    const unsigned P = (unsigned)a.pix_per_map;
    const int nmax = a.n_maps - 1;
    const long pix0 = g0 * HW;
    // Per accumulator tile j (32 consecutive pixels, shorter than a map: at most ONE map boundary inside it) everything
    // but the lane's pixel offset dq = r0 + 8k is wave-uniform: first map n0 and pixel-in-map p0 of the tile's first pixel,
    // the images of n0 and n0 + 1.  (Aligned tiles: the whole workgroup tile lies in one map.)
    auto tile_base = [&](const int j, unsigned& n0, int& p0, long& b0, long& b1) {
        const unsigned q0t = (unsigned)(wm * 224 + 32 * j);
        const unsigned rr = q0t / (unsigned)HW, c0 = q0t - rr * HW;
        const unsigned g = (unsigned)g0 + rr;
        n0 = g / (unsigned)HW;
        p0 = (int)((g - n0 * HW) * HW + c0);
        long img0, img1;
        if constexpr (AL) {
            img0 = img1 = a.map2img ? a.map2img[min((int)n0, nmax)] : (long)n0;
        } else {       // map-straddling tiles: the two images of the tile from the workgroup's table (filled in the prologue)
            img0 = tab[(wm * 7 + j) * 4 + 2]; img1 = tab[(wm * 7 + j) * 4 + 3];
        }
        b0 = (img0 * P + p0) * (long)ncol + oc4;
        b1 = (img1 * P + p0 - (long)P) * (long)ncol + oc4;
    };
    // ---- multiplicand loads run ahead of the tiles: all 28 up front for aligned tiles; map-straddling tiles (two
    // candidate addresses per load, more live registers) keep a ring of 3 tiles (issued two tiles ahead: such a load is
    // older than the stores it would otherwise queue behind in the in-order vmcnt)
    constexpr int RING = AL ? 7 : 3;
    f32x4 xv[RING][4];
    auto load_x = [&](const int j) {
        unsigned n0; int p0; long b0, b1;
        tile_base(j, n0, p0, b0, b1);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int dq = r0 + 8 * k;
            const long gp = pix0 + wm * 224 + 32 * j + dq;
            const bool ok = col_ok && (AL || gp < total_pix);
            const long xi = ((AL || p0 + dq < (int)P) ? b0 : b1) + (long)dq * ncol;
            xv[j % RING][k] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok && !(LRPX_EPI_EXP & 1)) xv[j % RING][k] = *reinterpret_cast<const f32x4*>(X + xi);
        }
    };
#pragma unroll
    for (int j = 0; j < (AL ? 7 : 2); ++j) load_x(j);
    const unsigned n_first = (unsigned)g0 / (unsigned)HW;    // map of the workgroup tile's first pixel
    float m_al = 0.f;
    // map-straddling tiles: the 224 pixels of the workgroup touch at most the maps n_first, + 1, + 2 (a map has >= 196 pixels).  Their
    // maxima are kept per lane and reduced ONCE behind the last tile: amax_update reads the word first, and a wait for that read is a
    // wait for every store issued before it (one in-order counter) - two such waits per tile made the wave sit out the write latency
    // of each of its 7 tiles
    float mm0 = 0.f, mm1 = 0.f, mm2 = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        if constexpr (!AL) { if (j + 2 < 7) load_x(j + 2); }
#pragma unroll
        for (int e = 0; e < 16; ++e) scr[((e & 3) + 8 * (e >> 2) + 4 * lh) * PITCH_F + li] = acc[j][e];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private patch: LDS ops of one wave run in order
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *reinterpret_cast<const f32x4*>(scr + (r0 + 8 * k) * PITCH_F + 4 * qd);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // ... and the next tile's writes come after these reads
        unsigned nt0; int p0; long b0_, b1_;
        tile_base(j, nt0, p0, b0_, b1_);
        float m0 = 0.f, m1 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 xk = xv[j % RING][k];
            f32x4 r;
            if constexpr (EPI == EPI_GUIDED) {     // ReLU hook of the layer below (a.relu == 2: the plain autograd mask)
#pragma unroll
                for (int c = 0; c < 4; ++c) r[c] = (xk[c] > 0.f && (a.relu == 2 || v[k][c] > 0.f)) ? v[k][c] : 0.f;
            } else {
                r = f32x4{v[k][0] * xk[0], v[k][1] * xk[1], v[k][2] * xk[2], v[k][3] * xk[3]};
            }
            const int dq = r0 + 8 * k;
            const long gp = pix0 + wm * 224 + 32 * j + dq;
            if (col_ok && (AL || gp < total_pix)) {
                float* op = reinterpret_cast<float*>(Ou + ((unsigned)((32 * j + 8 * k) * ostr * 4) + ooff));     // scalar base + 32-bit offset
#if LRPX_EPI_EXP & 2
                if (r[0] == 1.2345e-30f) *reinterpret_cast<f32x4*>(op) = r;
#elif LRPXH_NT_STORE & 1
                __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(op));
#else
                *reinterpret_cast<f32x4*>(op) = r;
#endif
                const float m = fmaxf(fmaxf(fabsf(r[0]), fabsf(r[1])), fmaxf(fabsf(r[2]), fabsf(r[3])));
                if (AL || p0 + dq < (int)P) m0 = fmaxf(m0, m); else m1 = fmaxf(m1, m);
            }
        }
        if (oamax) {
            if constexpr (AL) {
                m_al = fmaxf(m_al, m0);
            } else {
                const int k0 = (int)(nt0 - n_first);               // 0 .. 2, wave-uniform
                mm0 = fmaxf(mm0, k0 == 0 ? m0 : 0.f);
                mm1 = fmaxf(mm1, k0 == 0 ? m1 : (k0 == 1 ? m0 : 0.f));
                mm2 = fmaxf(mm2, k0 == 1 ? m1 : (k0 == 2 ? m0 : 0.f));
            }
        }
    }
    if constexpr (AL) {
        if (oamax) {
            m_al = wave_max(m_al);
            if (lane == 0 && (int)n_first <= nmax) amax_update(&oamax[n_first], m_al);
        }
    } else {
        if (oamax) {
            mm0 = wave_max(mm0); mm1 = wave_max(mm1); mm2 = wave_max(mm2);
            if (lane == 0 && (int)n_first <= nmax) amax_update(&oamax[n_first], mm0);
            if (lane == 0 && (int)n_first + 1 <= nmax) amax_update(&oamax[n_first + 1], mm1);
            if (lane == 0 && (int)n_first + 2 <= nmax) amax_update(&oamax[n_first + 2], mm2);
        }
    }
}

// ---- fp6 cross products (LRPXH_XP6) ----
typedef unsigned u32x6_ __attribute__((ext_vector_type(6)));
typedef unsigned u32x3_ __attribute__((ext_vector_type(3)));
typedef unsigned u32x16_ __attribute__((ext_vector_type(16)));
// One 16-channel slice v * sc (scaled into the fp16 range): hw = the fp16 hi halves (8 packed pairs), rw = the residuals
// (x - hi) * 2^11 as fp16 pairs (|.| <= |x|: the same block serves both), and the slice's block scale bs - a float whose EXPONENT
// FIELD e is what the conversion and the matrix core use: values / 2^(e-127) lie in [3.75, 7.5] for the largest |x| (e2m3 saturates
// at 7.5).  The E8M0 byte of the matrix core carries the 2^-11 of the residual scaling: e - 11 (clamped at 0: such a slice is
// < 2^-116 of the fp16 range).
__device__ __forceinline__ void x6_split(const f32x4 (&v)[4], const float sc, unsigned (&hw)[8], unsigned (&rw)[8], float& bs, unsigned& sbyte) {
    float m = 0.f;
#pragma unroll
    for (int c = 0; c < 16; c += 2) {
        const f32x2_ x2 = f32x2_{v[c >> 2][c & 3], v[c >> 2][(c & 3) + 1]} * f32x2_{sc, sc};
        const f16x2_ h = __builtin_convertvector(x2, f16x2_);
        const f32x2_ hf = __builtin_convertvector(h, f32x2_);
        const f16x2_ r = __builtin_convertvector((x2 - hf) * f32x2_{2048.f, 2048.f}, f16x2_);
        hw[c / 2] = __builtin_bit_cast(unsigned, h);
        rw[c / 2] = __builtin_bit_cast(unsigned, r);
        m = fmaxf(m, fmaxf(fabsf(x2[0]), fabsf(x2[1])));
    }
    bs = m * (16.f / 15.f * 0.25f);
    const int e = (int)((__builtin_bit_cast(unsigned, bs) >> 23) & 0xffu);
    sbyte = (unsigned)max(e - 11, 0);
}
// 32 fp16 -> 32 fp6 e2m3, field c = x_c / 2^(e-127), field 16 + c = r_c / 2^(e-127) (v_cvt_scalef32_pk32_fp6_f16: element i ->
// field i, round to nearest even, saturating).  Inline asm with an EARLY-CLOBBER result: the compiler (ROCm 7.2) lets the 6
// result registers of the conversion builtins overlap their 16 / 32 source registers, and the multi-pass instruction then reads
// sources it has already overwritten (seen with __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32 as wrong cross products in exactly
// those unrolled item copies whose allocation overlapped: v[126:131] <- v[112:127], v[128:143]).
__device__ __forceinline__ u32x6_ x6_pack(const unsigned (&hw)[8], const unsigned (&rw)[8], const float bs) {
    const u32x16_ src = {hw[0], hw[1], hw[2], hw[3], hw[4], hw[5], hw[6], hw[7], rw[0], rw[1], rw[2], rw[3], rw[4], rw[5], rw[6], rw[7]};
    u32x6_ q;
    asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(q) : "v"(src), "v"(bs));
    return q;
}
// host / pack-time encoder of one e2m3 value (round to nearest even, saturating at 7.5): |v| < 2: steps of 1/8 (subnormals and the
// first binade), [2, 4): 1/4, [4, 8): 1/2
__host__ __device__ __forceinline__ unsigned fp6_e2m3_encode(float v) {
    const unsigned sgn = v < 0.f ? 32u : 0u;
    const float a = fabsf(v);
    float c;
    if (!(a < 2.f)) c = !(a < 4.f) ? rintf(a * 2.f) + 16.f : rintf(a * 4.f) + 8.f;
    else c = rintf(a * 8.f);
    if (!(c < 31.f)) c = 31.f;                      // (also NaN)
    return sgn | (unsigned)c;
}

// lane's channel within its 32-channel block for the ROW-PERMUTED weight pack of the mode-3 relevance kernels (pack_weights_f16f8_kernel,
// BWD_POS): fragment row / column rho carries channel 16 ((rho >> 2) & 1) + 4 (rho >> 3) + (rho & 3)
__device__ __forceinline__ int perm_row_channel(const int rho) { return 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3); }

// REL_MUL / GUIDED epilogue of the MAP-ALIGNED kernels (56 / 112 / 224: the workgroup tile lies inside one map) with scalar-base
// addressing: element (tile j, register e) of a lane sits at  X[(img P + p0 + wm 224 + 32 j + dq(e)) ncol + ocb 32] + (4 lh ncol + li) -
// ONE wave-uniform 64-bit base per tensor plus a 32-bit byte offset (uniform element part + the lane's part: one v_add_u32) - so every
// access is `global_load / global_store_dword v, v_off, s[base]`: no 64-bit vector add per access (the generic dword epilogue: 224
// v_lshl_add_u64 per wave) and half the address bytes on the way to the texture unit.  Same arithmetic, same order: bit-identical.
template <int HW, int EPI, bool F8, bool PERM = false, bool B6 = false>
__device__ __forceinline__ void epi_rel_mul_al(const ConvArgs& a, f32x16 (&acc)[7], const int wm, const int ocb, const int lane,
                                               const long g0, const long total_pix, unsigned* __restrict__ oamax, const float inv_w,
                                               const unsigned* __restrict__ in_amax) {
    // PERM: the weights come from the row-permuted pack (the B-fragment column of lane l holds channel perm_row_channel(l % 32) of the
    // block): the lane's channel, nothing else, changes - the same 128-byte lines per instruction
    const int li = PERM ? perm_row_channel(lane & 31) : (lane & 31), lh = lane >> 5;
    const int ncol = a.oc_split;
    const int oc = ocb * 32 + li;
    if (oc >= ncol) return;                                     // (padding columns of the last channel block)
    const unsigned P = (unsigned)a.pix_per_map;
    const int nmax = a.n_maps - 1;
    const unsigned n = (unsigned)g0 / (unsigned)HW;             // the tile's map
    const unsigned p0 = ((unsigned)g0 - n * HW) * HW;           // pixel-in-map of the tile's first pixel
    const long img = a.map2img ? a.map2img[min((int)n, nmax)] : (long)n;
    float f = 1.f;                                              // (B6: exact bf16 splits, nothing was scaled)
    if constexpr (!B6) f = exp2i(-split_scale_exp<F8>(in_amax[min((int)n, nmax)])) * inv_w;
    const int ch = a.out_chunk;
    const int ostr = ch > 0 ? ch : ncol;
    // uniform bases (bytes) and the lane's offsets (bytes, 32-bit)
    const char* __restrict__ Xu = reinterpret_cast<const char*>(a.X) + ((img * P + p0 + wm * 224) * (long)ncol + ocb * 32) * 4;
    char* __restrict__ Ou = reinterpret_cast<char*>(a.out1 ? a.out1 : a.out0) +
                            ((g0 * HW + wm * 224) * (long)ostr + (ch > 0 ? (long)((ocb * 32) / ch) * total_pix * ch + (ocb * 32) % ch : (long)ocb * 32)) * 4;
    const unsigned xoff = (unsigned)((4 * lh * ncol + li) * 4);
    const unsigned ooff = (unsigned)(((long)(4 * lh) * ostr + (ch > 0 ? (long)(li / ch) * total_pix * ch + li % ch : (long)li)) * 4);
    float xv[7][16];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned ub = (unsigned)((32 * j + (e & 3) + 8 * (e >> 2)) * ncol * 4);         // uniform, < 2^20
            xv[j][e] = (LRPX_EPI_EXP & 1) ? 1.f : *reinterpret_cast<const float*>(Xu + (ub + xoff));
        }
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float v = B6 ? acc[j][e] : acc[j][e] * f;
            const float r = EPI == EPI_GUIDED ? ((xv[j][e] > 0.f && (a.relu == 2 || v > 0.f)) ? v : 0.f) : xv[j][e] * v;
            const unsigned ub = (unsigned)((32 * j + (e & 3) + 8 * (e >> 2)) * ostr * 4);         // uniform
            float* op = reinterpret_cast<float*>(Ou + (ub + ooff));
#if LRPX_EPI_EXP & 2
            if (r == 1.2345e-30f) *op = r;
#elif LRPXH_NT_STORE & 2
            __builtin_nontemporal_store(r, op);
#else
            *op = r;
#endif
            m = fmaxf(m, fabsf(r));
        }
    if (oamax) {
        m = wave_max(m);
        if (lane == 0 && (int)n <= nmax) amax_update(&oamax[n], m);
    }
}

// FWD_DUAL epilogue of the map-aligned forward kernels, addressed like epi_rel_mul_al (one scalar base, 32-bit offsets): channel blocks
// below oc_split are activations - out0 = ReLU(acc + bias), per-image maximum to out0_amax -, the blocks above it are Z+ - out1 = acc.
// A channel block is one or the other as a whole (oc_split is a multiple of 32 for every VGG16 layer; the launcher checks).
template <int HW, bool F8, bool B6 = false>
__device__ __forceinline__ void epi_fwd_dual_al(const ConvArgs& a, f32x16 (&acc)[7], const int wm, const int ocb, const int lane,
                                                const long g0, const float inv_w, const unsigned* __restrict__ in_amax) {
    const int li = lane & 31, lh = lane >> 5;
    const int ncol = a.oc_split;
    const int nmax = a.n_maps - 1;
    const unsigned n = (unsigned)g0 / (unsigned)HW;             // the tile's image
    float f = 1.f;
    if constexpr (!B6) f = exp2i(-split_scale_exp<F8>(in_amax[min((int)n, nmax)])) * inv_w;
    const bool is_act = ocb * 32 < ncol;                        // wave-uniform
    const int ocl = (is_act ? ocb * 32 : ocb * 32 - ncol);      // first channel of the block inside its tensor
    if (ocl + li >= ncol) return;
    const float bias = (is_act && a.bias) ? a.bias[ocl + li] : 0.f;
    char* __restrict__ Ou = reinterpret_cast<char*>(is_act ? a.out0 : a.out1) + ((g0 * HW + wm * 224) * (long)ncol + ocl) * 4;
    const unsigned ooff = (unsigned)((4 * lh * ncol + li) * 4);
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            float v = B6 ? acc[j][e] : acc[j][e] * f;
            if (is_act) { v += bias; v = v > 0.f ? v : 0.f; m = fmaxf(m, v); }
            const unsigned ub = (unsigned)((32 * j + (e & 3) + 8 * (e >> 2)) * ncol * 4);         // uniform
            *reinterpret_cast<float*>(Ou + (ub + ooff)) = v;
        }
    if (is_act && a.out0_amax) {
        m = wave_max(m);
        if (lane == 0 && (int)n <= nmax) amax_update(&a.out0_amax[n], m);
    }
}

// REL_MUL epilogue of the mode-3 relevance kernels (BLK): TRANSPOSED accumulators and the BLOCKED layout (blocked.h).  The MFMAs were
// issued with the weights as the A operand and the pixels as the B operand - the same fragments, the arguments swapped - so the
// result tile is channels x pixels: a lane owns ONE pixel (tile pixel lane % 32) and, with the weight rows permuted at pack time
// (pack_weights_f16f8_kernel), the 16 CONTIGUOUS channels 32 ocb + 16 lh + e, e = 0..15: one 16-channel slice.  Multiplicand and
// output are blocked, so part k (4 channels) of the 32 pixels of a tile is one 512-byte run per lane half: 4 float4 loads + 4 float4
// stores per lane and tile, each wave instruction 8 whole 128-byte lines, nothing through LDS (the dword epilogue: 16 + 16 accesses
// of 4 bytes; the LDS-transposed float4 epilogue: the same 4 + 4 plus 16 ds_write + 4 ds_read per tile; the transposed result in
// NHWC - round 3's LRPXH_TR experiment - touched 32 lines per instruction and lost 12 % of the chain).  Arithmetic and its order are
// those of epi_rel_mul_al / epi_rel_mul_wide: out = x * (acc * 2^-kA 2^-kW), per-map maxima from the stored values.
template <int HW, bool AL, bool F8>
__device__ __forceinline__ void epi_rel_mul_blk(const ConvArgs& a, f32x16 (&acc)[7], const int wm, const int ocb, const int lane,
                                                const long g0, const long total_pix, unsigned* __restrict__ oamax,
                                                const int* __restrict__ tab, const float inv_w, const unsigned* __restrict__ in_amax) {
    const int li = lane & 31, lh = lane >> 5;
    const int ncol = a.oc_split;
    const int chunk = 2 * ocb + lh;                     // the lane's 16-channel slice
    const bool col_ok = chunk * 16 < ncol;
    const unsigned P = (unsigned)a.pix_per_map;
    const int nmax = a.n_maps - 1;
    const long cs_o = blk_chunk_stride(total_pix), cs_x = blk_chunk_stride(P);
    const long gq0 = g0 * HW + wm * 224;                // global pixel of the wave's first tile pixel: a multiple of 32
    // tile j, part k of the lane: Op + j * 512 + k * 128 floats (compile-time offsets from one pointer)
    float* __restrict__ Op = (a.out1 ? a.out1 : a.out0) + (long)chunk * cs_o + ((gq0 >> 5) << 9) + li * 4;
    const float* __restrict__ X = a.X;
    const long ximg = (long)(ncol >> 4) * cs_x;         // floats of one image's multiplicand
    auto store4 = [&](float* op, const f32x4 r) {
#if LRPX_EPI_EXP & 2
        if (r[0] == 1.2345e-30f) *reinterpret_cast<f32x4*>(op) = r;
#elif LRPXH_NT_STORE & 1
        __builtin_nontemporal_store(r, reinterpret_cast<f32x4*>(op));
#else
        *reinterpret_cast<f32x4*>(op) = r;
#endif
    };
    if constexpr (AL) {
        // the workgroup tile lies inside one map: one image, one operand scale, and the tile's first pixel-in-map is a multiple of 32
        const unsigned n = (unsigned)g0 / (unsigned)HW;
        const unsigned p0 = ((unsigned)g0 - n * HW) * HW + wm * 224;
        const long img = a.map2img ? a.map2img[min((int)n, nmax)] : (long)n;
        const float f = exp2i(-split_scale_exp<F8>(in_amax[min((int)n, nmax)])) * inv_w;
        const float* __restrict__ Xp = X + img * ximg + (long)chunk * cs_x + ((long)(p0 >> 5) << 9) + li * 4;
        f32x4 xv[7][4];
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xv[j][k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (col_ok && !(LRPX_EPI_EXP & 1)) xv[j][k] = *reinterpret_cast<const f32x4*>(Xp + j * 512 + k * 128);
            }
        float m = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            f32x4 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float v = acc[j][4 * k + c] * f; r[k][c] = xv[j][k][c] * v; }
                m = fmaxf(m, fmaxf(fmaxf(fabsf(r[k][0]), fabsf(r[k][1])), fmaxf(fabsf(r[k][2]), fabsf(r[k][3]))));
            }
            if (col_ok) {
#pragma unroll
                for (int k = 0; k < 4; ++k) store4(Op + j * 512 + k * 128, r[k]);
            }
        }
        if (!col_ok) m = 0.f;
        if (oamax) {
            m = wave_max(m);
            if (lane == 0 && (int)n <= nmax) amax_update(&oamax[n], m);
        }
    } else {
        // map-straddling tiles (28 x 28 / 14 x 14): a 32-pixel accumulator tile is shorter than a map - at most ONE map boundary inside
        // it, so a lane's pixel belongs to the map of the tile's first pixel or to the next one (scales and images of both from the
        // workgroup's table); its pixel-in-map is per lane, the multiplicand address with it (blocks of 32 pixels per image)
        const unsigned n_first = (unsigned)g0 / (unsigned)HW;
        auto pixel = [&](const int j, long& gp, long& xoff, float& f, int& kmap) {
            const unsigned q0t = (unsigned)(wm * 224 + 32 * j);
            const unsigned rr = q0t / (unsigned)HW, c0 = q0t - rr * HW;
            const unsigned g = (unsigned)g0 + rr;
            const unsigned n0 = g / (unsigned)HW;
            const int p = (int)((g - n0 * HW) * HW + c0) + li;
            const bool second = p >= (int)P;
            const int pp = second ? p - (int)P : p;
            const long img = tab[(wm * 7 + j) * 4 + (second ? 3 : 2)];
            gp = gq0 + 32 * j + li;
            xoff = img * ximg + (long)chunk * cs_x + blk_pix_off(pp);
            f = __builtin_bit_cast(float, tab[(wm * 7 + j) * 4 + (second ? 1 : 0)]) * inv_w;
            kmap = (int)(n0 - n_first) + (second ? 1 : 0);         // 0 .. 2: the 224 pixels of a wave touch at most three maps
        };
#ifndef LRPXH_BLK_RING
#define LRPXH_BLK_RING 3          // multiplicand tiles in flight in the map-straddling epilogue (7: all loads before the first store)
#endif
        constexpr int RING = LRPXH_BLK_RING;
        f32x4 xv[RING][4];
        auto load_x = [&](const int j) {
            long gp, xoff; float f; int kmap;
            pixel(j, gp, xoff, f, kmap);
            const bool ok = col_ok && gp < total_pix;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                xv[j % RING][k] = f32x4{0.f, 0.f, 0.f, 0.f};
                if (ok && !(LRPX_EPI_EXP & 1)) xv[j % RING][k] = *reinterpret_cast<const f32x4*>(X + xoff + k * 128);
            }
        };
#pragma unroll
        for (int j = 0; j < RING - 1; ++j) load_x(j);
        // per-map maxima in registers until the last tile (amax_update reads the word first, and a wait for that read is a wait for
        // every store issued before it)
        float mm0 = 0.f, mm1 = 0.f, mm2 = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            if (j + RING - 1 < 7) load_x(j + RING - 1);
            long gp, xoff; float f; int kmap;
            pixel(j, gp, xoff, f, kmap);
            const bool ok = col_ok && gp < total_pix;
            float m = 0.f;
            f32x4 r[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
#pragma unroll
                for (int c = 0; c < 4; ++c) { const float v = acc[j][4 * k + c] * f; r[k][c] = xv[j % RING][k][c] * v; }
                m = fmaxf(m, fmaxf(fmaxf(fabsf(r[k][0]), fabsf(r[k][1])), fmaxf(fabsf(r[k][2]), fabsf(r[k][3]))));
            }
            if (ok) {
#pragma unroll
                for (int k = 0; k < 4; ++k) store4(Op + j * 512 + k * 128, r[k]);
            } else {
                m = 0.f;
            }
            mm0 = fmaxf(mm0, kmap == 0 ? m : 0.f);
            mm1 = fmaxf(mm1, kmap == 1 ? m : 0.f);
            mm2 = fmaxf(mm2, kmap == 2 ? m : 0.f);
        }
        if (oamax) {
            mm0 = wave_max(mm0); mm1 = wave_max(mm1); mm2 = wave_max(mm2);
            if (lane == 0 && (int)n_first <= nmax) amax_update(&oamax[n_first], mm0);
            if (lane == 0 && (int)n_first + 1 <= nmax) amax_update(&oamax[n_first + 1], mm1);
            if (lane == 0 && (int)n_first + 2 <= nmax) amax_update(&oamax[n_first + 2], mm2);
        }
    }
}

// F8 ("f16+f8x2"): the two CROSS products a0*b1 + a1*b0 - 2^-11 of the result - do not need fp16 operands: with both
// factors rounded to fp8 e4m3 (4 significand bits) their error is 2^-11 * 2^-4 per product, random sign; simulated
// through all 13 layers the maps move by < 1e-5 of their maximum (tolerance 1e-4; plain f16x3: ~1e-6).  They run on
// v_mfma_f32_32x32x64_f8f6f4, K = 64 = 4 (tap, 16-channel) slices: 5 fp8 MFMAs per K-chunk (f8_slot_*) instead of 18
// fp16 ones.  LDS pixel (80 B as before): 16 fp16 hi | 16 fp8 of x*2^-4 | 16 fp8 of (x-hi)*2^4 | pad.
//
// B6 (conv mode 1): the same tiling, staging, B queue and epilogues with EXACT operand splits on the bf16 matrix cores - every fp32
// operand a = a0 + a1 + a2 (three bf16 parts, 24 significand bits, fp32's exponent range: no operand scales, in_amax unused), the six
// products with i + j <= 2 accumulated in fp32 smallest first (conv_bf16x6.h:1-6: dropped terms <= 3 * 2^-24 of a product, what the
// rounding of an fp32 product itself costs).  LDS pixel = 3 planes x 32 B + 16 B pad = 112 B (pitch / 16 == 7 W (mod 16): conflict-free
// ds_read_b128 as with 80 B); weights from lrpx_pack_weights_bf16x3 (three planes per k-step, no header).
template <int HW, int MT, int NWN, bool DB, int EPI, bool POOL = false, bool F8 = false, bool B6 = false>
__global__ __launch_bounds__(64 * MT * NWN, (MT * NWN >= 8) ? 1 : 2) void conv_f16x3_kernel(ConvArgs a, int m_tiles, int n_blocks) {
    static_assert(!(F8 && B6), "conv_f16x3_kernel: F8 (fp16 + narrow cross products) and B6 (exact bf16 splits) exclude each other");
    LRPXH_T(t_start);
#ifdef LRPX_STAMP
    unsigned long long s_issue = 0, s_mfma = 0, s_commit = 0, s_barrier = 0, s_tap0 = 0, s_tap1 = 0, s_tap2 = 0;
#endif
    constexpr int KC = 16, TAPS = 9;
    using C = ConvCfg<HW, KC, MT, NWN, TAPS>;
    constexpr int W = C::W, H = C::H, NT = C::NT;
    constexpr int PSTRIDE = B6 ? 112 : 80;             // bytes per LDS pixel
    constexpr int PITCH = W * PSTRIDE + 256;           // pitch/16 == 5*W (B6: 7*W) (mod 16), >= (W+2) pixels
    constexpr int BUFB = C::NSLOT * PITCH;
    constexpr int NBUF = DB ? 2 : 1;
    constexpr int STAGE_SCRATCH = NBUF * BUFB + 16;    // 256 bytes behind the buffers take the writes of items with nothing to write
    constexpr int POOL_SCRATCH = STAGE_SCRATCH;
    constexpr bool AL = (H % C::R == 0);               // a workgroup tile never straddles two maps
#ifndef LRPXH_HOIST_MASK
#define LRPXH_HOIST_MASK 1
#endif
    // staging descriptors stay in registers (see item()): bit 0: 56 non-pooled, 1: 56 pooled, 2: 112 pooled, 3: 224 pooled
#ifndef LRPXH_F8_HOIST
#define LRPXH_F8_HOIST 1      // measured with the row-wise item layout: 56x56 F8 kernel 2.33 -> 2.12 ms per launch
#endif
    constexpr bool HOIST = (!F8 || LRPXH_F8_HOIST) && AL && (((LRPXH_HOIST_MASK & 1) && HW == 56 && !POOL) || ((LRPXH_HOIST_MASK & 2) && HW == 56 && POOL) ||
                                  ((LRPXH_HOIST_MASK & 4) && HW == 112 && POOL) || ((LRPXH_HOIST_MASK & 8) && HW == 224 && POOL));
#ifndef LRPXH_APIPE_MIN_HW
#define LRPXH_APIPE_MIN_HW 14
#endif
#ifndef LRPXB6_APIPE
#define LRPXB6_APIPE 1
#endif
    constexpr bool APIPE = B6 ? (LRPXB6_APIPE != 0) : ((HW >= LRPXH_APIPE_MIN_HW) && !F8);
    // X6: the cross products as block-scaled fp6.  LDS pixel: 16 fp16 hi (32 B) | 32 fp6 e2m3 = x_0..x_15, then (x_c - hi_c) * 2^11,
    // c = 0..15, all divided by the slice's block scale 2^s (24 B) | E8M0 byte s - 11 + 127 (dword at byte 60; byte 56 stays clear so that the 8 + 4 byte reads do not fuse into a slower ds_read_b96) | pad.  A staging item is one
    // pixel's whole 16-channel slice (4 float4): the block maximum and v_cvt_scalef32_pk32_fp6_f16 need the 16 values in one lane.
    constexpr bool X6 = F8 && (LRPXH_XP6 != 0);
    // BLK: the mode-3 relevance kernels (REL_MUL) read S and the multiplicand and write S in the BLOCKED layout (blocked.h) and
    // accumulate channels x pixels (TR: the arguments of every MFMA swapped, weight rows permuted at pack time): see epi_rel_mul_blk.
    // conv1_2 (HW == 224, pooled fp32 input, BLOCKED) keeps the plain accumulators and its NHWC multiplicand / 32-channel-chunk output:
    // its consumer, the first-layer kernel, walks 34-pixel halo rows, and 32-pixel blocks cost it 1.4x the lines (0.98 -> 1.42 ms)
    // for the 3 % the transposed epilogue gave conv1_2.
    constexpr bool BLK = X6 && (EPI == EPI_REL_MUL);
    constexpr bool TR = BLK && (HW != 224);
    constexpr int NV = X6 ? 4 : 1;                     // float4 loads per staging item
#ifndef LRPXH_X6_READ
#define LRPXH_X6_READ 1
#endif
    // fp6 operands from LDS as two conflict-free 16-byte reads (else 16 + 8 + 4 bytes: 2- and 4-way bank conflicts over pixels 80 bytes apart)
    constexpr bool X6RD = X6 && (EPI == EPI_REL_MUL) && (LRPXH_X6_READ != 0) && !(HW == 112 && MT == 2);      // (measured on the relevance chain only)
    extern __shared__ __attribute__((aligned(16))) char ldsb[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / NWN, wn = wave % NWN;
#ifdef LRPXB6_PRIO
    // (experiment, round 6) the two workgroups of a CU tend to phase-lock - the one that is alone in its matrix phase runs at twice the
    // speed and catches up with the other's barrier - so that both stage at the same time and the matrix pipe idles (busy 0.79).  A STATIC
    // asymmetry per SIMD: the wave in an odd hardware slot issues with priority, the other fills the gaps.
    if constexpr (B6) {
        const unsigned hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));      // HW_REG_HW_ID, WAVE_ID[3:0]
        if (hwid & 1u) __builtin_amdgcn_s_setprio(LRPXB6_PRIO); else __builtin_amdgcn_s_setprio(0);
    }
#endif
    // (the body stays in the __global__ function: as an inlined device function taking the argument struct by value it spilled 12 - 25
    // registers in every instantiation - the struct no longer lived in the kernel-argument segment - and ran 6 - 8 % slower)
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    // Workgroup ids go round-robin over the 8 XCDs (each with its own L2): XCD x walks a CONTIGUOUS range of tiles, so the
    // halo rows that neighbouring tiles share and the multiplicand tiles of an image's words are re-read from one L2
    // (round-robin tiles, mtile = (idx / n_blocks) * 8 + xcd: 28x28 layers +6 %, conv1_2 +4 %, chain 18.4 vs 18.1 ms)
#ifndef LRPXH_XCD_BLOCKED
#define LRPXH_XCD_BLOCKED 1
#endif
    int mtile = (idx / n_blocks) * 8 + xcd;
    if constexpr (LRPXH_XCD_BLOCKED != 0) {
        // balanced ranges: XCD x owns tiles [x * m / 8, (x + 1) * m / 8) (small grids - the forward trace - stay spread)
        const int t0 = (int)(((long)xcd * m_tiles) >> 3), t1 = (int)(((long)(xcd + 1) * m_tiles) >> 3);
        mtile = t0 + idx / n_blocks;
        if (mtile >= t1 && !(AL && a.tile_group > 1)) return;       // (grouped order: decided below)
    }
    const int nblk = idx % n_blocks;
#ifndef LRPXH_TILE_GROUP
#define LRPXH_TILE_GROUP 1
#endif
    if constexpr (AL && (LRPXH_TILE_GROUP != 0)) {
        // a.tile_group maps per image: the tiles (image row block r) of the words w of one image run back to back on ONE
        // XCD, so the image's multiplicand tile stays in that L2 for all of them instead of being fetched from HBM once
        // per word; an XCD walks a contiguous range of (image, row block) groups - consecutive row blocks of an image
        if (a.tile_group > 1) {
            constexpr int TPM = H / C::R;                          // tiles per map
            const int q = idx / n_blocks;                          // position in this XCD's sequence
            const int gl = q / a.tile_group, w = q - gl * a.tile_group;
            int G = gl * 8 + xcd;                                  // (image, row block) group
            const int n_groups = (a.n_maps / a.tile_group) * TPM;
            if constexpr (LRPXH_XCD_BLOCKED != 0) {
                const int g0x = (int)(((long)xcd * n_groups) >> 3), g1x = (int)(((long)(xcd + 1) * n_groups) >> 3);
                G = g0x + gl;
                if (G >= g1x) return;
            }
            const int i = G / TPM, r = G - i * TPM;
            mtile = (i * a.tile_group + w) * TPM + r;
            if (G >= n_groups) return;
        }
    }
    if (mtile >= m_tiles) return;
#ifdef LRPXH_START_SKEW
    // (experiment) the workgroups of the first round start spread over one tile time, so that the epilogues of the CUs - whose tiles
    // all take the same time - do not store in the same bursts
    if (idx / n_blocks < LRPXH_START_ROUNDS) {
        const int phase = (bid >> 3) & 7;
        for (int i = 0; i < phase * LRPXH_START_SKEW; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    const int ocb = nblk * NWN + wn;
    const bool wave_active = ocb * 32 < a.n_oc;
    const long total_pix = (long)a.n_maps * a.pix_per_map;
    // PLAIN epilogue with a.ksplit > 1 (forward trace of the 14x14 layers at small batches: 112 workgroups with 32
    // K-chunks each on a chip of 256 CUs): blockIdx.y takes a contiguous range of the K-chunks and writes its partial
    // sums to out0 + blockIdx.y * (pixels x columns); lrpx_fwd_dual_finish adds them in split order
    const int nchunk_all = a.cin / KC;
    int nchunk = nchunk_all, c_begin = 0;
    if constexpr (EPI == EPI_PLAIN) {
        if (a.ksplit > 1) {
            c_begin = (int)blockIdx.y * nchunk_all / a.ksplit;
            nchunk = ((int)blockIdx.y + 1) * nchunk_all / a.ksplit - c_begin;
            a.out0 += (long)blockIdx.y * total_pix * a.oc_split;
            a.ksplit = 1;            // (the epilogue stores its partial sums: conv_mfma.h adds atomically when ksplit > 1)
        }
    }
    const int li = lane & 31, lh = lane >> 5;
    const unsigned* __restrict__ in_amax = a.in_amax;
    // map-straddling tiles (28x28 / 14x14): per accumulator tile (wave row wm, tile j) the operand scales and the images of the
    // map of its first pixel and of the next map, looked up ONCE by 7 * MT threads while the first staging loads are in flight
    // and parked in LDS behind the staging buffers: [wm * 7 + j][2^-kA(n0), 2^-kA(n0 + 1), image(n0), image(n0 + 1)]
    int* tile_tab = reinterpret_cast<int*>(ldsb + NBUF * BUFB + 256 + 16);
    if constexpr (!AL) {
        if (tid < 7 * MT) {
            const unsigned q0t = (unsigned)((tid / 7) * 224 + 32 * (tid % 7));
            const unsigned g = (unsigned)((long)mtile * C::R) + q0t / (unsigned)HW;
            const int n0 = (int)(g / (unsigned)HW), nmax_ = a.n_maps - 1;
            const int na = min(n0, nmax_), nb = min(n0 + 1, nmax_);
            if constexpr (!B6) {
                tile_tab[tid * 4 + 0] = __builtin_bit_cast(int, exp2i(-split_scale_exp<F8>(in_amax[na])));
                tile_tab[tid * 4 + 1] = __builtin_bit_cast(int, exp2i(-split_scale_exp<F8>(in_amax[nb])));
            }
            tile_tab[tid * 4 + 2] = a.map2img ? a.map2img[na] : n0;
            tile_tab[tid * 4 + 3] = a.map2img ? a.map2img[nb] : n0 + 1;
        }
    }

    const long g0 = (long)mtile * C::R;
    const long v0 = g0 + g0 / H;
    int abase[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int q = wm * 224 + 32 * j + li;
        const int r = q / W, c = q % W;
        const long g = g0 + r;
        const int slot = (int)(g + g / H - v0) + 1;
        abase[j] = (slot - 1) * PITCH + c * PSTRIDE + lh * 16;
    }

    // ---- staging descriptors (16 channels = 4 float4 segments per pixel) ----
    // item `it` = (LDS row s, pixel px, 16-byte segment seg) -> LDS byte offset of its 4 fp16 in plane 0 (segment index in
    // bits 28-29; -1: never written, stays zero) and global pixel (-1: nothing to load).  Map-aligned tiles (224/112/
    // 56) derive both from `it` when needed - the tile sits inside one map, rows outside it are the zero padding;
    // tiles that can straddle maps (28/14: virtual zero row between maps) keep them in registers with the map's scale.
    //
    // POOL: the input is the relevance at the OUTPUT of the 2x2 max-pool under which this conv sits, S_lo
    // [n_maps][H/2*W/2][cin] (already divided by this layer's Z+ at the winner, see lrpx_vgg16_trace_derive), and
    // a.pool_am [n_img][H/2*W/2][cin] holds the window position (0..3, row-major) of each maximum: the Pool2d rule
    // (lrp_modules.py:182-195) is applied while staging - pixel (y, x) receives S_lo[y/2][x/2] if it is the winner of
    // its window, else 0 - so the 4x larger unpooled tensor (75 % zeros) never exists in HBM.  The global pixel is
    // then the LOW-resolution one, the window position rides in bits 26-27 of the LDS offset.
    constexpr int SEG = X6 ? 1 : KC / 4;
    constexpr int NITEM = C::NSLOT * W * SEG;
    // map-aligned tiles: staging items are laid out ROW-WISE over the threads - item slot u of thread tid is LDS row
    // u*RPS + tid/RI (narrow maps: RPS whole rows per slot) or row u/SPR, part u%SPR (wide maps: SPR slots per row) - so
    // that row, pixel and segment need no division, the row validity is wave-uniform and the descriptors are not worth
    // keeping in registers
#ifndef LRPXH_ROWMAP
#define LRPXH_ROWMAP 1
#endif
    constexpr int RI = W * SEG;                                   // items per LDS row
    // narrow maps under a wide workgroup (512 threads, 224 items per row): RPS = 2 rows per slot, 256 threads each
    constexpr int RPS = (RI <= NT / 2) ? 2 : 1;
    constexpr int TPR = NT / RPS;                                 // threads per row of a slot
    constexpr bool ROWMAP = AL && (LRPXH_ROWMAP != 0) && (RI <= NT ? (RI > NT / 2 || (RPS == 2 && RI > NT / 4)) : true);
    constexpr int SPR = (RI + NT - 1) / NT;                       // slots per row (1 when a row fits the workgroup)
    constexpr int U = ROWMAP ? ((C::NSLOT + RPS - 1) / RPS) * SPR : (NITEM + NT - 1) / NT;
    constexpr int UR = AL ? 1 : U;
    constexpr int HO = H / 2, WO = W / 2;
    int sdst[UR], sgp[UR];
    int sam[(POOL && !AL) ? U : 1];     // POOL: element offset into pool_am of the item's (image, low-res pixel)
    float ssc[UR];         // 2^kA of the item's map (one value per workgroup when tiles are map-aligned)
    const int n_al = (int)((unsigned)g0 / (unsigned)H), y_al = (int)g0 - n_al * H;
    long img_al = 0;
    if constexpr (AL) {
        ssc[0] = 1.f;
        if constexpr (!B6) ssc[0] = exp2i(split_scale_exp<F8>(in_amax[min(n_al, a.n_maps - 1)]));
        if constexpr (POOL) img_al = a.map2img ? a.map2img[min(n_al, a.n_maps - 1)] : n_al;
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = tid + u * NT;
            sdst[u] = -1; sgp[u] = -1; ssc[u] = 1.f;
            if constexpr (POOL) sam[u] = 0;
            if (it < NITEM) {
                const int s = it / (W * SEG);
                const int rem = it - s * (W * SEG);
                const int px = rem / SEG, seg = rem - px * SEG;
                const long v_ = v0 - 1 + s;
                const long n = v_ / (H + 1);
                const int y = (int)(v_ - n * (H + 1));
                if ((v_ >= 0) && (y < H) && (n < a.n_maps)) {
                    sdst[u] = (s * PITCH + (px + 1) * PSTRIDE + seg * 8) | (seg << 28);
                    sgp[u] = (int)((n * H + y - (g0 - 1)) * W + px);     // relative to the row above the tile's first
                    if constexpr (!B6) ssc[u] = exp2i(split_scale_exp<F8>(in_amax[n]));
                    if constexpr (POOL) {
                        const int lo = (y >> 1) * WO + (px >> 1);
                        const long img = a.map2img ? a.map2img[n] : n;
                        sdst[u] |= (((y & 1) << 1) | (px & 1)) << 26;
                        sgp[u] = (int)(n * (HO * WO) + lo);
                        sam[u] = (int)((img * (HO * WO) + lo) * a.cin);
                    }
                }
            }
        }
    }
    // Row-wise layout: everything about item u of a thread is (a value of the THREAD, the same for all its items) + (a
    // compile-time constant of u): LDS offset, relative pixel and element offset each cost one add per use, the column
    // predicate one compare; nothing worth keeping per item, nothing to multiply.  (The 12 items per thread of the 4-row
    // 112x112 tile spent 94 VALU instructions each on issue + commit, 41 of them conversions: as much time as the
    // chunk's MFMAs.)
    const int t_half = (ROWMAP && RPS == 2 && tid >= TPR) ? 1 : 0;
    const int t_rem = tid - t_half * TPR;                                   // position in the LDS row (RPS == 1: tid)
    const int t_px = t_rem / SEG, t_seg = t_rem - t_px * SEG;
    const int d_thr = (t_half * PITCH + (t_px + 1) * PSTRIDE + t_seg * 8) | (t_seg << 28);
    const int g_thr = t_half * W + t_px;
    const unsigned in_pix_stride_ = a.in_chunk_stride ? KC : a.cin;
    const unsigned e_thr = __umul24((unsigned)g_thr, in_pix_stride_) + (unsigned)t_seg * 4;   // element offset, thread part
    unsigned ethr_c = e_thr;
    // Copies the compiler cannot see through, refreshed once per K-chunk (LRPXH_REFRESH): what is derived from them is
    // recomputed per use (an add) instead of being hoisted out of the chunk loop into 2U registers - except in the
    // 56x56 kernel with a B queue of 8, where keeping the descriptors resident saves 6 % (tools/variant_sweep2.sh)
    int dthr_c = d_thr, gthr_c = g_thr, trem_c = t_rem, thalf_c = t_half;
#define LRPXH_REFRESH                                                                                    \
    if constexpr (!HOIST && AL && ROWMAP && !POOL) {                                                     \
        dthr_c = d_thr; gthr_c = g_thr; trem_c = t_rem; thalf_c = t_half; ethr_c = e_thr;                \
        asm volatile("" : "+v"(dthr_c), "+v"(gthr_c), "+v"(trem_c), "+v"(thalf_c), "+v"(ethr_c));        \
    }
    auto item = [&](const int u, int& dst, int& gp, int& amo) {
        amo = 0;
        if constexpr (AL && ROWMAP && !POOL) {
            // LDS row s = s_c + t_half, position in the row t_rem + rem_c
            constexpr int RPSv = RPS, SPRv = SPR;
            const int s_c = RPSv == 2 ? 2 * u : u / SPRv;
            const int rem_c = RPSv == 2 ? 0 : (u % SPRv) * NT;
            const int dt = dthr_c, gt = gthr_c, tr = trem_c, th = RPSv == 2 ? thalf_c : 0;
            const int y = y_al - 1 + s_c + th;      // (one row per slot: wave-uniform, the row checks are scalar)
            const bool ok = (tr < RI - rem_c) && (s_c + th < C::NSLOT) && (y >= 0) && (y < H);
            const int nm = ok ? 0 : -1;      // "| nm" instead of "ok ? x : -1": the latter compiles to an exec-masked branch
            dst = (dt + (s_c * PITCH + (rem_c / SEG) * PSTRIDE)) | nm;
            gp = (gt + (s_c * W + rem_c / SEG)) | nm;          // relative to the row above the tile's first (LDS row 0)
        } else if constexpr (AL) {
            int it = tid + u * NT;
            // recompute per use: hoisted out of the chunk loop the descriptors cost 2U registers and spill - except in the
            // 56x56 kernel with a B queue of 8, where keeping them resident saves 6 % (measured, tools/variant_sweep2.sh)
            if constexpr (!HOIST) asm volatile("" : "+v"(it));
            const int s = it / (W * SEG);
            const int rem = it - s * (W * SEG);
            const bool in_row = it < NITEM;
            const int px = rem / SEG, seg = rem - px * SEG;
            const int y = y_al - 1 + s;
            const bool ok = in_row && (y >= 0) && (y < H);
            const int nm = ok ? 0 : -1;
            dst = ((s * PITCH + (px + 1) * PSTRIDE + seg * 8) | (seg << 28)) | nm;
            gp = (s * W + px) | nm;
        } else {
            dst = sdst[u]; gp = sgp[u];
            if constexpr (POOL) amo = sam[u];
        }
    };
    // element offset of item u's load (row-wise layout): (thread part) + (uniform part of u); nothing to load: the tile's
    // first pixel
    auto item_e = [&](const int u) -> unsigned {
        constexpr int RPSv = RPS, SPRv = SPR;
        const int s_c = RPSv == 2 ? 2 * u : u / SPRv;
        const int rem_c = RPSv == 2 ? 0 : (u % SPRv) * NT;
        const int tr = trem_c, th = RPSv == 2 ? thalf_c : 0;
        const unsigned et = ethr_c;
        const int y = y_al - 1 + s_c + th;
        const bool ok = (tr < RI - rem_c) && (s_c + th < C::NSLOT) && (y >= 0) && (y < H);
        return ok ? et + (unsigned)(s_c * W + rem_c / SEG) * in_pix_stride_ : (unsigned)W * in_pix_stride_;
    };
    // BLK: the same item as a pixel index relative to the row above the tile's first (LDS row 0); nothing to load: the tile's first pixel
    auto item_p = [&](const int u) -> int {
        constexpr int RPSv = RPS, SPRv = SPR;
        const int s_c = RPSv == 2 ? 2 * u : u / SPRv;
        const int rem_c = RPSv == 2 ? 0 : (u % SPRv) * NT;
        const int tr = trem_c, th = RPSv == 2 ? thalf_c : 0;
        const int gt = gthr_c;
        const int y = y_al - 1 + s_c + th;
        const bool ok = (tr < RI - rem_c) && (s_c + th < C::NSLOT) && (y >= 0) && (y < H);
        return ok ? gt + (s_c * W + rem_c / SEG) : W;
    };
    // POOL: stage at LOW resolution - one item = 4 channels of one pooled pixel, loaded and split once, then written
    // (or zero) to the 4 pixels of its window: 4x fewer loads and splits than per-pixel staging.
    //   map-aligned tiles: pooled rows y_al/2 - 1 .. y_al/2 + R/2 (the first / last only reach the tile's halo row);
    //   tiles that can straddle maps: one candidate per LDS row s_top = -1 .. NSLOT-1 holding a window's TOP row (even
    //   y; its bottom row is the next LDS row, the same map), descriptors in registers.
    constexpr bool LOSTAGE = POOL;
    constexpr int RL = AL ? C::R / 2 + 2 : C::NSLOT + 1;
    constexpr int NITEM_LO = RL * WO * SEG;
    // row-wise item layout as above: RPSL pooled rows per slot (narrow) or SPRL slots per pooled row (wide)
    constexpr int RIL = WO * SEG;
    constexpr bool ROWMAP_LO = LOSTAGE && AL && (LRPXH_ROWMAP != 0);
    constexpr int RPSL = RIL <= NT ? NT / RIL : 1;
    constexpr int SPRL = (RIL + NT - 1) / NT;
    constexpr int UL = !LOSTAGE ? 1 : (ROWMAP_LO ? (RIL <= NT ? (RL + RPSL - 1) / RPSL : RL * SPRL) : (NITEM_LO + NT - 1) / NT);
    const int lo_q = tid / RIL, lo_r = tid - lo_q * RIL;          // (row within the slot, item within the row) of this thread
    constexpr int ULR = (LOSTAGE && !AL) ? UL : 1;
    int ldst[ULR], lgp[ULR], lam[ULR];
    float lsc[ULR];
    if constexpr (LOSTAGE && !AL) {
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int it = tid + u * NT;
            ldst[u] = 0; lgp[u] = -1; lam[u] = 0; lsc[u] = 1.f;
            if (it < NITEM_LO) {
                const int c_ = it / (WO * SEG);
                const int rem = it - c_ * (WO * SEG);
                const int pxl = rem / SEG, seg = rem - pxl * SEG;
                const int s0 = c_ - 1;
                const long v_ = v0 - 1 + s0;
                const long n = v_ >= 0 ? v_ / (H + 1) : 0;
                const int y = (int)(v_ - n * (H + 1));
                if ((v_ >= 0) && (y < H) && ((y & 1) == 0) && (n < a.n_maps)) {
                    const int rowmask = (s0 >= 0 ? 1 : 0) | (s0 + 1 < C::NSLOT ? 2 : 0);
                    const int lo = (y >> 1) * WO + pxl;
                    const long img = a.map2img ? a.map2img[n] : n;
                    ldst[u] = ((s0 < 0 ? 0 : s0) * PITCH + (2 * pxl + 1) * PSTRIDE + seg * 8) | (seg << 28) | (rowmask << 26);
                    lgp[u] = (int)(n * (HO * WO) + lo);
                    lam[u] = (int)((img * (HO * WO) + lo) * a.cin);
                    if constexpr (!B6) lsc[u] = exp2i(split_scale_exp<F8>(in_amax[n]));
                }
            }
        }
    }
    auto item_lo = [&](const int u, int& dst00, int& gp, int& amo, int& rowmask, float& sc) {
        if constexpr (AL) {
            int it = tid + u * NT;
            if constexpr (!HOIST) asm volatile("" : "+v"(it));
            int sl, rem;
            bool in_row;
            if constexpr (ROWMAP_LO) {
                int q_ = lo_q, r_ = lo_r;
                if constexpr (!HOIST) asm volatile("" : "+v"(q_), "+v"(r_));
                if constexpr (RIL <= NT) { sl = u * RPSL + q_; rem = r_; in_row = (q_ < RPSL) && (sl < RL); }
                else { sl = u / SPRL; rem = it - (u / SPRL) * SPRL * NT; in_row = rem < RIL; }
            } else {
                sl = it / (WO * SEG);
                rem = it - sl * (WO * SEG);
                in_row = it < NITEM_LO;
            }
            const int pxl = rem / SEG, seg = rem - pxl * SEG;
            const int ylo = (y_al >> 1) - 1 + sl;
            const bool ok = in_row && (ylo >= 0) && (ylo < HO);
            const int s0 = 2 * sl - 1;                               // LDS row of window row dy = 0 (dy = 1: s0 + 1)
            rowmask = ok ? ((s0 >= 0 ? 1 : 0) | (s0 + 1 < C::NSLOT ? 2 : 0)) : 0;
            dst00 = ((s0 < 0 ? 0 : s0) * PITCH + (2 * pxl + 1) * PSTRIDE + seg * 8) | (seg << 28);   // first row written
            gp = ok ? sl * WO + pxl : -1;        // relative to the pooled row above the tile's first
            amo = 0;
            sc = ssc[0];
        } else {
            dst00 = ldst[u]; gp = lgp[u]; amo = lam[u]; sc = lsc[u];
            rowmask = gp >= 0 ? (dst00 >> 26) & 3 : 0;
        }
    };
    // input addressing: NHWC (pixel stride cin, chunk step KC) or channel-chunked (pixel stride KC, chunk step = stride)
    // Addresses are (workgroup-uniform 64-bit base of the tile) + (32-bit element offset of the item): one v_mul_u32_u24 per
    // item instead of two v_mad_u64_u32 - the staging loads of the 4-row 112x112 tile (12 items per thread) spent 20 % of
    // a wave's time on address arithmetic (s_memtime stamps).  The base is the row ABOVE the tile's first row (LDS row 0),
    // so offsets are never negative; an item with nothing to load reads the tile's first pixel (always inside the tensor).
    const unsigned in_pix_stride = a.in_chunk_stride ? KC : a.cin;
    const long in_chunk_step = BLK ? blk_chunk_stride(POOL ? (long)a.n_maps * (HO * WO) : total_pix)
                                   : (a.in_chunk_stride ? a.in_chunk_stride : KC);
    const float* __restrict__ in_tile = a.in + (g0 - 1) * W * (long)in_pix_stride + c_begin * in_chunk_step;
    const unsigned char* __restrict__ am_tile = a.pool_am;
    // BLK (blocked.h): chunk c of pixel gp at a.in + c * in_chunk_step + blk_pix_off(gp); gp = blk_pix0 + (the item's pixel relative
    // to the row above the tile's first / to the pooled row above it); parts 1..3 of the slice 128 floats apart
    int blk_pix0 = (int)((g0 - 1) * W);
    if constexpr (BLK) in_tile = a.in;
    if constexpr (LOSTAGE && AL) {
        const long lo_row = (long)(y_al >> 1) - 1;
        if constexpr (!BLK) in_tile = a.in + ((long)n_al * (HO * WO) + lo_row * WO) * a.cin;
        blk_pix0 = (int)((long)n_al * (HO * WO) + lo_row * WO);
        am_tile = a.pool_am + (img_al * (HO * WO) + lo_row * WO) * a.cin;
    }
    constexpr int VSTEP = BLK ? 32 : 1;       // float4 units between the four parts of a staging item's slice
    f32x4 sv[LOSTAGE ? UL : U][NV];
    unsigned amv[(POOL && !X6) ? (LOSTAGE ? UL : U) : 1];
    u32x4_ amv4[(POOL && X6) ? (LOSTAGE ? UL : U) : 1];      // X6: the 16 winner bytes of the item's slice
#define LRPXH_ISSUE_LO(CHUNK) _Pragma("unroll") for (int u = 0; u < UL; ++u) LRPXH_ISSUE_LO1(u, CHUNK)
#define LRPXH_COMMIT_LO(BUFIDX) _Pragma("unroll") for (int u = 0; u < UL; ++u) LRPXH_COMMIT_LO1(u, BUFIDX)
#define LRPXH_ISSUE_LO1(u, CHUNK)                                                                            \
    {                                                                                                        \
        int dst_, gp_, amo_, rm_;                                                                            \
        float sc_;                                                                                           \
        item_lo(u, dst_, gp_, amo_, rm_, sc_);                                                               \
        const int sg_ = ((dst_ >> 28) & 3) * 4;                                                              \
        const float* sp_;                                                                                    \
        const unsigned char* ap_;                                                                            \
        if constexpr (AL) {                                                                                  \
            const unsigned e_ = __umul24((unsigned)(gp_ >= 0 ? gp_ : WO), (unsigned)a.cin) + (unsigned)sg_;  \
            sp_ = in_tile + (CHUNK) * KC + e_;                                                               \
            ap_ = am_tile + (CHUNK) * KC + e_;                                                               \
            if constexpr (BLK) sp_ = in_tile + (CHUNK) * in_chunk_step + blk_pix_off32(blk_pix0 + (gp_ >= 0 ? gp_ : WO)); \
        } else {                                                                                             \
            const long g_ = gp_ >= 0 ? gp_ : 0;                                                              \
            const int am_ = gp_ >= 0 ? amo_ : 0;                                                             \
            sp_ = a.in + g_ * a.cin + (CHUNK) * KC + sg_;                                                    \
            ap_ = a.pool_am + am_ + (CHUNK) * KC + sg_;                                                      \
            if constexpr (BLK) sp_ = in_tile + (CHUNK) * in_chunk_step + blk_pix_off32((int)g_);    \
        }                                                                                                    \
        _Pragma("unroll") for (int v_ = 0; v_ < NV; ++v_) sv[u][v_] = reinterpret_cast<const f32x4*>(sp_)[v_ * VSTEP]; \
        if constexpr (X6) amv4[u] = *reinterpret_cast<const u32x4_*>(ap_);                                   \
        else amv[u] = *reinterpret_cast<const unsigned*>(ap_);                                               \
    }
// The four window positions of an item differ only in the lanes that are kept: byte masks from one SWAR compare of the
// four winner bytes (values 0..3) per position, halfword masks by v_perm_b32, row bases selected once per item (an absent
// row - outside the LDS tile or nothing to load - goes to 256 scratch bytes behind the buffers), the position inside the
// window as an immediate offset: ~90 VALU per item instead of 170 (46 v_cndmask, 16 compares, ...: the commit of the
// pooled-input kernels cost 19-22 % of their time, timing experiment without commits).
#define LRPXH_COMMIT_LO1(u, BUFIDX)                                                                          \
    {                                                                                                        \
        int dst_, gp_, amo_, rm_;                                                                            \
        float sc_;                                                                                           \
        item_lo(u, dst_, gp_, amo_, rm_, sc_);                                                               \
        if constexpr (X6) { LRPXH_COMMIT_LO1_X6(u, BUFIDX) } else if constexpr (B6) { LRPXH_COMMIT_LO1_B6(u, BUFIDX) } else { \
        const f32x2_ xa_ = f32x2_{sv[u][0][0], sv[u][0][1]} * f32x2_{sc_, sc_}, xb_ = f32x2_{sv[u][0][2], sv[u][0][3]} * f32x2_{sc_, sc_}; \
        unsigned hw0_, hw1_, lw0_, lw1_;                                                                     \
        f32x2_ ha_, hb_;                                                                                     \
        split2_pk(xa_, hw0_, lw0_, ha_); split2_pk(xb_, hw1_, lw1_, hb_);                                    \
        unsigned w8_ = 0, wl8_ = 0;                                                                          \
        if constexpr (F8) {                                                                                  \
            const f32x2_ ya_ = xa_ * f32x2_{0.0625f, 0.0625f}, yb_ = xb_ * f32x2_{0.0625f, 0.0625f};         \
            const f32x2_ ra_ = (xa_ - ha_) * f32x2_{16.f, 16.f}, rb_ = (xb_ - hb_) * f32x2_{16.f, 16.f};     \
            w8_ = pack_fp8x4(ya_[0], ya_[1], yb_[0], yb_[1]);                                                \
            wl8_ = pack_fp8x4(ra_[0], ra_[1], rb_[0], rb_[1]);                                               \
        }                                                                                                    \
        const int o0_ = (BUFIDX) * BUFB + (dst_ & 0x03ffffff);                                               \
        const int sg4_ = ((dst_ >> 28) & 3) * 4;                                                             \
        const int rb0_ = ((rm_ & 1) && !(LRPXH_EXP & 4)) ? o0_ : POOL_SCRATCH;   /* window row dy = 0 */    \
        const int rb1_ = ((rm_ & 2) && !(LRPXH_EXP & 4)) ? o0_ + ((rm_ & 1) ? PITCH : 0) : POOL_SCRATCH;     \
        _Pragma("unroll") for (int pos = 0; pos < 4; ++pos) {                                                \
            const unsigned x_ = amv[u] ^ (0x01010101u * (unsigned)pos);       /* zero byte <=> winner == pos */ \
            const unsigned eq_ = ((x_ | (x_ >> 1)) & 0x01010101u) ^ 0x01010101u;                             \
            const unsigned bm_ = (eq_ << 8) - eq_;                            /* 0xff per winner byte */     \
            const unsigned m01_ = __builtin_amdgcn_perm(bm_, bm_, 0x01010000u);                              \
            const unsigned m23_ = __builtin_amdgcn_perm(bm_, bm_, 0x03030202u);                              \
            char* d_ = ldsb + ((pos >> 1) ? rb1_ : rb0_) + (pos & 1) * PSTRIDE;                              \
            *reinterpret_cast<u32x2_*>(d_) = u32x2_{hw0_ & m01_, hw1_ & m23_};                               \
            if constexpr (F8) {                                                                              \
                *reinterpret_cast<unsigned*>(d_ - sg4_ + 32) = w8_ & bm_;                                    \
                *reinterpret_cast<unsigned*>(d_ - sg4_ + 48) = wl8_ & bm_;                                   \
            } else {                                                                                         \
                *reinterpret_cast<u32x2_*>(d_ + 32) = u32x2_{lw0_ & m01_, lw1_ & m23_};                      \
            }                                                                                                \
        }                                                                                                    \
        }                                                                                                    \
    }
// B6: the three bf16 planes of the item's 4 channels, split once, then masked per window position (the masks of the fp16 path)
#define LRPXH_COMMIT_LO1_B6(u, BUFIDX)                                                                       \
    {                                                                                                        \
        unsigned pa_[3], pb_[3];                                                                             \
        split3_pk(f32x2_{sv[u][0][0], sv[u][0][1]}, pa_[0], pa_[1], pa_[2]);                                 \
        split3_pk(f32x2_{sv[u][0][2], sv[u][0][3]}, pb_[0], pb_[1], pb_[2]);                                 \
        const int o0_ = (BUFIDX) * BUFB + (dst_ & 0x03ffffff);                                               \
        const int rb0_ = (rm_ & 1) ? o0_ : POOL_SCRATCH;                         /* window row dy = 0 */    \
        const int rb1_ = (rm_ & 2) ? o0_ + ((rm_ & 1) ? PITCH : 0) : POOL_SCRATCH;                           \
        _Pragma("unroll") for (int pos = 0; pos < 4; ++pos) {                                                \
            const unsigned x_ = amv[u] ^ (0x01010101u * (unsigned)pos);       /* zero byte <=> winner == pos */ \
            const unsigned eq_ = ((x_ | (x_ >> 1)) & 0x01010101u) ^ 0x01010101u;                             \
            const unsigned bm_ = (eq_ << 8) - eq_;                            /* 0xff per winner byte */     \
            const unsigned m01_ = __builtin_amdgcn_perm(bm_, bm_, 0x01010000u);                              \
            const unsigned m23_ = __builtin_amdgcn_perm(bm_, bm_, 0x03030202u);                              \
            char* d_ = ldsb + ((pos >> 1) ? rb1_ : rb0_) + (pos & 1) * PSTRIDE;                              \
            _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_)                                              \
                *reinterpret_cast<u32x2_*>(d_ + 32 * pl_) = u32x2_{pa_[pl_] & m01_, pb_[pl_] & m23_};        \
        }                                                                                                    \
    }
// X6: the item is the whole 16-channel slice of a pooled pixel.  Split once; per window position the channels that did not win
// there are zeroed on the fp32 values (bit masks replicated from the SWAR compare of the 16 winner bytes), then hi = cvt_pk of the
// masked values and one fp6 conversion with the slice's common block scale (the maximum over all 16 channels: a position that
// keeps only small entries is rounded against the pooled pixel's largest one - the same absolute error as in a layer without pool)
#ifndef LRPXH_LO_SKIP
#define LRPXH_LO_SKIP 1       // wave-uniform skips in the pooled commit: a wave none of whose lanes has the window row does not compute it
#endif
#define LRPXH_COMMIT_LO1_X6(u, BUFIDX)                                                                       \
    if (!LRPXH_LO_SKIP || __builtin_amdgcn_ballot_w64(rm_ != 0) != 0) {                                      \
        unsigned hwu_[8], rwu_[8], sb_;                                                                      \
        float bs_;                                                                                           \
        x6_split(sv[u], sc_, hwu_, rwu_, bs_, sb_);                                                          \
        const int o0_ = (BUFIDX) * BUFB + (dst_ & 0x03ffffff);                                               \
        const int rb0_ = ((rm_ & 1) && !(LRPXH_EXP & 4)) ? o0_ : POOL_SCRATCH;   /* window row dy = 0 */    \
        const int rb1_ = ((rm_ & 2) && !(LRPXH_EXP & 4)) ? o0_ + ((rm_ & 1) ? PITCH : 0) : POOL_SCRATCH;     \
        const bool need0_ = !LRPXH_LO_SKIP || __builtin_amdgcn_ballot_w64((rm_ & 1) != 0) != 0;              \
        const bool need1_ = !LRPXH_LO_SKIP || __builtin_amdgcn_ballot_w64((rm_ & 2) != 0) != 0;              \
        _Pragma("unroll") for (int pos = 0; pos < 4; ++pos) if ((pos >> 1) ? need1_ : need0_) {              \
            unsigned hm_[8], rm2_[8];                                                                        \
            _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) {                                               \
                const unsigned x_ = amv4[u][q_] ^ (0x01010101u * (unsigned)pos);  /* zero byte <=> winner == pos */ \
                const unsigned eq_ = ((x_ | (x_ >> 1)) & 0x01010101u) ^ 0x01010101u;                         \
                /* v_perm_b32 selector bytes 0x0c / 0x0d yield the constants 0x00 / 0xff: halfword masks of channels 4q .. 4q + 3 */ \
                const unsigned sel_ = eq_ + 0x0c0c0c0cu;                                                     \
                const unsigned m01_ = __builtin_amdgcn_perm(0u, 0u, __builtin_amdgcn_perm(sel_, sel_, 0x01010000u)); \
                const unsigned m23_ = __builtin_amdgcn_perm(0u, 0u, __builtin_amdgcn_perm(sel_, sel_, 0x03030202u)); \
                hm_[2 * q_] = hwu_[2 * q_] & m01_; hm_[2 * q_ + 1] = hwu_[2 * q_ + 1] & m23_;                \
                rm2_[2 * q_] = rwu_[2 * q_] & m01_; rm2_[2 * q_ + 1] = rwu_[2 * q_ + 1] & m23_;              \
            }                                                                                                \
            const u32x6_ q6_ = x6_pack(hm_, rm2_, bs_);                                                      \
            char* d_ = ldsb + ((pos >> 1) ? rb1_ : rb0_) + (pos & 1) * PSTRIDE;                              \
            *reinterpret_cast<u32x4_*>(d_) = u32x4_{hm_[0], hm_[1], hm_[2], hm_[3]};                         \
            *reinterpret_cast<u32x4_*>(d_ + 16) = u32x4_{hm_[4], hm_[5], hm_[6], hm_[7]};                    \
            *reinterpret_cast<u32x4_*>(d_ + 32) = u32x4_{q6_[0], q6_[1], q6_[2], q6_[3]};                    \
            *reinterpret_cast<u32x4_*>(d_ + 48) = u32x4_{q6_[4], q6_[5], 0u, sb_};                           \
        }                                                                                                    \
    }
#define LRPXH_ISSUE(CHUNK) _Pragma("unroll") for (int u = 0; u < U; ++u) LRPXH_ISSUE1(u, CHUNK)
#define LRPXH_COMMIT(BUFIDX) _Pragma("unroll") for (int u = 0; u < U; ++u) LRPXH_COMMIT1(u, BUFIDX)
// Branch-free: an item that has nothing to load reads element 0 (and is zeroed at commit), an item that has nothing to
// write targets the 16 pad bytes of LDS pixel 0 - no exec-mask regions, so the scheduler can place these instructions
// between the MFMAs of a tap.
#define LRPXH_ISSUE1(u, CHUNK)                                                                               \
    {                                                                                                        \
        int dst_, gp_, amo_;                                                                                 \
        unsigned e_;                                                                                         \
        if constexpr (BLK) {                                                                                 \
            int rel_;                                                                                        \
            if constexpr (AL && ROWMAP && !POOL) rel_ = item_p(u);                                           \
            else { item(u, dst_, gp_, amo_); rel_ = gp_ >= 0 ? gp_ : W; }                                    \
            e_ = blk_pix_off32(blk_pix0 + rel_);                                                             \
        } else if constexpr (AL && ROWMAP && !POOL) {                                                        \
            e_ = item_e(u);                                                                                  \
        } else {                                                                                             \
            item(u, dst_, gp_, amo_);                                                                        \
            const int sg_ = dst_ >= 0 ? ((dst_ >> 28) & 3) * 4 : 0;                                          \
            e_ = __umul24((unsigned)(gp_ >= 0 ? gp_ : W), in_pix_stride) + (unsigned)sg_;                    \
        }                                                                                                    \
        _Pragma("unroll") for (int v_ = 0; v_ < NV; ++v_)                                                    \
            sv[u][v_] = reinterpret_cast<const f32x4*>(in_tile + (CHUNK) * in_chunk_step + e_)[v_ * VSTEP];  \
    }
#define LRPXH_COMMIT1(u, BUFIDX)                                                                             \
    {                                                                                                        \
        int dst_, gp_, amo_;                                                                                 \
        item(u, dst_, gp_, amo_);                                                                            \
        /* (an item with nothing to write converts whatever its registers hold into the scratch bytes) */    \
        char* d_ = ldsb + ((dst_ >= 0 && !(LRPXH_EXP & 4)) ? (BUFIDX) * BUFB + (dst_ & 0x03ffffff) : STAGE_SCRATCH); \
        if constexpr (X6) {                                                                                  \
            if (!LRPXH_LO_SKIP || __builtin_amdgcn_ballot_w64(dst_ >= 0) != 0) {                             \
            unsigned hwu_[8], rwu_[8], sb_;                                                                  \
            float bs_;                                                                                       \
            x6_split(sv[u], ssc[AL ? 0 : u], hwu_, rwu_, bs_, sb_);                                          \
            const u32x6_ q6_ = x6_pack(hwu_, rwu_, bs_);                                                     \
            *reinterpret_cast<u32x4_*>(d_) = u32x4_{hwu_[0], hwu_[1], hwu_[2], hwu_[3]};                     \
            *reinterpret_cast<u32x4_*>(d_ + 16) = u32x4_{hwu_[4], hwu_[5], hwu_[6], hwu_[7]};                \
            *reinterpret_cast<u32x4_*>(d_ + 32) = u32x4_{q6_[0], q6_[1], q6_[2], q6_[3]};                    \
            *reinterpret_cast<u32x4_*>(d_ + 48) = u32x4_{q6_[4], q6_[5], 0u, sb_};                           \
            }                                                                                                \
        } else if constexpr (B6) {                                                                           \
            unsigned pa_[3], pb_[3];                                                                         \
            split3_pk(f32x2_{sv[u][0][0], sv[u][0][1]}, pa_[0], pa_[1], pa_[2]);                             \
            split3_pk(f32x2_{sv[u][0][2], sv[u][0][3]}, pb_[0], pb_[1], pb_[2]);                             \
            _Pragma("unroll") for (int pl_ = 0; pl_ < 3; ++pl_)                                              \
                *reinterpret_cast<u32x2_*>(d_ + 32 * pl_) = u32x2_{pa_[pl_], pb_[pl_]};                      \
        } else {                                                                                             \
        const f32x2_ sc2_ = {ssc[AL ? 0 : u], ssc[AL ? 0 : u]};                                              \
        const f32x2_ xa_ = f32x2_{sv[u][0][0], sv[u][0][1]} * sc2_, xb_ = f32x2_{sv[u][0][2], sv[u][0][3]} * sc2_; \
        unsigned hw0_, hw1_, lw0_, lw1_;                                                                     \
        f32x2_ ha_, hb_;                                                                                     \
        split2_pk(xa_, hw0_, lw0_, ha_); split2_pk(xb_, hw1_, lw1_, hb_);                                    \
        *reinterpret_cast<u32x2_*>(d_) = u32x2_{hw0_, hw1_};                                                 \
        if constexpr (F8) {                                                                                  \
            const int sg4_ = ((dst_ >> 28) & 3) * 4;                                                         \
            const f32x2_ ya_ = xa_ * f32x2_{0.0625f, 0.0625f}, yb_ = xb_ * f32x2_{0.0625f, 0.0625f};         \
            const f32x2_ ra_ = (xa_ - ha_) * f32x2_{16.f, 16.f}, rb_ = (xb_ - hb_) * f32x2_{16.f, 16.f};     \
            *reinterpret_cast<unsigned*>(d_ - sg4_ + 32) = pack_fp8x4(ya_[0], ya_[1], yb_[0], yb_[1]);       \
            *reinterpret_cast<unsigned*>(d_ - sg4_ + 48) = pack_fp8x4(ra_[0], ra_[1], rb_[0], rb_[1]);       \
        } else {                                                                                             \
            *reinterpret_cast<u32x2_*>(d_ + 32) = u32x2_{lw0_, lw1_};                                        \
        }                                                                                                    \
        }                                                                                                    \
    }

    // STAGGER (8-wave workgroups, double-buffered): the two waves of a SIMD (wave w and w + 4 of the workgroup) share every
    // barrier, so with one schedule for all waves both sit in their MFMA phase together and in their staging phase
    // together - the matrix pipe idles while both convert and write LDS (measured: pipe busy 58 %).  Half a period of
    // skew fixes that without another barrier: the upper half of the waves ("group 1") commits its share of chunk c + 1
    // at the START of interval c (loaded one interval earlier) and then runs its MFMAs, the lower half runs its MFMAs
    // first and commits at the end - on every SIMD one wave's conversions now run in the shadow of the other's MFMAs.
    // Measured: +1 % (a single wave's MFMA phase does not saturate the pipe, so the overlap buys less than the idle share
    // suggests); grouping odd / even waves instead is 1 % slower than no skew: w and w + 4 are the SIMD partners.
    // Buffers: chunk c is read from buffer c & 1 between barrier c - 1 and barrier c by everyone; writes to it happen
    // after barrier c - 2 (group 1, chunk c: right behind that barrier; group 0: at the end of interval c - 1).
#ifndef LRPXH_STAGGER
#define LRPXH_STAGGER 1
#endif
#ifndef LRPXH_EXP
#define LRPXH_EXP 0       // timing experiments (wrong results): 1 = no staging commits in the K loop, 2 = no staging loads, 16 = no A-operand reads,
                          // 32 = no epilogue, 64 = no barrier in the (double-buffered) K loop
#endif
    constexpr bool STAG = DB && (LRPXH_STAGGER != 0) && (MT * NWN >= 8);
#ifdef LRPXH_ISSUE_LATE
    constexpr bool ISSUE_LATE = LRPXH_ISSUE_LATE != 0;
#else
    constexpr bool ISSUE_LATE = (HW == 112 && MT == 2);
#endif
    const int grp = STAG ? (wave >= MT * NWN / 2 ? 1 : 0) : 0;     // wave-uniform (waves w and w + 4 share a SIMD)
    if constexpr (!(LRPXH_EXP & 8)) { if constexpr (LOSTAGE) { LRPXH_ISSUE_LO(0) } else { LRPXH_ISSUE(0) } }
    else {      // (EXP 8: what the exposed first load of a tile costs - zeros instead)
#pragma unroll
        for (int u = 0; u < (LOSTAGE ? UL : U); ++u) {
#pragma unroll
            for (int v_ = 0; v_ < NV; ++v_) sv[u][v_] = f32x4{0.f, 0.f, 0.f, 0.f};
            if constexpr (POOL && !X6) amv[u] = 0;
            if constexpr (POOL && X6) amv4[u] = u32x4_{0, 0, 0, 0};
        }
    }
    for (int i = tid; i < NBUF * BUFB / 16; i += NT) reinterpret_cast<u32x4_*>(ldsb)[i] = u32x4_{0, 0, 0, 0};
    __syncthreads();
    if constexpr (LOSTAGE) { LRPXH_COMMIT_LO(0) } else { LRPXH_COMMIT(0) }
    if constexpr (STAG) {
        if (grp == 1 && nchunk > 1) { if constexpr (LOSTAGE) { LRPXH_ISSUE_LO(1) } else { LRPXH_ISSUE(1) } }
    }
    f32x16 acc[7];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

    // B fragments: per k-step two planes (hi, lo) of 64 lanes x 16 B, one contiguous stream per channel block
#ifdef LRPXH_NBQ
    constexpr int NBQ = LRPXH_NBQ;
#else
    // measured (tools/variant_sweep2.sh, chain of 320 maps): depth 2 -> 28.9 ms, 4 -> 26.9, 6 -> 25.9, 8 -> 25.0; the
    // map-straddling tiles (28/14) spill beyond 7
    constexpr int NBQ_H3 = !AL ? 7 : (HW == 224 ? (POOL ? 10 : 5) : (((HW == 112 && NWN == 2) || (HOIST && !POOL)) ? 8 : 9));
    // B6: three planes per entry (12 registers)
    // (measured, same-box A/B of builds: the map-aligned pooled-input kernels have the registers for a deeper queue - their staging items
    // are a quarter as many - conv1_2 5.16 -> 5.04 ms, conv2_2 4.41 -> 4.25; one entry more spills in every other instantiation)
#ifndef LRPXB6_NBQ
#define LRPXB6_NBQ ((POOL && AL) ? (HW == 112 ? 7 : 6) : ((HW <= 56) ? 5 : 4))
#endif
    constexpr int NBQ = B6 ? LRPXB6_NBQ : NBQ_H3;
#endif
    float inv_w = 1.f;
    if constexpr (!B6) inv_w = a.wp[0];
    // F8: one queue entry per tap ROW g = 7 planes of 64 lanes x 16 B: fp16 hi of dx = 0,1,2 | the B operands of fp8
    // MFMAs 2g and 2g+1 (2 planes each = the lane's two slots x 16 channels; row 2 has MFMA 4 only)
    constexpr int BP = F8 ? 7 : (B6 ? 3 : 2);       // planes per queue entry
    constexpr int BSTEPS = F8 ? 3 : TAPS;           // queue entries per K-chunk
    // measured (tools/variant_sweep3.sh, chain of 320 maps): 2 entries 24.1 ms, 3 entries 25.1 (spills), 4: 30.1; the
    // pooled-input 112 / 56 kernels have the registers for a third entry (2.39 -> 2.27 ms, 1.96 -> 1.89 ms)
#ifdef LRPXH_NBQ8
    constexpr int NQ = F8 ? LRPXH_NBQ8 : NBQ;
#else
#ifndef LRPXH_NQ_112P
#define LRPXH_NQ_112P 3
#endif
    // X6 (a staging item holds 16 floats): a third entry spills everywhere
    #ifndef LRPXH_X6_NQ3
#define LRPXH_X6_NQ3 0
#endif
    constexpr int NQ = F8 ? (((!X6 || (LRPXH_X6_NQ3 && HW == 112 && MT * NWN == 4)) && POOL && AL && HW <= 112) ? ((HW == 112 && MT * NWN == 4) ? LRPXH_NQ_112P : 3) : 2) : NBQ;
#endif
    // (a wave without a channel block of its own - n_oc not a multiple of the workgroup's channels - multiplies the last
    // valid block again and drops the result: one code path, see PRECISE below)
    const int ocb_w = wave_active ? ocb : (a.n_oc - 1) / 32;
    const u32x4_* wp = reinterpret_cast<const u32x4_*>(a.wp + (B6 ? 0 : F16X3_HEADER_FLOATS)) +
                       ((long)ocb_w * nchunk_all + c_begin) * (BSTEPS * BP * 64) + lane;
    const int last_step = nchunk * BSTEPS - 1;
    // plane p of queue entry `e`; X6: planes 4 and 6 hold 8 bytes of fp6 + the block-scale dword - 12 bytes, loaded as such
    auto ldb = [&](const long e, const int p) -> u32x4_ {
        if constexpr (X6) {
            if (p == 4 || p == 6) {
                const u32x3_ t = *reinterpret_cast<const u32x3_*>(wp + (e * BP + p) * 64);
                return u32x4_{t[0], t[1], t[2], 0u};
            }
        }
        return wp[(e * BP + p) * 64];
    };
    u32x4_ bq[NQ][BP];
#pragma unroll
    for (int i = 0; i < NQ; ++i)
#pragma unroll
        for (int p = 0; p < BP; ++p) bq[i][p] = u32x4_{0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NQ - 1; ++i)
#pragma unroll
        for (int p = 0; p < BP; ++p) bq[i][p] = ldb(min(i, last_step), p);
    // F8: LDS byte offsets of the lane's two fp8 tap slots of a tap row (dx = 0 / 1 for lanes 0-31; dx = 2 / 2 for lanes
    // 32-63, whose second slot carries zero weights)
    // F8: lanes 0-31 read the fp8 plane of x - hi (byte 48 of the pixel), lanes 32-63 that of x (byte 32; abase[] already
    // carries + 16 for them)
    const int c8 = lh ? 16 : 48;
    // X6: the operand of fp6 MFMA mm = 0..4 is the 32 bytes at byte 32 of ONE pixel per lane: tap 2mm for lanes 0-31, tap 2mm + 1
    // for lanes 32-63 (mm = 4: tap 8 again, against zero weights) - the lane halves differ by one pixel (mm = 0, 2, 3), by a row
    // less two pixels (mm = 1) or not at all (mm = 4); abase[] carries + 16 for lanes 32-63
    const int e6a = 32 - 16 * lh + lh * PSTRIDE, e6b = 32 - 16 * lh + lh * (PITCH - 2 * PSTRIDE), e6c = 32 - 16 * lh;
    __syncthreads();
    if constexpr (STAG) {
        if (grp == 1 && nchunk > 1) { if constexpr (LOSTAGE) { LRPXH_COMMIT_LO(1) } else { LRPXH_COMMIT(1) } }
    }

    LRPXH_T(t_loop);
#ifndef LRPXH_CHUNK_UNROLL
#define LRPXH_CHUNK_UNROLL 1
#endif
#pragma unroll LRPXH_CHUNK_UNROLL
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const bool more = chunk + 1 < nchunk;
        LRPXH_REFRESH
        LRPXH_T(ta);
        // PRECISE: the K loop has NO branch around a memory instruction.  `s_waitcnt vmcnt` counts in order, and at a
        // control-flow merge the compiler must assume the path with the fewest younger loads: with `if (next chunk
        // exists) issue` and `if (wave has a channel block) multiply` the first MFMA of every chunk waited for ALL
        // staging loads just issued (vmcnt(7) with 7 B loads behind them) and the commit for the B prefetch too - the
        // whole global-load latency sat in front of the matrix phase.  Now the staging loads of the next chunk are
        // issued unconditionally (past the last chunk they re-read it; nobody commits them); ISSUE_LATE: AFTER the B
        // loads of the second tap row, so they have two thirds of the chunk's matrix time to land before the next
        // in-order wait that covers them (else at the top of the interval: one third).
        // (measured, same-box A/B of three builds: chain of 320 maps 19.37 -> 19.22 ms with the precise waits; issuing after
        // the first B loads is worth 4 % on the 4-row 112x112 tile - 12 staging items per thread - and costs 3 % through 5-10
        // spilled registers elsewhere, so only that kernel does it)
        // group 0 loads the next chunk, group 1 (which has already committed the next chunk) the one after it
#define LRPXH_ISSUE_NEXT                                                                        \
        if constexpr (!(LRPXH_EXP & 2)) {                                                       \
            const int cn_ = min(chunk + 1 + grp, nchunk - 1);                                   \
            if constexpr (LOSTAGE) { LRPXH_ISSUE_LO(cn_) } else { LRPXH_ISSUE(cn_) }            \
            __builtin_amdgcn_sched_barrier(0);                                                  \
        }
        if constexpr (!ISSUE_LATE) { LRPXH_ISSUE_NEXT }
        LRPXH_T(tb);
        {
            const char* abuf = ldsb + (DB ? (chunk & 1) : 0) * BUFB;
#ifdef LRPX_STAMP
            unsigned long long tprev = tb;
#endif
            // APIPE (wide maps with few K-chunks: the partner wave of the SIMD is mostly outside its MFMA phase, so this
            // wave's LDS latency is not covered by the partner's MFMAs): A fragments are read one accumulator tile ahead
            if constexpr (F8) {
#ifndef LRPXH_F8_SGB
#define LRPXH_F8_SGB 0
#endif
                // (experiment) software pipeline by scheduling groups: the LDS reads of accumulator tile t+1 are placed
                // before the MFMAs of tile t
                if constexpr (LRPXH_F8_SGB != 0) __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
#ifndef LRPXH_F8_PIPE
#define LRPXH_F8_PIPE 3
#endif
                                // 4-wave workgroups (twice the staging registers per thread): depth 1 pays on the pooled-input 224x224 kernel
                // only (conv1_2 2.94 -> 2.83 ms with 5 spilled registers; conv2_1 +15 %, conv2_2 +2 %, conv3_1 +-0 with depth 1
                // or 2 and their 10-25 spills - same-box A/B)
#ifndef LRPXH_F8_PIPE_TALL
#define LRPXH_F8_PIPE_TALL 1      // 8-wave workgroups with MT >= 2 (tall tiles of the <= 128-channel layers)
#endif
#ifdef LRPXH_F8_PIPE4
                constexpr int PIPE_D = (MT * NWN >= 8) ? (MT >= 2 ? LRPXH_F8_PIPE_TALL : LRPXH_F8_PIPE) : LRPXH_F8_PIPE4;
#else
#ifndef LRPXH_PIPE_4W
#define LRPXH_PIPE_4W 0        // the other 4-wave kernels
#endif
#ifndef LRPXH_PIPE_112P
#define LRPXH_PIPE_112P 1      // conv2_2: 2.26 -> 2.12 ms (depth 3 with a B queue of 2: the same)
#endif
#ifndef LRPXH_F8_PIPE_POOLS
#define LRPXH_F8_PIPE_POOLS LRPXH_F8_PIPE      // the map-straddling pooled-input 8-wave kernel (conv4_3): the only one that spills
#endif
#ifndef LRPXH_PIPE_4W_X6
#define LRPXH_PIPE_4W_X6 1     // fp6 build: the explicit operand ring keeps fewer temporaries alive than the compiler's own schedule (35 -> 1..4 spills)
#endif
                constexpr int PIPE_D = (MT * NWN >= 8) ? (MT >= 2 ? LRPXH_F8_PIPE_TALL : ((POOL && !AL) ? LRPXH_F8_PIPE_POOLS : LRPXH_F8_PIPE))
                                                       : ((HW == 224 && POOL) ? 1 : ((HW == 112 && POOL) ? LRPXH_PIPE_112P : (X6 ? LRPXH_PIPE_4W_X6 : LRPXH_PIPE_4W)));
#endif
                if constexpr (PIPE_D != 0) {
                // Operand pipeline of depth D = LRPXH_F8_PIPE.  Left to itself the compiler (at 240+ VGPRs) keeps ONE set of
                // A-fragment registers and emits read -> s_waitcnt lgkmcnt(0) -> MFMA for every MFMA: each one waits for its
                // own LDS round trip and only the partner wave fills the pipe (busy 79 % of the phase).  Here the 98 MFMAs
                // of a K-chunk are one sequence of ops k = (tap row g, tile j, kind m); the A operand of op k + D is read
                // before MFMA k is issued (D operand sets in a ring); `sched_barrier(0)` fences keep the compiler's scheduler
                // from sinking the reads back to their uses (scheduling groups alone did not hold the order).  Measured
                // (chain of 320 maps): 56x56 -10 %, 28x28 -7 %, 14x14 -9 % per launch at D = 2 (D = 3: the same).
                constexpr int D = PIPE_D != 0 ? PIPE_D : 1;
                constexpr int TPX[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 8};
                // op k -> (g, j, m): tap rows 0 and 1 have 5 MFMAs per tile (2 fp8 + 3 fp16), row 2 has 4 (1 fp8 + 3 fp16)
                auto og = [](const int k) constexpr { return k < 35 ? 0 : (k < 70 ? 1 : 2); };
                auto oj = [](const int k) constexpr { return k < 70 ? (k % 35) / 5 : (k - 70) / 4; };
                auto om = [](const int k) constexpr { return k < 70 ? (k % 35) % 5 : ((k - 70) % 4 == 0 ? 0 : (k - 70) % 4 + 1); };   // 0,1: fp8; 2..4: fp16 dx 0..2
                auto rd = [&](const int k) {
                    const int g = og(k), j = oj(k), m = om(k);
                    if constexpr ((LRPXH_EXP & 16) != 0) return i32x8_{lane, lane + 1, lane + 2, lane + 3, lane, lane, 0x7f, 0};   // (EXP 16: no LDS reads)
                    if (m < 2) {                                   // fp8 operand: taps 4g + 2m, 4g + 2m + 1
                        const int t = 4 * g + 2 * m;
                        if constexpr (X6) {
                            const char* a6 = abuf + abase[j] + (t == 2 ? e6b : (t == 8 ? e6c : e6a)) + (TPX[t] / 3) * PITCH + (TPX[t] % 3) * PSTRIDE;
                            // 16 + 8 + 4 bytes: every register a read returns is used (a register that is loaded but dead gets
                            // re-used by the allocator while the read is in flight, and the wait for that write-after-write
                            // hazard is a full drain of the counter)
                            const u32x4_ x0 = *reinterpret_cast<const u32x4_*>(a6);
#ifndef LRPXH_X6_READ
#define LRPXH_X6_READ 1
#endif
                            if constexpr (X6RD) {
                                // (round 4: the default except for conv2_1's 4-row 112 x 112 kernel, which spills with the extra live register:
                                // 1.05 -> 1.19 ms; same-box A/B of the rest: conv4_3 1.68 -> 1.60, conv1_2 2.50 -> 2.44, conv5_x 0.51 -> 0.50, conv2_2
                                // 1.89 -> 1.87, the others +-0.01: profiles/r04_ab_x6_read.txt)
                                // two 16-byte reads, conflict-free over consecutive pixels 80 bytes apart; the clear dword
                                // behind the fields is kept alive past the MFMA by an empty asm (see the MFMA below).  The 8- and 4-byte
                                // reads of the default hit their banks 2 and 4 times over pixels 80 bytes apart (SQ_LDS_BANK_CONFLICT
                                // 0.33 of the LDS cycles against 0.03 with fp8) - but the kernels are not bound by the LDS: same-box A/B
                                // chain 17.33 / 17.27 ms with the conflicts, 17.46 / 17.38 without (the extra live register costs conv2_1
                                // 12 spilled VGPRs: 1.10 -> 1.26 ms; every other layer +-0.01)
                                const u32x4_ x1 = *reinterpret_cast<const u32x4_*>(a6 + 16);
                                return i32x8_{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[3], (int)x1[2]};
                            }
                            const u32x2_ x1 = *reinterpret_cast<const u32x2_*>(a6 + 16);
                            const unsigned xs = *reinterpret_cast<const unsigned*>(a6 + 28);
                            return i32x8_{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)xs, 0};
                        }
                        const char* a8 = abuf + abase[j] + c8;
                        const u32x4_ x0 = *reinterpret_cast<const u32x4_*>(a8 + (TPX[t] / 3) * PITCH + (TPX[t] % 3) * PSTRIDE);
                        const u32x4_ x1 = *reinterpret_cast<const u32x4_*>(a8 + (TPX[t + 1] / 3) * PITCH + (TPX[t + 1] % 3) * PSTRIDE);
                        return i32x8_{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)x1[2], (int)x1[3]};
                    }
                    const u32x4_ x0 = *reinterpret_cast<const u32x4_*>(abuf + abase[j] + g * PITCH + (m - 2) * PSTRIDE);
                    return i32x8_{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], 0, 0, 0, 0};
                };
                i32x8_ ring[D];
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const long nxt = (long)min(chunk * 3 + g + NQ - 1, last_step);
#pragma unroll
                    for (int p = 0; p < BP; ++p)
                        if (p < 5 || (g + NQ - 1) % 3 != 2) bq[NQ - 1][p] = ldb(nxt, p);
                    if (g == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if constexpr (ISSUE_LATE) { LRPXH_ISSUE_NEXT }
#pragma unroll
                        for (int d = 0; d < D; ++d) ring[d] = rd(d);
                    }
                    const f16x8 bh[3] = {__builtin_bit_cast(f16x8, bq[0][0]), __builtin_bit_cast(f16x8, bq[0][1]),
                                         __builtin_bit_cast(f16x8, bq[0][2])};
                    const i32x8_ bm0 = {(int)bq[0][3][0], (int)bq[0][3][1], (int)bq[0][3][2], (int)bq[0][3][3],
                                        (int)bq[0][4][0], (int)bq[0][4][1], (int)bq[0][4][2], (int)bq[0][4][3]};
                    const i32x8_ bm1 = {(int)bq[0][5][0], (int)bq[0][5][1], (int)bq[0][5][2], (int)bq[0][5][3],
                                        (int)bq[0][6][0], (int)bq[0][6][1], (int)bq[0][6][2], (int)bq[0][6][3]};
                    constexpr int K0[3] = {0, 35, 70}, KN[3] = {35, 35, 28};
#pragma unroll
                    for (int kk = 0; kk < KN[g]; ++kk) {
                        const int k = K0[g] + kk;
                        const int j = oj(k), m = om(k);
                        const i32x8_ cur = ring[k % D];
                        if (k + D < 98) {
                            ring[k % D] = rd(k + D);
                            __builtin_amdgcn_sched_barrier(0);       // (a fence for the compiler's scheduler, no instruction)
                        }
                        if (m < 2) {
                            if constexpr (X6) {      // fp6 x fp6, block scales: dword 6 of either operand (byte 0)
                                if constexpr (TR)
                                    acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(m == 0 ? bm0 : bm1, cur, acc[j], 2, 2, 0,
                                                                                             m == 0 ? bm0[6] : bm1[6], 0, cur[6]);
                                else
                                    acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cur, m == 0 ? bm0 : bm1, acc[j], 2, 2, 0, cur[6], 0,
                                                                                             m == 0 ? bm0[6] : bm1[6]);
                            } else {
                                acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cur, m == 0 ? bm0 : bm1, acc[j], 0, 0, 0, 0, 0, 0);
                            }
                            if constexpr (X6RD) {      // (the unused 4th dword of the second read stays allocated until here)
                                const int keep = cur[7];
                                asm volatile("" : : "v"(keep));
                            }
                        } else {
                            const u32x4_ c4 = {(unsigned)cur[0], (unsigned)cur[1], (unsigned)cur[2], (unsigned)cur[3]};
                            if constexpr (TR) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[m - 2], __builtin_bit_cast(f16x8, c4), acc[j], 0, 0, 0);
                            else acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, c4), bh[m - 2], acc[j], 0, 0, 0);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
#pragma unroll
                    for (int i = 0; i < NQ - 1; ++i)
#pragma unroll
                        for (int p = 0; p < BP; ++p) bq[i][p] = bq[i + 1][p];
                }
                } else {
#pragma unroll
                for (int g = 0; g < 3; ++g) {
                    const long nxt = (long)min(chunk * 3 + g + NQ - 1, last_step);
                    // (the entry of tap row 2 has one fp8 MFMA: its last two planes are padding and stay out of registers)
#pragma unroll
                    for (int p = 0; p < BP; ++p)
                        if (p < 5 || (g + NQ - 1) % 3 != 2) bq[NQ - 1][p] = ldb(nxt, p);
                    if (g == 0 && ISSUE_LATE) { __builtin_amdgcn_sched_barrier(0); LRPXH_ISSUE_NEXT }
                    const f16x8 bh0 = __builtin_bit_cast(f16x8, bq[0][0]);
                    const f16x8 bh1 = __builtin_bit_cast(f16x8, bq[0][1]);
                    const f16x8 bh2 = __builtin_bit_cast(f16x8, bq[0][2]);
                    const i32x8_ bm0 = {(int)bq[0][3][0], (int)bq[0][3][1], (int)bq[0][3][2], (int)bq[0][3][3],
                                        (int)bq[0][4][0], (int)bq[0][4][1], (int)bq[0][4][2], (int)bq[0][4][3]};
                    const i32x8_ bm1 = {(int)bq[0][5][0], (int)bq[0][5][1], (int)bq[0][5][2], (int)bq[0][5][3],
                                        (int)bq[0][6][0], (int)bq[0][6][1], (int)bq[0][6][2], (int)bq[0][6][3]};
#pragma unroll
                    for (int j = 0; j < 7; ++j) {
                        const char* ap = abuf + abase[j] + g * PITCH;
                        const char* a8 = abuf + abase[j] + c8;
                        // taps 2m, 2m+1 of fp8 MFMA m = 2g (+1); the slice of the non-existent tap 9 re-reads tap 8 against
                        // zero weights (leaving it out of the read with an exec mask was 45 % slower: every masked region
                        // fences the scheduler)
                        constexpr int TP[10] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 8};
#define LRPXH_TOFF(t) ((TP[t] / 3) * PITCH + (TP[t] % 3) * PSTRIDE)
                        u32x4_ p0, p1;
                        if constexpr (X6) {
                            const char* a6 = abuf + abase[j] + (g == 2 ? e6c : e6a) + LRPXH_TOFF(4 * g);
                            p0 = *reinterpret_cast<const u32x4_*>(a6);
                            const u32x4_ y_ = *reinterpret_cast<const u32x4_*>(a6 + 16);
                            p1 = u32x4_{y_[0], y_[1], y_[3], y_[2]};
                        } else {
                            p0 = *reinterpret_cast<const u32x4_*>(a8 + LRPXH_TOFF(4 * g));
                            p1 = *reinterpret_cast<const u32x4_*>(a8 + LRPXH_TOFF(4 * g + 1));
                        }
                        const f16x8 h0 = *reinterpret_cast<const f16x8*>(ap);
                        const f16x8 h1 = *reinterpret_cast<const f16x8*>(ap + PSTRIDE);
                        const f16x8 h2 = *reinterpret_cast<const f16x8*>(ap + 2 * PSTRIDE);
                        const i32x8_ am0 = {(int)p0[0], (int)p0[1], (int)p0[2], (int)p0[3], (int)p1[0], (int)p1[1], (int)p1[2], (int)p1[3]};
                        // small terms first: the cross products on the fp8 cores, then hi * hi_W
                        if constexpr (TR) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bm0, am0, acc[j], 2, 2, 0, bm0[6], 0, am0[6]);
                        else if constexpr (X6) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(am0, bm0, acc[j], 2, 2, 0, am0[6], 0, bm0[6]);
                        else acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(am0, bm0, acc[j], 0, 0, 0, 0, 0, 0);
                        if (g < 2) {          // (compile-time after unrolling)
                            u32x4_ q0, q1;
                            if constexpr (X6) {       // taps 4g + 2 / 4g + 3: g = 0: a row less two pixels apart, g = 1: one pixel
                                const char* a6 = abuf + abase[j] + (g == 0 ? e6b : e6a) + LRPXH_TOFF(g < 2 ? 4 * g + 2 : 0);
                                q0 = *reinterpret_cast<const u32x4_*>(a6);
                                const u32x4_ y_ = *reinterpret_cast<const u32x4_*>(a6 + 16);
                                q1 = u32x4_{y_[0], y_[1], y_[3], y_[2]};
                            } else {
                                q0 = *reinterpret_cast<const u32x4_*>(a8 + LRPXH_TOFF(g < 2 ? 4 * g + 2 : 0));
                                q1 = *reinterpret_cast<const u32x4_*>(a8 + LRPXH_TOFF(g < 2 ? 4 * g + 3 : 0));
                            }
#undef LRPXH_TOFF
                            const i32x8_ am1 = {(int)q0[0], (int)q0[1], (int)q0[2], (int)q0[3], (int)q1[0], (int)q1[1], (int)q1[2], (int)q1[3]};
                            if constexpr (TR) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(bm1, am1, acc[j], 2, 2, 0, bm1[6], 0, am1[6]);
                            else if constexpr (X6) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(am1, bm1, acc[j], 2, 2, 0, am1[6], 0, bm1[6]);
                            else acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(am1, bm1, acc[j], 0, 0, 0, 0, 0, 0);
                        }
                        if constexpr (TR) {
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh0, h0, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh1, h1, acc[j], 0, 0, 0);
                            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh2, h2, acc[j], 0, 0, 0);
                        } else {
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h0, bh0, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h1, bh1, acc[j], 0, 0, 0);
                        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(h2, bh2, acc[j], 0, 0, 0);
                        }
                        if constexpr (LRPXH_F8_SGB != 0) {
                            const int gn = (j == 6) ? g + 1 : g;                     // tap row of the next tile
                            if (gn < 2) __builtin_amdgcn_sched_group_barrier(0x100, 7, 0);
                            else if (gn == 2) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
                            if (g < 2) __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
                            else __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                        }
                    }
#pragma unroll
                    for (int i = 0; i < NQ - 1; ++i)
#pragma unroll
                        for (int p = 0; p < BP; ++p) bq[i][p] = bq[i + 1][p];
                }
                }
            } else if constexpr (B6) {
            // six bf16 products per (tap, accumulator tile), smallest first; A fragments (three planes) one tile ahead of their MFMAs
            bf16x8 n0, n1, n2;
            if constexpr (APIPE) {
                n0 = *reinterpret_cast<const bf16x8*>(abuf + abase[0]);
                n1 = *reinterpret_cast<const bf16x8*>(abuf + abase[0] + 32);
                n2 = *reinterpret_cast<const bf16x8*>(abuf + abase[0] + 64);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            }
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const long nxt = (long)min(chunk * TAPS + tap + NBQ - 1, last_step) * 3;
#pragma unroll
                for (int p = 0; p < 3; ++p) bq[NBQ - 1][p] = wp[(nxt + p) * 64];
                if (tap == 0 && ISSUE_LATE) { __builtin_amdgcn_sched_barrier(0); LRPXH_ISSUE_NEXT }
                const bf16x8 b0 = __builtin_bit_cast(bf16x8, bq[0][0]);
                const bf16x8 b1 = __builtin_bit_cast(bf16x8, bq[0][1]);
                const bf16x8 b2 = __builtin_bit_cast(bf16x8, bq[0][2]);
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    bf16x8 a0, a1, a2;
                    if constexpr (APIPE) {
                        a0 = n0; a1 = n1; a2 = n2;
                        if (!(tap == TAPS - 1 && j == 6)) {
                            const int jn = (j + 1) % 7, tn = tap + (j == 6 ? 1 : 0);
                            const char* apn = abuf + abase[jn] + (tn / 3) * PITCH + (tn % 3) * PSTRIDE;
                            n2 = *reinterpret_cast<const bf16x8*>(apn + 64);
                            n1 = *reinterpret_cast<const bf16x8*>(apn + 32);
                            n0 = *reinterpret_cast<const bf16x8*>(apn);
                        }
                    } else {
                        const char* ap = abuf + abase[j] + (tap / 3) * PITCH + (tap % 3) * PSTRIDE;
                        a0 = *reinterpret_cast<const bf16x8*>(ap);
                        a1 = *reinterpret_cast<const bf16x8*>(ap + 32);
                        a2 = *reinterpret_cast<const bf16x8*>(ap + 64);
                    }
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[j], 0, 0, 0);
                    if constexpr (APIPE) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);   // the 3 reads of the NEXT tile ...
                        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);   // ... then the 6 MFMAs of this one
                    }
                }
#pragma unroll
                for (int i = 0; i < NBQ - 1; ++i)
#pragma unroll
                    for (int p = 0; p < 3; ++p) bq[i][p] = bq[i + 1][p];
            }
            } else {
#ifndef LRPXH_APIPE_D
#define LRPXH_APIPE_D 1       // A fragments read this many accumulator tiles ahead of their MFMAs (non-F8 path: forward trace, mode 2)
#endif
            constexpr int AD = LRPXH_APIPE_D;
            f16x8 nq[AD][2];          // ring: nq[t % AD] = fragments (hi, lo) of op t = tap * 7 + j
            if constexpr (APIPE) {
#pragma unroll
                for (int d = 0; d < AD; ++d) {
                    const int jn = d % 7, tn = d / 7;
                    const char* apn = abuf + abase[jn] + (tn / 3) * PITCH + (tn % 3) * PSTRIDE;
                    nq[d][0] = *reinterpret_cast<const f16x8*>(apn);
                    nq[d][1] = *reinterpret_cast<const f16x8*>(apn + 32);
                }
                __builtin_amdgcn_sched_group_barrier(0x100, 2 * AD, 0);
            }
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                const int tapoff = (tap / 3) * PITCH + (tap % 3) * PSTRIDE;
                const long nxt = (long)min(chunk * TAPS + tap + NBQ - 1, last_step) * 2;
#pragma unroll
                for (int p = 0; p < 2; ++p) bq[NBQ - 1][p] = wp[(nxt + p) * 64];
                if (tap == 0 && ISSUE_LATE) { __builtin_amdgcn_sched_barrier(0); LRPXH_ISSUE_NEXT }
                const f16x8 b0 = __builtin_bit_cast(f16x8, bq[0][0]);
                const f16x8 b1 = __builtin_bit_cast(f16x8, bq[0][1]);
#pragma unroll
                for (int j = 0; j < 7; ++j) {
                    f16x8 a0, a1;
                    if constexpr (APIPE) {
                        const int t = tap * 7 + j;
                        a0 = nq[t % AD][0]; a1 = nq[t % AD][1];
                        if (t + AD < TAPS * 7) {
                            const int jn = (t + AD) % 7, tn = (t + AD) / 7;
                            const char* apn = abuf + abase[jn] + (tn / 3) * PITCH + (tn % 3) * PSTRIDE;
                            nq[t % AD][1] = *reinterpret_cast<const f16x8*>(apn + 32);
                            nq[t % AD][0] = *reinterpret_cast<const f16x8*>(apn);
                        }
                    } else {
                        const char* ap = abuf + abase[j] + tapoff;
                        a0 = *reinterpret_cast<const f16x8*>(ap);
                        a1 = *reinterpret_cast<const f16x8*>(ap + 32);
                    }
                    // small terms first
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b0, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b1, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, b0, acc[j], 0, 0, 0);
                    if constexpr (APIPE) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // the 2 reads of a LATER tile ...
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);   // ... then the 3 MFMAs of this one
                    }
                }
#pragma unroll
                for (int i = 0; i < NBQ - 1; ++i)
#pragma unroll
                    for (int p = 0; p < 2; ++p) bq[i][p] = bq[i + 1][p];
#ifdef LRPX_STAMP
                if (tap == 2 || tap == 5 || tap == 8) {
                    LRPXH_T(tt);
                    if (tap == 2) s_tap0 += tt - tprev; else if (tap == 5) s_tap1 += tt - tprev; else s_tap2 += tt - tprev;
                    tprev = tt;
                }
#endif
            }
            }   // !F8
        }
        LRPXH_T(tc);
        if constexpr (DB) {
            if (more && grp == 0 && !(LRPXH_EXP & 1)) { if constexpr (LOSTAGE) { LRPXH_COMMIT_LO((chunk + 1) & 1) } else { LRPXH_COMMIT((chunk + 1) & 1) } }
            LRPXH_T(td);
            if constexpr (!(LRPXH_EXP & 64)) __syncthreads();      // (EXP 64: no barrier in the K loop)
            if constexpr (STAG) {     // group 1: chunk + 2 into the buffer everyone has just finished reading
                if (grp == 1 && chunk + 2 < nchunk && !(LRPXH_EXP & 1)) { if constexpr (LOSTAGE) { LRPXH_COMMIT_LO(chunk & 1) } else { LRPXH_COMMIT(chunk & 1) } }
            }
            LRPXH_T(te);
#ifdef LRPX_STAMP
            s_issue += tb - ta; s_mfma += tc - tb; s_commit += td - tc; s_barrier += te - td;
#endif
        } else {
            __syncthreads();                       // every wave is done reading the single buffer
            LRPXH_T(td0);
            if (more && !(LRPXH_EXP & 1)) { if constexpr (LOSTAGE) { LRPXH_COMMIT_LO(0) } else { LRPXH_COMMIT(0) } }
            LRPXH_T(td);
            __syncthreads();
            LRPXH_T(te);
#ifdef LRPX_STAMP
            s_issue += tb - ta; s_mfma += tc - tb; s_commit += td - td0; s_barrier += (te - td) + (td0 - tc);
#endif
        }
    }
#undef LRPXH_ISSUE_NEXT
#undef LRPXH_REFRESH
#undef LRPXH_ISSUE
#undef LRPXH_COMMIT
#undef LRPXH_ISSUE_LO
#undef LRPXH_COMMIT_LO
    if (!wave_active) return;
    if constexpr ((LRPXH_EXP & 32) != 0) {        // (EXP 32: no epilogue)
        float sacc = 0.f;
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) sacc += acc[j][e];
        if (sacc == 1.2345e-30f) a.out0[0] = sacc;
        return;
    }
    LRPXH_T(t_epi);
    if constexpr (BLK && !TR) {
        epi_rel_mul_al<HW, EPI, F8, true>(a, acc, wm, ocb, lane, g0, total_pix, a.out1 ? a.out1_amax : nullptr, inv_w, in_amax);
#ifdef LRPX_STAMP
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LRPXH_T(t_endp);
            if (lane == 0 && HW == LRPX_STAMP_HW) {
                atomicAdd(&g_stamp_h3[0], t_loop - t_start);
                atomicAdd(&g_stamp_h3[1], s_issue);
                atomicAdd(&g_stamp_h3[2], s_mfma);
                atomicAdd(&g_stamp_h3[3], s_commit);
                atomicAdd(&g_stamp_h3[4], s_barrier);
                atomicAdd(&g_stamp_h3[5], t_endp - t_epi);
                atomicAdd(&g_stamp_h3[6], t_endp - t_start);
                atomicAdd(&g_stamp_h3[7], 1ull);
            }
        }
#endif
        return;
    }
    if constexpr (TR) {
        // the map-straddling branch of epi_rel_mul_blk numbers a wave's maps from the WORKGROUP's first row and keeps three maxima: with
        // more than one pixel sub-tile per workgroup (MT > 1) the later waves would drop theirs; every such instantiation is AL
        static_assert(AL || MT == 1, "epi_rel_mul_blk: non-aligned (map-straddling) tiles need MT == 1");
        epi_rel_mul_blk<HW, AL, F8>(a, acc, wm, ocb, lane, g0, total_pix, a.out1 ? a.out1_amax : nullptr, tile_tab, inv_w, in_amax);
#ifdef LRPXH_END_SLEEP
        for (int i = 0; i < LRPXH_END_SLEEP; ++i) __builtin_amdgcn_s_sleep(127);
#endif
#ifdef LRPX_STAMP
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LRPXH_T(t_endt);
            if (lane == 0 && HW == LRPX_STAMP_HW) {
                atomicAdd(&g_stamp_h3[0], t_loop - t_start);
                atomicAdd(&g_stamp_h3[1], s_issue);
                atomicAdd(&g_stamp_h3[2], s_mfma);
                atomicAdd(&g_stamp_h3[3], s_commit);
                atomicAdd(&g_stamp_h3[4], s_barrier);
                atomicAdd(&g_stamp_h3[5], t_endt - t_epi);
                atomicAdd(&g_stamp_h3[6], t_endt - t_start);
                atomicAdd(&g_stamp_h3[7], 1ull);
            }
        }
#endif
        return;
    }

#ifndef LRPXH_AL_EPI
#define LRPXH_AL_EPI 1
#endif
    // (measured, same box: conv1_2 2.77 -> 2.60 ms, conv2_2 2.07 -> 2.00, conv3_1 0.96 -> 0.93, the 8-wave 56x56 layers +-0; the
    // 4-row 112x112 kernel of conv2_1 spills 9 registers with it, 1.09 -> 1.14: it keeps the generic epilogue)
#ifndef LRPXH_AL_EPI_H3
#define LRPXH_AL_EPI_H3 1      // mode 2 (three fp16 products) too: round 5
#endif
    if constexpr ((X6 || (!F8 && (LRPXH_AL_EPI_H3 != 0))) && AL && (LRPXH_AL_EPI != 0) && !(HW == 112 && MT == 2 && !B6) && (EPI == EPI_REL_MUL || EPI == EPI_GUIDED)) {
        unsigned* __restrict__ oamax_a = (EPI == EPI_GUIDED) ? a.out0_amax : (a.out1 ? a.out1_amax : nullptr);
        epi_rel_mul_al<HW, EPI, F8, false, B6>(a, acc, wm, ocb, lane, g0, total_pix, oamax_a, inv_w, in_amax);
#ifdef LRPX_STAMP
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LRPXH_T(t_enda);
            if (lane == 0 && HW == LRPX_STAMP_HW) {
                atomicAdd(&g_stamp_h3[0], t_loop - t_start);
                atomicAdd(&g_stamp_h3[1], s_issue);
                atomicAdd(&g_stamp_h3[2], s_mfma);
                atomicAdd(&g_stamp_h3[3], s_commit);
                atomicAdd(&g_stamp_h3[4], s_barrier);
                atomicAdd(&g_stamp_h3[5], t_enda - t_epi);
                atomicAdd(&g_stamp_h3[6], t_enda - t_start);
                atomicAdd(&g_stamp_h3[7], 1ull);
            }
        }
#endif
        return;
    }
#ifndef LRPXH_AL_FWD
#define LRPXH_AL_FWD 1
#endif
    if constexpr (AL && (LRPXH_AL_FWD != 0) && EPI == EPI_FWD_DUAL) {
        if ((a.oc_split & 31) == 0) {                      // (uniform; every VGG16 layer)
            epi_fwd_dual_al<HW, F8, B6>(a, acc, wm, ocb, lane, g0, inv_w, in_amax);
            return;
        }
    }
    EpiCtx cx;
    cx.oc = ocb * 32 + li;
    cx.lane = lane;
    cx.q0 = wm * 224 + 4 * lh;
    cx.g0 = (int)g0;
    cx.pix0 = g0 * W;
    cx.total_pix = total_pix;
    cx.xi_base = 0;
    if constexpr (AL) {
        const unsigned rr = (unsigned)cx.q0 / (unsigned)HW, cc = (unsigned)cx.q0 - rr * HW;
        const unsigned g = (unsigned)cx.g0 + rr;
        const unsigned n = g / (unsigned)HW;
        const long img = a.map2img ? a.map2img[n] : n;
        cx.xi_base = (img * a.pix_per_map + (long)((g - n * HW) * HW + cc)) * a.oc_split + cx.oc;
    }

    // ---- undo the operand scales: acc * 2^-kA(map of the pixel) * 2^-kW ----
    const int nmax = a.n_maps - 1;
    if constexpr (B6) {
        // exact splits: nothing to undo
    } else if constexpr (AL) {
        const unsigned n = (unsigned)g0 / (unsigned)H;
        const float inv_a = exp2i(-split_scale_exp<F8>(in_amax[min((int)n, nmax)]));
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][e] = acc[j][e] * inv_a * inv_w;
    } else {
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            // a 32-pixel accumulator tile is shorter than a map: at most one boundary inside it (cf. epi_gather).  The two
            // scales of the tile come from the workgroup's table (the prologue looked them up: 14 dependent global loads per
            // wave sat here, behind the K loop - conv4_3 -8 %, conv4_2 -2 % in a build without them)
            const unsigned q0t = (unsigned)(wm * 224 + 32 * j);            // the tile's first pixel (lanes 32-63: 4 pixels on)
            const unsigned rr = q0t / (unsigned)HW, c0 = q0t - rr * HW;
            const unsigned g = (unsigned)cx.g0 + rr;
            const unsigned n0 = g / (unsigned)HW;
            const int p0 = (int)((g - n0 * HW) * HW + c0) + 4 * lh;
            const float i0 = __builtin_bit_cast(float, tile_tab[(wm * 7 + j) * 4 + 0]);
            const float i1 = __builtin_bit_cast(float, tile_tab[(wm * 7 + j) * 4 + 1]);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int dq = (e & 3) + 8 * (e >> 2);
                acc[j][e] = acc[j][e] * (p0 + dq < a.pix_per_map ? i0 : i1) * inv_w;
            }
        }
    }

    // ---- epilogue (software-pipelined per tile); max|out1| per map goes to a.out1_amax for the next f16x3 consumer ----
    unsigned* __restrict__ oamax = (EPI == EPI_FWD_DUAL || EPI == EPI_GUIDED) ? a.out0_amax
                                   : (((EPI == EPI_REL || EPI == EPI_REL_MUL) && a.out1) ? a.out1_amax : nullptr);
    EpiMax mx0 = {0.f, 0.f}, mx1 = mx0, mx2 = mx0, mx3 = mx0, mx4 = mx0, mx5 = mx0, mx6 = mx0;   // (scalars: an array
#ifndef LRPXH_WIDE_EPI
#define LRPXH_WIDE_EPI 1
#endif
    // measured per layer (chain of 320 maps, tools/ab_chain.sh): the map-straddling 28x28 / 14x14 kernels gain 4-6 % (and
    // lose their 11-13 spilled VGPRs: the two-map dword epilogue held two candidate addresses per element); the aligned
    // 56 / 112 / 224 kernels are 1-5 % SLOWER with it (their dword epilogue has compile-time offsets from one base), so
    // LRPXH_WIDE_EPI = 1 selects it for the straddling kernels only, 2 everywhere
    if constexpr ((EPI == EPI_REL_MUL || EPI == EPI_GUIDED) && ((LRPXH_WIDE_EPI == 1 && !AL) || LRPXH_WIDE_EPI == 2)) {
        // (the K loop ends with a barrier: nobody reads the staging buffers any more; 32 x 36 floats per wave)
        float* scr = reinterpret_cast<float*>(ldsb) + wave * (32 * 36);
        epi_rel_mul_wide<HW, AL, EPI>(a, acc, scr, wm, ocb, lane, g0, total_pix, oamax, tile_tab);
#ifdef LRPXH_END_SLEEP
        for (int i = 0; i < LRPXH_END_SLEEP; ++i) __builtin_amdgcn_s_sleep(127);      // (experiment: is the store drain at wave end exposed?)
#endif
#ifdef LRPX_STAMP
        {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            LRPXH_T(t_endw);
            if (lane == 0 && HW == LRPX_STAMP_HW) {
                atomicAdd(&g_stamp_h3[0], t_loop - t_start);
                atomicAdd(&g_stamp_h3[1], s_issue);
                atomicAdd(&g_stamp_h3[2], s_mfma);
                atomicAdd(&g_stamp_h3[3], s_commit);
                atomicAdd(&g_stamp_h3[4], s_barrier);
                atomicAdd(&g_stamp_h3[5], t_endw - t_epi);
                atomicAdd(&g_stamp_h3[6], t_endw - t_start);
                atomicAdd(&g_stamp_h3[7], 1ull);
            }
        }
#endif
        return;
    }
    if constexpr (EPI == EPI_REL_MUL || EPI == EPI_GUIDED) {                             //  would live in scratch)
        // one multiplicand per element: all 112 loads are issued before the first store (stores share the in-order
        // vmcnt with loads - a load behind a store waits for the store's write acknowledgement)
        EpiRegs r0, r1, r2, r3, r4, r5, r6;
        epi_gather<EPI, HW, TAPS, AL>(a, cx, 0, r0);
        epi_gather<EPI, HW, TAPS, AL>(a, cx, 1, r1);
        epi_gather<EPI, HW, TAPS, AL>(a, cx, 2, r2);
        epi_gather<EPI, HW, TAPS, AL>(a, cx, 3, r3);
        epi_gather<EPI, HW, TAPS, AL>(a, cx, 4, r4);
        epi_gather<EPI, HW, TAPS, AL>(a, cx, 5, r5);
        epi_gather<EPI, HW, TAPS, AL>(a, cx, 6, r6);
        epi_finish<EPI, HW, TAPS, AL>(a, cx, 0, acc[0], r0, &mx0);
        epi_finish<EPI, HW, TAPS, AL>(a, cx, 1, acc[1], r1, &mx1);
        epi_finish<EPI, HW, TAPS, AL>(a, cx, 2, acc[2], r2, &mx2);
        epi_finish<EPI, HW, TAPS, AL>(a, cx, 3, acc[3], r3, &mx3);
        epi_finish<EPI, HW, TAPS, AL>(a, cx, 4, acc[4], r4, &mx4);
        epi_finish<EPI, HW, TAPS, AL>(a, cx, 5, acc[5], r5, &mx5);
        epi_finish<EPI, HW, TAPS, AL>(a, cx, 6, acc[6], r6, &mx6);
    } else {
    EpiRegs ra, rb;
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 0, ra);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 1, rb);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 0, acc[0], ra, &mx0);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 2, ra);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 1, acc[1], rb, &mx1);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 3, rb);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 2, acc[2], ra, &mx2);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 4, ra);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 3, acc[3], rb, &mx3);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 5, rb);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 4, acc[4], ra, &mx4);
    epi_gather<EPI, HW, TAPS, AL>(a, cx, 6, ra);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 5, acc[5], rb, &mx5);
    epi_finish<EPI, HW, TAPS, AL>(a, cx, 6, acc[6], ra, &mx6);
    }
    if (oamax) {
        if constexpr (AL) {
            float m = fmaxf(fmaxf(fmaxf(mx0.m0, mx1.m0), fmaxf(mx2.m0, mx3.m0)), fmaxf(fmaxf(mx4.m0, mx5.m0), mx6.m0));
            m = wave_max(m);
            const unsigned n = (unsigned)g0 / (unsigned)H;
            if (lane == 0 && (int)n <= nmax) amax_update(&oamax[n], m);
        } else {
#define LRPXH_AMAX(J, MX)                                                                                       \
            {                                                                                                   \
                const unsigned q0t = (unsigned)(wm * 224 + 32 * (J));                                           \
                const unsigned n0 = ((unsigned)cx.g0 + q0t / (unsigned)HW) / (unsigned)HW;                       \
                const float m0 = wave_max(MX.m0), m1 = wave_max(MX.m1);                                         \
                if (lane == 0 && (int)n0 <= nmax) amax_update(&oamax[n0], m0);                                       \
                if (lane == 0 && (int)n0 + 1 <= nmax) amax_update(&oamax[n0 + 1], m1);                            \
            }
            LRPXH_AMAX(0, mx0) LRPXH_AMAX(1, mx1) LRPXH_AMAX(2, mx2) LRPXH_AMAX(3, mx3)
            LRPXH_AMAX(4, mx4) LRPXH_AMAX(5, mx5) LRPXH_AMAX(6, mx6)
#undef LRPXH_AMAX
        }
    }
#ifdef LRPXH_END_SLEEP
    for (int i = 0; i < LRPXH_END_SLEEP; ++i) __builtin_amdgcn_s_sleep(127);
#endif
#ifdef LRPX_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LRPXH_T(t_end);
    if (lane == 0 && HW == LRPX_STAMP_HW) {
        atomicAdd(&g_stamp_h3[0], t_loop - t_start);
        atomicAdd(&g_stamp_h3[1], s_issue);
        atomicAdd(&g_stamp_h3[2], s_mfma);
        atomicAdd(&g_stamp_h3[3], s_commit);
        atomicAdd(&g_stamp_h3[4], s_barrier);
        atomicAdd(&g_stamp_h3[5], t_end - t_epi);
        atomicAdd(&g_stamp_h3[6], t_end - t_start);
        atomicAdd(&g_stamp_h3[7], 1ull);
        atomicAdd(&g_stamp_h3[8], s_tap0); atomicAdd(&g_stamp_h3[9], s_tap1); atomicAdd(&g_stamp_h3[10], s_tap2);
    }
#endif
}

template <int HW, int MT, int NWN, bool DB, int EPI, bool POOL = false, bool F8 = false, bool B6 = false>
int launch_conv_f16x3(const ConvArgs& a, hipStream_t stream) {
    using C = ConvCfg<HW, 16, MT, NWN, 9>;
    constexpr int LDS = (DB ? 2 : 1) * C::NSLOT * (HW * (B6 ? 112 : 80) + 256) + 256 + 16 + 7 * MT * 16;     // buffers + scratch + tile table
    static_assert(LDS <= 160 * 1024, "conv_f16x3: the tile does not fit the CU's LDS");
    const long m_tiles = ceil_div((long)a.n_maps * HW, C::R);
    const int n_blocks = (int)ceil_div(a.n_oc, 32 * NWN);
    long grid = ceil_div(m_tiles, 8) * 8 * n_blocks;
    using CC = ConvCfg<HW, 16, MT, NWN, 9>;
    if (HW % CC::R == 0 && a.tile_group > 1) {       // grouped order: every XCD walks whole (image, row block) groups
        const long n_groups = (long)(a.n_maps / a.tile_group) * (HW / CC::R);
        grid = ceil_div(n_groups, 8) * 8 * a.tile_group * n_blocks;
    }
    auto kern = conv_f16x3_kernel<HW, MT, NWN, DB, EPI, POOL, F8, B6>;
    static LdsOnce attr_once;
    LRPX_TRY(reserve_lds_once(attr_once, kern, LDS, "conv_f16x3"));
    const unsigned ks = (EPI == EPI_PLAIN && a.ksplit > 1) ? (unsigned)a.ksplit : 1u;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, ks), dim3(64 * MT * NWN), LDS, stream, a, (int)m_tiles, n_blocks);
    return check_launch("conv_f16x3");
}

}  // namespace lrpx
