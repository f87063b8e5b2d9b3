// liblrpx core: error state, weight packing, layout / pooling / elementwise kernels of the LRP path.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>

#include "common.h"
#include "conv_mfma.h"
#include "conv_bf16x6.h"
#include "blocked.h"
#include "conv_f16x3.h"

namespace lrpx {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

bool ptr_checks_all() {
    static const bool on = [] { const char* v = getenv("LRPX_CHECK_PTRS"); return v && *v && atoi(v) != 0; }();
    return on;
}

int check_dev_ptrs(const char* fn, std::initializer_list<PtrArg> args) {
    int dev = -1;
    for (const PtrArg& a : args) {
        if (!a.p) continue;                      // (null is the entry point's own business: optional arguments exist)
        hipPointerAttribute_t at;
        const hipError_t e = hipPointerGetAttributes(&at, a.p);
        if (e != hipSuccess) {
            (void)hipGetLastError();             // (the failed query must not surface as the next launch's error)
            set_error("%s: argument `%s` = %p is not memory the GPU can address (a pageable host pointer?): %s", fn, a.name, a.p, hipGetErrorString(e));
            return LRPX_EINVAL;
        }
        if (at.type == hipMemoryTypeHost) continue;          // pinned host memory: device-accessible
        if (at.type != hipMemoryTypeDevice && at.type != hipMemoryTypeManaged && at.type != hipMemoryTypeUnified) {
            set_error("%s: argument `%s` = %p is not device memory (hipPointerGetAttributes: type %d - a pageable host pointer?)", fn, a.name, a.p, (int)at.type);
            return LRPX_EINVAL;
        }
        if (at.type == hipMemoryTypeDevice) {
            if (dev < 0 && hipGetDevice(&dev) != hipSuccess) dev = -1;
            if (dev >= 0 && at.device != dev) {
                set_error("%s: argument `%s` = %p lives on device %d, the calling thread's current device is %d", fn, a.name, a.p, at.device, dev);
                return LRPX_EINVAL;
            }
        }
    }
    return LRPX_OK;
}

const Switches& switches() {
    static const Switches sw = [] {
        auto num = [](const char* name, int dflt) { const char* v = getenv(name); return v ? atoi(v) : dflt; };
        auto set = [](const char* name) { return getenv(name) ? 1 : 0; };
        Switches w;
        w.wide = num("LRPX_WIDE", 7);
        w.fwd_ksplit14 = num("LRPX_FWD_KSPLIT", 8);
        w.fwd_ksplit28 = num("LRPX_FWD_KSPLIT28", 1);
        w.fwd_wide = num("LRPX_FWD_WIDE", 0);
        w.conv11_f16 = num("LRPX_CONV11_F16", 1);
        w.first_valu = set("LRPX_FIRST_VALU");
        w.s21_nhwc = set("LRPX_S21_NHWC");
        w.guided_poolbwd = set("LRPX_GUIDED_POOLBWD");
        w.dense_wide = num("LRPX_DENSE_WIDE", 0);
        w.dense_ks_rel = num("LRPX_DENSE_KS_REL", 1);
        w.dense_n256 = num("LRPX_DENSE_N256", 1);
        w.dense_rt = num("LRPX_DENSE_RT", 0);
        w.dense_1wave = set("LRPX_DENSE_1WAVE");
        w.linear_valu = set("LRPX_LINEAR_VALU");
        w.x6_legacy = set("LRPX_X6_LEGACY");
        w.b6_wide = num("LRPX_B6_WIDE", 0);
        w.b6_rel_ksplit14 = num("LRPX_B6_REL_KSPLIT14", 2);
        w.b6_fwd_ksplit28 = num("LRPX_B6_FWD_KSPLIT28", 4);
        w.b6_fwd_ksplit56 = num("LRPX_B6_FWD_KSPLIT56", 2);
        return w;
    }();
    return sw;
}

// ------------------------------------------------------------------------------------------------
// weight packing: fragment-major [ocb][chunk][tap][ks][lane][4]; lane l of a k-step holds
// B[k = 8*ks + 4*(l>>5) + e][oc = 32*ocb + (l&31)], e = 0..3  (operand map of v_mfma_f32_32x32x2_f32
// with the k index permuted identically on the A side, see conv_mfma.h)
// ------------------------------------------------------------------------------------------------
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin,
                                    int taps, int mode, int kc, int n_oc_pad, int k_pad, long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int e = idx & 3;
    const int lane = (idx >> 2) & 63;
    long rest = idx >> 8;
    const int ksteps = kc / 8;
    const int ks = rest % ksteps; rest /= ksteps;
    const int tap = rest % taps; rest /= taps;
    const int nchunk = k_pad / kc;
    const int chunk = rest % nchunk; rest /= nchunk;
    const int ocb = (int)rest;
    const int oc = ocb * 32 + (lane & 31);
    const int k = chunk * kc + ks * 8 + 4 * (lane >> 5) + e;
    float v = 0.f;
    // conv weights are (cout, cin, 3, 3): w[((co*cin + ci)*9) + tap]
    switch (mode) {
        case LRPX_PACK_FWD_DUAL:   // GEMM k = ci, oc in [0,2cout)
            if (k < cin && oc < 2 * cout) {
                int co = oc < cout ? oc : oc - cout;
                float x = w[((long)co * cin + k) * taps + tap];
                v = oc < cout ? x : fmaxf(x, 0.f);
            }
            break;
        case LRPX_PACK_FWD:
            if (k < cin && oc < cout) v = w[((long)oc * cin + k) * taps + tap];
            break;
        case LRPX_PACK_FWD_DUAL_FIRST:   // input channels [0,cin) = x+, [cin,2cin) = x-
            if (k < 2 * cin && oc < 2 * cout) {
                int co = oc < cout ? oc : oc - cout;
                int ci = k < cin ? k : k - cin;
                float x = w[((long)co * cin + ci) * taps + tap];
                v = oc < cout ? x : (k < cin ? fmaxf(x, 0.f) : fminf(x, 0.f));
            }
            break;
        case LRPX_PACK_BWD_POS:    // GEMM k = co, oc = ci, kernel flipped
            if (k < cout && oc < cin) v = fmaxf(w[((long)k * cin + oc) * taps + (taps - 1 - tap)], 0.f);
            break;
        case LRPX_PACK_BWD_PLAIN:
            if (k < cout && oc < cin) v = w[((long)k * cin + oc) * taps + (taps - 1 - tap)];
            break;
        case LRPX_PACK_BWD_FIRST:  // oc in [0,cin): W+, [cin,2cin): W-
            if (k < cout && oc < 2 * cin) {
                int ci = oc < cin ? oc : oc - cin;
                float x = w[((long)k * cin + ci) * taps + (taps - 1 - tap)];
                v = oc < cin ? fmaxf(x, 0.f) : fminf(x, 0.f);
            }
            break;
        case LRPX_PACK_DENSE_T:    // W is (k = cout rows, n = cin cols): B[k][oc] = W[k][oc]
            if (k < cout && oc < cin) v = w[(long)k * cin + oc];
            break;
        case LRPX_PACK_DENSE:      // W is (n = cout rows, k = cin cols): B[k][oc] = W[oc][k]
            if (k < cin && oc < cout) v = w[(long)oc * cin + k];
            break;
    }
    out[idx] = v;
}

// bf16x3 split of the packed weights for conv_bf16x6.h: [ocb][chunk 16][tap][plane 3][lane 64][8 bf16]; lane l holds
// B[k = 16*chunk + 8*(l>>5) + j][oc = 32*ocb + (l&31)], j = 0..7  (operand map of v_mfma_f32_32x32x16_bf16)
__global__ void pack_weights_bf16x3_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int cout,
                                           int cin, int taps, int mode, int k_pad, long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // one thread per (.., lane, j): writes the 3 planes
    if (idx >= total) return;
    const int j = idx & 7;
    const int lane = (idx >> 3) & 63;
    long rest = idx >> 9;
    const int tap = rest % taps; rest /= taps;
    const int nchunk = k_pad / 16;
    const int chunk = rest % nchunk; rest /= nchunk;
    const int ocb = (int)rest;
    const int oc = ocb * 32 + (lane & 31);
    const int k = chunk * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    switch (mode) {
        case LRPX_PACK_BWD_POS:
            if (k < cout && oc < cin) v = fmaxf(w[((long)k * cin + oc) * taps + (taps - 1 - tap)], 0.f);
            break;
        case LRPX_PACK_BWD_PLAIN:
            if (k < cout && oc < cin) v = w[((long)k * cin + oc) * taps + (taps - 1 - tap)];
            break;
        case LRPX_PACK_FWD:
            if (k < cin && oc < cout) v = w[((long)oc * cin + k) * taps + tap];
            break;
        case LRPX_PACK_FWD_DUAL:
            if (k < cin && oc < 2 * cout) {
                const int co = oc < cout ? oc : oc - cout;
                const float x = w[((long)co * cin + k) * taps + tap];
                v = oc < cout ? x : fmaxf(x, 0.f);
            }
            break;
    }
    unsigned short p0, p1, p2;
    split3(v, p0, p1, p2);
    const long base = ((((long)ocb * nchunk + chunk) * taps + tap) * 3) * 512 + lane * 8 + j;
    out[base] = p0; out[base + 512] = p1; out[base + 1024] = p2;
}

// max |x| over a flat tensor -> *out (float bits; non-negative floats order like unsigned integers)
__global__ void amax_flat_kernel(const float* __restrict__ x, long n, unsigned* __restrict__ out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        m = fmaxf(m, fabsf(x[i]));
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) amax_update(out, m);
}

// End of a K-split forward conv (conv_f16x3.h, PLAIN epilogue with ksplit): part[s][pixel][2*cout] partial sums of
// [conv(x, W) | conv(x, W+)] -> act = ReLU(sum + bias) [pixel][cout], zpos [pixel][cout], max of act per image.  The splits
// are added in order (deterministic).  Whole blocks of 256 float4 items inside one image (host-checked).
__global__ void fwd_dual_finish_kernel(const float* __restrict__ part, int nsplit, long split_stride,
                                       const float* __restrict__ bias, float* __restrict__ act, float* __restrict__ zpos,
                                       int cout, long per_img4, unsigned* __restrict__ amax);

int amax_flat(const float* x, long n, unsigned* out, hipStream_t s) {
    hipLaunchKernelGGL(amax_flat_kernel, dim3(256), dim3(256), 0, s, x, n, out);
    return check_launch("amax_flat");
}

// lane-local |max| of 4 values -> per-map amax.  Waves whose 64 lanes sit in one map (every layer of the path) reduce
// in registers and issue one atomic; mixed waves fall back to one atomic per lane.
__device__ __forceinline__ void amax_commit(unsigned* __restrict__ amax, long n, float m) {
    // blocks of 256 threads; every launch that records an amax has whole blocks (host-checked)
    __shared__ float red[4];
    __shared__ long redn[4];
    const int wv = threadIdx.x >> 6;
    const long n0 = __shfl(n, 0, 64);
    const bool wave_uniform = __all(n == n0);
    float mw = wave_max(wave_uniform ? m : 0.f);
    if (!wave_uniform) amax_update(&amax[n], m);          // (never on the path: maps are multiples of 1024 floats)
    if ((threadIdx.x & 63) == 0) { red[wv] = mw; redn[wv] = n0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (redn[0] == redn[1] && redn[0] == redn[2] && redn[0] == redn[3]) {
            amax_update(&amax[redn[0]], fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));
        } else {
            for (int i = 0; i < 4; ++i) amax_update(&amax[redn[i]], red[i]);
        }
    }
}

__global__ void fwd_dual_finish_kernel(const float* __restrict__ part, int nsplit, long split_stride,
                                       const float* __restrict__ bias, float* __restrict__ act, float* __restrict__ zpos,
                                       int cout, long per_img4, unsigned* __restrict__ amax) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // float4 items over [pixel][2*cout]
    const int c4n = cout / 2;                                          // float4 per pixel
    const long pix = idx / c4n;
    const int c = (int)(idx - pix * c4n) * 4;                          // column in [0, 2*cout)
    // Pairwise (binary-tree) order over the splits, the same for every element and every launch: deterministic, and the
    // last additions meet partial sums of equal length (a running sum over 8 splits would round the nearly complete sum
    // seven times).  One split after the other: all loads of a thread in flight at once ran 73 instead of 43 us - the
    // partial tensors lie 12.8 MB apart.
    f32x4 lvl[5];
    f32x4 v = reinterpret_cast<const f32x4*>(part)[idx];
    for (int s = 0; s < nsplit; ++s) {
        if (s) v = reinterpret_cast<const f32x4*>(part + s * split_stride)[idx];
        int k = s, l = 0;
        while (k & 1) { v = lvl[l] + v; k >>= 1; ++l; }
        lvl[l] = v;
    }
    // (nsplit is a power of two <= 16, host-checked: the total sits in `v`)
    float m = 0.f;
    if (c < cout) {
        const f32x4 b = *reinterpret_cast<const f32x4*>(bias + c);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = fmaxf(v[e] + b[e], 0.f); m = fmaxf(m, v[e]); }
        *reinterpret_cast<f32x4*>(act + pix * cout + c) = v;
    } else {
        *reinterpret_cast<f32x4*>(zpos + pix * cout + (c - cout)) = v;
    }
    if (amax) amax_commit(amax, idx / per_img4, m);
}

// ITER float4 per thread at block stride (ITER > 1 only when 256*ITER divides a map: one amax update per block of
// 256*ITER float4 instead of one per 256 - the updates all land on n_maps words)
template <int ITER>
__global__ void amax_maps_kernel(const float* __restrict__ s, long per4, long total, unsigned* __restrict__ amax) {
    const long base = (long)blockIdx.x * (blockDim.x * ITER) + threadIdx.x;   // float4 units
    float m = 0.f;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const long idx = base + (long)it * blockDim.x;
        if (idx < total) {
            const f32x4 v = reinterpret_cast<const f32x4*>(s)[idx];
            m = fmaxf(m, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
        }
    }
    amax_commit(amax, (base < total ? base : total - 1) / per4, m);
}

// f16x2 split of the packed weights for conv_f16x3.h: 64-byte header {2^-kW, amax bits, ...} then
// [ocb][chunk 16][tap][plane 2][lane 64][8 fp16]; lane map as pack_weights_bf16x3_kernel
__global__ void pack_weights_f16x2_kernel(const float* __restrict__ w, float* __restrict__ header, int cout, int cin,
                                          int taps, int mode, int k_pad, long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int kw = f16_scale_exp(reinterpret_cast<const unsigned*>(header)[1]);
    if (idx == 0) header[0] = exp2i(-kw);
    unsigned short* out = reinterpret_cast<unsigned short*>(header + F16X3_HEADER_FLOATS);
    const int j = idx & 7;
    const int lane = (idx >> 3) & 63;
    long rest = idx >> 9;
    const int tap = rest % taps; rest /= taps;
    const int nchunk = k_pad / 16;
    const int chunk = rest % nchunk; rest /= nchunk;
    const int ocb = (int)rest;
    const int oc = ocb * 32 + (lane & 31);
    const int k = chunk * 16 + 8 * (lane >> 5) + j;
    float v = 0.f;
    switch (mode) {
        case LRPX_PACK_BWD_POS:
            if (k < cout && oc < cin) v = fmaxf(w[((long)k * cin + oc) * taps + (taps - 1 - tap)], 0.f);
            break;
        case LRPX_PACK_BWD_PLAIN:
            if (k < cout && oc < cin) v = w[((long)k * cin + oc) * taps + (taps - 1 - tap)];
            break;
        case LRPX_PACK_FWD:
            if (k < cin && oc < cout) v = w[((long)oc * cin + k) * taps + tap];
            break;
        case LRPX_PACK_FWD_DUAL:
            if (k < cin && oc < 2 * cout) {
                const int co = oc < cout ? oc : oc - cout;
                const float x = w[((long)co * cin + k) * taps + tap];
                v = oc < cout ? x : fmaxf(x, 0.f);
            }
            break;
        case LRPX_PACK_FWD_DUAL_FIRST:   // input channels [0,cin) = x+, [cin,2cin) = x- (the signed image): Z = x+ W+ + x- W-
            if (k < 2 * cin && oc < 2 * cout) {
                const int co = oc < cout ? oc : oc - cout;
                const int ci = k < cin ? k : k - cin;
                const float x = w[((long)co * cin + ci) * taps + tap];
                v = oc < cout ? x : (k < cin ? fmaxf(x, 0.f) : fminf(x, 0.f));
            }
            break;
    }
    _Float16 hi, lo;
    split2(v * exp2i(kw), hi, lo);
    const long base = ((((long)ocb * nchunk + chunk) * taps + tap) * 2) * 512 + lane * 8 + j;
    out[base] = __builtin_bit_cast(unsigned short, hi);
    out[base + 512] = __builtin_bit_cast(unsigned short, lo);
}

// weights for the F8 variant of conv_f16x3.h (fp16 hi.hi product + two fp8 cross products): 64-byte header {2^-kW, amax
// bits}, then [ocb][chunk 16][tap row g 3][plane 7][lane 64][16 B]: planes 0-2 = fp16 hi of taps (g, dx = 0,1,2) (lane
// map as the f16x2 pack), planes 3/4 and 5/6 = B operands of the fp8 MFMAs 2g and 2g+1 of the chunk (conv_f16x3.h,
// f8_slot_*: per lane two (tap, 16 channels) slots of fp8 e4m3: W*2^-4 for an "L" slot, (W - hi)*2^4 for an "S" slot,
// zero for a pad slot).  W is scaled into [2^11, 2^12) first.
__global__ void pack_weights_f16f8_kernel(const float* __restrict__ w, float* __restrict__ header, int cout, int cin,
                                          int mode, int k_pad, long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int kw = split_scale_exp<true>(reinterpret_cast<const unsigned*>(header)[1]);
    if (idx == 0) header[0] = exp2i(-kw);
    const float sc = exp2i(kw);
    const int lane = idx & 63;
    long rest = idx >> 6;
    const int g = rest % 3; rest /= 3;
    const int nchunk = k_pad / 16;
    const int chunk = rest % nchunk; rest /= nchunk;
    const int ocb = (int)rest;
    // BWD_POS (the relevance pass, REL_MUL epilogue): the kernel multiplies with the weights as the A operand (transposed result:
    // a lane owns one pixel and the 16 channels of rows (e & 3) + 8 (e >> 2) + 4 lh, e = 0..15), so fragment row rho carries
    // output channel 16 ((rho >> 2) & 1) + 4 (rho >> 3) + (rho & 3) of the block: a lane's 16 results are then the CONTIGUOUS
    // channels 16 lh .. 16 lh + 15 - one slice of the blocked layout (blocked.h)
    const int rho = lane & 31;
    const int row_ch = mode == LRPX_PACK_BWD_POS ? 16 * ((rho >> 2) & 1) + 4 * (rho >> 3) + (rho & 3) : rho;
    const int oc = ocb * 32 + row_ch, lh = lane >> 5;
    auto wv = [&](int k, int dx) -> float {      // scaled weight of (input channel k of this pass, tap (g, dx))
        const int tap = g * 3 + dx;
        float v = 0.f;
        if (mode == LRPX_PACK_BWD_POS) { if (k < cout && oc < cin) v = fmaxf(w[((long)k * cin + oc) * 9 + (8 - tap)], 0.f); }
        else if (mode == LRPX_PACK_BWD_PLAIN) { if (k < cout && oc < cin) v = w[((long)k * cin + oc) * 9 + (8 - tap)]; }
        return v * sc;
    };
    unsigned char* out = reinterpret_cast<unsigned char*>(header + F16X3_HEADER_FLOATS) +
                         ((((long)ocb * nchunk + chunk) * 3 + g) * 7) * 1024 + lane * 16;
    for (int dx = 0; dx < 3; ++dx) {
        unsigned short* o = reinterpret_cast<unsigned short*>(out + dx * 1024);
        for (int j = 0; j < 8; ++j) o[j] = __builtin_bit_cast(unsigned short, (_Float16)wv(chunk * 16 + 8 * lh + j, dx));
    }
#if LRPXH_XP6
    // fp6 cross products: planes 3/4 = B operand of MFMA m = 2g, planes 5/6 = MFMA 2g + 1 (none for g = 2).  A lane's 32 e2m3
    // values belong to ONE tap - 2m for lanes 0-31, 2m + 1 for lanes 32-63 (tap 9: zeros) - in the field order of the staging
    // code: field c meets x_c and holds (W_c - hi(W_c)) * 2^11, field 16 + c meets (x_c - hi) * 2^11 and holds W_c; all divided
    // by the lane's block scale 2^(e-127), e in the dword behind the 24 bytes (conv_f16x3.h, x6_split)
    for (int mm = 0; mm < 2; ++mm) {
        const int m = 2 * g + mm, tap = 2 * m + lh;
        unsigned* o = reinterpret_cast<unsigned*>(out + (3 + 2 * mm) * 1024);       // 16 B here, 16 B one plane on
        unsigned words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (m < 5 && tap <= 8) {
            float ws[16], wr[16], mx = 0.f;
            for (int c = 0; c < 16; ++c) {
                const int k = chunk * 16 + c;
                float v = 0.f;
                if (mode == LRPX_PACK_BWD_POS) { if (k < cout && oc < cin) v = fmaxf(w[((long)k * cin + oc) * 9 + (8 - tap)], 0.f); }
                else if (mode == LRPX_PACK_BWD_PLAIN) { if (k < cout && oc < cin) v = w[((long)k * cin + oc) * 9 + (8 - tap)]; }
                v *= sc;
                ws[c] = v;
                wr[c] = (v - (float)(_Float16)v) * 2048.f;
                mx = fmaxf(mx, fmaxf(fabsf(v), fabsf(wr[c])));
            }
            const float bs = mx * (16.f / 15.f * 0.25f);
            const int e = (int)((__builtin_bit_cast(unsigned, bs) >> 23) & 0xffu);
            if (e > 0 && e < 255) {
                const float inv = exp2i(127 - e);
                for (int c = 0; c < 16; ++c) {
                    const unsigned f0 = fp6_e2m3_encode(wr[c] * inv), f1 = fp6_e2m3_encode(ws[c] * inv);
                    const int b0 = 6 * c, b1 = 6 * (16 + c);
                    words[b0 >> 5] |= f0 << (b0 & 31);
                    if ((b0 & 31) > 26) words[(b0 >> 5) + 1] |= f0 >> (32 - (b0 & 31));
                    words[b1 >> 5] |= f1 << (b1 & 31);
                    if ((b1 & 31) > 26) words[(b1 >> 5) + 1] |= f1 >> (32 - (b1 & 31));
                }
                words[6] = (unsigned)e;
            }
        }
        for (int q = 0; q < 4; ++q) { o[q] = words[q]; o[256 + q] = words[4 + q]; }
    }
#else
    // planes 3/4: B operand of fp8 MFMA m = 2g (taps 2m, 2m+1; lanes 0-31: "L" slices, lanes 32-63: "S" slices);
    // planes 5/6: MFMA 2g+1 (none for g = 2)
    for (int mm = 0; mm < 2; ++mm) {
        const int m = 2 * g + mm;
        for (int i = 0; i < 2; ++i) {
            const int tap = f8_slot_tap(m, i);
            const int kind = (m < 5 && tap <= 8) ? lh : 2;
            unsigned* o = reinterpret_cast<unsigned*>(out + (3 + 2 * mm + i) * 1024);
            for (int q = 0; q < 4; ++q) {
                float v[4];
                for (int e = 0; e < 4; ++e) {
                    float x = 0.f;
                    if (kind != 2) {
                        const int k = chunk * 16 + q * 4 + e;
                        float wsc = 0.f;
                        if (mode == LRPX_PACK_BWD_POS) { if (k < cout && oc < cin) wsc = fmaxf(w[((long)k * cin + oc) * 9 + (8 - tap)], 0.f); }
                        else if (mode == LRPX_PACK_BWD_PLAIN) { if (k < cout && oc < cin) wsc = w[((long)k * cin + oc) * 9 + (8 - tap)]; }
                        wsc *= sc;
                        // kind 0 ("L": activations' residual x - hi): weight operand W * 2^-4;  kind 1 ("S"): (W - hi_W) * 2^4
                        x = kind == 0 ? wsc * 0.0625f : (wsc - (float)(_Float16)wsc) * 16.f;
                    }
                    v[e] = x;
                }
                o[q] = pack_fp8x4(v[0], v[1], v[2], v[3]);
            }
        }
    }
#endif
}

static void pack_dims(int cout, int cin, int mode, int kc, int* n_oc_pad, int* k_pad) {
    int n_oc, k;
    switch (mode) {
        case LRPX_PACK_FWD_DUAL: n_oc = 2 * cout; k = cin; break;
        case LRPX_PACK_FWD: n_oc = cout; k = cin; break;
        case LRPX_PACK_FWD_DUAL_FIRST: n_oc = 2 * cout; k = 2 * cin; break;
        case LRPX_PACK_BWD_POS: case LRPX_PACK_BWD_PLAIN: n_oc = cin; k = cout; break;
        case LRPX_PACK_BWD_FIRST: n_oc = 2 * cin; k = cout; break;
        case LRPX_PACK_DENSE_T: n_oc = cin; k = cout; break;
        default: n_oc = cout; k = cin; break;  // DENSE
    }
    *n_oc_pad = round_up(n_oc, 32);
    *k_pad = round_up(k, kc);
}

// ------------------------------------------------------------------------------------------------
// layout kernels
// ------------------------------------------------------------------------------------------------
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, int P, int c_pad,
                                    long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over n*P*c_pad
    if (idx >= total) return;
    int ch = idx % c_pad;
    long np = idx / c_pad;
    long n = np / P, p = np - n * P;
    dst[idx] = ch < c ? src[(n * c + ch) * P + p] : 0.f;
}

// signed first-layer input split into its positive and negative parts: channels [0,c) = max(x,0),
// [c,2c) = min(x,0), rest zero (PosNetConv.forward clamps, LRPtools/lrp_modules.py:81-84)
__global__ void nchw_to_nhwc_posneg_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, int P,
                                           int c_pad, long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over n*P*c_pad
    if (idx >= total) return;
    int ch = idx % c_pad;
    long np = idx / c_pad;
    long n = np / P, p = np - n * P;
    float v = 0.f;
    if (ch < c) v = fmaxf(src[(n * c + ch) * P + p], 0.f);
    else if (ch < 2 * c) v = fminf(src[(n * c + ch - c) * P + p], 0.f);
    dst[idx] = v;
}

__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int c, int P, int c_src,
                                    long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over n*c*P
    if (idx >= total) return;
    long p = idx % P;
    long nc = idx / P;
    long n = nc / c;
    int ch = nc - n * c;
    dst[idx] = src[(n * P + p) * c_src + ch];
}

// MaxPool2d(2,2) forward, NHWC, 4 channels per thread
__global__ void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int ho, int wo, int c4,
                                   long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over n*ho*wo*c4
    if (idx >= total) return;
    int cc = idx % c4;
    long r = idx / c4;
    int xo = r % wo; r /= wo;
    int yo = r % ho;
    long n = r / ho;
    const int wi = 2 * wo;
    const f32x4* base = reinterpret_cast<const f32x4*>(x) + ((n * 2 * ho + 2 * yo) * wi + 2 * xo) * c4 + cc;
    f32x4 a = base[0], b = base[c4], c_ = base[(long)wi * c4], d = base[(long)wi * c4 + c4];
    f32x4 m;
#pragma unroll
    for (int e = 0; e < 4; ++e) m[e] = fmaxf(fmaxf(a[e], b[e]), fmaxf(c_[e], d[e]));
    reinterpret_cast<f32x4*>(y)[idx] = m;
}

// Pool2d rule as a scatter (lrp_modules.py:182-195 with S already at the winners): s_hi[n][2yo+dy][2xo+dx][c] =
// s_lo[n][yo][xo][c] if (dy,dx) is the window position am[img][yo][xo][c], else 0.  One thread = one pooled pixel x 4 ch.
__global__ void unpool_winner_kernel(const float* __restrict__ s_lo, const unsigned char* __restrict__ am,
                                     const int* __restrict__ map2img, float* __restrict__ s_hi, int ho, int wo, int c4,
                                     long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over n*ho*wo*c4
    if (idx >= total) return;
    const int cc = idx % c4;
    long r = idx / c4;
    const int xo = r % wo; r /= wo;
    const int yo = r % ho;
    const long n = r / ho;
    const long img = map2img ? map2img[n] : n;
    const f32x4 v = reinterpret_cast<const f32x4*>(s_lo)[idx];
    const unsigned pk = reinterpret_cast<const unsigned*>(am)[((img * ho + yo) * wo + xo) * c4 + cc];
    const int wi = 2 * wo;
    const long b0 = ((n * 2 * ho + 2 * yo) * wi + 2 * xo) * c4 + cc;
    const long off[4] = {0, c4, (long)wi * c4, (long)wi * c4 + c4};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = ((pk >> (8 * e)) & 0xffu) == (unsigned)k ? v[e] : 0.f;
        reinterpret_cast<f32x4*>(s_hi)[b0 + off[k]] = o;
    }
}

// NHWC <-> BLOCKED (blocked.h): a wave moves 16 pixels x 16 channels - 64-byte pieces on the NHWC side, 256-byte runs on the
// blocked side.  `n_groups` tensors of P pixels each (per-image multiplicands: one block set per image; an S tensor: ONE group of
// n_maps * P pixels); to_blocked = 0: the way back.
__global__ void blocked_convert_kernel(const float* __restrict__ src, float* __restrict__ dst, int P, int C, long total,
                                       int to_blocked) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= total) return;
    const int l = (int)(t & 63);
    long r = t >> 6;
    const int npg = (P + 15) >> 4, nch = C >> 4;
    const int pg = (int)(r % npg); r /= npg;
    const int chunk = (int)(r % nch);
    const long grp = r / nch;
    const int pix = pg * 16 + (l >> 2), k = l & 3;
    if (pix >= P) return;
    const long cs = blk_chunk_stride(P);
    const long nh = (grp * P + pix) * C + chunk * 16 + k * 4;
    const long bl = grp * (long)nch * cs + (long)chunk * cs + blk_pix_off(pix) + k * 128;
    if (to_blocked) *reinterpret_cast<f32x4*>(dst + bl) = *reinterpret_cast<const f32x4*>(src + nh);
    else *reinterpret_cast<f32x4*>(dst + nh) = *reinterpret_cast<const f32x4*>(src + bl);
}

// r / z of the split-product chains, with 0 where z == 0.  The reference divides by z + 1e-7 [z == 0] (LRPtools/utils.py:16-18), which
// makes S = R / 1e-7 at such a pixel - and multiplies it with zeros only: Z+_c(p) = sum_{i,taps} x_i W+[c,i] = 0 with every term >= 0 means
// every product that S_c(p) meets in R_in = x * convT(S, W+) is zero (the first layer's x+ W+ + x- W- likewise).  The VALUE there never
// reaches a result, but in the fp16 split-product kernels it would set the map's operand scale 1e7 above everything that matters
// (relevance on a dead feature channel: tests/test_gpu_vgg.py::test_chain_hostile_weights_all_modes).  So those kernels' S is 0 there.
__device__ __forceinline__ float div_safe0(float r, float z) { return z == 0.f ? 0.f : r / z; }

// winners of a 2x2 max-pool + the fused multiplicand max / safe(Z+ at the winner); one thread = one pooled pixel x 4 ch
__global__ void pool_winner_kernel(const float* __restrict__ x, const float* __restrict__ z, float* __restrict__ xzw,
                                   unsigned char* __restrict__ am, int ho, int wo, int c4, long total, float* __restrict__ xzw_blk,
                                   float* __restrict__ y_pool) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over n*ho*wo*c4
    if (idx >= total) return;
    const int cc = idx % c4;
    long r = idx / c4;
    const int xo = r % wo; r /= wo;
    const int yo = r % ho;
    const long n = r / ho;
    const int wi = 2 * wo;
    const long b0 = ((n * 2 * ho + 2 * yo) * wi + 2 * xo) * c4 + cc;
    const long off[4] = {0, c4, (long)wi * c4, (long)wi * c4 + c4};
    f32x4 w4[4], z4[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        w4[k] = reinterpret_cast<const f32x4*>(x)[b0 + off[k]];
        z4[k] = reinterpret_cast<const f32x4*>(z)[b0 + off[k]];
    }
    f32x4 o;
    unsigned pk = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float m = w4[0][e], zz = z4[0][e];
        unsigned a_ = 0;
        if (w4[1][e] > m) { m = w4[1][e]; zz = z4[1][e]; a_ = 1; }
        if (w4[2][e] > m) { m = w4[2][e]; zz = z4[2][e]; a_ = 2; }
        if (w4[3][e] > m) { m = w4[3][e]; zz = z4[3][e]; a_ = 3; }
        o[e] = div_safe0(m, zz);
        pk |= a_ << (8 * e);
    }
    reinterpret_cast<f32x4*>(xzw)[idx] = o;
    reinterpret_cast<unsigned*>(am)[idx] = pk;
    if (y_pool) {           // the pooled activations themselves (the forward pass: this kernel then replaces maxpool_fwd_kernel - same expression)
        f32x4 m4;
#pragma unroll
        for (int e = 0; e < 4; ++e) m4[e] = fmaxf(fmaxf(w4[0][e], w4[1][e]), fmaxf(w4[2][e], w4[3][e]));
        reinterpret_cast<f32x4*>(y_pool)[idx] = m4;
    }
    if (xzw_blk) {          // the same multiplicand in the BLOCKED layout (blocked.h), one block set per image
        const int P = ho * wo;
        *reinterpret_cast<f32x4*>(xzw_blk + n * (long)(c4 >> 2) * blk_chunk_stride(P) + blk_off((long)yo * wo + xo, 4 * cc, blk_chunk_stride(P))) = o;
    }
}

// Pool2d rule + division by the Z+ of the conv below.  One thread = ITER x (one hi-res pixel x 4 channels); a block
// covers 256*ITER consecutive float4 (ITER > 1 only when that divides a map: one amax update per block).
template <int ITER>
__global__ void maxpool_relevance_kernel(const float* __restrict__ x, const float* __restrict__ r_out,
                                         const float* __restrict__ zdiv, const int* __restrict__ map2img,
                                         float* __restrict__ r_in, float* __restrict__ s_out, int ho, int wo, int c4,
                                         long total, int s_chunk4, long total_pix, unsigned* __restrict__ amax) {
    const long base = (long)blockIdx.x * (blockDim.x * ITER) + threadIdx.x;   // over n_maps*(2ho)*(2wo)*c4
    if (base >= total) return;    // (total is a multiple of the block span whenever amax is used)
    const int wi = 2 * wo, hi = 2 * ho;
    float mabs = 0.f;
    long n_first = 0;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const long idx = base + (long)it * blockDim.x;
        if (ITER > 1 && idx >= total) break;
        int cc = idx % c4;
        long r = idx / c4;
        int xi = r % wi; r /= wi;
        int yi = r % hi;
        long n = r / hi;
        if (it == 0) n_first = n;
        long img = map2img ? map2img[n] : n;
        const int yo = yi >> 1, xo = xi >> 1;
        const int pos = (yi & 1) * 2 + (xi & 1);   // position of this pixel inside its window, row-major
        const f32x4* xb = reinterpret_cast<const f32x4*>(x) + ((img * hi + 2 * yo) * wi + 2 * xo) * c4 + cc;
        f32x4 w4[4] = {xb[0], xb[c4], xb[(long)wi * c4], xb[(long)wi * c4 + c4]};
        f32x4 ro = reinterpret_cast<const f32x4*>(r_out)[((n * ho + yo) * wo + xo) * c4 + cc];
        f32x4 z = {1.f, 1.f, 1.f, 1.f};
        if (zdiv) z = reinterpret_cast<const f32x4*>(zdiv)[((img * hi + yi) * wi + xi) * c4 + cc];
        f32x4 ri, so;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            // first maximum in row-major window order wins (strict >), as max_pool2d's backward does
            float m = w4[0][e];
            int am = 0;
            if (w4[1][e] > m) { m = w4[1][e]; am = 1; }
            if (w4[2][e] > m) { m = w4[2][e]; am = 2; }
            if (w4[3][e] > m) { m = w4[3][e]; am = 3; }
            float v = 0.f;
            if (am == pos) v = m * (ro[e] / stab_safe(m));
            ri[e] = v;
            so[e] = zdiv ? v / stab_safe(z[e]) : v;
        }
        if (r_in) reinterpret_cast<f32x4*>(r_in)[idx] = ri;
        if (s_out) {
            long o = idx;
            if (s_chunk4) {   // channel-chunked: [c4 / s_chunk4][pixel][s_chunk4] in float4 units
                const long pix = idx / c4;
                o = ((long)(cc / s_chunk4) * total_pix + pix) * s_chunk4 + (cc % s_chunk4);
            }
            reinterpret_cast<f32x4*>(s_out)[o] = so;
            mabs = fmaxf(mabs, fmaxf(fmaxf(fabsf(so[0]), fabsf(so[1])), fmaxf(fabsf(so[2]), fabsf(so[3]))));
        }
    }
    if (amax) amax_commit(amax, n_first, mabs);
}

// ITER consecutive float4 per thread at block stride (ITER > 1 only when 256*ITER divides a map: one amax update/block)
template <int ITER>
__global__ void divide_stab_kernel(const float* __restrict__ r, const float* __restrict__ z,
                                   const int* __restrict__ map2img, float* __restrict__ s, long per4, int stab,
                                   long total, unsigned* __restrict__ amax) {
    const long base = (long)blockIdx.x * (blockDim.x * ITER) + threadIdx.x;   // float4 units over n_maps*per4
    if (base >= total) return;     // (total is a multiple of the block span whenever amax is used)
    const long n = base / per4;
    const long img = map2img ? map2img[n] : n;
    float mabs = 0.f;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const long idx = base + (long)it * blockDim.x;
        if (ITER > 1 && idx >= total) break;
        const long i = idx - n * per4;
        f32x4 rv = reinterpret_cast<const f32x4*>(r)[idx];
        f32x4 zv = reinterpret_cast<const f32x4*>(z)[img * per4 + i];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float zz = zv[e];
            zz = stab == STAB_SAFE ? stab_safe(zz) : (stab == STAB_EPS ? stab_eps(zz) : zz);
            o[e] = stab == STAB_SAFE0 ? div_safe0(rv[e], zv[e]) : rv[e] / zz;
        }
        reinterpret_cast<f32x4*>(s)[idx] = o;
        mabs = fmaxf(mabs, fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3]))));
    }
    if (amax) amax_commit(amax, n, mabs);
}

// S = R / safe(Z+) at the top of the mode-3 chain, written BLOCKED (blocked.h) with the per-map maxima.
// r: [n_maps][P][C] NHWC, z: [n_img][P][C].  A wave moves 8 pixels x 32 channels per step: 8 lanes read one pixel's 128 contiguous
// bytes (whole lines in), and write per (16-channel chunk, 4-channel part) 8 pixels x 16 bytes = one 128-byte run (whole lines out).
// It walks ITER such steps - fewer pixels than a map has, so they belong to the map of its first pixel or to the next one: two maxima.
template <int ITER>
__global__ void divide_stab_blocked_kernel(const float* __restrict__ r, const float* __restrict__ z,
                                           const int* __restrict__ map2img, float* __restrict__ s, int P, int C, long n_pix,
                                           unsigned* __restrict__ amax, int n_maps) {
    const long gw = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int l = threadIdx.x & 63;
    const int ncp = C >> 5;                                  // pairs of 16-channel chunks
    const int cp = (int)(gw % ncp);
    const long pix0 = (gw / ncp) * (8 * ITER);
    if (pix0 >= n_pix) return;
    const long cs = blk_chunk_stride(n_pix);
    const long n_w0 = pix0 / P;
    const int q = l & 7;                                     // 4-channel quad of the 32 channels: chunk 2 cp + (q >> 2), part q & 3
    const int c = cp * 32 + q * 4;
    float m_lo = 0.f, m_hi = 0.f;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const long pix = pix0 + it * 8 + (l >> 3);
        if (pix >= n_pix) break;
        const long n = pix / P, p = pix - n * P;
        const long img = map2img ? map2img[n] : n;
        const f32x4 rv = *reinterpret_cast<const f32x4*>(r + pix * C + c);
        const f32x4 zv = *reinterpret_cast<const f32x4*>(z + (img * P + p) * C + c);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = div_safe0(rv[e], zv[e]);
        *reinterpret_cast<f32x4*>(s + (long)(2 * cp + (q >> 2)) * cs + blk_pix_off(pix) + (q & 3) * 128) = o;
        const float m = fmaxf(fmaxf(fabsf(o[0]), fabsf(o[1])), fmaxf(fabsf(o[2]), fabsf(o[3])));
        if (n == n_w0) m_lo = fmaxf(m_lo, m); else m_hi = fmaxf(m_hi, m);
    }
    if (amax) {
        m_lo = wave_max(m_lo); m_hi = wave_max(m_hi);
        if (l == 0) {
            amax_update(&amax[n_w0], m_lo);
            if (n_w0 + 1 < n_maps) amax_update(&amax[n_w0 + 1], m_hi);
        }
    }
}

// x / safe(z) per image (the multiplicand of a conv directly above another conv), NHWC and BLOCKED in one pass
__global__ void divide_safe_blk_kernel(const float* __restrict__ r, const float* __restrict__ z, float* __restrict__ s,
                                       float* __restrict__ s_blk, int P, int c4, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;      // float4 units over n_img * P * c4
    if (idx >= total) return;
    const int cq = (int)(idx % c4);
    const long rp = idx / c4;
    const long n = rp / P;
    const int pix = (int)(rp - n * P);
    const f32x4 rv = reinterpret_cast<const f32x4*>(r)[idx], zv = reinterpret_cast<const f32x4*>(z)[idx];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = div_safe0(rv[e], zv[e]);
    reinterpret_cast<f32x4*>(s)[idx] = o;
    const long cs = blk_chunk_stride(P);
    *reinterpret_cast<f32x4*>(s_blk + n * (long)(c4 >> 2) * cs + blk_off(pix, 4 * cq, cs)) = o;
}

__global__ void cumsum_maps_kernel(const float* __restrict__ in, float* __restrict__ out, int t_per_img, long per4,
                                   long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // float4 units over n_img*per4
    if (idx >= total) return;
    long b = idx / per4, i = idx - b * per4;
    f32x4 run = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < t_per_img; ++t) {
        long o = (b * t_per_img + t) * per4 + i;
        f32x4 v = reinterpret_cast<const f32x4*>(in)[o];
        run = (t == 0) ? v : run + v;
        reinterpret_cast<f32x4*>(out)[o] = run;
    }
}

// Variable caption lengths: `in` holds only the valid (image, word) rows, image b's len[b] rows starting at row off[b]; `out` is the
// padded [n_img][t_per_img] layout: running sums (accumulate, lrp_wrapper.py:64-82 quirk) or copies of the valid rows, exact zeros
// behind an image's last word.
__global__ void scatter_maps_kernel(const float* __restrict__ in, float* __restrict__ out, int t_per_img,
                                    const int* __restrict__ lens, const int* __restrict__ offs, long per4, long total,
                                    int accumulate) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // float4 units over n_img*per4
    if (idx >= total) return;
    long b = idx / per4, i = idx - b * per4;
    const int len = min(lens[b], t_per_img);
    const long src0 = offs[b];
    f32x4 run = {0.f, 0.f, 0.f, 0.f};
    for (int t = 0; t < t_per_img; ++t) {
        f32x4 o = {0.f, 0.f, 0.f, 0.f};
        if (t < len) {
            f32x4 v = reinterpret_cast<const f32x4*>(in)[(src0 + t) * per4 + i];
            run = (t == 0 || !accumulate) ? v : run + v;
            o = run;
        }
        reinterpret_cast<f32x4*>(out)[(b * t_per_img + t) * per4 + i] = o;
    }
}

// dst[r][:] = src[rows[r]][:]  (per-row operands of the compacted (word, pixel) rules; 4-byte items of any type)
__global__ void gather_rows_kernel(const uint32_t* __restrict__ src, const int* __restrict__ rows, uint32_t* __restrict__ dst,
                                   int width, long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const long r = idx / width;
    dst[idx] = src[(long)rows[r] * width + (idx - r * width)];
}

// Grad-CAM (models/gridTDmodel.py:1760-1771): cam[p] = relu(sum_c F[img][p][c] * mean_p'(G[row][p'][c])) / (max + 1e-6).
// One workgroup per (image, word) row; P <= 256 pixels, any C (multiple of 4).
__global__ __launch_bounds__(256) void gradcam_kernel(const float* __restrict__ feats, const float* __restrict__ grads,
                                                      const int* __restrict__ map2img, float* __restrict__ cam, int P,
                                                      int C) {
    extern __shared__ float wsh[];            // [C] channel weights, then [4] wave maxima
    const int row = blockIdx.x, tid = threadIdx.x;
    const long img = map2img ? map2img[row] : row;
    const float* G = grads + (long)row * P * C;
    for (int c = tid; c < C; c += 256) {
        float s_ = 0.f;
        for (int p = 0; p < P; ++p) s_ += G[(long)p * C + c];
        wsh[c] = s_ / (float)P;
    }
    __syncthreads();
    float v = 0.f;
    if (tid < P) {
        const float* Fp = feats + (img * P + tid) * C;
        for (int c = 0; c < C; ++c) v += Fp[c] * wsh[c];
        v = v > 0.f ? v : 0.f;
    }
    float m = wave_max(v);
    if ((tid & 63) == 0) wsh[C + (tid >> 6)] = m;
    __syncthreads();
    m = fmaxf(fmaxf(wsh[C], wsh[C + 1]), fmaxf(wsh[C + 2], wsh[C + 3]));
    if (tid < P) cam[(long)row * P + tid] = v / (m + 1e-6f);
}

// top ReLU hook of the guided pass: out = max(g,0) * [y > 0]   (models/gridTDmodel.py:1680-1686)
// plain != 0: the autograd ReLU backward only (g * [y > 0]; ExplainGridTDGradient.explain_cnn, :1507-1521)
__global__ void guided_gate_kernel(const float* __restrict__ g, const float* __restrict__ y,
                                   const int* __restrict__ map2img, float* __restrict__ out, long per4, long total,
                                   int plain) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // float4 units over n_maps*per4
    if (idx >= total) return;
    long n = idx / per4, i = idx - n * per4;
    long img = map2img ? map2img[n] : n;
    f32x4 gv = reinterpret_cast<const f32x4*>(g)[idx];
    f32x4 yv = reinterpret_cast<const f32x4*>(y)[img * per4 + i];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (yv[e] > 0.f && (plain || gv[e] > 0.f)) ? gv[e] : 0.f;
    reinterpret_cast<f32x4*>(out)[idx] = o;
}

// max_pool2d backward (gradient to the arg-max, first maximum wins) followed by the guided ReLU hook of the conv
// below: out = [this pixel is the window's arg-max and a > 0] * max(g, 0).  AMAX: max|out| per map for the fp16 scale of
// the conv below (whole blocks of 256 threads, host-checked).  (One thread per OUTPUT pixel: a variant with one thread per
// window - the window read once, four scattered stores - was 2.3x slower.)
template <bool AMAX, int ITER>
__global__ void maxpool_guided_bwd_kernel(const float* __restrict__ x, const float* __restrict__ g_out,
                                          const int* __restrict__ map2img, float* __restrict__ g_in, int ho, int wo,
                                          int c4, long total, int plain, unsigned* __restrict__ amax) {
    // ITER items per thread at block stride (AMAX: 256 * ITER divides a map, one amax update per block - one per 256
    // items made the updates of a map's single word the bottleneck: 5.9 instead of 1.8 ms per launch)
    const long base = (long)blockIdx.x * (blockDim.x * ITER) + threadIdx.x;   // over n_maps*(2ho)*(2wo)*c4
    const int wi = 2 * wo, hi = 2 * ho;
    float mabs = 0.f;
    long n = 0;
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
        const long idx = base + (long)it * blockDim.x;
        if (idx >= total) break;
        int cc = idx % c4;
        long r = idx / c4;
        int xi = r % wi; r /= wi;
        int yi = r % hi;
        n = r / hi;
        long img = map2img ? map2img[n] : n;
        const int yo = yi >> 1, xo = xi >> 1;
        const int pos = (yi & 1) * 2 + (xi & 1);
        const f32x4* xb = reinterpret_cast<const f32x4*>(x) + ((img * hi + 2 * yo) * wi + 2 * xo) * c4 + cc;
        f32x4 w4[4] = {xb[0], xb[c4], xb[(long)wi * c4], xb[(long)wi * c4 + c4]};
        f32x4 go = reinterpret_cast<const f32x4*>(g_out)[((n * ho + yo) * wo + xo) * c4 + cc];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float m = w4[0][e];
            int am = 0;
            if (w4[1][e] > m) { m = w4[1][e]; am = 1; }
            if (w4[2][e] > m) { m = w4[2][e]; am = 2; }
            if (w4[3][e] > m) { m = w4[3][e]; am = 3; }
            o[e] = (am == pos && m > 0.f && (plain || go[e] > 0.f)) ? go[e] : 0.f;
            mabs = fmaxf(mabs, fabsf(o[e]));
        }
        reinterpret_cast<f32x4*>(g_in)[idx] = o;
    }
    if constexpr (AMAX) amax_commit(amax, n, mabs);
}

__global__ void accumulate_kernel(float* __restrict__ dst, const float* __restrict__ src, long n4) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n4) return;
    reinterpret_cast<f32x4*>(dst)[idx] += reinterpret_cast<const f32x4*>(src)[idx];
}

// out[row][c] = in[row][c] + in[row][c + half]   (x+ and x- halves of a split relevance tensor)
__global__ void fold_halves_kernel(const float* __restrict__ in, float* __restrict__ out, int half, long total) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // over rows*half
    if (idx >= total) return;
    long row = idx / half;
    int c = idx - row * half;
    out[idx] = in[row * 2 * half + c] + in[row * 2 * half + half + c];
}

__global__ void check_kernel(const float* __restrict__ buf, long n, unsigned* flags) {
    long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    long stride = (long)gridDim.x * blockDim.x;
    unsigned f = 0;
    for (long i = idx; i < n; i += stride) {
        float v = buf[i];
        if (!isfinite(v)) f |= 1u;
        if (v != 0.f) f |= 2u;
    }
    if (f) atomicOr(flags, f);
}

static inline unsigned grid_for(long total, int block = 256) { return (unsigned)ceil_div(total, block); }

}  // namespace lrpx

using namespace lrpx;

extern "C" {

int lrpx_version(void) { return 100; }

const char* lrpx_last_error_string(void) { return g_err; }

size_t lrpx_packed_floats(int n_oc, int k, int taps, int kc) {
    return (size_t)round_up(n_oc, 32) * (size_t)round_up(k, kc) * (size_t)taps;
}

int lrpx_pack_weights(const float* w, int cout, int cin, int taps, int mode, int kc, float* packed, void* stream) {
    LRPX_CHECK_PTRS("lrpx_pack_weights", {w, "w"}, {packed, "packed"});
    LRPX_REQUIRE(w && packed, "pack_weights: null pointer");
    LRPX_REQUIRE(taps == 9 || taps == 1, "pack_weights: taps must be 1 or 9");
    LRPX_REQUIRE(kc == 8 || kc == 16 || kc == 32, "pack_weights: kc must be 8, 16 or 32");
    LRPX_REQUIRE(mode >= 0 && mode <= LRPX_PACK_FWD_DUAL_FIRST, "pack_weights: unknown mode %d", mode);
    int n_oc_pad, k_pad;
    pack_dims(cout, cin, mode, kc, &n_oc_pad, &k_pad);
    long total = (long)n_oc_pad * k_pad * taps;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, packed, cout,
                       cin, taps, mode, kc, n_oc_pad, k_pad, total);
    return check_launch("pack_weights");
}

size_t lrpx_packed_bf16x3_bytes(int n_oc, int k, int taps) {
    return (size_t)round_up(n_oc, 32) * (size_t)round_up(k, 16) * (size_t)taps * 3 * sizeof(unsigned short);
}

int lrpx_pack_weights_bf16x3(const float* w, int cout, int cin, int taps, int mode, void* packed, void* stream) {
    LRPX_CHECK_PTRS("lrpx_pack_weights_bf16x3", {w, "w"}, {packed, "packed"});
    LRPX_REQUIRE(w && packed && (taps == 9 || taps == 1), "pack_weights_bf16x3: bad arguments (3x3 kernels or dense matrices)");
    LRPX_REQUIRE(mode == LRPX_PACK_BWD_POS || mode == LRPX_PACK_BWD_PLAIN || mode == LRPX_PACK_FWD ||
                     mode == LRPX_PACK_FWD_DUAL, "pack_weights_bf16x3: mode %d not supported", mode);
    int n_oc_pad, k_pad;
    pack_dims(cout, cin, mode, 16, &n_oc_pad, &k_pad);
    long total = (long)n_oc_pad * k_pad * taps;     // threads: one per (oc, k, tap)
    hipLaunchKernelGGL(pack_weights_bf16x3_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w,
                       (unsigned short*)packed, cout, cin, taps, mode, k_pad, total);
    return check_launch("pack_weights_bf16x3");
}

int lrpx_nchw_to_nhwc(const float* src, float* dst, int n, int c, int hw_pix, int c_pad, void* stream) {
    LRPX_CHECK_PTRS("lrpx_nchw_to_nhwc", {src, "src"}, {dst, "dst"});
    LRPX_REQUIRE(src && dst && c <= c_pad, "nchw_to_nhwc: bad arguments");
    long total = (long)n * hw_pix * c_pad;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, c,
                       hw_pix, c_pad, total);
    return check_launch("nchw_to_nhwc");
}

int lrpx_nchw_to_nhwc_posneg(const float* src, float* dst, int n, int c, int hw_pix, int c_pad, void* stream) {
    LRPX_CHECK_PTRS("lrpx_nchw_to_nhwc_posneg", {src, "src"}, {dst, "dst"});
    LRPX_REQUIRE(src && dst && 2 * c <= c_pad, "nchw_to_nhwc_posneg: bad arguments");
    long total = (long)n * hw_pix * c_pad;
    hipLaunchKernelGGL(nchw_to_nhwc_posneg_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst,
                       c, hw_pix, c_pad, total);
    return check_launch("nchw_to_nhwc_posneg");
}

int lrpx_nhwc_to_nchw(const float* src, float* dst, int n, int c, int hw_pix, int c_src, void* stream) {
    LRPX_CHECK_PTRS("lrpx_nhwc_to_nchw", {src, "src"}, {dst, "dst"});
    LRPX_REQUIRE(src && dst && c <= c_src, "nhwc_to_nchw: bad arguments");
    long total = (long)n * hw_pix * c;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, c,
                       hw_pix, c_src, total);
    return check_launch("nhwc_to_nchw");
}

int lrpx_maxpool2x2_fwd(const float* x, float* y, int n, int h, int w, int c, void* stream) {
    LRPX_CHECK_PTRS("lrpx_maxpool2x2_fwd", {x, "x"}, {y, "y"});
    LRPX_REQUIRE(x && y && (h % 2 == 0) && (w % 2 == 0) && (c % 4 == 0), "maxpool2x2_fwd: need even h,w and c%%4==0");
    long total = (long)n * (h / 2) * (w / 2) * (c / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, y, h / 2,
                       w / 2, c / 4, total);
    return check_launch("maxpool2x2_fwd");
}

}  // extern "C"
namespace lrpx {
int maxpool_relevance_amax(const float* x, const float* r_out, const float* zdiv, const int32_t* map2img, float* r_in,
                           float* s_out, int n_maps, int h_out, int w_out, int c, int s_chunk, unsigned* amax,
                           hipStream_t stream) {
    LRPX_REQUIRE(x && r_out && (r_in || s_out) && (c % 4 == 0), "maxpool2x2_relevance: bad arguments");
    LRPX_REQUIRE(s_chunk == 0 || (s_chunk % 4 == 0 && c % s_chunk == 0), "maxpool2x2_relevance: bad s_chunk");
    const long per_map = (long)(2 * h_out) * (2 * w_out) * (c / 4);
    const long total = (long)n_maps * per_map;
    LRPX_REQUIRE(!amax || (s_out && per_map % 256 == 0), "maxpool2x2_relevance: amax needs s_out and whole blocks");
    if (amax && per_map % 2048 == 0) {      // 8 float4 per thread: 8x fewer amax updates (they all hit n_maps words)
        hipLaunchKernelGGL(maxpool_relevance_kernel<8>, dim3(grid_for(total, 2048)), dim3(256), 0, stream, x, r_out,
                           zdiv, map2img, r_in, s_out, h_out, w_out, c / 4, total, s_chunk / 4,
                           (long)n_maps * (2 * h_out) * (2 * w_out), amax);
    } else {
        hipLaunchKernelGGL(maxpool_relevance_kernel<1>, dim3(grid_for(total)), dim3(256), 0, stream, x, r_out,
                           zdiv, map2img, r_in, s_out, h_out, w_out, c / 4, total, s_chunk / 4,
                           (long)n_maps * (2 * h_out) * (2 * w_out), amax);
    }
    return check_launch("maxpool2x2_relevance");
}
int divide_stab_amax(const float* r, const float* z, const int32_t* map2img, float* s, int n_maps, long pix_c, int stab,
                     unsigned* amax, hipStream_t stream) {
    LRPX_REQUIRE(r && z && s && (pix_c % 4 == 0), "divide_stab: bad arguments");
    const long per4 = pix_c / 4, total = (long)n_maps * per4;
    LRPX_REQUIRE(!amax || per4 % 256 == 0, "divide_stab: amax needs whole blocks per map");
    if (amax && per4 % (256 * 7) == 0) {       // 14x14x512 features: 14 blocks per map instead of 98
        hipLaunchKernelGGL(divide_stab_kernel<7>, dim3(grid_for(total, 256 * 7)), dim3(256), 0, stream, r, z, map2img, s,
                           per4, stab, total, amax);
    } else {
        hipLaunchKernelGGL(divide_stab_kernel<1>, dim3(grid_for(total)), dim3(256), 0, stream, r, z, map2img, s, per4,
                           stab, total, amax);
    }
    return check_launch("divide_stab");
}
int divide_stab_blocked(const float* r, const float* z, const int32_t* map2img, float* s_blk, int n_maps, int pix, int c,
                        unsigned* amax, hipStream_t stream) {
    constexpr int ITER = 16;           // 128 pixels per wave
    LRPX_REQUIRE(r && z && s_blk && n_maps > 0 && pix >= 8 * ITER && c > 0 && c % 32 == 0,
                 "divide_stab_blocked: bad arguments (c %% 32, >= %d pixels per map)", 8 * ITER);
    const long n_pix = (long)n_maps * pix;
    const long waves = (long)(c / 32) * ceil_div(n_pix, 8 * ITER);
    hipLaunchKernelGGL(divide_stab_blocked_kernel<ITER>, dim3(grid_for(waves * 64)), dim3(256), 0, stream, r, z, map2img, s_blk, pix, c,
                       n_pix, amax, n_maps);
    return check_launch("divide_stab_blocked");
}
// the two producers of the per-image multiplicands (lrpx_vgg16_trace_derive) with the BLOCKED copy written in the same pass
int pool_winner_blk(const float* x, const float* z, float* xzw, uint8_t* am, float* xzw_blk, int n, int h_out, int w_out, int c, hipStream_t stream,
                    float* y_pool) {
    LRPX_REQUIRE(x && z && xzw && am && n > 0 && h_out > 0 && w_out > 0 && c % 16 == 0, "pool_winner: bad arguments");
    const long total = (long)n * h_out * w_out * (c / 4);
    hipLaunchKernelGGL(pool_winner_kernel, dim3(grid_for(total)), dim3(256), 0, stream, x, z, xzw, am, h_out, w_out, c / 4, total, xzw_blk, y_pool);
    return check_launch("pool_winner");
}
int divide_safe_blk(const float* r, const float* z, float* s, float* s_blk, int n_img, int pix, int c, hipStream_t stream) {
    LRPX_REQUIRE(r && z && s && s_blk && n_img > 0 && pix > 0 && c % 16 == 0, "divide_safe_blk: bad arguments");
    const long total = (long)n_img * pix * (c / 4);
    hipLaunchKernelGGL(divide_safe_blk_kernel, dim3(grid_for(total)), dim3(256), 0, stream, r, z, s, s_blk, pix, c / 4, total);
    return check_launch("divide_safe_blk");
}
}  // namespace lrpx
extern "C" {

int lrpx_maxpool2x2_relevance(const float* x, const float* r_out, const float* zdiv, const int32_t* map2img,
                              float* r_in, float* s_out, int n_maps, int h_out, int w_out, int c, int s_chunk,
                              void* stream) {
    LRPX_CHECK_PTRS("lrpx_maxpool2x2_relevance", {x, "x"}, {r_out, "r_out"}, {zdiv, "zdiv"}, {map2img, "map2img"}, {r_in, "r_in"}, {s_out, "s_out"});
    return maxpool_relevance_amax(x, r_out, zdiv, map2img, r_in, s_out, n_maps, h_out, w_out, c, s_chunk, nullptr,
                                  (hipStream_t)stream);
}

size_t lrpx_blocked_floats(long n_pix, int c) { return (size_t)blk_floats(n_pix, c); }

static int blocked_convert(const float* src, float* dst, long n_groups, int pix_per_group, int c, int to_blocked, void* stream) {
    LRPX_REQUIRE(src && dst && n_groups > 0 && pix_per_group > 0 && c > 0 && c % 16 == 0, "blocked layout: bad arguments (c %% 16)");
    const long total = n_groups * (c / 16) * ((pix_per_group + 15) / 16) * 64;
    hipLaunchKernelGGL(blocked_convert_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, dst, pix_per_group, c,
                       total, to_blocked);
    return check_launch("blocked_convert");
}
int lrpx_nhwc_to_blocked(const float* src, float* dst, long n_groups, int pix_per_group, int c, void* stream) {
    LRPX_CHECK_PTRS("lrpx_nhwc_to_blocked", {src, "src"}, {dst, "dst"});
    return blocked_convert(src, dst, n_groups, pix_per_group, c, 1, stream);
}
int lrpx_blocked_to_nhwc(const float* src, float* dst, long n_groups, int pix_per_group, int c, void* stream) {
    LRPX_CHECK_PTRS("lrpx_blocked_to_nhwc", {src, "src"}, {dst, "dst"});
    return blocked_convert(src, dst, n_groups, pix_per_group, c, 0, stream);
}

int lrpx_pool_winner(const float* x, const float* z, float* xzw, uint8_t* am, int n, int h_out, int w_out, int c,
                     void* stream) {
    LRPX_CHECK_PTRS("lrpx_pool_winner", {x, "x"}, {z, "z"}, {xzw, "xzw"}, {am, "am"});
    LRPX_REQUIRE(x && z && xzw && am && n > 0 && h_out > 0 && w_out > 0 && c % 4 == 0, "pool_winner: bad arguments");
    const long total = (long)n * h_out * w_out * (c / 4);
    hipLaunchKernelGGL(pool_winner_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x, z, xzw, am,
                       h_out, w_out, c / 4, total, (float*)nullptr, (float*)nullptr);
    return check_launch("pool_winner");
}

int lrpx_unpool_winner(const float* s_lo, const uint8_t* am, const int32_t* map2img, float* s_hi, int n_maps, int h_out,
                       int w_out, int c, void* stream) {
    LRPX_CHECK_PTRS("lrpx_unpool_winner", {s_lo, "s_lo"}, {am, "am"}, {map2img, "map2img"}, {s_hi, "s_hi"});
    LRPX_REQUIRE(s_lo && am && s_hi && n_maps > 0 && h_out > 0 && w_out > 0 && c % 4 == 0, "unpool_winner: bad arguments");
    const long total = (long)n_maps * h_out * w_out * (c / 4);
    hipLaunchKernelGGL(unpool_winner_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, s_lo, am, map2img,
                       s_hi, h_out, w_out, c / 4, total);
    return check_launch("unpool_winner");
}

int lrpx_amax_maps(const float* s, int n_maps, long per, uint32_t* amax, void* stream) {
    LRPX_CHECK_PTRS("lrpx_amax_maps", {s, "s"}, {amax, "amax"});
    LRPX_REQUIRE(s && amax && n_maps > 0 && per > 0 && per % 4 == 0, "amax_maps: bad arguments (per %% 4)");
    if (hipMemsetAsync(amax, 0, (size_t)n_maps * sizeof(uint32_t), (hipStream_t)stream) != hipSuccess) {
        set_error("amax_maps: memset failed");
        return LRPX_ELAUNCH;
    }
    const long per4 = per / 4, total = (long)n_maps * per4;
    // ITER float4 per thread where 256 * ITER divides a map (one atomic per block of 256 * ITER float4, ITER loads in flight per thread)
    const long q = per4 % 256 == 0 ? per4 / 256 : 0;
    auto go = [&](auto kern, int iter) {
        hipLaunchKernelGGL(kern, dim3(grid_for(total, 256 * iter)), dim3(256), 0, (hipStream_t)stream, s, per4, total, amax);
    };
    if (q && q % 8 == 0) go(amax_maps_kernel<8>, 8);
    else if (q && q % 7 == 0) go(amax_maps_kernel<7>, 7);
    else if (q && q % 9 == 0) go(amax_maps_kernel<9>, 9);          // (36 rows x 512: the bottom-up projector rules)
    else if (q && q % 6 == 0) go(amax_maps_kernel<6>, 6);
    else if (q && q % 4 == 0) go(amax_maps_kernel<4>, 4);
    else if (q && q % 3 == 0) go(amax_maps_kernel<3>, 3);
    else if (q && q % 2 == 0) go(amax_maps_kernel<2>, 2);
    else go(amax_maps_kernel<1>, 1);
    return check_launch("amax_maps");
}

size_t lrpx_packed_f16x2_bytes(int n_oc, int k, int taps) {
    return F16X3_HEADER_FLOATS * sizeof(float) +
           (size_t)round_up(n_oc, 32) * (size_t)round_up(k, 16) * (size_t)taps * 2 * sizeof(unsigned short);
}

size_t lrpx_packed_f16f8_bytes(int n_oc, int k) {
    return F16X3_HEADER_FLOATS * sizeof(float) + (size_t)(round_up(n_oc, 32) / 32) * (size_t)(round_up(k, 16) / 16) * 3 * 7 * 1024;
}

int lrpx_pack_weights_f16f8(const float* w, int cout, int cin, int mode, void* packed, void* stream) {
    LRPX_CHECK_PTRS("lrpx_pack_weights_f16f8", {w, "w"}, {packed, "packed"});
    LRPX_REQUIRE(w && packed, "pack_weights_f16f8: bad arguments");
    LRPX_REQUIRE(mode == LRPX_PACK_BWD_POS || mode == LRPX_PACK_BWD_PLAIN, "pack_weights_f16f8: mode %d not supported", mode);
    int n_oc_pad, k_pad;
    pack_dims(cout, cin, mode, 16, &n_oc_pad, &k_pad);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(packed, 0, F16X3_HEADER_FLOATS * sizeof(float), st) != hipSuccess) {
        set_error("pack_weights_f16f8: memset failed");
        return LRPX_ELAUNCH;
    }
    hipLaunchKernelGGL(amax_flat_kernel, dim3(256), dim3(256), 0, st, w, (long)cout * cin * 9, (unsigned*)packed + 1);
    const long total = (long)(n_oc_pad / 32) * (k_pad / 16) * 3 * 64;
    hipLaunchKernelGGL(pack_weights_f16f8_kernel, dim3(grid_for(total)), dim3(256), 0, st, w, (float*)packed, cout, cin,
                       mode, k_pad, total);
    return check_launch("pack_weights_f16f8");
}

int lrpx_pack_weights_f16x2(const float* w, int cout, int cin, int taps, int mode, void* packed, void* stream) {
    LRPX_CHECK_PTRS("lrpx_pack_weights_f16x2", {w, "w"}, {packed, "packed"});
    LRPX_REQUIRE(w && packed && (taps == 9 || taps == 1), "pack_weights_f16x2: bad arguments (3x3 kernels or dense matrices)");
    LRPX_REQUIRE(mode == LRPX_PACK_BWD_POS || mode == LRPX_PACK_BWD_PLAIN || mode == LRPX_PACK_FWD ||
                     mode == LRPX_PACK_FWD_DUAL || mode == LRPX_PACK_FWD_DUAL_FIRST, "pack_weights_f16x2: mode %d not supported", mode);
    int n_oc_pad, k_pad;
    pack_dims(cout, cin, mode, 16, &n_oc_pad, &k_pad);
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(packed, 0, F16X3_HEADER_FLOATS * sizeof(float), st) != hipSuccess) {
        set_error("pack_weights_f16x2: memset failed");
        return LRPX_ELAUNCH;
    }
    // layer scale from max|w| (an upper bound for every mode's transformed weights)
    const long nw = (long)cout * cin * taps;
    hipLaunchKernelGGL(amax_flat_kernel, dim3(256), dim3(256), 0, st, w, nw, (unsigned*)packed + 1);
    long total = (long)n_oc_pad * k_pad * taps;
    hipLaunchKernelGGL(pack_weights_f16x2_kernel, dim3(grid_for(total)), dim3(256), 0, st, w, (float*)packed, cout, cin,
                       taps, mode, k_pad, total);
    return check_launch("pack_weights_f16x2");
}

int lrpx_divide_stab(const float* r, const float* z, const int32_t* map2img, float* s, int n_maps, long pix_c,
                     int stab, void* stream) {
    LRPX_CHECK_PTRS("lrpx_divide_stab", {r, "r"}, {z, "z"}, {map2img, "map2img"}, {s, "s"});
    LRPX_REQUIRE(stab >= LRPX_STAB_NONE && stab <= LRPX_STAB_EPS, "divide_stab: unknown stabiliser %d", stab);
    return divide_stab_amax(r, z, map2img, s, n_maps, pix_c, stab, nullptr, (hipStream_t)stream);
}

int lrpx_cumsum_maps(const float* in, float* out, int n_img, int t_per_img, long per, void* stream) {
    LRPX_CHECK_PTRS("lrpx_cumsum_maps", {in, "in"}, {out, "out"});
    LRPX_REQUIRE(in && out && (per % 4 == 0) && t_per_img > 0, "cumsum_maps: bad arguments");
    long total = (long)n_img * (per / 4);
    hipLaunchKernelGGL(cumsum_maps_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out,
                       t_per_img, per / 4, total);
    return check_launch("cumsum_maps");
}

int lrpx_scatter_maps(const float* in, float* out, int n_img, int t_per_img, const int32_t* lens, const int32_t* offs,
                      long per, int accumulate, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_scatter_maps", {in, "in"}, {out, "out"}, {lens, "lens"}, {offs, "offs"});
    LRPX_REQUIRE(in && out && lens && offs && (per % 4 == 0) && t_per_img > 0 && n_img > 0, "scatter_maps: bad arguments");
    long total = (long)n_img * (per / 4);
    hipLaunchKernelGGL(scatter_maps_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, in, out, t_per_img,
                       lens, offs, per / 4, total, accumulate);
    return check_launch("scatter_maps");
}

int lrpx_zero(void* dst, size_t bytes, void* stream) {
    LRPX_REQUIRE(dst || bytes == 0, "zero: null pointer");
    LRPX_CHECK_PTRS_OPT("lrpx_zero", {dst, "dst"});
    if (bytes && hipMemsetAsync(dst, 0, bytes, (hipStream_t)stream) != hipSuccess) {
        set_error("zero: hipMemsetAsync failed");
        return LRPX_ELAUNCH;
    }
    return LRPX_OK;
}

int lrpx_gather_rows(const void* src, const int32_t* rows, void* dst, int n_rows, int width, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gather_rows", {src, "src"}, {rows, "rows"}, {dst, "dst"});
    LRPX_REQUIRE(src && rows && dst && n_rows > 0 && width > 0, "gather_rows: bad arguments");
    long total = (long)n_rows * width;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const uint32_t*)src,
                       rows, (uint32_t*)dst, width, total);
    return check_launch("gather_rows");
}

extern "C++" {
namespace lrpx {
int guided_gate(const float* g, const float* y, const int* map2img, float* out, int n_maps, long per, int plain,
                hipStream_t s) {
    long total = (long)n_maps * (per / 4);
    hipLaunchKernelGGL(guided_gate_kernel, dim3(grid_for(total)), dim3(256), 0, s, g, y, map2img, out, per / 4, total,
                       plain);
    return check_launch("guided_gate");
}
int fwd_dual_finish(const float* part, int nsplit, const float* bias, float* act, float* zpos, int n_img, long pix_per_img,
                    int cout, unsigned* amax, hipStream_t s) {
    const long per_img4 = pix_per_img * (cout / 2), total4 = (long)n_img * per_img4;
    LRPX_REQUIRE(cout % 4 == 0 && per_img4 % 256 == 0, "fwd_dual_finish: %ld float4 items per image are not whole blocks", per_img4);
    LRPX_REQUIRE(nsplit >= 1 && nsplit <= 16 && (nsplit & (nsplit - 1)) == 0, "fwd_dual_finish: %d splits (a power of two <= 16)", nsplit);
    hipLaunchKernelGGL(fwd_dual_finish_kernel, dim3((unsigned)(total4 / 256)), dim3(256), 0, s, part, nsplit,
                       (long)n_img * pix_per_img * 2 * cout, bias, act, zpos, cout, per_img4, amax);
    return check_launch("fwd_dual_finish");
}
// out[m][i] = x[img(m)][i] * (sum over the K splits of part[s][m][i]), pairwise in a fixed order (as fwd_dual_finish): the REL_MUL epilogue
// of a K-split relevance conv (lrpx_vgg16_relevance_ex, LRPX_B6_REL_KSPLIT14)
__global__ void rel_mul_finish_kernel(const float* __restrict__ part, int nsplit, long split_stride, const float* __restrict__ x,
                                      const int* __restrict__ map2img, float* __restrict__ out, long per_map4, long total4) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total4) return;
    f32x4 lvl[5];
    f32x4 v = reinterpret_cast<const f32x4*>(part)[idx];
    for (int s = 0; s < nsplit; ++s) {
        if (s) v = reinterpret_cast<const f32x4*>(part + s * split_stride)[idx];
        int k = s, l = 0;
        while (k & 1) { v = lvl[l] + v; k >>= 1; ++l; }
        lvl[l] = v;
    }
    const long m = idx / per_map4, w = idx - m * per_map4;
    const long img = map2img ? map2img[m] : m;
    const f32x4 xv = reinterpret_cast<const f32x4*>(x)[img * per_map4 + w];
    reinterpret_cast<f32x4*>(out)[idx] = f32x4{xv[0] * v[0], xv[1] * v[1], xv[2] * v[2], xv[3] * v[3]};
}
int rel_mul_finish(const float* part, int nsplit, const float* x, const int* map2img, float* out, int n_maps, long per_map, hipStream_t s) {
    LRPX_REQUIRE(per_map % 4 == 0 && nsplit >= 1 && nsplit <= 16 && (nsplit & (nsplit - 1)) == 0, "rel_mul_finish: bad arguments");
    const long total4 = (long)n_maps * (per_map / 4);
    hipLaunchKernelGGL(rel_mul_finish_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, part, nsplit, (long)n_maps * per_map, x,
                       map2img, out, per_map / 4, total4);
    return check_launch("rel_mul_finish");
}
int maxpool_guided_bwd(const float* x, const float* g_out, const int* map2img, float* g_in, int n_maps, int ho, int wo,
                       int c, int plain, unsigned* amax, hipStream_t s) {
    const long per = (long)4 * ho * wo * (c / 4);       // float4 items per map of the unpooled tensor
    const long total = (long)n_maps * per;
    const bool fused = amax && per % 2048 == 0;          // whole blocks of 8 x 256 items inside one map
    if (fused)
        hipLaunchKernelGGL((maxpool_guided_bwd_kernel<true, 8>), dim3(grid_for(total, 2048)), dim3(256), 0, s, x, g_out, map2img,
                           g_in, ho, wo, c / 4, total, plain, amax);
    else
        hipLaunchKernelGGL((maxpool_guided_bwd_kernel<false, 1>), dim3(grid_for(total)), dim3(256), 0, s, x, g_out, map2img, g_in,
                           ho, wo, c / 4, total, plain, amax);
    LRPX_TRY(check_launch("maxpool_guided_bwd"));
    if (amax && !fused) return lrpx_amax_maps(g_in, n_maps, per * 4, amax, s);   // (odd sizes: one streaming read)
    return LRPX_OK;
}
}  // namespace lrpx
}  // extern "C++"

int lrpx_gradcam(const float* feats, const float* grads, const int32_t* map2img, float* cam, int rows, int P, int C,
                 void* stream) {
    LRPX_CHECK_PTRS("lrpx_gradcam", {feats, "feats"}, {grads, "grads"}, {map2img, "map2img"}, {cam, "cam"});
    LRPX_REQUIRE(feats && grads && cam && rows > 0 && P > 0 && P <= 256 && C > 0, "gradcam: bad arguments (P <= 256)");
    hipLaunchKernelGGL(gradcam_kernel, dim3(rows), dim3(256), (C + 4) * sizeof(float), (hipStream_t)stream, feats, grads,
                       map2img, cam, P, C);
    return check_launch("gradcam");
}

int lrpx_accumulate(float* dst, const float* src, long n, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_accumulate", {dst, "dst"}, {src, "src"});
    LRPX_REQUIRE(dst && src && n > 0 && n % 4 == 0, "accumulate: bad arguments");
    hipLaunchKernelGGL(accumulate_kernel, dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, dst, src, n / 4);
    return check_launch("accumulate");
}

int lrpx_fold_halves(const float* in, float* out, long rows, int half, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_fold_halves", {in, "in"}, {out, "out"});
    LRPX_REQUIRE(in && out && rows > 0 && half > 0, "fold_halves: bad arguments");
    hipLaunchKernelGGL(fold_halves_kernel, dim3(grid_for(rows * half)), dim3(256), 0, (hipStream_t)stream, in, out, half,
                       rows * half);
    return check_launch("fold_halves");
}

int lrpx_check(const float* buf, long n, int flags, void* stream) {
    LRPX_CHECK_PTRS("lrpx_check", {buf, "buf"});
    LRPX_REQUIRE(buf && n > 0, "check: bad arguments");
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    static unsigned* dflags = nullptr;   // init-once 4-byte scratch word owned by the library
    if (!dflags && hipMalloc(&dflags, sizeof(unsigned)) != hipSuccess) {
        set_error("check: cannot allocate flag word");
        return LRPX_ELAUNCH;
    }
    hipStream_t st = (hipStream_t)stream;
    (void)hipMemsetAsync(dflags, 0, sizeof(unsigned), st);
    hipLaunchKernelGGL(check_kernel, dim3(1024), dim3(256), 0, st, buf, n, dflags);
    unsigned h = 0;
    (void)hipMemcpyAsync(&h, dflags, sizeof(unsigned), hipMemcpyDeviceToHost, st);
    if (hipStreamSynchronize(st) != hipSuccess) {
        set_error("check: stream failure: %s", hipGetErrorString(hipGetLastError()));
        return LRPX_ELAUNCH;
    }
    if ((flags & 1) && (h & 1u)) { set_error("check: NaN/Inf in relevance"); return LRPX_ENONFINITE; }
    if ((flags & 2) && !(h & 2u)) { set_error("check: relevance is all zero"); return LRPX_EZERO; }
    return LRPX_OK;
}

}  // extern "C"
