// Micro-benchmark: what does the A-operand traffic from LDS cost the matrix pipe on this (power-limited) chip, and what would a wave
// tile of TWO channel blocks (every operand read feeding two MFMAs, 14 accumulator tiles, one wave per SIMD) gain?
// The instruction mix of one K-chunk of the mode-3 relevance kernels (conv_f16x3.h): per accumulator tile 9 x v_mfma_f32_32x32x16_f16 +
// 5 x v_mfma_scale_f32_32x32x64_f8f6f4 (fp6 e2m3), operands read from an LDS tile at the kernel's addresses (80-byte pixels, 16 + 8 + 4
// byte reads for the fp6 operands) through a depth-3 operand ring, B operands resident in registers (the L2 stream is not modelled).
//   variant A: NB = 1, 8 waves per CU (2 per SIMD), NO LDS reads (operands in registers)          - the pipe alone
//   variant B: NB = 1, 8 waves per CU, one operand read per MFMA                                  - the shipped design
//   variant C: NB = 2, 4 waves per CU (1 per SIMD), one operand read per TWO MFMAs                - two channel blocks per wave
//   variant D: NB = 2, 4 waves per CU, no LDS reads                                               - (its ceiling)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_lds tools/micro/mfma_lds.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

constexpr int W = 56, PSTRIDE = 80, PITCH = W * PSTRIDE + 256, NSLOT = 6, BUFB = NSLOT * PITCH;

template <int NB, int WAVES, bool RD, int BQ = 0>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const unsigned* __restrict__ src, float* __restrict__ dst, int iters, const u32x4* __restrict__ wsrc = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
    for (int i = tid; i < 2 * BUFB / 4; i += 64 * WAVES) reinterpret_cast<unsigned*>(lds)[i] = src[i & 4095] & 0x3b3b3b3bu;   // finite fp16 / fp6
    __syncthreads();
    int abase[7];
    for (int j = 0; j < 7; ++j) { const int q = 32 * j + li; abase[j] = (q / W) * PITCH + (q % W) * PSTRIDE + lh * 16; }
    f32x16 acc[NB][7];
    for (int n = 0; n < NB; ++n) for (int i = 0; i < 7; ++i) for (int j = 0; j < 16; ++j) acc[n][i][j] = 0.f;
    f16x8 bh[NB][3]; i32x8 bm[NB][2];
    for (int n = 0; n < NB; ++n) {
        for (int t = 0; t < 3; ++t) for (int j = 0; j < 8; ++j) bh[n][t][j] = (_Float16)(float)((src[lane * 8 + j + t + 11 * n] & 255) * 0.01f);
        for (int t = 0; t < 2; ++t) for (int j = 0; j < 8; ++j) bm[n][t][j] = (int)(src[lane * 8 + j + 5 * t + 17 * n] & 0x3b3b3b3bu);
        bm[n][0][6] = 127; bm[n][1][6] = 127;
    }
    i32x8 nord[3];
    for (int d = 0; d < 3; ++d) for (int j = 0; j < 8; ++j) nord[d][j] = (int)(src[lane * 8 + j + 29 * d] & 0x3b3b3b3bu);
    for (int d = 0; d < 3; ++d) nord[d][6] = 127;
    const int e6a = 32 - 16 * lh + lh * PSTRIDE, e6b = 32 - 16 * lh + lh * (PITCH - 2 * PSTRIDE), e6c = 32 - 16 * lh;
    // BQ: the weight stream of the kernel: per tap row 7 planes of 64 lanes x 16 bytes from global memory (L2-resident here), entry
    // g + 1 loaded while entry g is multiplied
    u32x4 bq[2][7];
    const u32x4* wp = wsrc + lane;
    if constexpr (BQ != 0) {
#pragma unroll
        for (int p = 0; p < 7; ++p) { bq[0][p] = wp[p * 64]; bq[1][p] = bq[0][p]; }
    }
    for (int it = 0; it < iters; ++it) {
        auto chunk_body = [&](auto par_c, const int it) {
        constexpr int PAR0 = decltype(par_c)::value;
        const char* abuf = lds + (it & 1) * BUFB;
        constexpr int D = 3;
        auto og = [](const int k) constexpr { return k < 35 ? 0 : (k < 70 ? 1 : 2); };
        auto oj = [](const int k) constexpr { return k < 70 ? (k % 35) / 5 : (k - 70) / 4; };
        auto om = [](const int k) constexpr { return k < 70 ? (k % 35) % 5 : ((k - 70) % 4 == 0 ? 0 : (k - 70) % 4 + 1); };
        auto rd = [&](const int k) {
            const int g = og(k), j = oj(k), m = om(k);
            if constexpr (!RD) return nord[k % 3];
            if (m < 2) {
                const int t = 4 * g + 2 * m;
                const int tp = t > 8 ? 8 : t;
                const char* a6 = abuf + abase[j] + (t == 2 ? e6b : (t == 8 ? e6c : e6a)) + (tp / 3) * PITCH + (tp % 3) * PSTRIDE;
                const u32x4 x0 = *reinterpret_cast<const u32x4*>(a6);
                const u32x2 x1 = *reinterpret_cast<const u32x2*>(a6 + 16);
                const unsigned xs = *reinterpret_cast<const unsigned*>(a6 + 28);
                return i32x8{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], (int)x1[0], (int)x1[1], (int)(xs & 0x7f) | 64, 0};
            }
            const u32x4 x0 = *reinterpret_cast<const u32x4*>(abuf + abase[j] + g * PITCH + (m - 2) * PSTRIDE);
            return i32x8{(int)x0[0], (int)x0[1], (int)x0[2], (int)x0[3], 0, 0, 0, 0};
        };
        i32x8 ring[D];
#pragma unroll
        for (int d = 0; d < D; ++d) ring[d] = rd(d);
        constexpr int K0[3] = {0, 35, 70}, KN[3] = {35, 35, 28};
#pragma unroll
        for (int g = 0; g < 3; ++g) {
            // which queue entry is current: BQ 1 rotates (entry 0 always current, moves behind the row); BQ 2 alternates with static
            // indices (the step parity: `it` advances by 2 per loop trip in that variant, see the loop increment)
            constexpr int dummy = 0; (void)dummy;
            const int s_par = (BQ == 2) ? ((PAR0 * 3 + g) & 1) : 0;
            if constexpr (BQ != 0) {
                const long e = ((long)(it * 3 + g + 1) % 96) * 7;
#pragma unroll
                for (int p = 0; p < 7; ++p) bq[BQ == 2 ? (s_par ^ 1) : 1][p] = wp[(e + p) * 64];
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const u32x4* c = bq[s_par];
                    bh[n][0] = __builtin_bit_cast(f16x8, c[0]); bh[n][1] = __builtin_bit_cast(f16x8, c[1]); bh[n][2] = __builtin_bit_cast(f16x8, c[2]);
                    bm[n][0] = i32x8{(int)c[3][0], (int)c[3][1], (int)c[3][2], (int)c[3][3], (int)c[4][0], (int)c[4][1], 127, 0};
                    bm[n][1] = i32x8{(int)c[5][0], (int)c[5][1], (int)c[5][2], (int)c[5][3], (int)c[6][0], (int)c[6][1], 127, 0};
                }
            }
#pragma unroll
            for (int k2 = 0; k2 < KN[g]; ++k2) {
                const int kk = K0[g] + k2;
                const int j = oj(kk), m = om(kk);
                const i32x8 cur = ring[kk % D];
                if (kk + D < 98) { ring[kk % D] = rd(kk + D); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    if (m < 2) acc[n][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(cur, bm[n][m], acc[n][j], 2, 2, 0, cur[6], 0, bm[n][m][6]);
                    else {
                        const u32x4 c4 = {(unsigned)cur[0], (unsigned)cur[1], (unsigned)cur[2], (unsigned)cur[3]};
                        acc[n][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, c4), bh[n][m - 2], acc[n][j], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (BQ == 1) {
#pragma unroll
                for (int p = 0; p < 7; ++p) bq[0][p] = bq[1][p];
            }
        }
        };      // chunk_body
        if constexpr (BQ == 2) { chunk_body(std::integral_constant<int, 0>{}, it); chunk_body(std::integral_constant<int, 1>{}, it + 1); ++it; }
        else chunk_body(std::integral_constant<int, 0>{}, it);
    }
    float s = 0.f;
    for (int n = 0; n < NB; ++n) for (int i = 0; i < 7; ++i) for (int j = 0; j < 16; ++j) s += acc[n][i][j];
    dst[blockIdx.x * 64 * WAVES + threadIdx.x] = s;
}

template <int NB, int WAVES, bool RD, int BQ = 0> double run(const unsigned* src, float* dst, int iters, const char* name, const u32x4* w = nullptr) {
    auto kern = k<NB, WAVES, RD, BQ>;
    const int ldsb = 2 * BUFB;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, ldsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * 4;          // 4 rounds of one workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), ldsb, 0, src, dst, 10, w);
    hipDeviceSynchronize();
    double best = 1e30, tot = 0;
    for (int rep = 0; rep < 8; ++rep) {            // ~ sustained: 8 launches back to back
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), ldsb, 0, src, dst, iters, w);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 2) { tot += ms; if (ms < best) best = ms; }
    }
    const double ms = tot / 6;
    // algorithmic flop: 9 taps x 32x32x16 x 2 per (tile, block, chunk)
    const double alg = (double)grid * WAVES * NB * 7 * iters * 9 * 32768.0 / (ms * 1e-3) / 1e12;
    printf("%-58s %8.3f ms  %7.0f algorithmic TFLOP/s (%.0f fp16-equivalent executed)\n", name, ms, alg, alg * 1.5);
    hipError_t e = hipGetLastError(); if (e != hipSuccess) printf("  error: %s\n", hipGetErrorString(e));
    return alg;
}

int main() {
    unsigned* src; float* dst;
    hipMalloc(&src, 4096 * 4 + 4096); hipMalloc(&dst, 1024 * 512 * 4);
    unsigned h[4096 + 1024];
    srand(1);
    for (int i = 0; i < 4096 + 1024; ++i) h[i] = (unsigned)rand() * 2654435761u;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    u32x4* w; hipMalloc(&w, 96 * 7 * 64 * 16 + 4096);
    { unsigned* hw = (unsigned*)malloc(96 * 7 * 64 * 16); for (int i = 0; i < 96 * 7 * 64 * 4; ++i) hw[i] = ((unsigned)rand() * 2654435761u) & 0x3b3b3b3bu; hipMemcpy(w, hw, 96 * 7 * 64 * 16, hipMemcpyHostToDevice); free(hw); }
    const int iters = 600;
    for (int rep = 0; rep < 2; ++rep) {
        run<1, 8, false>(src, dst, iters, "A: 1 block / wave, 8 waves / CU, operands in registers");
        run<1, 8, true>(src, dst, iters, "B: 1 block / wave, 8 waves / CU, LDS read per MFMA (shipped)");
        run<2, 4, true>(src, dst, iters, "C: 2 blocks / wave, 4 waves / CU, LDS read per 2 MFMAs");
        run<2, 4, false>(src, dst, iters, "D: 2 blocks / wave, 4 waves / CU, operands in registers");
        run<1, 4, true>(src, dst, 2 * iters, "E: 1 block / wave, 4 waves / CU (1 per SIMD), LDS read per MFMA");
        run<1, 8, true, 1>(src, dst, iters, "F: as B + the weight stream from L2, queue ROTATED by moves (shipped)", w);
        run<1, 8, true, 2>(src, dst, iters, "G: as B + the weight stream from L2, queue double-buffered (no moves)", w);
    }
    return 0;
}
