// Micro-benchmark: how long does one K-chunk of the conv kernel's MFMA phase take on the whole chip when the two
// cross products run as fp8 K=64 MFMAs instead of fp16 K=16?  (7 accumulator tiles per wave, random operands:
// the sustained clock depends on the data.)
//   variant 0: 27 x v_mfma_f32_32x32x16_f16 per chunk per tile        (f16x3 as shipped)
//   variant 1:  9 x f16  + 6 x v_mfma_f32_32x32x64_f8f6f4 (fp8 e4m3)  (hi.hi in fp16, cross terms in fp8, 12 tap slots)
//   variant 2:  9 x f16  only                                          (floor)
//   variant 3: 18 x f16 + 5 x fp8 per 2 chunks (2 taps x 32 channels)  (per-chunk cost = half)
//   variant 4:  9 x f16  + 5 x fp8                                     (as shipped: 18 cross-term slices in 5 MFMAs)
//   variant 5:  9 x f16  + 5 x fp6 e2m3 (same instruction, cbsz = blgp = 2: 4x the fp16 rate)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_mix tools/micro/mfma_mix.hip   (run: ./tools/micro/mfma_mix)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <int VAR>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ dst, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[7];
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    f16x8 a[7], b;
    i32x8 a8[7], b8;
    for (int i = 0; i < 7; ++i) {
        for (int j = 0; j < 8; ++j) { a[i][j] = (_Float16)src[(lane * 7 + i) * 8 + j]; a8[i][j] = __float_as_int(src[lane * 56 + i * 8 + j]) & 0x3f3f3f3f; }
    }
    for (int j = 0; j < 8; ++j) { b[j] = (_Float16)src[lane * 8 + j + 1]; b8[j] = __float_as_int(src[lane * 8 + j + 3]) & 0x3f3f3f3f; }
    for (int it = 0; it < iters; ++it) {
        constexpr int NF16 = VAR == 0 ? 27 : (VAR == 3 ? 18 : 9);
        constexpr int NF8 = VAR == 1 ? 6 : ((VAR == 3 || VAR == 4) ? 5 : 0);
        constexpr int NF6 = VAR == 5 ? 5 : 0;
#pragma unroll
        for (int t = 0; t < NF16; ++t)
#pragma unroll
            for (int i = 0; i < 7; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b, acc[i], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NF8; ++t)
#pragma unroll
            for (int i = 0; i < 7; ++i)
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8, acc[i], 0, 0, 0, 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NF6; ++t)
#pragma unroll
            for (int i = 0; i < 7; ++i)
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i], b8, acc[i], 2, 2, 0, 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    dst[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VAR> float run(const float* src, float* dst, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<VAR>, dim3(512), dim3(256), 0, 0, src, dst, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<VAR>, dim3(512), dim3(256), 0, 0, src, dst, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float *src, *dst;
    hipMalloc(&src, 64 * 64 * 4); hipMalloc(&dst, 512 * 256 * 4);
    float h[64 * 64];
    srand(1);
    for (int i = 0; i < 64 * 64; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * 4.f;
    hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice);
    const int iters = 4000;     // chunks
    for (int rep = 0; rep < 2; ++rep) {
        const float t0 = run<0>(src, dst, iters), t1 = run<1>(src, dst, iters), t2 = run<2>(src, dst, iters), t3 = run<3>(src, dst, iters / 2);
        // cycles per chunk per wave-tile group if the pipe were alone: 2 waves per SIMD share it
        printf("27 f16: %.3f ms   9 f16 + 6 fp8: %.3f ms (x%.2f)   9 f16: %.3f ms   (18 f16 + 5 fp8)/2: %.3f ms (x%.2f)\n", t0, t1, t0 / t1, t2, t3, t0 / t3);
        const float t4 = run<4>(src, dst, iters), t5 = run<5>(src, dst, iters);
        printf("   9 f16 + 5 fp8: %.3f ms   9 f16 + 5 fp6: %.3f ms (x%.2f of the fp8 mix)\n", t4, t5, t4 / t5);
        const double mf = 512.0 * 4 * iters * 7;   // waves * chunks * tiles
        printf("   f16 rate in variant 0: %.0f TFLOP/s\n", mf * 27 * 32768 / (t0 * 1e-3) / 1e12);
        // the shipped mix (variant 4) as the conv kernels count it: algorithmic = 9 x 32768 flop per (tile, chunk); executed
        // fp16-equivalent = 9 f16 MFMAs + 5 fp8 MFMAs at half weight = 19 x 32768
        printf("   variant 4 (shipped f16+f8 mix), pipe alone: %.0f algorithmic TFLOP/s = %.0f fp16-equivalent TFLOP/s\n",
               mf * 9 * 32768 / (t4 * 1e-3) / 1e12, mf * 19 * 32768 / (t4 * 1e-3) / 1e12);
    }
    {   // sustained: the shipped mix back to back for ~5 s (power-limited clock)
        const double mf = 512.0 * 4 * iters * 7;
        double tot = 0; int n = 0; float last = 0;
        while (tot < 5000.0) { last = run<4>(src, dst, iters); tot += last; ++n; }
        printf("sustained variant 4 over %.1f s: last launch %.3f ms = %.0f algorithmic TFLOP/s = %.0f fp16-equivalent TFLOP/s\n",
               tot * 1e-3, last, mf * 9 * 32768 / (last * 1e-3) / 1e12, mf * 19 * 32768 / (last * 1e-3) / 1e12);
    }
    float o; hipMemcpy(&o, dst, 4, hipMemcpyDeviceToHost); printf("checksum %g\n", o);
    return 0;
}
