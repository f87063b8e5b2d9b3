// Micro-benchmark: the sustained rate of v_mfma_f32_32x32x16_f16 on the whole chip with operands in registers - the ceiling of the
// dense f16x3 kernels (3 such MFMAs per 16 of K) on a part whose clock follows the power the matrix cores draw.
//   arg 1: 0 = operands all zero, 1 = random finite fp16 (what a GEMM of real data toggles), 2 = random, hi/lo split pattern (small lo halves)
//   arg 2: waves per SIMD (1 or 2)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_f16_peak tools/micro/mfma_f16_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const float* __restrict__ src, float* __restrict__ dst, int iters, int mode) {
    const int tid = threadIdx.x;
    f16x8 a[4], b[4];
    for (int t = 0; t < 4; ++t)
        for (int j = 0; j < 8; ++j) {
            const float va = mode ? src[(tid * 8 + j + 97 * t) & 65535] : 0.f, vb = mode ? src[(tid * 8 + j + 131 * t + 7) & 65535] : 0.f;
            // mode 2: tiles 2, 3 carry the residuals of a split (11 bits smaller), as the lo halves of the f16x3 kernels do
            a[t][j] = (_Float16)((mode == 2 && t >= 2) ? va * 4.8e-4f : va);
            b[t][j] = (_Float16)((mode == 2 && t >= 2) ? vb * 4.8e-4f : vb);
        }
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(i + r) & 3], b[(i >> 1) & 3], acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) dst[tid] = s;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 1, wps = argc > 2 ? atoi(argv[2]) : 2;
    float *src, *dst;
    hipMalloc(&src, 65536 * 4); hipMalloc(&dst, 4096);
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) { float u = 0.f; for (int q = 0; q < 12; ++q) u += rand() / (float)RAND_MAX; h[i] = u - 6.f; }   // ~N(0,1)
    hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {2000, 8000, 32000}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (wps == 2) hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 0, 0, src, dst, iters, mode);
            else hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, src, dst, iters, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = 2.0 * 32 * 32 * 16 * 32.0 * iters * 256 * 4 * wps;
            if (rep) printf("mode %d, %d waves/SIMD, %d iterations: %.3f ms = %.0f TFLOP/s of fp16 MFMA = %.2f GHz at one MFMA per 32 cycles\n", mode, wps, iters, ms,
                            fl / ms / 1e9, fl / ms / 1e9 * 1e12 / (1024.0 * 1024 * 1e9) / 1.0);
        }
    }
    return 0;
}
