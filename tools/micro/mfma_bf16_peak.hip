// Micro-benchmark: the sustained rate of v_mfma_f32_32x32x16_bf16 on the whole chip with operands in registers - the ceiling of the
// bf16x6 kernels (conv mode 1: 6 such MFMAs per 16 of K and fp32 product) on a part whose clock follows the power the matrix cores draw.
//   arg 1: 0 = operands all zero, 1 = random finite bf16, 2 = the three planes of an EXACT 3-way split of random fp32 data and the six
//          products a2b0 a1b1 a0b2 a1b0 a0b1 a0b0 of the bf16x6 kernels (what mode 1 really multiplies)
//   arg 2: waves per SIMD (1 or 2)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/mfma_bf16_peak tools/micro/mfma_bf16_peak.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline unsigned short to_bf16(float x) { __bf16 b = (__bf16)x; return __builtin_bit_cast(unsigned short, b); }
__device__ inline float from_bf16(unsigned short u) { return __builtin_bit_cast(float, (unsigned)u << 16); }

template <int WAVES>
__global__ __launch_bounds__(64 * WAVES, 1) void k(const float* __restrict__ src, float* __restrict__ dst, int iters, int mode) {
    const int tid = threadIdx.x;
    bf16x8 a[3], b[3];
    for (int j = 0; j < 8; ++j) {
        float va = mode ? src[(tid * 8 + j) & 65535] : 0.f, vb = mode ? src[(tid * 8 + j + 7919) & 65535] : 0.f;
        for (int t = 0; t < 3; ++t) {
            const unsigned short pa = to_bf16(va), pb = to_bf16(vb);
            a[t][j] = (short)pa; b[t][j] = (short)pb;
            if (mode == 2) { va -= from_bf16(pa); vb -= from_bf16(pb); }                       // the next plane carries the residual
            else { va = src[(tid * 8 + j + 97 * (t + 1)) & 65535]; vb = src[(tid * 8 + j + 131 * (t + 1)) & 65535]; if (!mode) va = vb = 0.f; }
        }
    }
    f32x16 acc[7];
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc[i], 0, 0, 0);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 7; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
    if (s == 12345.678f) dst[tid] = s;
}

int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 2, wps = argc > 2 ? atoi(argv[2]) : 2;
    float *src, *dst;
    hipMalloc(&src, 65536 * 4); hipMalloc(&dst, 4096);
    float* h = (float*)malloc(65536 * 4);
    srand(1);
    for (int i = 0; i < 65536; ++i) { float u = 0.f; for (int q = 0; q < 12; ++q) u += rand() / (float)RAND_MAX; h[i] = u - 6.f; }   // ~N(0,1)
    hipMemcpy(src, h, 65536 * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int iters : {500, 2000, 8000}) {
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (wps == 2) hipLaunchKernelGGL(k<8>, dim3(256), dim3(512), 0, 0, src, dst, iters, mode);
            else hipLaunchKernelGGL(k<4>, dim3(256), dim3(256), 0, 0, src, dst, iters, mode);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double fl = 2.0 * 32 * 32 * 16 * 42.0 * iters * 256 * 4 * wps;
            if (rep) printf("mode %d, %d waves/SIMD, %d iterations: %.3f ms = %.0f TFLOP/s of bf16 MFMA = %.0f fp32-equivalent TFLOP/s at 6 products\n", mode, wps, iters, ms,
                            fl / ms / 1e9, fl / ms / 1e9 / 6.0);
        }
    }
    return 0;
}
