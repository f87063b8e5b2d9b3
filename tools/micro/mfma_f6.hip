// Probe of the block-scaled fp6 (e2m3) matrix-core path of gfx950 for the cross products of the split-operand convolution:
//   1. v_cvt_scalef32_2xpk16_fp6_f32: value encoding, packing order and the meaning of the scale operand;
//   2. v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 operands: which lane holds which k, what a lane's scale byte multiplies;
//   3. cycles per MFMA (one wave per SIMD, 4 independent accumulators) for fp16 K=16, fp8 K=64 and fp6 K=64, scaled or not.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_f6.hip -o gpurun_out/mfma_f6 && gpurun_out/mfma_f6
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

static float fp6_decode(unsigned c) {          // e2m3, bias 1
    const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
    const float v = e == 0 ? m * 0.125f : (1.f + m * 0.125f) * (float)(1 << (e - 1));
    return s ? -v : v;
}
static unsigned fp6_at(const unsigned* w, int i) {      // i-th 6-bit field of a little-endian bit string
    const int bit = 6 * i;
    unsigned long long two = w[bit / 32] | ((unsigned long long)(bit / 32 + 1 < 6 ? w[bit / 32 + 1] : 0) << 32);
    return (unsigned)(two >> (bit % 32)) & 63u;
}

__global__ void cvt_kernel(const float* in, unsigned* out, float sc) {
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x * 32 + i]; b[i] = in[threadIdx.x * 32 + 16 + i]; }
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, sc);
    for (int i = 0; i < 6; ++i) out[threadIdx.x * 6 + i] = r[i];
}

// one MFMA: A, B as 8 dwords per lane (fp6: the first 6), scale words per lane
__global__ void mfma_kernel(const i32x8* A, const i32x8* B, const int* sA, const int* sB, f32x16* C) {
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(A[threadIdx.x], B[threadIdx.x], acc, 2, 2, 0, sA[threadIdx.x], 0, sB[threadIdx.x]);
    C[threadIdx.x] = acc;
}

template <int KIND>
__global__ __launch_bounds__(256) void time_kernel(unsigned long long* cyc, float* sink, int iters, int sa, int sb) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = 0x11111111 * (threadIdx.x & 3); b[i] = 0x01010101 * (threadIdx.x & 7); }
    const f16x8 ha = __builtin_bit_cast(f16x8, i32x4{a[0], a[1], a[2], a[3]});
    const f16x8 hb = __builtin_bit_cast(f16x8, i32x4{b[0], b[1], b[2], b[3]});
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ha, hb, acc[j], 0, 0, 0);
            if constexpr (KIND == 1) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 0, 0, 0, 0, 0);
            if constexpr (KIND == 2) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 0, 0, 0, sa, 0, sb);
            if constexpr (KIND == 3) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 2, 2, 0, sa, 0, sb);
            if constexpr (KIND == 4) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 2, 0, 0, sa, 0, sb);
            if constexpr (KIND == 5) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 4, 4, 0, sa, 0, sb);
            if constexpr (KIND == 6) acc[j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc[j], 2, 2, 0, 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
    for (int j = 0; j < 4; ++j)
        for (int e = 0; e < 16; ++e) s += acc[j][e];
    sink[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
static void run_time(const char* name, unsigned long long* dC, float* dS) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(time_kernel<KIND>, dim3(256), dim3(256), 0, 0, dC, dS, 100, 127, 127);
    hipEventRecord(e0);
    hipLaunchKernelGGL(time_kernel<KIND>, dim3(256), dim3(256), 0, 0, dC, dS, iters, 127, 127);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c;
    hipMemcpy(&c, dC, 8, hipMemcpyDeviceToHost);
    printf("%-44s %8.1f ns per MFMA per SIMD  (%.3f ms for %d x 4; s_memrealtime ticks per MFMA %.2f)\n", name, ms * 1e6 / (iters * 4.0), ms, iters,
           (double)c / (iters * 4.0));
}

int main() {
    // ---- 1. conversion ----
    {
        std::vector<float> in(64 * 32);
        const float vals[32] = {0.f, 0.0625f, 0.1f, 0.125f, 0.19f, 0.25f, 0.8f, 0.93f, 0.95f, 1.f, 1.06f, 1.0625f, 1.1f, 1.9f, 2.f, 3.3f,
                                3.9f, 4.f, 6.9f, 7.5f, 7.8f, 9.f, 100.f, -0.3f, -1.3f, -7.4f, -30.f, 1e-8f, 5.f, 5.3f, 0.4375f, 0.06f};
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 32; ++i) in[l * 32 + i] = vals[i];
        float* dI; unsigned* dO;
        hipMalloc(&dI, in.size() * 4); hipMalloc(&dO, 64 * 6 * 4);
        hipMemcpy(dI, in.data(), in.size() * 4, hipMemcpyHostToDevice);
        for (float sc : {1.f, 4.f, 0.25f, 3.f}) {
            hipLaunchKernelGGL(cvt_kernel, dim3(1), dim3(64), 0, 0, dI, dO, sc);
            unsigned out[6];
            hipMemcpy(out, dO, sizeof(out), hipMemcpyDeviceToHost);
            printf("cvt_scalef32_2xpk16_fp6_f32, scale %.2f:\n ", sc);
            for (int i = 0; i < 32; ++i) printf(" %g->%g", vals[i], fp6_decode(fp6_at(out, i)));
            printf("\n");
        }
    }
    // ---- 2. MFMA layout / scale ----
    {
        // A[i][k], B[k][j] small integers representable in e2m3 (0, +-0.5, +-1, +-2 ...)
        const float pool[8] = {0.f, 0.5f, 1.f, 1.5f, 2.f, -1.f, -0.5f, 3.f};
        auto enc = [](float v) -> unsigned {
            for (unsigned c = 0; c < 64; ++c) if (fp6_decode(c) == v && !(c == 32)) return c;
            return 0;
        };
        std::vector<float> A(32 * 64), B(64 * 32);
        srand(3);
        for (auto& v : A) v = pool[rand() % 8];
        for (auto& v : B) v = pool[rand() % 8];
        // hypothesis H0: lane l holds row (l & 31), k = 32 * (l >> 5) + 0..31, consecutive 6-bit fields
        std::vector<unsigned> pa(64 * 8, 0), pb(64 * 8, 0);
        std::vector<int> sa(64), sb(64);
        for (int l = 0; l < 64; ++l) {
            for (int t = 0; t < 32; ++t) {
                const int k = 32 * (l >> 5) + t, bit = 6 * t;
                const unsigned long long ca = enc(A[(l & 31) * 64 + k]), cb = enc(B[k * 32 + (l & 31)]);
                pa[l * 8 + bit / 32] |= (unsigned)(ca << (bit % 32));
                if (bit % 32 > 26) pa[l * 8 + bit / 32 + 1] |= (unsigned)(ca >> (32 - bit % 32));
                pb[l * 8 + bit / 32] |= (unsigned)(cb << (bit % 32));
                if (bit % 32 > 26) pb[l * 8 + bit / 32 + 1] |= (unsigned)(cb >> (32 - bit % 32));
            }
            sa[l] = 127 + (l % 5) - 2;          // per-lane scale exponents: distinct per row and per k half
            sb[l] = 127 + (l % 3) - 1;
        }
        i32x8 *dA, *dB; int *dsa, *dsb; f32x16* dC;
        hipMalloc(&dA, 64 * 32); hipMalloc(&dB, 64 * 32); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dC, 64 * 64);
        hipMemcpy(dA, pa.data(), 64 * 32, hipMemcpyHostToDevice);
        hipMemcpy(dB, pb.data(), 64 * 32, hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 256, hipMemcpyHostToDevice);
        hipMemcpy(dsb, sb.data(), 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dC);
        float C[64 * 16];
        hipMemcpy(C, dC, sizeof(C), hipMemcpyDeviceToHost);
        // expected under H0 with scale(l) applying to row / column (l & 31), k block (l >> 5)
        double worst = 0;
        int bad = 0;
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 16; ++e) {
                const int row = (e & 3) + 8 * (e >> 2) + 4 * (l >> 5), col = l & 31;
                double want = 0;
                for (int k = 0; k < 64; ++k) {
                    const int la = row + 32 * (k / 32), lb = col + 32 * (k / 32);
                    want += (double)A[row * 64 + k] * B[k * 32 + col] * std::ldexp(1.0, sa[la] - 127) * std::ldexp(1.0, sb[lb] - 127);
                }
                const double d = std::fabs(want - C[l * 16 + e]);
                if (d > 1e-6 * (1 + std::fabs(want))) { if (bad < 6) printf("  mismatch row %d col %d: got %g want %g\n", row, col, C[l * 16 + e], want); ++bad; }
                worst = d > worst ? d : worst;
            }
        printf("MFMA fp6 x fp6, H0 (lane l: row/col l&31, k = 32*(l>>5)+t, scale byte 0 of the lane's word = 2^(s-127) of its 32 values): %d mismatches, worst |diff| %.3g\n",
               bad, worst);
    }
    // ---- 3. cycles ----
    {
        unsigned long long* dC; float* dS;
        hipMalloc(&dC, 8); hipMalloc(&dS, 256 * 256 * 4);
        run_time<0>("f16 32x32x16", dC, dS);
        run_time<1>("f8f6f4 32x32x64 fp8 x fp8 (no scale)", dC, dS);
        run_time<2>("f8f6f4 32x32x64 fp8 x fp8 scaled", dC, dS);
        run_time<3>("f8f6f4 32x32x64 fp6 x fp6 scaled", dC, dS);
        run_time<6>("f8f6f4 32x32x64 fp6 x fp6 (no scale)", dC, dS);
        run_time<4>("f8f6f4 32x32x64 fp6 x fp8 scaled", dC, dS);
        run_time<5>("f8f6f4 32x32x64 fp4 x fp4 scaled", dC, dS);
    }
    return 0;
}
