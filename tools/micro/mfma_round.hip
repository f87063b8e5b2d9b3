// How does v_mfma_f32_32x32x16_f16 round its fp32 accumulation?  One wave chains N MFMAs into the same accumulator (the
// K loop of conv_f16x3.h) on random fp16 operands; the host evaluates the same sums in fp64.  Printed per chain length:
// rms and MEAN SIGNED error of the results in units of ulp(|result|)-free relative terms - a mean far from zero means the
// adds truncate (errors grow ~N), a zero mean with rms ~ sqrt(N) means round-to-nearest.  Variants: all-positive operands
// (the Z+ half of the forward trace) and signed operands; sequential chain against 8 blocks of N/8 summed at the end.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_round.hip -o gpurun_out/mfma_round && gpurun_out/mfma_round
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A: [n][32 rows][16 k] fp16, B: [n][16 k][32 cols]; out [32][32]
__global__ void chain(const _Float16* A, const _Float16* B, float* out, int n, int blocks) {
    const int lane = threadIdx.x, li = lane & 31, lh = lane >> 5;
    f32x16 total;
    for (int e = 0; e < 16; ++e) total[e] = 0.f;
    const int per = n / blocks;
    for (int b = 0; b < blocks; ++b) {
        f32x16 acc;
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        for (int s = b * per; s < (b + 1) * per; ++s) {
            f16x8 a, w;
            for (int j = 0; j < 8; ++j) {
                a[j] = A[((long)s * 32 + li) * 16 + lh * 8 + j];        // A operand: row li, k = 8*lh + j
                w[j] = B[((long)s * 16 + lh * 8 + j) * 32 + li];        // B operand: k = 8*lh + j, col li
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, w, acc, 0, 0, 0);
        }
        for (int e = 0; e < 16; ++e) total[e] = (blocks == 1) ? acc[e] : total[e] + acc[e];
    }
    for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;       // C layout of the 32x32 MFMA: lane -> column li, rows by e
        out[row * 32 + li] = total[e];
    }
}

int main() {
    const int NMAX = 1024;
    std::vector<_Float16> hA((size_t)NMAX * 32 * 16), hB((size_t)NMAX * 16 * 32);
    _Float16 *dA, *dB; float* dO;
    hipMalloc(&dA, hA.size() * 2); hipMalloc(&dB, hB.size() * 2); hipMalloc(&dO, 32 * 32 * 4);
    for (int positive = 0; positive < 2; ++positive) {
        srand(7);
        for (auto& v : hA) { float x = (float)rand() / RAND_MAX; v = (_Float16)(positive ? x : 2 * x - 1); }
        for (auto& v : hB) { float x = (float)rand() / RAND_MAX; v = (_Float16)(positive ? x : 2 * x - 1); }
        hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice);
        hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice);
        for (int n : {8, 36, 72, 144, 288, 864}) {
            for (int blocks : {1, 8}) {
                if (n % blocks) continue;
                hipLaunchKernelGGL(chain, dim3(1), dim3(64), 0, 0, dA, dB, dO, n, blocks);
                float out[32 * 32];
                hipMemcpy(out, dO, sizeof(out), hipMemcpyDeviceToHost);
                double se = 0, s2 = 0, l2 = 0;
                for (int r = 0; r < 32; ++r)
                    for (int c = 0; c < 32; ++c) {
                        double ref = 0, mag = 0;
                        for (int s = 0; s < n; ++s)
                            for (int k = 0; k < 16; ++k) {
                                const double p = (double)(float)hA[((size_t)s * 32 + r) * 16 + k] * (double)(float)hB[((size_t)s * 16 + k) * 32 + c];
                                ref += p; mag += fabs(p);
                            }
                        const double scale = positive ? ref : sqrt((double)n * 16) / 3.0;     // typical |sum|
                        const double e = (out[r * 32 + c] - ref) / scale;
                        se += e; s2 += e * e; l2 += 1;
                    }
                printf("%s  n=%4d MFMAs (K=%5d) %s: rms %.3e  mean %+.3e  (2^-24 = 5.96e-8)\n", positive ? "positive" : "signed  ", n, n * 16,
                       blocks == 1 ? "one chain   " : "8 blocks    ", sqrt(s2 / l2), se / l2);
            }
        }
    }
    return 0;
}
