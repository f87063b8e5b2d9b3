// What does the store phase at the end of a workgroup cost?  256 x R workgroups of 512 threads, launch_bounds(512, 1) with 100 KB of
// LDS (one workgroup per CU, like the 8-wave relevance kernels): each spins on the matrix cores for `busy` MFMAs per wave, then writes
// its 224 pixels x 256 channels x 4 bytes (229 KB) in one of several ways, then ends.  Reported: time per workgroup "round" minus the
// round without stores = what the stores (issue + drain behind s_endpgm + the dispatch of the next workgroup) add per tile.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/store_drain.hip -o /tmp/store_drain && /tmp/store_drain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0 none, 1 dword (lane = channel, 16 pixel rows per tile), 2 float4 (lane = 4 channels), 3 float4 nontemporal, 4 dword nontemporal
__global__ __launch_bounds__(512, 1) void k(float* out, int busy, int rounds_unused) {
    extern __shared__ char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc[7];
    for (int j = 0; j < 7; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(lane & 3); b[i] = (_Float16)(wave & 1); }
    for (int it = 0; it < busy; ++it)
#pragma unroll
        for (int j = 0; j < 7; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    if (lds[threadIdx.x] == 77) acc[0][0] += 1.f;           // (keeps the LDS allocation)
    const long tile = blockIdx.x;
    float* base = out + tile * (224L * 256);
    const int li = lane & 31, lh = lane >> 5;
    if (MODE == 1 || MODE == 4) {
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                float* p = base + (32 * j + (e & 3) + 8 * (e >> 2) + 4 * lh) * 256L + wave * 32 + li;
                if (MODE == 4) __builtin_nontemporal_store(acc[j][e], p); else *p = acc[j][e];
            }
    } else if (MODE == 2 || MODE == 3) {
        const int qd = lane & 7, r0 = lane >> 3;
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                f32x4 v = {acc[j][4 * kq], acc[j][4 * kq + 1], acc[j][4 * kq + 2], acc[j][4 * kq + 3]};
                f32x4* p = reinterpret_cast<f32x4*>(base + (32 * j + r0 + 8 * kq) * 256L + wave * 32 + 4 * qd);
                if (MODE == 3) __builtin_nontemporal_store(v, p); else *p = v;
            }
    } else {
        float s = 0.f;
        for (int j = 0; j < 7; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
        if (s == 1.2345e-30f) base[threadIdx.x] = s;
    }
}

template <int MODE>
static float run(float* d, int busy, int rounds) {
    auto kern = k<MODE>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 100 * 1024, 0, d, busy, rounds);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256 * rounds), dim3(512), 100 * 1024, 0, d, busy, rounds);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1000.f / rounds;          // us per round of 256 workgroups
}

int main() {
    const int rounds = 16;
    float* d; hipMalloc(&d, 256L * rounds * 224 * 256 * 4);
    for (int busy : {250, 500}) {
        const float t0 = run<0>(d, busy, rounds), t1 = run<1>(d, busy, rounds), t4 = run<4>(d, busy, rounds), t2 = run<2>(d, busy, rounds), t3 = run<3>(d, busy, rounds);
        printf("busy %4d MFMA x 7 per wave: round without stores %6.1f us | + dword stores %5.1f us | + dword nt %5.1f | + float4 %5.1f | + float4 nt %5.1f   (229 KB per workgroup, 60 MB per round)\n",
               busy, t0, t1 - t0, t4 - t0, t2 - t0, t3 - t0);
    }
    return 0;
}
