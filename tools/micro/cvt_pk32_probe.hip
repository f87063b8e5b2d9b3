// Probe of v_cvt_scalef32_pk32_fp6_f16 (32 fp16 -> 32 fp6 e2m3): element i -> field i, divides by 2^exponent(scale), round to nearest even,
// saturates at 7.5.   hipcc --offload-arch=gfx950 -O2 tools/micro/cvt_pk32_probe.hip -o tools/micro/cvt_pk32_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
static float fp6_decode(unsigned c) { const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7; const float v = e == 0 ? m * 0.125f : (1.f + m * 0.125f) * (float)(1 << (e - 1)); return s ? -v : v; }
static unsigned fp6_at(const unsigned* w, int i) { const int bit = 6 * i; unsigned long long two = w[bit / 32] | ((unsigned long long)(bit / 32 + 1 < 6 ? w[bit / 32 + 1] : 0) << 32); return (unsigned)(two >> (bit % 32)) & 63u; }
__global__ void k(const _Float16* in, unsigned* out, float sc) {
    f16x32 a;
    for (int i = 0; i < 32; ++i) a[i] = in[i];
    u32x6 r;
    asm("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=&v"(r) : "v"(a), "v"(sc));
    for (int i = 0; i < 6; ++i) out[i] = r[i];
}
int main() {
    _Float16 h[32]; for (int i = 0; i < 32; ++i) h[i] = (_Float16)(0.25f * i - 2.f);
    _Float16* d; unsigned* o; hipMalloc(&d, 64); hipMalloc(&o, 24); hipMemcpy(d, h, 64, hipMemcpyHostToDevice);
    for (float sc : {1.f, 0.5f}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, sc);
        unsigned out[6]; hipMemcpy(out, o, 24, hipMemcpyDeviceToHost);
        printf("scale %.2f:", sc); for (int i = 0; i < 32; ++i) printf(" %g->%g", (float)h[i], fp6_decode(fp6_at(out, i))); printf("\n");
    }
    return 0;
}
