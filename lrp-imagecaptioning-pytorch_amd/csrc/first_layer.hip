// Relevance rule of the FIRST VGG16 conv (3 signed input channels <- 64 channels at 224x224):
//   R[n][c][y][x] = x+[c] * convT(S, W+)[c] + x-[c] * convT(S, W-)[c]        (LRPtools/lrp_modules.py:124-150)
// Only 6 output values per pixel (K = 64*9 = 576 each): a 32-wide MFMA tile would waste 26/32 of its work, so
// this layer is a direct VALU convolution: one thread per output pixel, the S halo tile staged through LDS per
// 16-channel chunk, the 3456 weights read through the scalar cache (wave-uniform addresses).
// HBM: reads S once (12.8 MB per map) — this kernel sits at the HBM/VALU balance point.
#include "common.h"
#include "conv_mfma.h"

namespace lrpx {

constexpr int FL_TH = 8, FL_TW = 32, FL_KC = 16, FL_STRIDE = FL_KC + 4;
constexpr int FL_HP = FL_TH + 2, FL_WP = FL_TW + 2;

// w6: [chunk 4][tap 9][ch 16][6] = {W+ for c=0..2, W- for c=0..2} with the kernel flipped (transposed conv)
__global__ void pack_first_layer_kernel(const float* __restrict__ w, float* __restrict__ w6, int cout, int plain) {
    int idx = blockIdx.x * blockDim.x + threadIdx.x;     // over cout*9*6
    if (idx >= cout * 9 * 6) return;
    const int o = idx % 6;
    int r = idx / 6;
    const int ch = r % FL_KC; r /= FL_KC;
    const int tap = r % 9;
    const int chunk = r / 9;
    const int co = chunk * FL_KC + ch, c = o % 3;
    const float x = w[((long)co * 3 + c) * 9 + (8 - tap)];
    if (plain) w6[idx] = o < 3 ? x : 0.f;      // plain transposed conv (guided backprop): second triple unused
    else w6[idx] = o < 3 ? fmaxf(x, 0.f) : fminf(x, 0.f);
}

__global__ __launch_bounds__(256) void first_layer_rel_kernel(const float* __restrict__ S, const float* __restrict__ w6,
                                                              const float* __restrict__ X8,
                                                              const int* __restrict__ map2img, float* __restrict__ out,
                                                              int cin, int plain, long chunk_stride) {
    constexpr int HW = 224;
    __shared__ __attribute__((aligned(16))) float lds[FL_HP * FL_WP * FL_STRIDE];
    const int n = blockIdx.z, ty0 = blockIdx.y * FL_TH, tx0 = blockIdx.x * FL_TW;
    const int tid = threadIdx.x, ly = tid >> 5, lx = tid & 31;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nchunk = cin / FL_KC;
    // staging items of this thread: (halo pixel, 16-byte segment) -> LDS offset and global element offset (without the
    // chunk term), fixed for all chunks.  The loads of chunk c+1 are in flight while chunk c is computed: one load at a
    // time with a wait after each (the previous version) left the kernel bound by global-load latency.
    constexpr int NIT = (FL_HP * FL_WP * (FL_KC / 4) + 255) / 256;
    int lds_off[NIT];
    long g_off[NIT];            // < 0: outside the image (stays zero) or no item
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        const int it = tid + k * 256;
        const int seg = it & 3, p = it >> 2;
        const int py = p / FL_WP, px = p - py * FL_WP;
        const int gy = ty0 + py - 1, gx = tx0 + px - 1;
        lds_off[k] = p * FL_STRIDE + seg * 4;
        const bool in_tile = it < FL_HP * FL_WP * (FL_KC / 4);
        const bool in_img = gy >= 0 && gy < HW && gx >= 0 && gx < HW;
        const long pix = ((long)n * HW + gy) * HW + gx;
        // channel-chunked S [cin/16][n_maps*HW*HW][16]: contiguous 64-byte runs per pixel; else NHWC
        g_off[k] = !in_tile ? -2 : (!in_img ? -1 : (chunk_stride ? pix * FL_KC + seg * 4 : pix * cin + seg * 4));
    }
    const long cstep = chunk_stride ? chunk_stride : FL_KC;
    f32x4 stage[NIT];
#pragma unroll
    for (int k = 0; k < NIT; ++k) {
        stage[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (g_off[k] >= 0) stage[k] = *reinterpret_cast<const f32x4*>(S + g_off[k]);
    }
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        __syncthreads();                              // the previous chunk has been consumed
#pragma unroll
        for (int k = 0; k < NIT; ++k)
            if (g_off[k] >= -1) *reinterpret_cast<f32x4*>(lds + lds_off[k]) = stage[k];
        __syncthreads();
        if (chunk + 1 < nchunk) {
#pragma unroll
            for (int k = 0; k < NIT; ++k)
                if (g_off[k] >= 0) stage[k] = *reinterpret_cast<const f32x4*>(S + g_off[k] + (long)(chunk + 1) * cstep);
        }
        const float* wc = w6 + chunk * 9 * FL_KC * 6;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* sp = lds + ((ly + tap / 3) * FL_WP + lx + tap % 3) * FL_STRIDE;
#pragma unroll
            for (int c4 = 0; c4 < FL_KC / 4; ++c4) {
                const f32x4 s = *reinterpret_cast<const f32x4*>(sp + c4 * 4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float* wk = wc + (tap * FL_KC + c4 * 4 + e) * 6;   // wave-uniform -> scalar loads
#pragma unroll
                    for (int o = 0; o < 6; ++o) acc[o] += s[e] * wk[o];
                }
            }
        }
    }
    const int y = ty0 + ly, x = tx0 + lx;
    const long img = map2img ? map2img[n] : n;
    const long p = (long)y * HW + x;
    if (plain) {     // image gradient: no input multiplication
#pragma unroll
        for (int c = 0; c < 3; ++c) out[((long)n * 3 + c) * HW * HW + p] = acc[c];
        return;
    }
    const float* xp = X8 + (img * HW * HW + p) * 8;          // [x+ (3) | x- (3) | 0 0]
    const f32x4 xa = *reinterpret_cast<const f32x4*>(xp), xb = *reinterpret_cast<const f32x4*>(xp + 4);
    const float xpos[3] = {xa[0], xa[1], xa[2]}, xneg[3] = {xa[3], xb[0], xb[1]};
#pragma unroll
    for (int c = 0; c < 3; ++c) out[((long)n * 3 + c) * HW * HW + p] = xpos[c] * acc[c] + xneg[c] * acc[3 + c];
}

int first_layer_pack(const float* w, float* w6, int cout, int plain, hipStream_t s) {
    hipLaunchKernelGGL(pack_first_layer_kernel, dim3((cout * 54 + 255) / 256), dim3(256), 0, s, w, w6, cout, plain);
    return check_launch("pack_first_layer");
}

int first_layer_relevance(const float* S, const float* w6, const float* X8, const int* map2img, float* out, int n_maps,
                          int cin, int plain, int s_chunked, hipStream_t s) {
    const long chunk_stride = s_chunked ? (long)n_maps * 224 * 224 * FL_KC : 0;
    hipLaunchKernelGGL(first_layer_rel_kernel, dim3(224 / FL_TW, 224 / FL_TH, n_maps), dim3(256), 0, s, S, w6, X8,
                       map2img, out, cin, plain, chunk_stride);
    return check_launch("first_layer_relevance");
}

}  // namespace lrpx
