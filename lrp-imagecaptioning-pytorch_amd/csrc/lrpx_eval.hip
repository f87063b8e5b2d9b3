// Device-side consumers of the relevance maps: the reductions the reference's evaluation
// experiments run on every map after copying it to the host (evaluation.py) - here the maps never leave HBM.
//   spatial reduce      evaluation.py:124-134 (mean over channels), :410-412 (mean of the positive / negative part)
//   _project_maxabs     evaluation.py:338-343
//   block_image         evaluation.py:57-80   (8x8 patch sums, the k most relevant patches are masked out)
//   _calculate_overlaped_pixels   evaluation.py:313-336 (share of the thresholded relevance inside a bounding box)
//   tpfp statistics     evaluation.py:506-513 (mean, mean |x|, mean of the positives, max, quantiles)
//   heat maps           LRPtools/utils.py:67-145 `gamma` + `heatmap` (+ `project`), as the explainers' visualize_explanations
//                       call them (models/gridTDmodel.py:1196-1198): (N,C,H,W) relevance -> (N,H,W,3) colours
// All HBM-bound single-pass kernels: one workgroup per map (maps are 50 176 pixels), block reductions in LDS.
#include <math.h>

#include <hipcub/hipcub.hpp>

#include "common.h"

namespace lrpx {

typedef float f32x4e __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// 256-thread blocks; red: >= 4 floats of LDS
__device__ __forceinline__ float bsum(float v, float* red) {
    v = wsum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}
__device__ __forceinline__ float bmax(float v, float* red) {
    v = wmax(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// (n, c, hw) -> (n, hw): mode 0 mean_c(x), 1 mean_c(max(x, 0)), 2 mean_c(max(-x, 0))
__global__ void spatial_reduce_kernel(const float* __restrict__ maps, int c, long hw, int mode, float* __restrict__ out,
                                      long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over n*hw
    if (idx >= total) return;
    const long n = idx / hw, p = idx - n * hw;
    float s = 0.f;
    for (int k = 0; k < c; ++k) {
        float v = maps[(n * c + k) * hw + p];
        if (mode == 1) v = fmaxf(v, 0.f);
        else if (mode == 2) v = fmaxf(-v, 0.f);
        s += v;
    }
    out[idx] = s / (float)c;
}

__global__ __launch_bounds__(256) void project_maxabs_kernel(float* __restrict__ x, long per) {
    __shared__ float red[4];
    float* xm = x + (long)blockIdx.x * per;
    float m = 0.f;
    for (long i = threadIdx.x; i < per; i += 256) m = fmaxf(m, fabsf(xm[i]));
    m = bmax(m, red);
    if (m == 0.f) return;                  // `np.zeros(x.shape)`: the map is all zeros already
    for (long i = threadIdx.x; i < per; i += 256) xm[i] = xm[i] / m;
}

// one workgroup per map: patch sums in LDS, k rounds of arg-max (first index wins), then the mask
__global__ __launch_bounds__(256) void patch_mask_kernel(const float* __restrict__ sp, int h, int w, int patch, int k,
                                                         float* __restrict__ mask) {
    extern __shared__ float sm[];          // [np] patch sums, [np] selected flag, [4] + [4] reduction scratch
    const int nph = h / patch, npw = w / patch, np_ = nph * npw, tid = threadIdx.x;
    float* psum = sm;
    float* sel = sm + np_;
    float* redv = sel + np_;
    int* redi = reinterpret_cast<int*>(redv + 4);
    const float* s = sp + (long)blockIdx.x * h * w;
    for (int q = tid; q < np_; q += 256) {
        const int py = q / npw, px = q - py * npw;
        float a = 0.f;
        for (int y = 0; y < patch; ++y)
            for (int x = 0; x < patch; ++x) a += s[(long)(py * patch + y) * w + px * patch + x];
        psum[q] = a; sel[q] = 0.f;
    }
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int q = tid; q < np_; q += 256)
            if (sel[q] == 0.f && (psum[q] > bv)) { bv = psum[q]; bi = q; }
        // wave arg-max (larger value, then smaller index), then across the 4 waves
#pragma unroll
        for (int o = 32; o; o >>= 1) {
            const float ov = __shfl_xor(bv, o, 64);
            const int oi = __shfl_xor(bi, o, 64);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        __syncthreads();
        if ((tid & 63) == 0) { redv[tid >> 6] = bv; redi[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            float v = redv[0]; int i = redi[0];
            for (int j = 1; j < 4; ++j)
                if (redv[j] > v || (redv[j] == v && redi[j] < i)) { v = redv[j]; i = redi[j]; }
            if (i != 0x7fffffff) sel[i] = 1.f;
        }
        __syncthreads();
    }
    float* m = mask + (long)blockIdx.x * h * w;
    for (int i = tid; i < h * w; i += 256) {
        const int y = i / w, x = i - y * w;
        m[i] = sel[(y / patch) * npw + x / patch] != 0.f ? 0.f : 1.f;
    }
}

// ratio[n][j] = sum(rel[rel > thr_j] inside box) / sum(rel[rel > thr_j]), 0 if the total is 0, capped at 1
__global__ __launch_bounds__(256) void bbox_ratio_kernel(const float* __restrict__ sp, int h, int w,
                                                         const int* __restrict__ boxes, const float* __restrict__ thr,
                                                         int nthr, float* __restrict__ out) {
    __shared__ float red[4];
    const float* s = sp + (long)blockIdx.x * h * w;
    const int x0 = boxes[blockIdx.x * 4], y0 = boxes[blockIdx.x * 4 + 1], x1 = boxes[blockIdx.x * 4 + 2],
              y1 = boxes[blockIdx.x * 4 + 3];
    for (int j = 0; j < nthr; ++j) {
        const float t = thr[j];
        float tot = 0.f, ins = 0.f;
        for (int i = threadIdx.x; i < h * w; i += 256) {
            const float v = s[i];
            if (!(v <= t)) {                    // `relevance[relevance <= threshold] = 0`
                const int y = i / w, x = i - y * w;
                tot += v;
                if (y >= y0 && y < y1 && x >= x0 && x < x1) ins += v;
            }
        }
        tot = bsum(tot, red);
        ins = bsum(ins, red);
        if (threadIdx.x == 0) {
            float r = 0.f;
            if (tot != 0.f) { r = ins / tot; if (r > 1.f) r = 1.f; }
            out[(long)blockIdx.x * nthr + j] = r;
        }
    }
}

// out[n] = {mean, mean |x|, sum(max(x,0)) / count(x > 0) (0 if none), max}
__global__ __launch_bounds__(256) void map_stats_kernel(const float* __restrict__ sp, long per, float* __restrict__ out) {
    __shared__ float red[4];
    const float* s = sp + (long)blockIdx.x * per;
    float sm_ = 0.f, sa = 0.f, spos = 0.f, cnt = 0.f, mx = -INFINITY;
    for (long i = threadIdx.x; i < per; i += 256) {
        const float v = s[i];
        sm_ += v; sa += fabsf(v); mx = fmaxf(mx, v);
        if (v > 0.f) { spos += v; cnt += 1.f; }
    }
    sm_ = bsum(sm_, red); sa = bsum(sa, red); spos = bsum(spos, red); cnt = bsum(cnt, red); mx = bmax(mx, red);
    if (threadIdx.x == 0) {
        float* o = out + (long)blockIdx.x * 4;
        o[0] = sm_ / (float)per; o[1] = sa / (float)per; o[2] = cnt > 0.f ? spos / cnt : 0.f; o[3] = mx;
    }
}

// np.quantile(map, q) with the default 'linear' method (evaluation.py:510, :543 on the channel-mean map; 100 points):
// the maps are sorted per segment (rocPRIM segmented radix sort: a utility sort, not a hot kernel) and the order
// statistics around q*(n-1) are interpolated the way numpy's _lerp does.
__global__ void segment_offsets_kernel(int* __restrict__ off, int n_seg, long per) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n_seg) off[i] = (int)((long)i * per);
}

__global__ void quantile_lerp_kernel(const float* __restrict__ sorted, long per, const double* __restrict__ q, int nq,
                                     float* __restrict__ out) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nq) return;
    const float* s = sorted + (long)blockIdx.y * per;
    const double pos = q[j] * (double)(per - 1);
    long lo = (long)floor(pos);
    lo = lo < 0 ? 0 : (lo > per - 1 ? per - 1 : lo);
    const long hi = lo + 1 > per - 1 ? per - 1 : lo + 1;
    const double t = pos - (double)lo;
    const double a = s[lo], b = s[hi], d = b - a;
    out[(long)blockIdx.y * nq + j] = (float)(t >= 0.5 ? b - d * (1.0 - t) : a + d * t);
}

// gamma(X) (utils.py:97-145, minamp = 0, maxamp = max|X| of the map) followed by heatmap(.) (utils.py:67-90: sum over
// the channels, project to [0,255] with the map's own max |.|, integer colour-map lookup).  One workgroup per map;
// `tmp` holds the channel sums between the two passes.
__global__ __launch_bounds__(256) void heatmap_kernel(const float* __restrict__ maps, int c, long hw, float gamma_,
                                                      const float* __restrict__ lut, int nlut, float* __restrict__ tmp,
                                                      float* __restrict__ out) {
    __shared__ float red[4];
    const float* m = maps + (long)blockIdx.x * c * hw;
    float* t = tmp + (long)blockIdx.x * hw;
    float amax = 0.f;
    for (long i = threadIdx.x; i < c * hw; i += 256) amax = fmaxf(amax, fabsf(m[i]));
    amax = bmax(amax, red);
    float tmax = 0.f;
    for (long p = threadIdx.x; p < hw; p += 256) {
        float s = 0.f;
        for (int k = 0; k < c; ++k) {
            float x = m[k * hw + p];
            if (amax != 0.f) {                                   // `if maxamp == 0: return X`
                x = x / amax;
                x = (x >= 0.f ? powf(x, gamma_) : -powf(-x, gamma_)) * amax;
            }
            s += x;
        }
        t[p] = s;
        tmax = fmaxf(tmax, fabsf(s));
    }
    tmax = bmax(tmax, red);
    float* o = out + (long)blockIdx.x * hw * 3;
    for (long p = threadIdx.x; p < hw; p += 256) {
        float x = t[p];
        if (tmax != 0.f) x = x / tmax;                           // project(): only maps with a non-zero maximum are scaled
        x = fminf(fmaxf((x + 1.f) * 0.5f, 0.f), 1.f) * 255.f;   // (X+1)/2, clip, output_range (0, 255)
        int idx = (int)x;                                        // .astype(np.int64) truncates
        idx = idx < 0 ? 0 : (idx >= nlut ? nlut - 1 : idx);
        o[p * 3] = lut[idx * 3]; o[p * 3 + 1] = lut[idx * 3 + 1]; o[p * 3 + 2] = lut[idx * 3 + 2];
    }
}

}  // namespace lrpx

using namespace lrpx;

extern "C" {

int lrpx_spatial_reduce(const float* maps, int n, int c, long hw, int mode, float* out, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_spatial_reduce", {maps, "maps"}, {out, "out"});
    LRPX_REQUIRE(maps && out && n > 0 && c > 0 && hw > 0 && mode >= 0 && mode <= 2, "spatial_reduce: bad arguments");
    const long total = (long)n * hw;
    hipLaunchKernelGGL(spatial_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, maps,
                       c, hw, mode, out, total);
    return check_launch("spatial_reduce");
}

int lrpx_project_maxabs(float* x, int n, long per, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_project_maxabs", {x, "x"});
    LRPX_REQUIRE(x && n > 0 && per > 0, "project_maxabs: bad arguments");
    hipLaunchKernelGGL(project_maxabs_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, x, per);
    return check_launch("project_maxabs");
}

int lrpx_patch_mask(const float* spatial, int n, int h, int w, int patch, int k, float* mask, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_patch_mask", {spatial, "spatial"}, {mask, "mask"});
    LRPX_REQUIRE(spatial && mask && n > 0 && patch > 0 && h % patch == 0 && w % patch == 0, "patch_mask: bad arguments");
    const int np_ = (h / patch) * (w / patch);
    LRPX_REQUIRE(k >= 0 && k <= np_ && np_ <= 8192, "patch_mask: k must not exceed the number of patches (<= 8192)");
    hipLaunchKernelGGL(patch_mask_kernel, dim3(n), dim3(256), (size_t)(2 * np_ + 8) * sizeof(float), (hipStream_t)stream,
                       spatial, h, w, patch, k, mask);
    return check_launch("patch_mask");
}

int lrpx_bbox_ratio(const float* spatial, int n, int h, int w, const int32_t* boxes, const float* thresholds, int nthr,
                    float* out, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_bbox_ratio", {spatial, "spatial"}, {boxes, "boxes"}, {thresholds, "thresholds"}, {out, "out"});
    LRPX_REQUIRE(spatial && boxes && thresholds && out && n > 0 && h > 0 && w > 0 && nthr > 0, "bbox_ratio: bad arguments");
    hipLaunchKernelGGL(bbox_ratio_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, spatial, h, w, boxes, thresholds, nthr,
                       out);
    return check_launch("bbox_ratio");
}

int lrpx_map_stats(const float* spatial, int n, long per, float* out4, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_map_stats", {spatial, "spatial"}, {out4, "out4"});
    LRPX_REQUIRE(spatial && out4 && n > 0 && per > 0, "map_stats: bad arguments");
    hipLaunchKernelGGL(map_stats_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, spatial, per, out4);
    return check_launch("map_stats");
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t lrpx_map_quantiles_workspace(int n, long per) {
    if (n <= 0 || per <= 0 || (long)n * per >= 0x7fffffffL) return 0;
    size_t tmp = 0;
    const float* kin = nullptr; float* kout = nullptr; const int* off = nullptr;
    if (hipcub::DeviceSegmentedRadixSort::SortKeys(nullptr, tmp, kin, kout, (int)((long)n * per), n, off, off + 1) !=
        hipSuccess)
        return 0;
    return align256((size_t)n * per * sizeof(float)) + align256((size_t)(n + 1) * sizeof(int)) + align256(tmp);
}

int lrpx_map_quantiles(const float* spatial, int n, long per, const double* q, int nq, float* out, void* workspace,
                       size_t workspace_bytes, void* stream) {
    LRPX_CHECK_PTRS("lrpx_map_quantiles", {spatial, "spatial"}, {out, "out"}, {workspace, "workspace"});
    LRPX_REQUIRE(spatial && q && out && workspace && n > 0 && per > 0 && nq > 0, "map_quantiles: bad arguments");
    LRPX_REQUIRE((long)n * per < 0x7fffffffL, "map_quantiles: more than 2^31 values in one call");
    const size_t need = lrpx_map_quantiles_workspace(n, per);
    LRPX_REQUIRE(need && workspace_bytes >= need, "map_quantiles: workspace too small (lrpx_map_quantiles_workspace)");
    char* w = static_cast<char*>(workspace);
    float* sorted = reinterpret_cast<float*>(w);
    int* off = reinterpret_cast<int*>(w + align256((size_t)n * per * sizeof(float)));
    void* tmp = w + align256((size_t)n * per * sizeof(float)) + align256((size_t)(n + 1) * sizeof(int));
    size_t tmp_bytes = workspace_bytes - (size_t)(static_cast<char*>(tmp) - w);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(segment_offsets_kernel, dim3((n + 256) / 256), dim3(256), 0, st, off, n, per);
    if (hipcub::DeviceSegmentedRadixSort::SortKeys(tmp, tmp_bytes, spatial, sorted, (int)((long)n * per), n, off, off + 1,
                                                   0, 32, st) != hipSuccess) {
        set_error("map_quantiles: segmented sort failed");
        return LRPX_ELAUNCH;
    }
    hipLaunchKernelGGL(quantile_lerp_kernel, dim3((nq + 127) / 128, n), dim3(128), 0, st, sorted, per, q, nq, out);
    return check_launch("map_quantiles");
}

int lrpx_heatmap(const float* maps, int n, int c, long hw, float gamma, const float* lut, int nlut, float* tmp, float* out,
                 void* stream) {
    LRPX_CHECK_PTRS("lrpx_heatmap", {maps, "maps"}, {lut, "lut"}, {tmp, "tmp"}, {out, "out"});
    LRPX_REQUIRE(maps && lut && tmp && out && n > 0 && c > 0 && hw > 0 && nlut > 0 && gamma > 0.f, "heatmap: bad arguments");
    hipLaunchKernelGGL(heatmap_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, maps, c, hw, gamma, lut, nlut, tmp, out);
    return check_launch("heatmap");
}

}  // extern "C"
