// Guided-Grad-CAM combination (ExplainGridTDGuidedGradCam.explain_cnn, models/gridTDmodel.py:1814-1836;
// ExplainAOAGuidedGradCam.explain_cnn, models/aoamodel.py:1729-1751):
//     guided_results = guided_gradient * pyramid_expand(cam, upscale = 16)          (cam: 14x14 Grad-CAM heat map)
// skimage.transform.pyramid_expand (scikit-image, un-vendored dependency of the reference; absent from this image) is
// `resize(order=1, mode='reflect')` followed by a Gaussian smoothing with sigma = 2 * upscale / 6, truncate 4 - both
// linear and separable, so the whole expansion is E = M cam M^T with one (HW x p) matrix M = Gauss (HW x HW) . Bilinear
// (HW x p) that the host builds once (ops.pyramid_expand_matrix).  One workgroup per map: tmp = M cam (HW x p, LDS), then
// every pixel E[y][x] = sum_j tmp[y][j] M[x][j] times the three channels of the guided gradient.  HBM-bound: reads and
// writes the maps once (1.2 MB per map).
#include "common.h"

namespace lrpx {

template <int PMAX>
__global__ __launch_bounds__(256) void guided_gradcam_kernel(const float* __restrict__ g, const float* __restrict__ cam,
                                                             const float* __restrict__ M, float* __restrict__ out,
                                                             int p, int hw, int channels) {
    extern __shared__ float lds[];
    float* Ms = lds;                    // [hw][p]
    float* tmp = lds + hw * p;          // [hw][p]   tmp[y][j] = sum_i M[y][i] cam[i][j]
    float* cs = tmp + hw * p;           // [p][p]
    const int n = blockIdx.x, tid = threadIdx.x;
    for (int e = tid; e < hw * p; e += 256) Ms[e] = M[e];
    for (int e = tid; e < p * p; e += 256) cs[e] = cam[(long)n * p * p + e];
    __syncthreads();
    for (int e = tid; e < hw * p; e += 256) {
        const int y = e / p, j = e - y * p;
        float s = 0.f;
        for (int i = 0; i < p; ++i) s = fmaf(Ms[y * p + i], cs[i * p + j], s);
        tmp[e] = s;
    }
    __syncthreads();
    const long per = (long)hw * hw;
    const float* gn = g + (long)n * channels * per;
    float* on = out + (long)n * channels * per;
    for (long q = tid; q < per; q += 256) {
        const int y = (int)(q / hw), x = (int)(q - (long)y * hw);
        float ev = 0.f;
        for (int j = 0; j < p; ++j) ev = fmaf(tmp[y * p + j], Ms[x * p + j], ev);
        for (int c = 0; c < channels; ++c) on[c * per + q] = gn[c * per + q] * ev;
    }
}

}  // namespace lrpx

using namespace lrpx;

extern "C" int lrpx_guided_gradcam(const float* guided, const float* cam, const float* expand_m, float* out, int rows, int p,
                                   int hw, int channels, void* stream) {
    LRPX_REQUIRE(guided && cam && expand_m && out && rows > 0 && p > 0 && p <= 32 && hw > 0 && channels > 0,
                 "guided_gradcam: bad arguments");
    const int lds = (2 * hw * p + p * p) * (int)sizeof(float);
    LRPX_REQUIRE(lds <= 64 * 1024, "guided_gradcam: hw * p too large for the LDS image (%d bytes)", lds);
    hipLaunchKernelGGL((guided_gradcam_kernel<32>), dim3(rows), dim3(256), lds, (hipStream_t)stream, guided, cam, expand_m,
                       out, p, hw, channels);
    return check_launch("guided_gradcam");
}
