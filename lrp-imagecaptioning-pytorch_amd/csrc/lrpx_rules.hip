// Relevance rules of the layers the VGG16 path never reaches (ResNet encoders of the reference):
//   Linear       epsilon rule with the in-place zero nudge      LRPtools/lrp_modules.py:9-37
//   BatchNorm2d / BatchNorm1d   |xw| / (|xw| + |b|) split      LRPtools/lrp_modules.py:197-246
//   Add          proportional split, 0.5 / 0.5 on zero sums     LRPtools/lrp_modules.py:256-280
//   Dropout      |R_out - R_in| < 1e-7 check                    LRPtools/lrp_modules.py:248-254
//   AvgPool2d    Z = avgpool(X), R = X * avgpool^T(R_out / Z)    LRPtools/lrp_modules.py:172-195 (Pool2d, the nn.AvgPool2d branch)
// (Flatten, :282-291, is a copy: lrpx_scale with factor 1.)
// All of them are HBM-bound: one streaming pass each, the Linear rule streams W twice (Z = x W^T, then (R/Z) W) with the
// handful of batch rows held in registers.  Arithmetic follows the reference expression by expression (IEEE division and
// square root, no re-association), so the elementwise rules are bit-exact against PyTorch's CPU kernels.
#include <math.h>

#include "common.h"

#pragma clang fp contract(off)   // expression-by-expression arithmetic, as the reference's elementwise ops evaluate it

namespace lrpx {

static constexpr float kEps = 0.01f;      // LRPtools/utils.py:10 EPSILON
static constexpr float kZEps = 1e-7f;     // LRPtools/utils.py:11 Z_EPSILON
static constexpr float kRect = -1e-6f;    // LRPtools/utils.py:14 RELEVANCE_RECT

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// lrp_modules.py:14  input_.masked_fill_(input_ == 0, RELEVANCE_RECT) - on the SAVED input, in place (quirk h)
__global__ void nudge_zero_kernel(float* __restrict__ x, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && x[i] == 0.f) x[i] = kRect;
}

// Z = x W^T (:16), stabiliser (:17-21), S = R_out / Z (:22).  One wave per output feature, RB rows in registers, lanes
// stride over the input features (coalesced 256-byte reads of the W row and of the x rows).
template <int RB>
__global__ __launch_bounds__(256) void linear_rule_s_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ r_out, float* __restrict__ s,
                                                            int N, int I, int O) {
    const int lane = threadIdx.x & 63;
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (o >= O) return;
    const float* wr = w + (long)o * I;
    for (int b0 = 0; b0 < N; b0 += RB) {
        float acc[RB];
#pragma unroll
        for (int b = 0; b < RB; ++b) acc[b] = 0.f;
        for (int k = lane; k < I; k += 64) {
            const float wv = wr[k];
#pragma unroll
            for (int b = 0; b < RB; ++b) {
                const int row = min(b0 + b, N - 1);
                acc[b] = fmaf(x[(long)row * I + k], wv, acc[b]);
            }
        }
#pragma unroll
        for (int b = 0; b < RB; ++b) {
            float z = wsum(acc[b]);
            if (lane == 0 && b0 + b < N) {
                if (bias) {
                    z += bias[o];                                          // not ignore_bias (:20-21)
                } else {
                    z += kEps * (z > 0.f ? 1.f : (z < 0.f ? -1.f : 0.f));  // Z += eps * sign(Z) (:18)
                    if (z == 0.f) z = kEps;                                // exact zeros -> eps (:19)
                }
                s[(long)(b0 + b) * O + o] = r_out[(long)(b0 + b) * O + o] / z;
            }
        }
    }
}

// C = S W (:23), R = x * C (:24).  One thread per input feature (coalesced reads of W rows), RB rows in registers, S
// staged through LDS in chunks of OC outputs.
template <int RB, int OC>
__global__ __launch_bounds__(256) void linear_rule_back_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ s, float* __restrict__ r_in,
                                                               int N, int I, int O) {
    __shared__ float ss[RB][OC];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int b0 = blockIdx.y * RB;
    float acc[RB];
#pragma unroll
    for (int b = 0; b < RB; ++b) acc[b] = 0.f;
    for (int o0 = 0; o0 < O; o0 += OC) {
        __syncthreads();
        for (int e = threadIdx.x; e < RB * OC; e += 256) {
            const int b = e / OC, oo = e % OC;
            ss[b][oo] = (b0 + b < N && o0 + oo < O) ? s[(long)(b0 + b) * O + o0 + oo] : 0.f;
        }
        __syncthreads();
        if (i < I) {
            const int on = min(OC, O - o0);
            for (int oo = 0; oo < on; ++oo) {
                const float wv = w[(long)(o0 + oo) * I + i];
#pragma unroll
                for (int b = 0; b < RB; ++b) acc[b] = fmaf(ss[b][oo], wv, acc[b]);
            }
        }
    }
    if (i < I) {
#pragma unroll
        for (int b = 0; b < RB; ++b)
            if (b0 + b < N) r_in[(long)(b0 + b) * I + i] = x[(long)(b0 + b) * I + i] * acc[b];
    }
}

// lrp_modules.py:205-216 / :231-242.  Element e of the result lies in channel (e / inner) % C; with bcast_x the input and
// the relevance are indexed by e % inner only (the (N,C) / (1,C,L) inputs of BatchNorm1d against w[:, None, None]).
__global__ void batchnorm_rule_kernel(const float* __restrict__ x, const float* __restrict__ r_out,
                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                      const float* __restrict__ mean, const float* __restrict__ var, float eps,
                                      float* __restrict__ r_in, long total, int C, long inner, int bcast_x) {
    const long e = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= total) return;
    const int c = (int)((e / inner) % C);
    const long xi = bcast_x ? e % inner : e;
    const float sd = sqrtf(var[c] + eps);
    const float wc = gamma[c] / sd;                                   // :210
    const float bc = beta[c] - (mean[c] * gamma[c]) / sd;             // :211
    const float xw = fabsf(x[xi] * wc);                               // :212
    const float den = xw + fabsf(bc);
    r_in[e] = (xw / (den + kZEps * (den == 0.f ? 1.f : 0.f))) * r_out[xi];   // safe_divide (utils.py:16-18), :214-215
}

// lrp_modules.py:260-275
__global__ void add_rule_kernel(const float* __restrict__ x1, const float* __restrict__ x2,
                                const float* __restrict__ r_out, float* __restrict__ r1, float* __restrict__ r2, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float a = x1[i], b = x2[i], r = r_out[i];
    float out = a + b;
    const float half = out == 0.f ? 0.5f : 0.f;                       // out_mask (:263-264)
    out += kEps * (out > 0.f ? 1.f : (out < 0.f ? -1.f : 0.f));       // :266
    float v1 = (r * a) / out, v2 = (r * b) / out;                     // :269-270
    if (v1 != v1) v1 = 0.f;                                           // :271-272
    if (v2 != v2) v2 = 0.f;
    const float rm = r * half;                                        // R_mask (:268)
    r1[i] = v1 + rm;
    r2[i] = v2 + rm;
}

// max |a - b| as float bits (non-negative floats order like unsigned integers); NaN differences count as +inf
__global__ void max_abs_diff_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                    unsigned* __restrict__ out) {
    float m = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const float d = fabsf(a[i] - b[i]);
        m = (d != d) ? INFINITY : fmaxf(m, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(out, __float_as_uint(m));
}


// ---- Pool2d rule for nn.AvgPool2d (lrp_modules.py:176-177 clone, :182-195 rule; dispatch entry :327) -----------------------
// Z = avgpool(X), S = safe_divide(R_out, Z) (utils.py:16-18), Z.backward(S), R = X * X.grad.  NCHW fp32 as the module sees it,
// any kernel / stride / padding / count_include_pad / ceil_mode / divisor_override.  Window bounds and divisor follow ATen's
// CPU kernels statement by statement (the padded window end is clipped at H + pad BEFORE the divisor is taken, then the window
// is clipped to the image), the window sum runs row-major and the input gradient adds its windows in ascending (oh, ow) order
// - what the reference's autograd executes - so the result is bit-exact against it.
struct AvgPoolGeom { int H, W, OH, OW, kh, kw, sh, sw, ph, pw, count_include_pad, divisor_override; };

__device__ __forceinline__ void avg_window(const AvgPoolGeom& g, int oh, int ow, int& h0, int& h1, int& w0, int& w1, int& div) {
    h0 = oh * g.sh - g.ph; w0 = ow * g.sw - g.pw;
    h1 = min(h0 + g.kh, g.H + g.ph); w1 = min(w0 + g.kw, g.W + g.pw);
    const int pool_size = (h1 - h0) * (w1 - w0);
    h0 = max(h0, 0); w0 = max(w0, 0); h1 = min(h1, g.H); w1 = min(w1, g.W);
    div = g.divisor_override ? g.divisor_override : (g.count_include_pad ? pool_size : (h1 - h0) * (w1 - w0));
}

// one thread per output element: sd = (R_out / (Z + 1e-7 [Z == 0])) / divisor  (what every input of the window receives)
__global__ void avgpool_rule_s_kernel(const float* __restrict__ x, const float* __restrict__ r_out, float* __restrict__ sd,
                                      long planes, AvgPoolGeom g) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)g.OH * g.OW;
    if (i >= planes * per) return;
    const long pl = i / per;
    const int o = (int)(i - pl * per), oh = o / g.OW, ow = o - oh * g.OW;
    int h0, h1, w0, w1, div;
    avg_window(g, oh, ow, h0, h1, w0, w1, div);
    float z = 0.f;
    if (h0 < h1 && w0 < w1) {
        const float* xp = x + pl * g.H * g.W;
        float sum = 0.f;
        for (int h = h0; h < h1; ++h)
            for (int w = w0; w < w1; ++w) sum += xp[h * g.W + w];
        z = sum / (float)div;
    }
    const float s = r_out[i] / (z + kZEps * (z == 0.f ? 1.f : 0.f));
    sd[i] = (h0 < h1 && w0 < w1) ? s / (float)div : 0.f;
}

// one thread per input element: X.grad = sum of sd over the windows that contain it (ascending), R = X * X.grad
__global__ void avgpool_rule_back_kernel(const float* __restrict__ x, const float* __restrict__ sd, float* __restrict__ r_in,
                                         long planes, AvgPoolGeom g) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long per = (long)g.H * g.W;
    if (i >= planes * per) return;
    const long pl = i / per;
    const int p = (int)(i - pl * per), h = p / g.W, w = p - h * g.W;
    // windows with h0 <= h < h0 + kh (h < H: the clips at H + pad and at H never cut an image row out of a window)
    const int oh_lo = max(0, (h + g.ph - g.kh + g.sh) / g.sh), oh_hi = min(g.OH - 1, (h + g.ph) / g.sh);
    const int ow_lo = max(0, (w + g.pw - g.kw + g.sw) / g.sw), ow_hi = min(g.OW - 1, (w + g.pw) / g.sw);
    const float* sp = sd + pl * g.OH * g.OW;
    float grad = 0.f;
    for (int oh = oh_lo; oh <= oh_hi; ++oh)
        for (int ow = ow_lo; ow <= ow_hi; ++ow) grad += sp[oh * g.OW + ow];
    r_in[i] = x[i] * grad;
}

}  // namespace lrpx

using namespace lrpx;

extern "C" {

int lrpx_linear_eps_rule(float* x, const float* w, const float* bias, const float* r_out, float* s_ws, float* r_in,
                         int n_rows, int n_in, int n_out, void* stream) {
    LRPX_CHECK_PTRS("lrpx_linear_eps_rule", {x, "x"}, {w, "w"}, {bias, "bias"}, {r_out, "r_out"}, {s_ws, "s_ws"}, {r_in, "r_in"});
    LRPX_REQUIRE(x && w && r_out && s_ws && r_in && n_rows > 0 && n_in > 0 && n_out > 0, "linear_eps_rule: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const long nx = (long)n_rows * n_in;
    hipLaunchKernelGGL(nudge_zero_kernel, dim3((unsigned)ceil_div(nx, 256)), dim3(256), 0, st, x, nx);
    hipLaunchKernelGGL((linear_rule_s_kernel<8>), dim3((n_out + 3) / 4), dim3(256), 0, st, x, w, bias, r_out, s_ws, n_rows,
                       n_in, n_out);
    hipLaunchKernelGGL((linear_rule_back_kernel<8, 128>), dim3((n_in + 255) / 256, (n_rows + 7) / 8), dim3(256), 0, st, x,
                       w, s_ws, r_in, n_rows, n_in, n_out);
    return check_launch("linear_eps_rule");
}

int lrpx_batchnorm_rule(const float* x, const float* r_out, const float* gamma, const float* beta, const float* mean,
                        const float* var, float eps, float* r_in, long n_outer, int channels, long inner, int broadcast_x,
                        void* stream) {
    LRPX_CHECK_PTRS("lrpx_batchnorm_rule", {x, "x"}, {r_out, "r_out"}, {gamma, "gamma"}, {beta, "beta"}, {mean, "mean"}, {var, "var"}, {r_in, "r_in"});
    LRPX_REQUIRE(x && r_out && gamma && beta && mean && var && r_in && n_outer > 0 && channels > 0 && inner > 0,
                 "batchnorm_rule: bad arguments");
    LRPX_REQUIRE(!broadcast_x || n_outer == 1, "batchnorm_rule: a broadcast input has no outer dimension");
    const long total = n_outer * channels * inner;
    hipLaunchKernelGGL(batchnorm_rule_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, x,
                       r_out, gamma, beta, mean, var, eps, r_in, total, channels, inner, broadcast_x);
    return check_launch("batchnorm_rule");
}

int lrpx_add_rule(const float* x1, const float* x2, const float* r_out, float* r1, float* r2, long n, void* stream) {
    LRPX_CHECK_PTRS("lrpx_add_rule", {x1, "x1"}, {x2, "x2"}, {r_out, "r_out"}, {r1, "r1"}, {r2, "r2"});
    LRPX_REQUIRE(x1 && x2 && r_out && r1 && r2 && n > 0, "add_rule: bad arguments");
    hipLaunchKernelGGL(add_rule_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x1, x2, r_out,
                       r1, r2, n);
    return check_launch("add_rule");
}

int lrpx_avgpool_rule(const float* x, const float* r_out, float* s_ws, float* r_in, long planes, int h, int w, int oh, int ow,
                      int kh, int kw, int sh, int sw, int ph, int pw, int count_include_pad, int divisor_override, void* stream) {
    LRPX_CHECK_PTRS("lrpx_avgpool_rule", {x, "x"}, {r_out, "r_out"}, {s_ws, "s_ws"}, {r_in, "r_in"});
    LRPX_REQUIRE(x && r_out && s_ws && r_in && planes > 0 && h > 0 && w > 0 && oh > 0 && ow > 0, "avgpool_rule: bad arguments");
    LRPX_REQUIRE(kh > 0 && kw > 0 && sh > 0 && sw > 0 && ph >= 0 && pw >= 0 && 2 * ph <= kh && 2 * pw <= kw && divisor_override >= 0,
                 "avgpool_rule: bad window (kernel %dx%d stride %dx%d padding %dx%d)", kh, kw, sh, sw, ph, pw);
    // every window must start inside the image or its left padding (what ATen's output-size rule guarantees)
    LRPX_REQUIRE((long)(oh - 1) * sh < h + ph && (long)(ow - 1) * sw < w + pw, "avgpool_rule: output %dx%d does not fit input %dx%d", oh, ow, h, w);
    LRPX_REQUIRE(planes * (long)h * w < (1L << 40), "avgpool_rule: tensor too large");
    const AvgPoolGeom g = {h, w, oh, ow, kh, kw, sh, sw, ph, pw, count_include_pad ? 1 : 0, divisor_override};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(avgpool_rule_s_kernel, dim3((unsigned)ceil_div(planes * oh * ow, 256)), dim3(256), 0, st, x, r_out, s_ws, planes, g);
    hipLaunchKernelGGL(avgpool_rule_back_kernel, dim3((unsigned)ceil_div(planes * h * w, 256)), dim3(256), 0, st, x, s_ws, r_in, planes, g);
    return check_launch("avgpool_rule");
}

int lrpx_max_abs_diff(const float* a, const float* b, long n, float* out_dev, void* stream) {
    LRPX_REQUIRE(a && b && out_dev && n > 0, "max_abs_diff: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(out_dev, 0, sizeof(float), st) != hipSuccess) {
        set_error("max_abs_diff: cannot zero the result");
        return LRPX_ELAUNCH;
    }
    const unsigned blocks = (unsigned)min((long)2048, ceil_div(n, 256));
    hipLaunchKernelGGL(max_abs_diff_kernel, dim3(blocks), dim3(256), 0, st, a, b, n, reinterpret_cast<unsigned*>(out_dev));
    return check_launch("max_abs_diff");
}

}  // extern "C"
