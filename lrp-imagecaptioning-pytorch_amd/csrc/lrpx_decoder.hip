// Decoder-side kernels of the LRP hot path (gridTD adaptive-attention decoder, AoA decoder).
//
// Everything here is small next to the VGG16 relevance pass (<1 % of the FLOPs), HBM / latency bound,
// and written as fused per-time-step kernels batched over all (image, word) rows so that the
// reference's ~50 k tiny `lrp_linear_eps` calls per image (models/gridTDmodel.py:1060-1128) become
// ~5 launches per lock-step.  Row r = b*T + t is the explanation of word t of image b; at lock-step s
// every row with t >= s works on time index i = t - s (models/gridTDmodel.py:1060 `for i in range(t+1)[::-1]`).
//
// Trace tensors are [B][T(+1)][...] fp32; token ids int64 (torch.long).
#include <math.h>

#include "common.h"
#include "conv_mfma.h"
#include "conv_launch.h"

namespace lrpx {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float eps_id(float r, float x, float z) { return (x / stab_eps(z)) * r; }   // weight=eye rule

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
// block-wide reductions for 256-thread blocks (4 waves); `red` has >= 4 floats of LDS
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ------------------------------------------------------------------------------------------------
// skinny linear: out[b][n] = act(sum_k x[b][k] * w[n][k] + bias[n]) for a handful of rows b.
// One wave per NPW output features; lanes split K in float4; rows live in register accumulators.
// Weight-read bound (each weight byte is read once per 16-row chunk, from L2).
// ------------------------------------------------------------------------------------------------
template <int NPW, int BCH>
__global__ __launch_bounds__(256) void linear_small_kernel(const float* __restrict__ x, long ldx,
                                                           const float* __restrict__ w, const float* __restrict__ bias,
                                                           float* __restrict__ out, long ldo, int B, int K, int N,
                                                           int act) {
    // one workgroup per NPW output features; its 4 waves split K (a 16-row problem has too few outputs to fill the
    // chip with one wave per feature group, and one wave's serial K loop is a chain of L2 round trips)
    __shared__ float red[4][NPW * BCH];
    const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
    const int n0 = blockIdx.x * NPW;
    const int K4 = K >> 2;
    const int kq = (K4 + 3) / 4, k_lo = wv_ * kq, k_hi = min(K4, k_lo + kq);
    for (int b0 = 0; b0 < B; b0 += BCH) {
        float acc[NPW][BCH];
#pragma unroll
        for (int j = 0; j < NPW; ++j)
#pragma unroll
            for (int bb = 0; bb < BCH; ++bb) acc[j][bb] = 0.f;
        for (int k4 = k_lo + lane; k4 < k_hi; k4 += 64) {
            f32x4 wv[NPW];
#pragma unroll
            for (int j = 0; j < NPW; ++j) {
                const int n = min(n0 + j, N - 1);
                wv[j] = reinterpret_cast<const f32x4*>(w + (long)n * K)[k4];
            }
#pragma unroll
            for (int bb = 0; bb < BCH; ++bb) {
                const int b = min(b0 + bb, B - 1);
                const f32x4 xv = reinterpret_cast<const f32x4*>(x + (long)b * ldx)[k4];
#pragma unroll
                for (int j = 0; j < NPW; ++j)
                    acc[j][bb] += wv[j][0] * xv[0] + wv[j][1] * xv[1] + wv[j][2] * xv[2] + wv[j][3] * xv[3];
            }
        }
        __syncthreads();                 // (previous row chunk's reduction has been read)
#pragma unroll
        for (int j = 0; j < NPW; ++j)
#pragma unroll
            for (int bb = 0; bb < BCH; ++bb) {
                const float v = wave_sum(acc[j][bb]);
                if (lane == 0) red[wv_][j * BCH + bb] = v;
            }
        __syncthreads();
        if (threadIdx.x < NPW * BCH) {
            const int j = threadIdx.x / BCH, bb = threadIdx.x % BCH;
            const int n = n0 + j, b = b0 + bb;
            float v = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
            if (n < N && b < B) {
                if (bias) v += bias[n];
                if (act == 1) v = v > 0.f ? v : 0.f;
                out[(long)b * ldo + n] = v;
            }
        }
    }
}

// The K loop of the skinny linears on the fp32 matrix cores (linear_mfma_kernel below and the fused step kernels further down share it:
// the same dot products in the same order, bit-identical traces).  The 4 waves split K; a lane loads ONE float4 of its weight row and one
// of its x row per 16 k (lane = (feature or row) lane & 15, k quad lane >> 4: component j of the two float4s is the operand pair of MFMA j).
// Round 6: FOUR accumulators per row tile take the wave's k-steps in turn (k-step s -> accumulator s % 4): a sum of 2048 products is formed
// from 16 partial sums of 128 instead of 4 of 512 - the gate pre-activations of a gridTD trace moved from 8e-7 to 3e-7 of their fp64 values
// (what an fp32 CPU GEMM gives: tests/diag_t20_words.py), and with them the worst T = 20 r_words row of the goldens from 1.0e-5 to 1.4e-6 of
// fp64.  No faster: the kernel is at the rate its weights stream in (11 us per gridTD gate linear, 5 us of them the launch floor).
template <int RT>
__device__ __forceinline__ void linear_mfma_core(const float* __restrict__ x, long ldx, const float* __restrict__ w, int B, int K,
                                                 int N, float (&red)[4][RT][16][17], const int n0) {
    const int lane = threadIdx.x & 63, wv_ = threadIdx.x >> 6;
    const int nl = lane & 15, kq = lane >> 4;
    const int kslice = ((K / 16 + 3) / 4) * 16;                        // K % 16 == 0 (host-checked)
    const int k_lo = wv_ * kslice, k_hi = min(K, k_lo + kslice);
    const float* __restrict__ wrow = w + (long)min(n0 + nl, N - 1) * K + kq * 4;
    const float* __restrict__ xrow[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) xrow[r] = x + (long)min(r * 16 + nl, B - 1) * ldx + kq * 4;
    constexpr int NA = 4;                                              // accumulators per row tile (k-step s -> accumulator s % NA)
    constexpr int NW = RT <= 2 ? 8 : 4;                                // k-steps of weights in flight per wave (16 measured no faster: the
                                                                       // kernel streams its 13 - 17 MB of fp32 weights at ~2.8 TB/s either way)
    constexpr int NX = 4;                                              // k-steps of x rows in flight (L2 hits)
    f32x4 acc[RT][NA];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int u = 0; u < NA; ++u) acc[r][u] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k0 = k_lo; k0 < k_hi; k0 += 16 * NW) {
        f32x4 wq[NW];
#pragma unroll
        for (int u = 0; u < NW; ++u) wq[u] = *reinterpret_cast<const f32x4*>(wrow + min(k0 + 16 * u, k_hi - 16));   // (past the slice: re-read, MFMAs skipped)
#pragma unroll
        for (int v = 0; v < NW; v += NX) {
            f32x4 xq[RT][NX];
#pragma unroll
            for (int u = 0; u < NX; ++u)
#pragma unroll
                for (int r = 0; r < RT; ++r) xq[r][u] = *reinterpret_cast<const f32x4*>(xrow[r] + min(k0 + 16 * (v + u), k_hi - 16));
#pragma unroll
            for (int u = 0; u < NX; ++u) {
                if (k0 + 16 * (v + u) < k_hi) {                        // (wave-uniform)
#pragma unroll
                    for (int r = 0; r < RT; ++r)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[r][u % NA] = __builtin_amdgcn_mfma_f32_16x16x4f32(xq[r][u][j], wq[v + u][j], acc[r][u % NA], 0, 0, 0);
                }
            }
        }
    }
    // result layout: column (feature) lane & 15, rows 4 * (lane >> 4) + i
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) red[wv_][r][4 * kq + i][nl] = (acc[r][0][i] + acc[r][1][i]) + (acc[r][2][i] + acc[r][3][i]);
    __syncthreads();
}

// The same skinny linear on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32 products): M = 16 rows b per row tile,
// N = 16 output features per workgroup, the 4 waves split K.  A lane loads ONE float4 of its weight row and one of its x
// row per 16 k (lane = (feature or row) lane & 15, k quad lane >> 4: component j of the two float4s is the operand pair of
// MFMA j) - no cross-lane reduction at all: the VALU version spends 64 wave reductions (384 ds_bpermute) per wave and
// ran 22 us for a 16 x 1536 x 2560 problem (0.7 TB/s of weights).  The four K-slices meet in LDS, summed in wave order.
template <int RT>
__global__ __launch_bounds__(256) void linear_mfma_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ w,
                                                          const float* __restrict__ bias, float* __restrict__ out, long ldo,
                                                          int B, int K, int N, int act) {
    __shared__ float red[4][RT][16][17];
    const int n0 = blockIdx.x * 16;
    linear_mfma_core<RT>(x, ldx, w, B, K, N, red, n0);
    for (int e = threadIdx.x; e < RT * 256; e += 256) {
        const int r = e >> 8, row = (e >> 4) & 15, col = e & 15;
        const int b = r * 16 + row, n = n0 + col;
        if (b < B && n < N) {
            float v = (red[0][r][row][col] + red[1][r][row][col]) + (red[2][r][row][col] + red[3][r][row][col]);
            if (bias) v += bias[n];
            if (act == 1) v = v > 0.f ? v : 0.f;
            out[(long)b * ldo + n] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// image-side constants: avg over pixels, relu
// ------------------------------------------------------------------------------------------------
__global__ void mean_pixels_kernel(const float* __restrict__ f, float* __restrict__ avg, int P, int C, float scale) {
    // block = 64 channels x 4 pixel slices (a lone thread per channel walked the P pixels as a chain of dependent adds on
    // 32 workgroups: 76 us for 6 MB)
    __shared__ float part[4][64];
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
    float s = 0.f;
    if (c < C)
        for (int p = q; p < P; p += 4) s += f[((long)b * P + p) * C + c];
    part[q][threadIdx.x & 63] = s;
    __syncthreads();
    if (q == 0 && c < C) avg[(long)b * C + c] = ((part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x])) * scale;
}

__global__ void relu_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = fmaxf(x[i], 0.f);
}

// ------------------------------------------------------------------------------------------------
// gridTD forward (trace) — models/gridTDmodel.py:933-1012
// ------------------------------------------------------------------------------------------------
struct GridFwd {
    int B, T, H, E, P;
    // trace ([B][T..] tensors)
    float *xh1, *xh2;                     // [B][T][2E+2H], [B][T][3H]
    float *h1, *c1, *h2, *c2;             // [B][T+1][H]
    float *g1, *i1, *f1, *g2, *i2, *f2;   // [B][T][H]
    float *s, *ctx, *ctx_hat, *hc;        // [B][T][H]
    float *alpha, *beta;                  // [B][T][P], [B][T]
    float *o1, *o2, *sgate;               // [B][T][H] output gates / sentinel gate (gradient explainers only; may be null)
};

// xh1[b,t] = [h2[b,t] | glob[b] | emb[tok[b,t]] | h1[b,t]]
__global__ void gridtd_fwd_pre_kernel(GridFwd g, int t, const float* __restrict__ glob, const float* __restrict__ emb,
                                      const long long* __restrict__ tok, int tok_ld) {
    const int b = blockIdx.x;
    const int W = 2 * g.E + 2 * g.H;
    float* dst = g.xh1 + ((long)b * g.T + t) * W;
    const long st = ((long)b * (g.T + 1) + t) * g.H;
    const long long k = tok[(long)b * tok_ld + t];
    for (int c = threadIdx.x; c < W; c += blockDim.x) {
        float v;
        if (c < g.H) v = g.h2[st + c];
        else if (c < g.H + g.E) v = glob[(long)b * g.E + (c - g.H)];
        else if (c < g.H + 2 * g.E) v = emb[k * g.E + (c - g.H - g.E)];
        else v = g.h1[st + (c - g.H - 2 * g.E)];
        dst[c] = v;
    }
}

// LSTM point-wise part (models/gridTDmodel.py:777-784) + sentinel s = sigmoid(gate) * tanh(c) (:982-983)
// zz: [B][4H (+H gate)] pre-activations incl. bias
__global__ void gridtd_fwd_lstm_kernel(GridFwd g, int t, const float* __restrict__ zz, int ldz, int which) {
    const int b = blockIdx.x;
    const int H = g.H;
    const long st0 = ((long)b * (g.T + 1) + t) * H, st1 = st0 + H, tr = ((long)b * g.T + t) * H;
    float *hh = which == 1 ? g.h1 : g.h2, *cc = which == 1 ? g.c1 : g.c2;
    float *gg = which == 1 ? g.g1 : g.g2, *ii = which == 1 ? g.i1 : g.i2, *ff = which == 1 ? g.f1 : g.f2;
    const float* z = zz + (long)b * ldz;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float i = sigmoidf_(z[c]), f = sigmoidf_(z[H + c]), zg = z[2 * H + c], o = sigmoidf_(z[3 * H + c]);
        const float cn = f * cc[st0 + c] + i * tanhf(zg);
        const float hn = o * tanhf(cn);
        cc[st1 + c] = cn; hh[st1 + c] = hn;
        gg[tr + c] = zg; ii[tr + c] = i; ff[tr + c] = f;
        float* oo = which == 1 ? g.o1 : g.o2;
        if (oo) oo[tr + c] = o;
        if (which == 1) {
            const float sg = sigmoidf_(z[4 * H + c]);
            g.s[tr + c] = sg * tanhf(cn);
            if (g.sgate) g.sgate[tr + c] = sg;
        } else {
            g.hc[tr + c] = hn + g.ctx_hat[tr + c];           // fc input (:990)
        }
    }
}

// AdaptiveAttention.forward (models/gridTDmodel.py:71-103), two kernels so that B images fill the chip:
//  (1) scores: grid (B, pixel blocks): h_proj[k] = W_g[k].h1, s_proj[k] = W_s[k].s + b_s[k],
//      z[k] = w_h . tanh(att_img[k,:] + h_proj[k])   (ht_proj is broadcast along the row, :82);
//      att_img = W_v_proj(V)+b_v is time-invariant and precomputed.
//  (2) context: grid (B, channel blocks): sentinel score, both softmaxes, context, c_hat and
//      xh2[b,t] = [ctx_hat | h1_new | h2_old].
// (latency-bound at B = 16: with 28 pixels / 128 channels per block the two kernels ran 30 + 29 us per time step on 112 / 64
// workgroups, each wave walking 7 pixels or each thread 98 pixels in sequence; now one pixel per wave and 8 pixel
// slices per channel)
constexpr int ATT_PB = 4;     // pixels per block in (1): one per wave
constexpr int ATT_CB = 32;    // channels per block in (2): 8 pixel slices x 32 channels per block

__global__ __launch_bounds__(256) void gridtd_fwd_att_scores_kernel(
    GridFwd g, int t, const float* __restrict__ att_img, const float* __restrict__ Wg, const float* __restrict__ Ws,
    const float* __restrict__ bs, const float* __restrict__ wh, float* __restrict__ scr, const float* __restrict__ sg_in) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, H = g.H, P = g.P, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* h1n = sm;          // H
    float* sv = h1n + H;      // H
    const long st1 = ((long)b * (g.T + 1) + t + 1) * H, tr = ((long)b * g.T + t) * H;
    if (sg_in) {
        // fused step (gridtd_linear_lstm_kernel): the sentinel s = sigmoid(gate) * tanh(c1) (:982-983) is formed here from the gate the
        // gate linear left in sg_in [B][H] and the cell state it wrote - the expression of gridtd_fwd_lstm_kernel on the same values;
        // the first pixel block keeps it for the trace
        for (int c = tid; c < H; c += 256) {
            const float sgv = sg_in[(long)b * H + c];
            const float s = sgv * tanhf(g.c1[st1 + c]);
            h1n[c] = g.h1[st1 + c]; sv[c] = s;
            if (blockIdx.y == 0) { g.s[tr + c] = s; if (g.sgate) g.sgate[tr + c] = sgv; }
        }
    } else {
        for (int c = tid; c < H; c += 256) { h1n[c] = g.h1[st1 + c]; sv[c] = g.s[tr + c]; }
    }
    __syncthreads();
    float* zsc = scr + (long)b * 3 * P;       // [z | h_proj | s_proj]
    const int k0 = blockIdx.y * ATT_PB, k1 = min(k0 + ATT_PB, P);
    for (int k = k0 + wv; k < k1; k += 4) {
        float a = 0.f, c2 = 0.f;
        for (int c = lane; c < H; c += 64) { a += Wg[(long)k * H + c] * h1n[c]; c2 += Ws[(long)k * H + c] * sv[c]; }
        a = wave_sum(a); c2 = wave_sum(c2);
        float z = 0.f;
        for (int j = lane; j < P; j += 64) z += wh[j] * tanhf(att_img[((long)b * P + k) * P + j] + a);
        z = wave_sum(z);
        if (lane == 0) { zsc[k] = z; zsc[P + k] = a; zsc[2 * P + k] = c2 + bs[k]; }
    }
}

__global__ __launch_bounds__(256) void gridtd_fwd_att_context_kernel(GridFwd g, int t, const float* __restrict__ Vp,
                                                                     const float* __restrict__ wh,
                                                                     const float* __restrict__ scr) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, H = g.H, P = g.P, tid = threadIdx.x;
    float* zsc = sm;            // P+1
    float* alpha = zsc + P + 1; // P
    float* red = zsc + (2 * P + 1 > 256 ? 2 * P + 1 : 256);     // 8 (behind the 256 partial sums that reuse zsc / alpha)
    const float* sc = scr + (long)b * 3 * P;
    for (int k = tid; k < P; k += 256) zsc[k] = sc[k];
    {   // sentinel score = w_h . tanh(s_proj + h_proj)   (:94)
        float a = 0.f;
        for (int j = tid; j < P; j += 256) a += wh[j] * tanhf(sc[2 * P + j] + sc[P + j]);
        a = block_sum(a, red);
        if (tid == 0) zsc[P] = a;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int k = tid; k < P; k += 256) m = fmaxf(m, zsc[k]);
    m = block_max(m, red);
    float e = 0.f;
    for (int k = tid; k < P; k += 256) e += expf(zsc[k] - m);
    const float denom = block_sum(e, red);
    const float m2 = fmaxf(m, zsc[P]);
    float e2 = 0.f;
    for (int k = tid; k <= P; k += 256) e2 += expf(zsc[k] - m2);
    const float denom2 = block_sum(e2, red);
    const float beta = expf(zsc[P] - m2) / denom2;
    __syncthreads();
    for (int k = tid; k < P; k += 256) {
        const float a = expf(zsc[k] - m) / denom;
        alpha[k] = a;
        if (blockIdx.y == 0) g.alpha[((long)b * g.T + t) * P + k] = a;
    }
    if (tid == 0 && blockIdx.y == 0) g.beta[(long)b * g.T + t] = beta;
    __syncthreads();
    // context for this block's 32 channels: thread = (pixel slice tid / 32, channel tid % 32) - 128-byte runs per slice -
    // partial sums through LDS, summed in slice order
    const int c = blockIdx.y * ATT_CB + (tid & 31), part = tid >> 5;
    float a = 0.f;
    if (c < H)
        for (int k = part; k < P; k += 8) a += Vp[((long)b * P + k) * H + c] * alpha[k];
    float* psum = zsc;                     // (P + 1 >= 8 * 32 floats are not needed any more: alpha is its own array)
    __syncthreads();
    psum[tid] = a;
    __syncthreads();
    if (part == 0) {
        a = psum[tid];
#pragma unroll
        for (int q = 1; q < 8; ++q) a += psum[q * 32 + tid];
    }
    if (c < H && part == 0) {
        const long tr = ((long)b * g.T + t) * H, st0 = ((long)b * (g.T + 1) + t) * H;
        const float sv = g.s[tr + c], h1n = g.h1[st0 + H + c];
        const float ch = beta * sv + (1.f - beta) * a;
        float* x2 = g.xh2 + ((long)b * g.T + t) * 3 * H;
        g.ctx[tr + c] = a; g.ctx_hat[tr + c] = ch;
        x2[c] = ch; x2[H + c] = h1n; x2[2 * H + c] = g.h2[st0 + c];
    }
}

// logit of the target word: dot(fc.weight[k], hc[b,t]) + fc.bias[k]   (one wave per row)
__global__ void target_logit_kernel(const float* __restrict__ hc, const float* __restrict__ fcw,
                                    const float* __restrict__ fcb, const long long* __restrict__ tok, int tok_ld,
                                    float* __restrict__ logit, int B, int T, int H) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= B * T) return;
    const int b = row / T, t = row - b * T, lane = threadIdx.x & 63;
    const long long k = tok[(long)b * tok_ld + t + 1];
    float a = 0.f;
    for (int c = lane; c < H; c += 64) a += fcw[k * H + c] * hc[(long)row * H + c];
    a = wave_sum(a);
    if (lane == 0) logit[row] = a + fcb[k];
}

// argmax over the vocabulary (first maximum wins, like torch.argmax / topk(1)) — one block per row
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, long ld, int n,
                                                          long long* __restrict__ out) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    const float* r = x + (long)blockIdx.x * ld;
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int i = threadIdx.x; i < n; i += 256) { const float v = r[i]; if (v > best) { best = v; idx = i; } }
    bv[threadIdx.x] = best; bi[threadIdx.x] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const float v = bv[threadIdx.x + s]; const int i2 = bi[threadIdx.x + s];
            if (v > bv[threadIdx.x] || (v == bv[threadIdx.x] && i2 < bi[threadIdx.x])) { bv[threadIdx.x] = v; bi[threadIdx.x] = i2; }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[blockIdx.x] = bi[0];
}

// ------------------------------------------------------------------------------------------------
// LRP-inference decoding (models/gridTDmodel.py:548-577, :631-702): per step the predicted word's logit is
// redistributed to the two fc summands, both relevance vectors become weights around 1, and the logits are
// recomputed from the re-weighted fc input.
// ------------------------------------------------------------------------------------------------
// xg[b] = [h2_old | glob | emb | h1_NEW]: the sentinel gate of sample_lrp sees the new h1 (:672), not h_{t-1}
__global__ void gridtd_gate_input_kernel(GridFwd g, int t, float* __restrict__ xg) {
    const int b = blockIdx.x;
    const int W = 2 * g.E + 2 * g.H;
    const float* src = g.xh1 + ((long)b * g.T + t) * W;
    const float* h1n = g.h1 + ((long)b * (g.T + 1) + t + 1) * g.H;
    for (int c = threadIdx.x; c < W; c += blockDim.x) xg[(long)b * W + c] = c < W - g.H ? src[c] : h1n[c - (W - g.H)];
}

// s = sigmoid(gate) * tanh(c1_new)   (:672-673)
__global__ void gridtd_sentinel_kernel(GridFwd g, int t, const float* __restrict__ zg, int ldz) {
    const int b = blockIdx.x;
    const long st1 = ((long)b * (g.T + 1) + t + 1) * g.H, tr = ((long)b * g.T + t) * g.H;
    for (int c = threadIdx.x; c < g.H; c += blockDim.x) {
        const float sg = sigmoidf_(zg[(long)b * ldz + c]);
        g.s[tr + c] = sg * tanhf(g.c1[st1 + c]);
        if (g.sgate) g.sgate[tr + c] = sg;
    }
}

// get_lrp_weight_step (:548-577) + the re-weighted fc input (:687).  One block per image, H = 512, 256 threads.
// h / ctx: the two fc summands of row b (row strides ldh / ldc).  lsm: the AoA model hands the rule the log-softmax of
// the scores instead of the scores (models/aoamodel.py:721-723): relevance and stabilised output are log p(k).
__global__ __launch_bounds__(256) void lrp_reweight_kernel(const float* __restrict__ pred, long ld, int V,
                                                           const float* __restrict__ hrow, long ldh,
                                                           const float* __restrict__ crow, long ldc,
                                                           const float* __restrict__ fcw,
                                                           const unsigned char* __restrict__ skip,
                                                           float* __restrict__ hcw, int H, int lsm) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    __shared__ float red[8];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* r = pred + (long)b * ld;
    const float* hp = hrow + (long)b * ldh;
    const float* cp = crow + (long)b * ldc;
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int i = tid; i < V; i += 256) { const float v = r[i]; if (v > best) { best = v; idx = i; } }
    bv[tid] = best; bi[tid] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float v = bv[tid + s]; const int i2 = bi[tid + s];
            if (v > bv[tid] || (v == bv[tid] && i2 < bi[tid])) { bv[tid] = v; bi[tid] = i2; }
        }
        __syncthreads();
    }
    const int k = min(bi[0], V - 1);            // (all-NaN row: keep the index in range)
    float pk = bv[0];
    float* out = hcw + (long)b * H;
    if (skip[k]) {                                 // stop words and specials: both weights are 1 (:557-558)
        for (int c = tid; c < H; c += 256) out[c] = cp[c] * 1.f + 1.f * hp[c];
        return;
    }
    if (lsm) {                                     // log_softmax(pred)[k] = -log sum exp(pred - max)
        float se = 0.f;
        for (int i = tid; i < V; i += 256) se += expf(r[i] - pk);
        se = block_sum(se, red);
        pk = -logf(se);
        __syncthreads();
    }
    const float zt = stab_eps(pk);
    float rh[2], rc[2], mh = 0.f, mc = 0.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = tid + j * 256;
        const float h2 = hp[c], ch = cp[c];
        const float hc = h2 + ch;
        const float r_hc = (fcw[(long)k * H + c] * hc / zt) * pk;     // one-hot epsilon rule through fc
        rh[j] = eps_id(r_hc, h2, hc);
        rc[j] = eps_id(r_hc, ch, hc);
        mh = fmaxf(mh, fabsf(rh[j])); mc = fmaxf(mc, fabsf(rc[j]));
    }
    mh = block_max(mh, red);
    __syncthreads();
    mc = block_max(mc, red);
    if (mh == 0.f) mh = 1.f;                       // LRPtools/utils.py:59
    if (mc == 0.f) mc = 1.f;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int c = tid + j * 256;
        const float w_h = rh[j] / mh + 1.f, w_c = rc[j] / mc + 1.f;
        out[c] = cp[c] * w_c + w_h * hp[c];
    }
}

// argmax of log_softmax(x) and its value (sample_next_word greedy, :522-526): one block per row
__global__ __launch_bounds__(256) void argmax_logprob_rows_kernel(const float* __restrict__ x, long ld, int n,
                                                                  long long* __restrict__ out, float* __restrict__ lp) {
    __shared__ float bv[256];
    __shared__ int bi[256];
    __shared__ float red[8];
    const int tid = threadIdx.x;
    const float* r = x + (long)blockIdx.x * ld;
    float best = -INFINITY; int idx = 0x7fffffff;
    for (int i = tid; i < n; i += 256) { const float v = r[i]; if (v > best) { best = v; idx = i; } }
    bv[tid] = best; bi[tid] = idx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float v = bv[tid + s]; const int i2 = bi[tid + s];
            if (v > bv[tid] || (v == bv[tid] && i2 < bi[tid])) { bv[tid] = v; bi[tid] = i2; }
        }
        __syncthreads();
    }
    const float mx = bv[0];
    float se = 0.f;
    for (int i = tid; i < n; i += 256) se += expf(r[i] - mx);
    se = block_sum(se, red);
    if (tid == 0) { out[blockIdx.x] = min(bi[0], n - 1); lp[blockIdx.x] = -logf(se); }   // x[k] - mx = 0
}


// Beam-search step (GridTDModel.beam_search, models/gridTDmodel.py:437-444; AOAModel.beam_search): the k best of
// cum[r] + log_softmax(x[r])[w] over the n_rows live beams (flat index r * n + w, value).  One workgroup: row maxima and
// log-sum-exps, a per-thread top-k (k <= 4) over the flat range, then k rounds of a block-wide arg-max over the list heads
// (1024 threads: one workgroup has to walk n_rows x vocab scores three times, so the width of the group is the speed).
// Ties: lower flat index first.
template <int KMAX>
__global__ __launch_bounds__(1024) void beam_topk_kernel(const float* __restrict__ x, long ld, int n_rows, int n,
                                                         const float* __restrict__ cum, int k,
                                                         long long* __restrict__ out_idx, float* __restrict__ out_val) {
    constexpr int NT = 1024, NW = NT / 64;
    __shared__ float red[NW];
    __shared__ float lse[8];
    __shared__ float cv[NW];
    __shared__ int ci[NW];
    const int tid = threadIdx.x;
    for (int r = 0; r < n_rows; ++r) {
        const float* xr = x + (long)r * ld;
        float m = -INFINITY;
        for (int i = tid; i < n; i += NT) m = fmaxf(m, xr[i]);
        m = wave_max(m);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = m;
        __syncthreads();
        m = red[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) m = fmaxf(m, red[w]);
        float se = 0.f;
        for (int i = tid; i < n; i += NT) se += expf(xr[i] - m);
        se = wave_sum(se);
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6] = se;
        __syncthreads();
        if (tid == 0) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[w];
            lse[r] = m + logf(t);
        }
    }
    __syncthreads();
    float bv[KMAX];
    int bi[KMAX];
#pragma unroll
    for (int j = 0; j < KMAX; ++j) { bv[j] = -INFINITY; bi[j] = 0x7fffffff; }
    for (int r = 0; r < n_rows; ++r) {
        const float off = (cum ? cum[r] : 0.f), l = lse[r];
        const float* xr = x + (long)r * ld;
        for (int w = tid; w < n; w += NT) {
            float v = off + (xr[w] - l);
            int vi = r * n + w;
#pragma unroll
            for (int j = 0; j < KMAX; ++j) {          // insertion into the sorted list (descending value, ascending index)
                const bool better = v > bv[j] || (v == bv[j] && vi < bi[j]);
                const float tv = better ? bv[j] : v; const int ti = better ? bi[j] : vi;
                bv[j] = better ? v : bv[j]; bi[j] = better ? vi : bi[j];
                v = tv; vi = ti;
            }
        }
    }
    // merge: k rounds of a block-wide arg-max over the heads of the 1024 sorted lists (value descending, flat index ascending on ties)
    int head = 0;
    for (int o = 0; o < k; ++o) {
        float v = -INFINITY; int vi = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < KMAX; ++j) if (j == head) { v = bv[j]; vi = bi[j]; }
        float wv = v; int wi = vi;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(wv, off, 64); const int oi = __shfl_xor(wi, off, 64);
            const bool take = ov > wv || (ov == wv && oi < wi);
            wv = take ? ov : wv; wi = take ? oi : wi;
        }
        __syncthreads();                               // (cv / ci of the previous round have been read)
        if ((tid & 63) == 0) { cv[tid >> 6] = wv; ci[tid >> 6] = wi; }
        __syncthreads();
        float best = cv[0]; int bidx = ci[0];
#pragma unroll
        for (int w = 1; w < NW; ++w) {
            const bool take = cv[w] > best || (cv[w] == best && ci[w] < bidx);
            best = take ? cv[w] : best; bidx = take ? ci[w] : bidx;
        }
        if (bidx != 0x7fffffff && vi == bidx) ++head;                       // the owner of the winner moves on to its next candidate
        if (tid == 0) { out_idx[o] = bidx != 0x7fffffff ? bidx : 0; out_val[o] = best; }
    }
}

// ------------------------------------------------------------------------------------------------
// gridTD relevance — models/gridTDmodel.py:1014-1135.  One block (256 threads) per row, H = 512.
// ------------------------------------------------------------------------------------------------
struct GridRel {
    int B, T, H, E, P;
    const int* lens;                      // [B] caption length per image (null: T)
    // trace (const)
    const float *xh1, *xh2, *h2, *c1, *c2, *g1, *i1, *f1, *g2, *i2, *f2, *s, *ctx, *ctx_hat, *hc, *beta;
    // state per row
    float *r_h2n, *r_c2, *r_c1, *r_ch0, *r_h2p, *r_glob;   // [rows][H]   (r_glob: [rows][E])
    float *A, *rx;                                          // GEMM in [rows][H], GEMM out [rows][<=2E+2H]
    float *wacc;                                            // [rows][T][H]  r_ctx / z~(ctx)
    float *r_words;                                         // [rows][T]
};

__device__ __forceinline__ bool row_active(const GridRel& g, int b, int t, int s) {
    const int len = g.lens ? g.lens[b] : g.T;
    return t < len && t >= s;
}

// :1033-1059 — one-hot relevance at the target logit through fc, split between h2 and ctx_hat
__global__ void gridtd_rel_init_kernel(GridRel g, const float* __restrict__ fcw, const float* __restrict__ logit,
                                       const long long* __restrict__ tok, int tok_ld) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H;
    const long long k = tok[(long)b * tok_ld + t + 1];
    const float lg = logit[row];
    const float zt = stab_eps(lg);
    const long tr = (long)row * H, st1 = ((long)b * (g.T + 1) + t + 1) * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float hc = g.hc[tr + c];
        const float r_hc = (fcw[k * H + c] * hc / zt) * lg;
        g.r_h2n[tr + c] = eps_id(r_hc, g.h2[st1 + c], hc);
        g.r_ch0[tr + c] = eps_id(r_hc, g.ctx_hat[tr + c], hc);
        g.r_c2[tr + c] = 0.f; g.r_c1[tr + c] = 0.f;
    }
    for (int c = threadIdx.x; c < g.E; c += blockDim.x) g.r_glob[(long)row * g.E + c] = 0.f;
    for (int c = threadIdx.x; c < g.T; c += blockDim.x) g.r_words[(long)row * g.T + c] = 0.f;
}

// :1061-1069 LanguageLSTM cell: r_c2 += r_h2; split into the g-gate path and the c-path; A = r_g2 / z~(g2)
__global__ void gridtd_rel_a_kernel(GridRel g, int s) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H;
    const long tr = (long)row * H;
    if (!row_active(g, b, t, s)) {
        for (int c = threadIdx.x; c < H; c += blockDim.x) g.A[tr + c] = 0.f;
        return;
    }
    const int i = t - s;
    const long ti = ((long)b * g.T + i) * H, sc1 = ((long)b * (g.T + 1) + i + 1) * H, sc0 = sc1 - H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float rc = g.r_c2[tr + c] + g.r_h2n[tr + c];
        const float cn = g.c2[sc1 + c];
        const float rg = eps_id(rc, g.i2[ti + c] * tanhf(g.g2[ti + c]), cn);
        g.r_c2[tr + c] = eps_id(rc, g.f2[ti + c] * g.c2[sc0 + c], cn);
        g.A[tr + c] = rg / stab_eps(g.g2[ti + c]);
    }
}

// :1074-1105 after the LanguageLSTM dense rule: split r_xh2, sentinel/context split, AdaLSTM cell
__global__ void gridtd_rel_b_kernel(GridRel g, int s) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H;
    const long tr = (long)row * H;
    if (!row_active(g, b, t, s)) {
        for (int c = threadIdx.x; c < H; c += blockDim.x) g.A[tr + c] = 0.f;
        return;
    }
    const int i = t - s;
    const long ti = ((long)b * g.T + i) * H, sc1 = ((long)b * (g.T + 1) + i + 1) * H, sc0 = sc1 - H;
    const float* rx = g.rx + (long)row * (3 * H);                 // r_xh2 = [ctx_hat | h1 | h2], row stride 3H
    const float beta = g.beta[(long)b * g.T + i];
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        g.r_h2p[tr + c] = rx[2 * H + c];                           // :1074
        const float r_h1 = rx[H + c];                              // :1075
        const float r_ch = (s == 0 ? g.r_ch0[tr + c] : 0.f) + rx[c];   // :1076
        const float ch = g.ctx_hat[ti + c], cx = g.ctx[ti + c];
        const float r_s = eps_id(r_ch, beta * g.s[ti + c], ch);    // :1077-1080
        const float r_cx = eps_id(r_ch, cx * (1.f - beta), ch);    // :1081-1084
        g.wacc[((long)row * g.T + i) * H + c] = r_cx / stab_eps(cx);   // pixel spread (:1091-1095) is finished in rel_pix
        float rc = g.r_c1[tr + c] + r_s;                           // :1096
        rc = rc + r_h1;                                            // :1097
        const float cn = g.c1[sc1 + c];
        const float rg = eps_id(rc, g.i1[ti + c] * tanhf(g.g1[ti + c]), cn);
        g.r_c1[tr + c] = eps_id(rc, g.f1[ti + c] * g.c1[sc0 + c], cn);
        g.A[tr + c] = rg / stab_eps(g.g1[ti + c]);
    }
}

// :1110-1115 after the AdaLSTM dense rule: r_xh1 = [h2 | glob | emb | h1]
__global__ __launch_bounds__(256) void gridtd_rel_c_kernel(GridRel g, int s) {
    __shared__ float red[8];
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H, E = g.E;
    if (!row_active(g, b, t, s)) return;
    const int i = t - s;
    const long tr = (long)row * H;
    const float* rx = g.rx + (long)row * (2 * E + 2 * H);
    for (int c = threadIdx.x; c < H; c += 256) g.r_h2n[tr + c] = g.r_h2p[tr + c] + rx[c];           // :1111
    float acc = 0.f;
    for (int c = threadIdx.x; c < E; c += 256) {
        g.r_glob[(long)row * E + c] += rx[H + c];                                                    // :1114
        acc += rx[H + E + c];                                                                        // :1115, :1129
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) g.r_words[(long)row * g.T + i] = acc;
}

// gridtd_rel_c_kernel of lock-step s and gridtd_rel_a_kernel of lock-step s + 1 in one launch (same expressions, same order: a thread owns
// the same channels c in both; the r_h2n it writes is the one it reads).  One launch less per lock-step: they are ~5 us dependent launches.
__global__ __launch_bounds__(256) void gridtd_rel_ca_kernel(GridRel g, int s) {
    __shared__ float red[8];
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H, E = g.E;
    const long tr = (long)row * H;
    if (!row_active(g, b, t, s)) {                 // (then not active at s + 1 either)
        for (int c = threadIdx.x; c < H; c += 256) g.A[tr + c] = 0.f;
        return;
    }
    const int i = t - s;
    const float* rx = g.rx + (long)row * (2 * E + 2 * H);
    const bool nxt = row_active(g, b, t, s + 1);
    const int i1 = i - 1;                          // time step of lock-step s + 1
    const long ti = ((long)b * g.T + i1) * H, sc1 = ((long)b * (g.T + 1) + i1 + 1) * H, sc0 = sc1 - H;
    for (int c = threadIdx.x; c < H; c += 256) {
        const float r_h2n = g.r_h2p[tr + c] + rx[c];                                                  // :1111
        g.r_h2n[tr + c] = r_h2n;
        if (nxt) {                                                                                    // :1061-1069 at s + 1
            const float rc = g.r_c2[tr + c] + r_h2n;
            const float cn = g.c2[sc1 + c];
            const float rg = eps_id(rc, g.i2[ti + c] * tanhf(g.g2[ti + c]), cn);
            g.r_c2[tr + c] = eps_id(rc, g.f2[ti + c] * g.c2[sc0 + c], cn);
            g.A[tr + c] = rg / stab_eps(g.g2[ti + c]);
        } else {
            g.A[tr + c] = 0.f;
        }
    }
    float acc = 0.f;
    for (int c = threadIdx.x; c < E; c += 256) {
        g.r_glob[(long)row * E + c] += rx[H + c];                                                    // :1114
        acc += rx[H + E + c];                                                                        // :1115, :1129
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) g.r_words[(long)row * g.T + i] = acc;
}

// :1116-1119 prologue: A = r_glob / z~(glob_pre)
__global__ void gridtd_rel_glob_kernel(GridRel g, const float* __restrict__ glob_pre, float* __restrict__ Aglob) {
    const int row = blockIdx.x, b = row / g.T;
    for (int c = threadIdx.x; c < g.E; c += blockDim.x)
        Aglob[(long)row * g.E + c] = g.r_glob[(long)row * g.E + c] / stab_eps(glob_pre[(long)b * g.E + c]);
}

// U[row][c] = r_avg / (P * z~(avg))  — the eye rule of :1121-1124 folded into the projector GEMM's bracket
__global__ void gridtd_rel_u_kernel(const float* __restrict__ r_avg, const float* __restrict__ avg,
                                    float* __restrict__ U, int T, int C, int P) {
    const int row = blockIdx.x, b = row / T;
    for (int c = threadIdx.x; c < C; c += blockDim.x)
        U[(long)row * C + c] = r_avg[(long)row * C + c] / ((float)P * stab_eps(avg[(long)b * C + c]));
}

// pixel spread :1091-1095 summed over i, and the prologue of the projector rule (:1125-1128):
// Aproj[row][k][c] = Vp[b][k][c] * sum_{i<=t} alpha[b][i][k] * wacc[row][i][c] / z~(proj_pre[b][k][c])
__global__ __launch_bounds__(256) void gridtd_rel_pix_kernel(GridRel g, const float* __restrict__ Vp,
                                                             const float* __restrict__ proj_pre,
                                                             const float* __restrict__ alpha,
                                                             float* __restrict__ Aproj, int kchunk,
                                                             const int* __restrict__ rowlist) {
    // rowlist (variable caption lengths): workgroup x computes row rowlist[x] and writes it as row x of a COMPACT Aproj
    const int row = rowlist ? rowlist[blockIdx.x] : blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H, P = g.P;
    const long orow = blockIdx.x;
    const int len = g.lens ? g.lens[b] : g.T;
    const int k0 = blockIdx.y * kchunk, k1 = min(k0 + kchunk, P);
    extern __shared__ float al[];    // [t+1][kchunk]
    const bool act = t < len;
    const int n = act ? t + 1 : 0;
    for (int j = threadIdx.x; j < n * kchunk; j += 256) {
        const int i = j / kchunk, kk = j - i * kchunk;
        al[j] = (k0 + kk < P) ? alpha[((long)b * g.T + i) * P + k0 + kk] : 0.f;
    }
    __syncthreads();
    constexpr int TREG = 24;         // captions up to 24 words: the row's accumulated weights stay in registers
    for (int c = threadIdx.x; c < H; c += 256) {
        float wr[TREG];
        if (n <= TREG) {
#pragma unroll
            for (int i = 0; i < TREG; ++i) wr[i] = i < n ? g.wacc[((long)row * g.T + i) * H + c] : 0.f;
        }
        for (int k = k0; k < k1; ++k) {
            float a = 0.f;
            if (n <= TREG) {         // (same summation order as the loop below: i = n-1 .. 0; the tail adds exact zeros)
#pragma unroll
                for (int i = TREG - 1; i >= 0; --i) a += (i < n ? al[i * kchunk + (k - k0)] : 0.f) * wr[i];
            } else {
                for (int i = n - 1; i >= 0; --i) a += al[i * kchunk + (k - k0)] * g.wacc[((long)row * g.T + i) * H + c];
            }
            const long pi = ((long)b * P + k) * H + c;
            Aproj[(orow * P + k) * H + c] = act ? Vp[pi] * a / stab_eps(proj_pre[pi]) : 0.f;
        }
    }
}

// :1129-1132  r_words / max|r_words|
__global__ void rel_words_norm_kernel(float* __restrict__ r_words, int rows, int T) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int t = row % T;
    float m = 0.f;
    for (int i = 0; i <= t; ++i) m = fmaxf(m, fabsf(r_words[(long)row * T + i]));
    if (m > 0.f)
        for (int i = 0; i <= t; ++i) r_words[(long)row * T + i] /= m;
}

// ================================================================================================
// gridTD guided-backprop decoder: hand-written BPTT with alpha/beta constant (models/gridTDmodel.py:1588-1675)
// ================================================================================================
struct GridGrad {
    int B, T, H, E, P;
    const int* lens;
    const float *c1, *c2, *g1, *i1, *f1, *o1, *g2, *i2, *f2, *o2, *sgate, *beta;   // g* are pre-activations
    float *d_h2n, *d_c2, *d_c1, *d_ch0, *d_h2p, *d_glob;    // [rows][H] (d_glob: [rows][E])
    float *gates, *dx;                                       // GEMM in [rows][4H], out [rows][3H]
    float *wacc, *r_words;                                   // [rows][T][H], [rows][T]
};

__device__ __forceinline__ bool grad_row_active(const GridGrad& g, int b, int t, int s) {
    const int len = g.lens ? g.lens[b] : g.T;
    return t < len && t >= s;
}

// :1592-1625  d_word_pred one-hot = 1  ->  d(h2+ctx_hat) = fc.weight[k]
__global__ void gridtd_grad_init_kernel(GridGrad g, const float* __restrict__ fcw, const long long* __restrict__ tok,
                                        int tok_ld) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H;
    const long long k = tok[(long)b * tok_ld + t + 1];
    const long tr = (long)row * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float d = fcw[k * H + c];
        g.d_h2n[tr + c] = d; g.d_ch0[tr + c] = d; g.d_c2[tr + c] = 0.f; g.d_c1[tr + c] = 0.f;
    }
    for (int c = threadIdx.x; c < g.E; c += blockDim.x) g.d_glob[(long)row * g.E + c] = 0.f;
    for (int c = threadIdx.x; c < g.T; c += blockDim.x) g.r_words[(long)row * g.T + c] = 0.f;
}

// LSTM cell backward (:1627-1637 / :1647-1657): dh, running dc -> gate pre-activation gradients [i|f|g|o]
__device__ __forceinline__ void lstm_cell_bwd(float dh, float& dc, float c_new, float c_old, float i, float f, float gpre,
                                              float o, float* gates, int H, int c) {
    const float tc = tanhf(c_new), ga = tanhf(gpre);
    const float d_oa = dh * tc;
    dc = dc + dh * o * (1.f - tc * tc);
    const float d_fa = dc * c_old, d_ia = dc * ga, d_ga = dc * i;
    gates[c] = d_ia * i * (1.f - i);
    gates[H + c] = d_fa * f * (1.f - f);
    gates[2 * H + c] = d_ga * (1.f - ga * ga);
    gates[3 * H + c] = d_oa * o * (1.f - o);
    dc = dc * f;      // d_c[i] = d_c[i+1] * f   (:1630 / :1650)
}

__global__ void gridtd_grad_a_kernel(GridGrad g, int s) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H;
    float* gates = g.gates + (long)row * 4 * H;
    if (!grad_row_active(g, b, t, s)) {
        for (int c = threadIdx.x; c < 4 * H; c += blockDim.x) gates[c] = 0.f;
        return;
    }
    const int i = t - s;
    const long tr = (long)row * H, ti = ((long)b * g.T + i) * H, sc1 = ((long)b * (g.T + 1) + i + 1) * H, sc0 = sc1 - H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        float dc = g.d_c2[tr + c];
        lstm_cell_bwd(g.d_h2n[tr + c], dc, g.c2[sc1 + c], g.c2[sc0 + c], g.i2[ti + c], g.f2[ti + c], g.g2[ti + c],
                      g.o2[ti + c], gates, H, c);
        g.d_c2[tr + c] = dc;
    }
}

// dx = gates2 @ [W_ih | W_hh] = [d_ctx_hat | d_h1 | d_h2]   (:1638-1646), then the AdaLSTM cell backward
__global__ void gridtd_grad_b_kernel(GridGrad g, int s) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H;
    float* gates = g.gates + (long)row * 4 * H;
    if (!grad_row_active(g, b, t, s)) {
        for (int c = threadIdx.x; c < 4 * H; c += blockDim.x) gates[c] = 0.f;
        return;
    }
    const int i = t - s;
    const long tr = (long)row * H, ti = ((long)b * g.T + i) * H, sc1 = ((long)b * (g.T + 1) + i + 1) * H, sc0 = sc1 - H;
    const float* dx = g.dx + (long)row * 3 * H;
    const float beta = g.beta[(long)b * g.T + i];
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        g.d_h2p[tr + c] = dx[2 * H + c];                                    // :1638
        const float d_ch = (s == 0 ? g.d_ch0[tr + c] : 0.f) + dx[c];        // :1640
        g.wacc[((long)row * g.T + i) * H + c] = d_ch * (1.f - beta);        // d_context (:1641), spread in rel_pix
        const float d_s = d_ch * beta;                                      // :1644
        const float tc1 = tanhf(g.c1[sc1 + c]);
        float dc = g.d_c1[tr + c] + d_s * g.sgate[ti + c] * (1.f - tc1 * tc1);   // :1645
        lstm_cell_bwd(dx[H + c], dc, g.c1[sc1 + c], g.c1[sc0 + c], g.i1[ti + c], g.f1[ti + c], g.g1[ti + c],
                      g.o1[ti + c], gates, H, c);                           // :1646-1657
        g.d_c1[tr + c] = dc;
    }
}

// dx = gates1 @ W_ih = [d_h2 | d_glob | d_emb]  (:1659-1662)
__global__ __launch_bounds__(256) void gridtd_grad_c_kernel(GridGrad g, int s) {
    __shared__ float red[8];
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H, E = g.E;
    if (!grad_row_active(g, b, t, s)) return;
    const int i = t - s;
    const long tr = (long)row * H;
    const float* dx = g.dx + (long)row * 3 * H;      // row stride of the GEMM output buffer (H + 2E == 3H)
    for (int c = threadIdx.x; c < H; c += 256) g.d_h2n[tr + c] = g.d_h2p[tr + c] + dx[c];
    float acc = 0.f;
    for (int c = threadIdx.x; c < E; c += 256) {
        g.d_glob[(long)row * E + c] += dx[H + c];
        acc += dx[H + E + c];
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) g.r_words[(long)row * g.T + i] = acc;
}

// Aproj[row][k][c] = sum_{i<=t} alpha[b][i][k] * wacc[row][i][c]   (:1642-1643 summed over i)
__global__ __launch_bounds__(256) void spread_pixels_kernel(const float* __restrict__ wacc, const float* __restrict__ alpha,
                                                            const int* __restrict__ lens, float* __restrict__ Aproj,
                                                            int T, int H, int P, int kchunk,
                                                            const int* __restrict__ rowlist) {
    const int row = rowlist ? rowlist[blockIdx.x] : blockIdx.x, b = row / T, t = row - b * T;
    const long orow = blockIdx.x;
    const int len = lens ? lens[b] : T;
    const int k0 = blockIdx.y * kchunk, k1 = min(k0 + kchunk, P);
    extern __shared__ float al[];
    const int n = t < len ? t + 1 : 0;
    for (int j = threadIdx.x; j < n * kchunk; j += 256) {
        const int i = j / kchunk, kk = j - i * kchunk;
        al[j] = (k0 + kk < P) ? alpha[((long)b * T + i) * P + k0 + kk] : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < H; c += 256)
        for (int k = k0; k < k1; ++k) {
            float a = 0.f;
            for (int i = n - 1; i >= 0; --i) a += al[i * kchunk + (k - k0)] * wacc[((long)row * T + i) * H + c];
            Aproj[(orow * P + k) * H + c] = a;
        }
}

__global__ void scale_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float alpha) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] * alpha;
}

__global__ void positive_mask_kernel(const float* __restrict__ x, float* __restrict__ y, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = x[i] > 0.f ? 1.f : 0.f;
}

// ================================================================================================
// AoA decoder (models/aoamodel.py): trace :990-1062, relevance :1064-1156
// ================================================================================================
struct AoaFwd {
    int B, T, H, E, P, NH;
    float *xh;                   // [B][T][E+2H] = [emb | glob | h_t]          (xt ++ ht[:T], :1075)
    float *h, *c;                // [B][T+1][H]
    float *g, *i, *f;            // [B][T][H]
    float *ctx, *lin, *c_aoa, *hc;   // [B][T][H]  context, decoder_aoa_linear(context), gated, fc input
    float *alpha;                // [B][T][NH][P]
    float *o, *sg;               // optional: [B][T][H] output gate, sigmoid(aoa gate)   (gradient explainers)
};

// xh[b,t] = [emb[tok[b,t]] | glob[b] | h[b,t]]   (:1030, :1075)
__global__ void aoa_fwd_pre_kernel(AoaFwd g, int t, const float* __restrict__ glob, const float* __restrict__ emb,
                                   const long long* __restrict__ tok, int tok_ld) {
    const int b = blockIdx.x, W = g.E + 2 * g.H;
    float* dst = g.xh + ((long)b * g.T + t) * W;
    const long st = ((long)b * (g.T + 1) + t) * g.H;
    const long long k = tok[(long)b * tok_ld + t];
    for (int c = threadIdx.x; c < W; c += blockDim.x) {
        float v;
        if (c < g.E) v = emb[k * g.E + c];
        else if (c < g.E + g.H) v = glob[(long)b * g.H + (c - g.E)];
        else v = g.h[st + (c - g.E - g.H)];
        dst[c] = v;
    }
}

__global__ void aoa_fwd_lstm_kernel(AoaFwd g, int t, const float* __restrict__ zz, int ldz) {
    const int b = blockIdx.x, H = g.H;
    const long st0 = ((long)b * (g.T + 1) + t) * H, st1 = st0 + H, tr = ((long)b * g.T + t) * H;
    const float* z = zz + (long)b * ldz;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float i = sigmoidf_(z[c]), f = sigmoidf_(z[H + c]), zg = z[2 * H + c], o = sigmoidf_(z[3 * H + c]);
        const float cn = f * g.c[st0 + c] + i * tanhf(zg);
        g.c[st1 + c] = cn; g.h[st1 + c] = o * tanhf(cn);
        g.g[tr + c] = zg; g.i[tr + c] = i; g.f[tr + c] = f;
        if (g.o) g.o[tr + c] = o;
    }
}

// MultiHeadedDotAttention.forward (models/aoamodel.py:77-108) for a single query per image; one block per
// (image, head).  qg: [B][2H] = [q_proj(h) | aoa_linear_gate(h)]
__global__ __launch_bounds__(256) void aoa_fwd_attention_kernel(AoaFwd g, int t, const float* __restrict__ qg, int ldq,
                                                                const float* __restrict__ key,
                                                                const float* __restrict__ value) {
    extern __shared__ float sm[];
    const int b = blockIdx.x, hd = blockIdx.y, H = g.H, P = g.P, dk = H / g.NH, tid = threadIdx.x;
    float* q = sm;          // dk
    float* sc = q + dk;     // P
    float* red = sc + P;    // 8
    for (int c = tid; c < dk; c += 256) q[c] = qg[(long)b * ldq + hd * dk + c];
    __syncthreads();
    const float inv = 1.f / sqrtf((float)dk);
    for (int k = tid; k < P; k += 256) {
        const float* kp = key + ((long)b * P + k) * H + hd * dk;
        float a = 0.f;
        for (int c = 0; c < dk; ++c) a += q[c] * kp[c];
        sc[k] = a * inv;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int k = tid; k < P; k += 256) m = fmaxf(m, sc[k]);
    m = block_max(m, red);
    float e = 0.f;
    for (int k = tid; k < P; k += 256) e += expf(sc[k] - m);
    const float denom = block_sum(e, red);
    __syncthreads();
    float* al = g.alpha + (((long)b * g.T + t) * g.NH + hd) * P;
    for (int k = tid; k < P; k += 256) { const float a = expf(sc[k] - m) / denom; sc[k] = a; al[k] = a; }
    __syncthreads();
    const long tr = ((long)b * g.T + t) * H;
    for (int c = tid; c < dk; c += 256) {
        float a = 0.f;
        for (int k = 0; k < P; ++k) a += sc[k] * value[((long)b * P + k) * H + hd * dk + c];
        g.ctx[tr + hd * dk + c] = a;
    }
}

// c_aoa = sigmoid(gate) * lin ; hc = c_aoa + h   (:1034-1038)
__global__ void aoa_fwd_post_kernel(AoaFwd g, int t, const float* __restrict__ qg, int ldq, const float* __restrict__ lin) {
    const int b = blockIdx.x, H = g.H;
    const long st1 = ((long)b * (g.T + 1) + t + 1) * H, tr = ((long)b * g.T + t) * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float l = lin[(long)b * H + c];
        const float sgv = sigmoidf_(qg[(long)b * ldq + H + c]);
        const float ca = sgv * l;
        g.lin[tr + c] = l; g.c_aoa[tr + c] = ca; g.hc[tr + c] = ca + g.h[st1 + c];
        if (g.sg) g.sg[tr + c] = sgv;
    }
}

// ---- the skinny linears of the AoA step with their point-wise neighbours in the epilogue ---------------------------------
// The bottom-up path is bound by the GPU's dispatch rate of dependent small kernels (~330 launches of 4 - 28 us per step of 640
// maps), so the seven launches of a decoder step become four: the gate linear applies the LSTM cell (its weight
// ROWS are interleaved - workgroup j holds the i, f, g, o rows of the hidden units 4j .. 4j+3 - so the four pre-activations of
// a unit meet in the workgroup's LDS reduction), decoder_aoa_linear applies the gated sum (fwd_post) and gathers the NEXT
// step's LSTM input row (fwd_pre).  The dot products run in the same order as in linear_mfma_kernel and the point-wise code is
// that of aoa_fwd_lstm / _post / _pre: results are bit-identical to the unfused step (token ids of the sampling goldens, traces).
// (linear_mfma_core: next to linear_mfma_kernel above)

// ---- gridTD: the two gate linears of a decoder step with their LSTM cells (gridtd_fwd_lstm_kernel) in the epilogue, and the next step's
// input row (gridtd_fwd_pre_kernel) behind the second: 7 launches per time step -> 4.  Weight rows interleaved as above (w_il / b_il:
// row 16 j + 4 gate + u = row gate * H + 4 j + u of [W_ih | W_hh]); the same dot products in the same order as linear_mfma_kernel and the
// point-wise expressions of gridtd_fwd_lstm_kernel: bit-identical traces.
// WHICH == 1, AdaLSTM (models/gridTDmodel.py:777-784, sentinel gate :982): workgroups 0 .. H/4 - 1 hold the cells of hidden units 4 j .. 4 j + 3;
// workgroups H/4 .. H/4 + H/16 - 1 the 16 sentinel-gate rows 4 H + 16 (j - H/4) .. of the UN-interleaved [.. ; x_gate | h_gate] - sigmoid(gate) goes
// to sg [B][H], the sentinel s = sigmoid(gate) tanh(c1) is formed by the attention kernel that reads it (the cell state comes from other workgroups).
// WHICH == 2, LanguageLSTM: cells + fc input hc = h2 + ctx_hat (:990), then xh1[b, t + 1] = [h2[b, t + 1] | glob[b] | emb[tok[b, t + 1]] | h1[b, t + 1]]:
// the workgroup's own four h2 columns from its registers, a slice of the other columns copied.
template <int RT, int WHICH>
__global__ __launch_bounds__(256) void gridtd_linear_lstm_kernel(GridFwd g, int t, const float* __restrict__ w_il, const float* __restrict__ b_il,
                                                                 const float* __restrict__ w_cat, const float* __restrict__ b_cat,
                                                                 float* __restrict__ sg, const float* __restrict__ glob,
                                                                 const float* __restrict__ emb, const long long* __restrict__ tok, int tok_ld) {
    __shared__ float red[4][RT][16][17];
    const int H = g.H, E = g.E;
    const int K = WHICH == 1 ? 2 * E + 2 * H : 3 * H;
    const float* x = WHICH == 1 ? g.xh1 + (long)t * K : g.xh2 + (long)t * K;
    const int n_cell = H / 4;                                   // workgroups that hold cells
    if (WHICH == 1 && (int)blockIdx.x >= n_cell) {              // sentinel-gate rows (workgroup-uniform branch)
        const int n0 = 4 * H + ((int)blockIdx.x - n_cell) * 16;
        linear_mfma_core<RT>(x, (long)g.T * K, w_cat, g.B, K, 5 * H, red, n0);
        for (int e = threadIdx.x; e < RT * 256; e += 256) {
            const int r = e >> 8, row = (e >> 4) & 15, col = e & 15;
            const int b = r * 16 + row, c = n0 - 4 * H + col;
            if (b >= g.B || c >= H) continue;
            float z = (red[0][r][row][col] + red[1][r][row][col]) + (red[2][r][row][col] + red[3][r][row][col]);
            if (b_cat) z += b_cat[n0 + col];
            sg[(long)b * H + c] = sigmoidf_(z);
        }
        return;
    }
    linear_mfma_core<RT>(x, (long)g.T * K, w_il, g.B, K, 4 * H, red, blockIdx.x * 16);
    float *hh = WHICH == 1 ? g.h1 : g.h2, *cc = WHICH == 1 ? g.c1 : g.c2;
    float *gg = WHICH == 1 ? g.g1 : g.g2, *ii = WHICH == 1 ? g.i1 : g.i2, *ff = WHICH == 1 ? g.f1 : g.f2, *oo = WHICH == 1 ? g.o1 : g.o2;
    for (int e = threadIdx.x; e < RT * 64; e += 256) {
        const int r = e >> 6, row = (e >> 2) & 15, u = e & 3;
        const int b = r * 16 + row, c = blockIdx.x * 4 + u;
        if (b >= g.B || c >= H) continue;
        float z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = 4 * q + u;
            z[q] = (red[0][r][row][col] + red[1][r][row][col]) + (red[2][r][row][col] + red[3][r][row][col]);
            if (b_il) z[q] += b_il[blockIdx.x * 16 + col];
        }
        const long st0 = ((long)b * (g.T + 1) + t) * H, st1 = st0 + H, tr = ((long)b * g.T + t) * H;
        const float i = sigmoidf_(z[0]), f = sigmoidf_(z[1]), zg = z[2], o = sigmoidf_(z[3]);
        const float cn = f * cc[st0 + c] + i * tanhf(zg);
        const float hn = o * tanhf(cn);
        cc[st1 + c] = cn; hh[st1 + c] = hn;
        gg[tr + c] = zg; ii[tr + c] = i; ff[tr + c] = f;
        if (oo) oo[tr + c] = o;
        if (WHICH == 2) {
            g.hc[tr + c] = hn + g.ctx_hat[tr + c];
            if (t + 1 < g.T) g.xh1[((long)b * g.T + t + 1) * (2 * E + 2 * H) + c] = hn;
        }
    }
    if (WHICH == 2 && t + 1 < g.T) {        // the other columns H .. 2 E + 2 H of the next input row: this workgroup's slice, every image
        const int W1 = 2 * E + 2 * H, rest = W1 - H;
        const int per = (rest + (int)gridDim.x - 1) / (int)gridDim.x;
        const int c0 = H + blockIdx.x * per, c1 = min(W1, c0 + per);
        for (int e = threadIdx.x; e < g.B * per; e += 256) {
            const int b = e / per, c = c0 + e - b * per;
            if (c >= c1) continue;
            float v;
            if (c < H + E) v = glob[(long)b * E + (c - H)];
            else if (c < H + 2 * E) v = emb[tok[(long)b * tok_ld + t + 1] * E + (c - H - E)];
            else v = g.h1[((long)b * (g.T + 1) + t + 1) * H + (c - H - 2 * E)];
            g.xh1[((long)b * g.T + t + 1) * W1 + c] = v;
        }
    }
}

// gate linear + LSTM cell (aoa_fwd_lstm_kernel).  w / bias: rows interleaved, row 16 j + 4 gate + u = original row gate * H + 4 j + u
template <int RT>
__global__ __launch_bounds__(256) void aoa_linear_lstm_kernel(AoaFwd g, int t, const float* __restrict__ w,
                                                              const float* __restrict__ bias) {
    __shared__ float red[4][RT][16][17];
    const int H = g.H, W = g.E + 2 * H;
    linear_mfma_core<RT>(g.xh + (long)t * W, (long)g.T * W, w, g.B, W, 4 * H, red, blockIdx.x * 16);
    for (int e = threadIdx.x; e < RT * 64; e += 256) {
        const int r = e >> 6, row = (e >> 2) & 15, u = e & 3;
        const int b = r * 16 + row, c = blockIdx.x * 4 + u;
        if (b >= g.B || c >= H) continue;
        float z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = 4 * q + u;
            z[q] = (red[0][r][row][col] + red[1][r][row][col]) + (red[2][r][row][col] + red[3][r][row][col]);
            if (bias) z[q] += bias[blockIdx.x * 16 + col];
        }
        const long st0 = ((long)b * (g.T + 1) + t) * H, st1 = st0 + H, tr = ((long)b * g.T + t) * H;
        const float i = sigmoidf_(z[0]), f = sigmoidf_(z[1]), zg = z[2], o = sigmoidf_(z[3]);
        const float cn = f * g.c[st0 + c] + i * tanhf(zg);
        g.c[st1 + c] = cn; g.h[st1 + c] = o * tanhf(cn);
        g.g[tr + c] = zg; g.i[tr + c] = i; g.f[tr + c] = f;
        if (g.o) g.o[tr + c] = o;
    }
}

// decoder_aoa_linear + gated sum (aoa_fwd_post_kernel) + the next step's input row (aoa_fwd_pre_kernel for t + 1)
template <int RT>
__global__ __launch_bounds__(256) void aoa_linear_post_kernel(AoaFwd g, int t, const float* __restrict__ w,
                                                              const float* __restrict__ bias, const float* __restrict__ qg, int ldq,
                                                              const float* __restrict__ glob, const float* __restrict__ emb,
                                                              const long long* __restrict__ tok, int tok_ld) {
    __shared__ float red[4][RT][16][17];
    const int H = g.H, W = g.E + 2 * H;
    linear_mfma_core<RT>(g.ctx + (long)t * H, (long)g.T * H, w, g.B, H, H, red, blockIdx.x * 16);
    for (int e = threadIdx.x; e < RT * 256; e += 256) {
        const int r = e >> 8, row = (e >> 4) & 15, col = e & 15;
        const int b = r * 16 + row, c = blockIdx.x * 16 + col;
        if (b >= g.B || c >= H) continue;
        float l = (red[0][r][row][col] + red[1][r][row][col]) + (red[2][r][row][col] + red[3][r][row][col]);
        if (bias) l += bias[c];
        const long st1 = ((long)b * (g.T + 1) + t + 1) * H, tr = ((long)b * g.T + t) * H;
        const float sgv = sigmoidf_(qg[(long)b * ldq + H + c]);
        const float ca = sgv * l;
        g.lin[tr + c] = l; g.c_aoa[tr + c] = ca; g.hc[tr + c] = ca + g.h[st1 + c];
        if (g.sg) g.sg[tr + c] = sgv;
    }
    if (t + 1 < g.T) {        // xh[b, t+1] = [emb[tok[b, t+1]] | glob[b] | h[b, t+1]]: this workgroup's slice of the columns, every image
        const int per = (W + (int)gridDim.x - 1) / (int)gridDim.x;
        const int c0 = blockIdx.x * per, c1 = min(W, c0 + per);
        for (int e = threadIdx.x; e < g.B * per; e += 256) {
            const int b = e / per, c = c0 + e - b * per;
            if (c >= c1) continue;
            const long long k = tok[(long)b * tok_ld + t + 1];
            float v;
            if (c < g.E) v = emb[k * g.E + c];
            else if (c < g.E + H) v = glob[(long)b * H + (c - g.E)];
            else v = g.h[((long)b * (g.T + 1) + t + 1) * H + (c - g.E - H)];
            g.xh[((long)b * g.T + t + 1) * W + c] = v;
        }
    }
}

// ---- the teacher-forced trace with the recurrence DECOUPLED from what hangs off it (round 5) ------------------------------------
// In this model the LSTM input is x_t = [emb(word_t) | global feature] and its own h_{t-1} (models/aoamodel.py:1030-1033): the
// attention, the AoA gate and the scores READ h_t but never feed the recurrence.  Under teacher forcing every x_t is known up front,
// so the only sequential work is  z_t = Zin[t] + W_hh h_{t-1}  with  Zin = [emb | glob] W_ih^T + b  computed for all (image, word)
// rows in ONE GEMM, and q / gate linear, attention, decoder_aoa_linear, gated sum run once over all B*T rows afterwards: a step of
// the reference's loop (:1019-1052) costs one launch of K = 512 instead of four launches with K = 1536 in the first.
__global__ void aoa_fwd_inputs_kernel(AoaFwd g, const float* __restrict__ glob, const float* __restrict__ emb,
                                      const long long* __restrict__ tok, int tok_ld, float* __restrict__ xin) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, W = g.E + 2 * g.H, EH = g.E + g.H;
    float* dst = g.xh + (long)row * W;
    const long long k = tok[(long)b * tok_ld + t];
    for (int c = threadIdx.x; c < EH; c += blockDim.x) {
        const float v = c < g.E ? emb[k * g.E + c] : glob[(long)b * g.H + (c - g.E)];
        dst[c] = v;
        if (xin) xin[(long)row * EH + c] = v;
    }
}

// z = zin[b, t] + W_hh h[b, t]  ->  LSTM cell (the point-wise code of aoa_fwd_lstm_kernel).  w: (4H, H), rows interleaved as in
// aoa_linear_lstm_kernel; zin: [B*T][4H] in the same interleaved column order (bias included)
// zin == null: the input part comes from a per-MODEL table instead of a per-step GEMM - the embedding part of x_t W_ih^T depends on the
// token alone: tab[token] = emb[token] W_ie^T (V x 4H, computed once per engine), gimg[b] = glob[b] W_ig^T + bias (one small linear
// per trace): z = (W_hh h + tab[tok[b, t]]) + gimg[b]
template <int RT>
__global__ __launch_bounds__(256) void aoa_rec_lstm_kernel(AoaFwd g, int t, const float* __restrict__ w,
                                                           const float* __restrict__ zin, const float* __restrict__ tab,
                                                           const float* __restrict__ gimg, const long long* __restrict__ tok, int tok_ld) {
    __shared__ float red[4][RT][16][17];
    const int H = g.H;
    linear_mfma_core<RT>(g.h + (long)t * H, (long)(g.T + 1) * H, w, g.B, H, 4 * H, red, blockIdx.x * 16);
    for (int e = threadIdx.x; e < RT * 64; e += 256) {
        const int r = e >> 6, row = (e >> 2) & 15, u = e & 3;
        const int b = r * 16 + row, c = blockIdx.x * 4 + u;
        if (b >= g.B || c >= H) continue;
        const float* zi = zin ? zin + ((long)b * g.T + t) * 4 * H + blockIdx.x * 16
                              : tab + (long)tok[(long)b * tok_ld + t] * 4 * H + blockIdx.x * 16;
        const float* gi = zin ? nullptr : gimg + (long)b * 4 * H + blockIdx.x * 16;
        float z[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int col = 4 * q + u;
            z[q] = ((red[0][r][row][col] + red[1][r][row][col]) + (red[2][r][row][col] + red[3][r][row][col])) + zi[col];
            if (gi) z[q] += gi[col];
        }
        const long st0 = ((long)b * (g.T + 1) + t) * H, st1 = st0 + H, tr = ((long)b * g.T + t) * H;
        const float i = sigmoidf_(z[0]), f = sigmoidf_(z[1]), zg = z[2], o = sigmoidf_(z[3]);
        const float cn = f * g.c[st0 + c] + i * tanhf(zg);
        g.c[st1 + c] = cn; g.h[st1 + c] = o * tanhf(cn);
        g.g[tr + c] = zg; g.i[tr + c] = i; g.f[tr + c] = f;
        if (g.o) g.o[tr + c] = o;
    }
}

// after the recurrence: hn[b, t] = h[b, t + 1] (contiguous rows for the q / gate GEMM), xh[b, t][E + H :] = h[b, t] (:1075)
__global__ __launch_bounds__(256) void aoa_fwd_gather_h_kernel(AoaFwd g, float* __restrict__ hn, unsigned* __restrict__ hn_amax) {
    __shared__ float red[8];
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H, W = g.E + 2 * H;
    const long st0 = ((long)b * (g.T + 1) + t) * H;
    float m = 0.f;
    for (int c = threadIdx.x; c < H; c += 256) {
        g.xh[(long)row * W + g.E + H + c] = g.h[st0 + c];
        const float v = g.h[st0 + H + c];
        hn[(long)row * H + c] = v;
        m = fmaxf(m, fabsf(v));
    }
    if (hn_amax) {          // max|hn[row]|: the operand scale of the q / gate GEMM (what lrpx_amax_maps would read the row again for)
        m = block_max(m, red);
        if (threadIdx.x == 0) hn_amax[row] = __float_as_uint(m);
    }
}

// aoa_fwd_attention_kernel for every (image, word) row at once: block (row, head); qg: [B*T][ldq]
__global__ __launch_bounds__(64) void aoa_fwd_attention_all_kernel(AoaFwd g, const float* __restrict__ qg, int ldq,
                                                                   const float* __restrict__ key, const float* __restrict__ value,
                                                                   unsigned* __restrict__ ctx_amax) {
    extern __shared__ float sm[];
    const int row = blockIdx.x, b = row / g.T, hd = blockIdx.y, H = g.H, P = g.P, dk = H / g.NH, tid = threadIdx.x;
    float* q = sm;          // dk
    float* sc = q + dk;     // P
    for (int c = tid; c < dk; c += 64) q[c] = qg[(long)row * ldq + hd * dk + c];
    __syncthreads();
    const float inv = 1.f / sqrtf((float)dk);
    for (int k = tid; k < P; k += 64) {
        const float* kp = key + ((long)b * P + k) * H + hd * dk;
        float a = 0.f;
        for (int c = 0; c < dk; ++c) a += q[c] * kp[c];
        sc[k] = a * inv;
    }
    __syncthreads();
    float m = -INFINITY;
    for (int k = tid; k < P; k += 64) m = fmaxf(m, sc[k]);
    m = wave_max(m);
    float e = 0.f;
    for (int k = tid; k < P; k += 64) e += expf(sc[k] - m);
    const float denom = wave_sum(e);
    float* al = g.alpha + ((long)row * g.NH + hd) * P;
    for (int k = tid; k < P; k += 64) { const float a = expf(sc[k] - m) / denom; sc[k] = a; al[k] = a; }
    __syncthreads();
    float mx = 0.f;
    for (int c = tid; c < dk; c += 64) {
        float a = 0.f;
        for (int k = 0; k < P; ++k) a += sc[k] * value[((long)b * P + k) * H + hd * dk + c];
        g.ctx[(long)row * H + hd * dk + c] = a;
        mx = fmaxf(mx, fabsf(a));
    }
    if (ctx_amax) {         // max|ctx[row]| over the heads (zero-initialised by the caller; non-negative floats order like unsigned integers)
        mx = wave_max(mx);
        if (tid == 0) atomicMax(&ctx_amax[row], __float_as_uint(mx));
    }
}

// aoa_fwd_post_kernel for every row: lin [B*T][H] = decoder_aoa_linear(ctx), qg [B*T][ldq] (gate in the second half)
__global__ __launch_bounds__(256) void aoa_fwd_post_all_kernel(AoaFwd g, const float* __restrict__ qg, int ldq, const float* __restrict__ lin,
                                                               unsigned* __restrict__ hc_amax) {
    __shared__ float red[8];
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H;
    const long st1 = ((long)b * (g.T + 1) + t + 1) * H, tr = (long)row * H;
    float m = 0.f;
    for (int c = threadIdx.x; c < H; c += 256) {
        const float l = lin[tr + c];
        const float sgv = sigmoidf_(qg[(long)row * ldq + H + c]);
        const float ca = sgv * l;
        const float hcv = ca + g.h[st1 + c];
        g.lin[tr + c] = l; g.c_aoa[tr + c] = ca; g.hc[tr + c] = hcv;
        if (g.sg) g.sg[tr + c] = sgv;
        m = fmaxf(m, fabsf(hcv));
    }
    if (hc_amax) {          // max|hc[row]|: the operand scale of the (T, V) score GEMM
        m = block_max(m, red);
        if (threadIdx.x == 0) hc_amax[row] = __float_as_uint(m);
    }
}

// ---- AoA gradient explainers (models/aoamodel.py:1435-1499), all (image, word) rows in lock-step -------------------
struct AoaGrad {
    int B, T, H, E, P, NH;
    const int* lens;
    const float *c, *g, *i, *f, *o, *sg, *lin, *alpha;
    float *d_h, *d_c, *dA, *dB, *gates, *dx, *d_glob, *r_words;
};

__device__ __forceinline__ bool aoa_grow_active(const AoaGrad& g, int b, int t, int s) {
    const int len = g.lens ? g.lens[b] : g.T;
    return t < len && t >= s;
}

// :1461-1470: d_word_pred one-hot -> fc row; gradient into the gated sum  c_aoa = sigmoid(gate) * lin
__global__ void aoa_grad_init_kernel(AoaGrad g, const float* __restrict__ fcw, const long long* __restrict__ tok,
                                     int tok_ld) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H;
    const long long k = tok[(long)b * tok_ld + t + 1];
    const long r = (long)row * H;          // trace tensors [B][T][H] are indexed by the same row
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float d = fcw[k * H + c], sgv = g.sg[r + c];
        g.d_h[r + c] = d; g.d_c[r + c] = 0.f; g.d_glob[r + c] = 0.f;
        g.dA[r + c] = d * sgv;
        g.dB[r + c] = d * g.lin[r + c] * (1.f - sgv) * sgv;
    }
    for (int c = threadIdx.x; c < g.T; c += blockDim.x) g.r_words[(long)row * g.T + c] = 0.f;
}

// phase 0 (:1474-1484): LSTM cell backward at time i = t - s -> gate gradients [i | f | g | o]
__global__ void aoa_grad_step0_kernel(AoaGrad g, int s) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H;
    const long r = (long)row * H;
    float* gt = g.gates + (long)row * 4 * H;
    if (!aoa_grow_active(g, b, t, s)) {
        for (int c = threadIdx.x; c < 4 * H; c += blockDim.x) gt[c] = 0.f;
        return;
    }
    const int i = t - s;
    const long ti = ((long)b * g.T + i) * H, ci = ((long)b * (g.T + 1) + i) * H, ci1 = ci + H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float dh = g.d_h[r + c], tc = tanhf(g.c[ci1 + c]);
        const float iv = g.i[ti + c], fv = g.f[ti + c], gv = tanhf(g.g[ti + c]), ov = g.o[ti + c];
        const float dc = g.d_c[r + c] + dh * ov * (1.f - tc * tc);
        gt[c] = dc * gv * iv * (1.f - iv);
        gt[H + c] = dc * g.c[ci + c] * fv * (1.f - fv);
        gt[2 * H + c] = dc * iv * (1.f - gv * gv);
        gt[3 * H + c] = dh * tc * ov * (1.f - ov);
        g.d_c[r + c] = dc * fv;
    }
}

// phase 1 (:1485-1488): dx = gates @ [W_ih | W_hh] = [d_emb (E) | d_glob (H) | d_h (H)]
__global__ void aoa_grad_step1_kernel(AoaGrad g, int s) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H, E = g.E;
    if (!aoa_grow_active(g, b, t, s)) return;
    const int i = t - s;
    const long r = (long)row * H;
    const float* dx = g.dx + (long)row * (E + 2 * H);
    __shared__ float red[4];
    float acc = 0.f;
    for (int c = threadIdx.x; c < E; c += blockDim.x) acc += dx[c];
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) g.r_words[(long)row * g.T + i] = acc;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        g.d_glob[r + c] = dx[E + c];            // assignment, not accumulation (:1487)
        g.d_h[r + c] = dx[E + H + c];
    }
}

__global__ void aoa_grad_pix_kernel(const float* __restrict__ alpha, int T, int NH, int P, int head,
                                    const float* __restrict__ v1, const float* __restrict__ v2,
                                    float* __restrict__ d_feat, int C4, long total,
                                    const int* __restrict__ rowlist) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;     // over (output rows)*P*C4
    if (idx >= total) return;
    const int c4 = idx % C4;
    const long rp = idx / C4;
    const int p = rp % P;
    const long row = rowlist ? rowlist[rp / P] : rp / P;
    const float a = alpha[(row * NH + head) * P + p];
    const f32x4 x1 = reinterpret_cast<const f32x4*>(v1)[row * C4 + c4], x2 = reinterpret_cast<const f32x4*>(v2)[row * C4 + c4];
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = a * x1[e] + x2[e];
    reinterpret_cast<f32x4*>(d_feat)[idx] = o;
}

__global__ void keep_cols_kernel(float* __restrict__ x, int ncol, int lo, int hi, long total) {
    const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = idx % ncol;
    if (c < lo || c >= hi) x[idx] = 0.f;
}

struct AoaRel {
    int B, T, H, E, P, NH;
    const int* lens;
    const float *xh, *h, *c, *g, *i, *ctx, *lin, *c_aoa, *hc, *alpha;
    float *r_hn, *r_glob, *A, *rx, *r_words;     // [rows][H], [rows][H], [rows][H], [rows][E+2H], [rows][T]
};

__device__ __forceinline__ bool aoa_row_active(const AoaRel& g, int b, int t, int s) {
    const int len = g.lens ? g.lens[b] : g.T;
    return t < len && t >= s;
}

// :1081-1110 fc one-hot rule, split into h and context_aoa; A = r_caoa / z~(aoa_linear output) for the dense rule
__global__ void aoa_rel_init_kernel(AoaRel g, const float* __restrict__ fcw, const float* __restrict__ logit,
                                    const long long* __restrict__ tok, int tok_ld) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H;
    const long long k = tok[(long)b * tok_ld + t + 1];
    const float lg = logit[row], zt = stab_eps(lg);
    const long tr = (long)row * H, st1 = ((long)b * (g.T + 1) + t + 1) * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float hc = g.hc[tr + c];
        const float r_hc = (fcw[k * H + c] * hc / zt) * lg;
        g.r_hn[tr + c] = eps_id(r_hc, g.h[st1 + c], hc);
        const float r_ca = eps_id(r_hc, g.c_aoa[tr + c], hc);
        g.A[tr + c] = r_ca / stab_eps(g.lin[tr + c]);
        g.r_glob[tr + c] = 0.f;
    }
    for (int c = threadIdx.x; c < g.T; c += blockDim.x) g.r_words[(long)row * g.T + c] = 0.f;
}

// lrp_mha (:812-862), single head: Aval[row][k][c] = (value*alpha / z~(ctx)) * r_ctx / z~(value)  inside head
// `head`, 0 elsewhere — the prologue of the v_proj dense rule (:1141-1144).  r_ctx: [rows][H]
__global__ __launch_bounds__(256) void aoa_rel_value_kernel(AoaRel g, const float* __restrict__ r_ctx,
                                                            const float* __restrict__ value, int head,
                                                            float* __restrict__ Aval, const int* __restrict__ rowlist, int head_only) {
    const int row = rowlist ? rowlist[blockIdx.x] : blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H, P = g.P, dk = H / g.NH;
    const long orow = blockIdx.x;
    const int len = g.lens ? g.lens[b] : g.T;
    const bool act = t < len;
    const float* al = g.alpha + (((long)b * g.T + t) * g.NH + head) * P;
    if (head_only) {
        // only the head's dk columns are non-zero: Aval [rows][P][dk] - the v_proj rule then contracts over K = dk with the head's
        // dk rows of W_v instead of K = H with zeros in 7/8 of the operand (the same products in the same order: bit-identical)
        for (int j = threadIdx.x; j < P * dk; j += 256) {
            const int k = j / dk, c = head * dk + (j - k * dk);
            float v = 0.f;
            if (act) {
                const float val = value[((long)b * P + k) * H + c];
                const float rv = eps_id(r_ctx[(long)row * H + c], val * al[k], g.ctx[(long)row * H + c]);
                v = rv / stab_eps(val);
            }
            Aval[(orow * P + k) * dk + (c - head * dk)] = v;
        }
        return;
    }
    for (int j = threadIdx.x; j < P * H; j += 256) {
        const int k = j / H, c = j - k * H;
        float v = 0.f;
        if (act && c >= head * dk && c < (head + 1) * dk) {
            const float val = value[((long)b * P + k) * H + c];
            const float rv = eps_id(r_ctx[(long)row * H + c], val * al[k], g.ctx[(long)row * H + c]);
            v = rv / stab_eps(val);
        }
        Aval[(orow * P + k) * H + c] = v;
    }
}

// :1116-1120  r_c = r_h (assignment); g-gate path; A = r_g / z~(g)
__global__ void aoa_rel_a_kernel(AoaRel g, int s) {
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H;
    const long tr = (long)row * H;
    if (!aoa_row_active(g, b, t, s)) {
        for (int c = threadIdx.x; c < H; c += blockDim.x) g.A[tr + c] = 0.f;
        return;
    }
    const int i = t - s;
    const long ti = ((long)b * g.T + i) * H, sc1 = ((long)b * (g.T + 1) + i + 1) * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float rg = eps_id(g.r_hn[tr + c], g.i[ti + c] * tanhf(g.g[ti + c]), g.c[sc1 + c]);
        g.A[tr + c] = rg / stab_eps(g.g[ti + c]);
    }
}

// :1129-1133  r_xh = [emb | glob | h]
__global__ __launch_bounds__(256) void aoa_rel_c_kernel(AoaRel g, int s) {
    __shared__ float red[8];
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H, E = g.E;
    if (!aoa_row_active(g, b, t, s)) return;
    const int i = t - s;
    const long tr = (long)row * H;
    const float* rx = g.rx + (long)row * (E + 2 * H);
    float acc = 0.f;
    for (int c = threadIdx.x; c < E; c += 256) acc += rx[c];
    for (int c = threadIdx.x; c < H; c += 256) {
        g.r_glob[tr + c] += rx[E + c];
        g.r_hn[tr + c] = rx[E + H + c];
    }
    acc = block_sum(acc, red);
    if (threadIdx.x == 0) g.r_words[(long)row * g.T + i] = acc;
}

// rel_c of lock-step s followed by rel_a of lock-step s + 1 for the same row (one launch instead of two; the same code, the
// element -> thread map of both parts is the same, so a thread reads back the r_hn it has just written)
__global__ __launch_bounds__(256) void aoa_rel_ca_kernel(AoaRel g, int s) {
    __shared__ float red[8];
    const int row = blockIdx.x, b = row / g.T, t = row - b * g.T, H = g.H, E = g.E;
    const long tr = (long)row * H;
    if (aoa_row_active(g, b, t, s)) {
        const int i = t - s;
        const float* rx = g.rx + (long)row * (E + 2 * H);
        float acc = 0.f;
        for (int c = threadIdx.x; c < E; c += 256) acc += rx[c];
        for (int c = threadIdx.x; c < H; c += 256) {
            g.r_glob[tr + c] += rx[E + c];
            g.r_hn[tr + c] = rx[E + H + c];
        }
        acc = block_sum(acc, red);
        if (threadIdx.x == 0) g.r_words[(long)row * g.T + i] = acc;
    }
    if (s + 1 >= g.T) return;
    if (!aoa_row_active(g, b, t, s + 1)) {
        for (int c = threadIdx.x; c < H; c += 256) g.A[tr + c] = 0.f;
        return;
    }
    const int i = t - s - 1;
    const long ti = ((long)b * g.T + i) * H, sc1 = ((long)b * (g.T + 1) + i + 1) * H;
    for (int c = threadIdx.x; c < H; c += 256) {
        const float rg = eps_id(g.r_hn[tr + c], g.i[ti + c] * tanhf(g.g[ti + c]), g.c[sc1 + c]);
        g.A[tr + c] = rg / stab_eps(g.g[ti + c]);
    }
}

// per-trace tables of the fused AoA lock-steps: what aoa_rel_a_kernel evaluates per (row, step) depends on (image, time index) only -
// q1 = i tanh(g) / z~(c[i + 1]) (eps_id's factor) and dg = z~(g) (:1116-1120) - and tmax[row] = t for a word the caption has, else -1
__global__ void aoa_rel_coef_kernel(AoaRel g, float* __restrict__ q1, float* __restrict__ dg, int* __restrict__ tmax) {
    const int row = blockIdx.x, b = row / g.T, i = row - b * g.T, H = g.H;
    const long ti = (long)row * H, sc1 = ((long)b * (g.T + 1) + i + 1) * H;
    for (int c = threadIdx.x; c < H; c += blockDim.x) {
        const float gg = g.g[ti + c];
        q1[ti + c] = (g.i[ti + c] * tanhf(gg)) / stab_eps(g.c[sc1 + c]);
        dg[ti + c] = stab_eps(gg);
    }
    if (threadIdx.x == 0) tmax[row] = i < (g.lens ? g.lens[b] : g.T) ? i : -1;
}

// the same after the fused AoA lock-steps (dense_f16x3.hip, FUSE): r_words[row][i] = the four 128-column partial sums of the
// embedding part in a fixed order, then :1129-1132; rows behind an image's last word stay zero
__global__ void rel_words_norm_parts_kernel(float* __restrict__ r_words, const float* __restrict__ wpart, const int* __restrict__ lens,
                                            int rows, int T) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= rows) return;
    const int b = row / T, t = row - b * T;
    const int len = lens ? lens[b] : T;
    float* rw = r_words + (long)row * T;
    if (t >= len) {
        for (int i = 0; i < T; ++i) rw[i] = 0.f;
        return;
    }
    float m = 0.f;
    for (int i = 0; i <= t; ++i) {
        const float* p = wpart + ((long)row * T + i) * 4;
        const float v = (p[0] + p[1]) + (p[2] + p[3]);
        rw[i] = v;
        m = fmaxf(m, fabsf(v));
    }
    for (int i = t + 1; i < T; ++i) rw[i] = 0.f;
    if (m > 0.f)
        for (int i = 0; i <= t; ++i) rw[i] /= m;
}


}  // namespace lrpx

using namespace lrpx;

extern "C" {

int lrpx_linear_small(const float* x, long ldx, const float* w, const float* bias, float* out, long ldo, int B, int K,
                      int N, int act, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_linear_small", {x, "x"}, {w, "w"}, {bias, "bias"}, {out, "out"});
    LRPX_REQUIRE(x && w && out && B > 0 && N > 0 && K > 0 && K % 4 == 0 && ldx % 4 == 0, "linear_small: bad arguments");
    const int lin_valu = switches().linear_valu;     // (A/B switch)
    if (K % 16 == 0 && B <= 64 && !lin_valu) {
        const dim3 grid((N + 15) / 16);
        hipStream_t st = (hipStream_t)stream;
        if (B <= 16) hipLaunchKernelGGL((linear_mfma_kernel<1>), grid, dim3(256), 0, st, x, ldx, w, bias, out, ldo, B, K, N, act);
        else if (B <= 32) hipLaunchKernelGGL((linear_mfma_kernel<2>), grid, dim3(256), 0, st, x, ldx, w, bias, out, ldo, B, K, N, act);
        else if (B <= 48) hipLaunchKernelGGL((linear_mfma_kernel<3>), grid, dim3(256), 0, st, x, ldx, w, bias, out, ldo, B, K, N, act);
        else hipLaunchKernelGGL((linear_mfma_kernel<4>), grid, dim3(256), 0, st, x, ldx, w, bias, out, ldo, B, K, N, act);
        return check_launch("linear_small");
    }
    hipLaunchKernelGGL((linear_small_kernel<4, 16>), dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, ldx, w,
                       bias, out, ldo, B, K, N, act);
    return check_launch("linear_small");
}

int lrpx_mean_pixels(const float* f, float* avg, int B, int P, int C, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_mean_pixels", {f, "f"}, {avg, "avg"});
    LRPX_REQUIRE(f && avg && B > 0, "mean_pixels: bad arguments");
    hipLaunchKernelGGL(mean_pixels_kernel, dim3((C + 63) / 64, B), dim3(256), 0, (hipStream_t)stream, f, avg, P, C,
                       1.0f / (float)P);
    return check_launch("mean_pixels");
}

int lrpx_relu(const float* x, float* y, long n, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_relu", {x, "x"}, {y, "y"});
    LRPX_REQUIRE(x && y && n > 0, "relu: bad arguments");
    hipLaunchKernelGGL(relu_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return check_launch("relu");
}

int lrpx_argmax_rows(const float* x, long ld, int rows, int n, long long* out, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_argmax_rows", {x, "x"}, {out, "out"});
    LRPX_REQUIRE(x && out && rows > 0 && n > 0, "argmax_rows: bad arguments");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ld, n, out);
    return check_launch("argmax_rows");
}

static GridFwd to_fwd(const lrpx_gridtd_trace* t) {
    GridFwd g;
    g.B = t->B; g.T = t->T; g.H = t->H; g.E = t->E; g.P = t->P;
    g.xh1 = t->xh1; g.xh2 = t->xh2; g.h1 = t->h1; g.c1 = t->c1; g.h2 = t->h2; g.c2 = t->c2;
    g.g1 = t->g1; g.i1 = t->i1; g.f1 = t->f1; g.g2 = t->g2; g.i2 = t->i2; g.f2 = t->f2;
    g.s = t->s; g.ctx = t->ctx; g.ctx_hat = t->ctx_hat; g.hc = t->hc; g.alpha = t->alpha; g.beta = t->beta;
    g.o1 = t->o1; g.o2 = t->o2; g.sgate = t->sgate;
    return g;
}

static int check_trace(const lrpx_gridtd_trace* t) {
    LRPX_REQUIRE(t && t->B > 0 && t->T > 0 && t->H == 512 && t->E % 4 == 0 && t->P > 0 && t->P <= 256,
                 "gridtd: unsupported trace dims (H must be 512, P <= 256)");
    LRPX_REQUIRE(t->xh1 && t->xh2 && t->h1 && t->c1 && t->h2 && t->c2 && t->g1 && t->i1 && t->f1 && t->g2 && t->i2 &&
                     t->f2 && t->s && t->ctx && t->ctx_hat && t->hc && t->alpha && t->beta,
                 "gridtd: null trace tensor");
    return LRPX_OK;
}

int lrpx_gridtd_fwd_pre(const lrpx_gridtd_trace* tr, int t, const float* glob, const float* emb, const long long* tok,
                        int tok_ld, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_fwd_pre", {glob, "glob"}, {emb, "emb"}, {tok, "tok"});
    LRPX_TRY(check_trace(tr));
    LRPX_REQUIRE(glob && emb && tok && t >= 0 && t < tr->T, "gridtd_fwd_pre: bad arguments");
    hipLaunchKernelGGL(gridtd_fwd_pre_kernel, dim3(tr->B), dim3(256), 0, (hipStream_t)stream, to_fwd(tr), t, glob, emb,
                       tok, tok_ld);
    return check_launch("gridtd_fwd_pre");
}

int lrpx_gridtd_fwd_gate_input(const lrpx_gridtd_trace* tr, int t, float* xg, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_fwd_gate_input", {xg, "xg"});
    LRPX_TRY(check_trace(tr));
    LRPX_REQUIRE(xg && t >= 0 && t < tr->T, "gridtd_fwd_gate_input: bad arguments");
    hipLaunchKernelGGL(gridtd_gate_input_kernel, dim3(tr->B), dim3(256), 0, (hipStream_t)stream, to_fwd(tr), t, xg);
    return check_launch("gridtd_fwd_gate_input");
}

int lrpx_gridtd_fwd_sentinel(const lrpx_gridtd_trace* tr, int t, const float* zg, int ldz, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_fwd_sentinel", {zg, "zg"});
    LRPX_TRY(check_trace(tr));
    LRPX_REQUIRE(zg && ldz >= tr->H && t >= 0 && t < tr->T, "gridtd_fwd_sentinel: bad arguments");
    hipLaunchKernelGGL(gridtd_sentinel_kernel, dim3(tr->B), dim3(256), 0, (hipStream_t)stream, to_fwd(tr), t, zg, ldz);
    return check_launch("gridtd_fwd_sentinel");
}

int lrpx_gridtd_lrp_reweight(const lrpx_gridtd_trace* tr, int t, const float* pred, long ld, int V, const float* fc_w,
                             const unsigned char* skip, float* hcw, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_lrp_reweight", {pred, "pred"}, {fc_w, "fc_w"}, {skip, "skip"}, {hcw, "hcw"});
    LRPX_TRY(check_trace(tr));
    LRPX_REQUIRE(pred && fc_w && skip && hcw && V > 0 && ld >= V && t >= 0 && t < tr->T,
                 "gridtd_lrp_reweight: bad arguments");
    hipLaunchKernelGGL(lrp_reweight_kernel, dim3(tr->B), dim3(256), 0, (hipStream_t)stream, pred, ld, V,
                       tr->h2 + (long)(t + 1) * tr->H, (long)(tr->T + 1) * tr->H, tr->ctx_hat + (long)t * tr->H,
                       (long)tr->T * tr->H, fc_w, skip, hcw, tr->H, 0);
    return check_launch("gridtd_lrp_reweight");
}

int lrpx_lrp_reweight_rows(const float* pred, long ld, int V, const float* h, long ldh, const float* ctx, long ldc,
                           const float* fc_w, const unsigned char* skip, float* hcw, int rows, int H, int log_softmax,
                           void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_lrp_reweight_rows", {pred, "pred"}, {h, "h"}, {ctx, "ctx"}, {fc_w, "fc_w"}, {skip, "skip"}, {hcw, "hcw"});
    LRPX_REQUIRE(pred && h && ctx && fc_w && skip && hcw && rows > 0 && V > 0 && ld >= V && H == 512 && ldh >= H &&
                     ldc >= H, "lrp_reweight_rows: bad arguments (H must be 512)");
    hipLaunchKernelGGL(lrp_reweight_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, pred, ld, V, h, ldh, ctx, ldc,
                       fc_w, skip, hcw, H, log_softmax ? 1 : 0);
    return check_launch("lrp_reweight_rows");
}

int lrpx_argmax_logprob_rows(const float* x, long ld, int rows, int n, long long* out, float* logprob, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_argmax_logprob_rows", {x, "x"}, {out, "out"}, {logprob, "logprob"});
    LRPX_REQUIRE(x && out && logprob && rows > 0 && n > 0, "argmax_logprob_rows: bad arguments");
    hipLaunchKernelGGL(argmax_logprob_rows_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, x, ld, n, out,
                       logprob);
    return check_launch("argmax_logprob_rows");
}

int lrpx_beam_topk(const float* x, long ld, int n_rows, int n, const float* cum, int k, long long* out_idx, float* out_val,
                   void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_beam_topk", {x, "x"}, {cum, "cum"}, {out_idx, "out_idx"}, {out_val, "out_val"});
    LRPX_REQUIRE(x && out_idx && out_val && n_rows > 0 && n_rows <= 8 && n > 0 && k > 0 && k <= 4 && (long)n_rows * n < 0x7fffffffL,
                 "beam_topk: bad arguments (at most 8 live beams, k <= 4)");
    hipLaunchKernelGGL((beam_topk_kernel<4>), dim3(1), dim3(1024), 0, (hipStream_t)stream, x, ld, n_rows, n, cum, k, out_idx,
                       out_val);
    return check_launch("beam_topk");
}

int lrpx_gridtd_fwd_lstm(const lrpx_gridtd_trace* tr, int t, const float* zz, int ldz, int which, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_fwd_lstm", {zz, "zz"});
    LRPX_TRY(check_trace(tr));
    LRPX_REQUIRE(zz && (which == 1 || which == 2) && t >= 0 && t < tr->T, "gridtd_fwd_lstm: bad arguments");
    hipLaunchKernelGGL(gridtd_fwd_lstm_kernel, dim3(tr->B), dim3(256), 0, (hipStream_t)stream, to_fwd(tr), t, zz, ldz,
                       which);
    return check_launch("gridtd_fwd_lstm");
}

}  // extern "C"
// sg: the sentinel gate [B][H] of the fused step (null: the trace's s is already there)
static int gridtd_attention(const lrpx_gridtd_trace* tr, int t, const float* Vp, const float* att_img, const float* Wg, const float* Ws,
                            const float* bs, const float* wh, float* scratch, const float* sg, void* stream);
extern "C" {
int lrpx_gridtd_fwd_attention(const lrpx_gridtd_trace* tr, int t, const float* Vp, const float* att_img,
                              const float* Wg, const float* Ws, const float* bs, const float* wh, float* scratch,
                              void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_fwd_attention", {Vp, "Vp"}, {att_img, "att_img"}, {Wg, "Wg"}, {Ws, "Ws"}, {bs, "bs"}, {wh, "wh"}, {scratch, "scratch"});
    return gridtd_attention(tr, t, Vp, att_img, Wg, Ws, bs, wh, scratch, nullptr, stream);
}
}  // extern "C"
static int gridtd_attention(const lrpx_gridtd_trace* tr, int t, const float* Vp, const float* att_img, const float* Wg, const float* Ws,
                            const float* bs, const float* wh, float* scratch, const float* sg, void* stream) {
    LRPX_TRY(check_trace(tr));
    LRPX_REQUIRE(Vp && att_img && Wg && Ws && bs && wh && scratch && t >= 0 && t < tr->T, "gridtd_fwd_attention: bad arguments");
    const GridFwd g = to_fwd(tr);
    hipLaunchKernelGGL(gridtd_fwd_att_scores_kernel, dim3(tr->B, (tr->P + ATT_PB - 1) / ATT_PB), dim3(256),
                       (size_t)2 * tr->H * sizeof(float), (hipStream_t)stream, g, t, att_img, Wg, Ws, bs, wh, scratch, sg);
    LRPX_TRY(check_launch("gridtd_fwd_att_scores"));
    hipLaunchKernelGGL(gridtd_fwd_att_context_kernel, dim3(tr->B, (tr->H + ATT_CB - 1) / ATT_CB), dim3(256),
                       (size_t)((2 * tr->P + 1 > 256 ? 2 * tr->P + 1 : 256) + 8) * sizeof(float), (hipStream_t)stream, g, t, Vp,
                       wh, scratch);
    return check_launch("gridtd_fwd_att_context");
}
extern "C" {

int lrpx_target_logit(const float* hc, const float* fcw, const float* fcb, const long long* tok, int tok_ld,
                      float* logit, int B, int T, int H, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_target_logit", {hc, "hc"}, {fcw, "fcw"}, {fcb, "fcb"}, {tok, "tok"}, {logit, "logit"});
    LRPX_REQUIRE(hc && fcw && fcb && tok && logit, "target_logit: null pointer");
    hipLaunchKernelGGL(target_logit_kernel, dim3((B * T + 3) / 4), dim3(256), 0, (hipStream_t)stream, hc, fcw, fcb, tok,
                       tok_ld, logit, B, T, H);
    return check_launch("target_logit");
}

static GridRel to_rel(const lrpx_gridtd_trace* t, const lrpx_gridtd_relstate* r) {
    GridRel g;
    g.B = t->B; g.T = t->T; g.H = t->H; g.E = t->E; g.P = t->P; g.lens = r->lens;
    g.xh1 = t->xh1; g.xh2 = t->xh2; g.h2 = t->h2; g.c1 = t->c1; g.c2 = t->c2;
    g.g1 = t->g1; g.i1 = t->i1; g.f1 = t->f1; g.g2 = t->g2; g.i2 = t->i2; g.f2 = t->f2;
    g.s = t->s; g.ctx = t->ctx; g.ctx_hat = t->ctx_hat; g.hc = t->hc; g.beta = t->beta;
    g.r_h2n = r->r_h2n; g.r_c2 = r->r_c2; g.r_c1 = r->r_c1; g.r_ch0 = r->r_ch0; g.r_h2p = r->r_h2p; g.r_glob = r->r_glob;
    g.A = r->A; g.rx = r->rx; g.wacc = r->wacc; g.r_words = r->r_words;
    return g;
}

static int check_rel(const lrpx_gridtd_trace* t, const lrpx_gridtd_relstate* r) {
    LRPX_TRY(check_trace(t));
    LRPX_REQUIRE(r && r->r_h2n && r->r_c2 && r->r_c1 && r->r_ch0 && r->r_h2p && r->r_glob && r->A && r->rx && r->wacc &&
                     r->r_words, "gridtd: null relevance-state tensor");
    return LRPX_OK;
}

int lrpx_gridtd_rel_init(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, const float* fcw,
                         const float* logit, const long long* tok, int tok_ld, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_rel_init", {fcw, "fcw"}, {logit, "logit"}, {tok, "tok"});
    LRPX_TRY(check_rel(tr, rs));
    LRPX_REQUIRE(fcw && logit && tok, "gridtd_rel_init: null pointer");
    hipLaunchKernelGGL(gridtd_rel_init_kernel, dim3(tr->B * tr->T), dim3(256), 0, (hipStream_t)stream, to_rel(tr, rs),
                       fcw, logit, tok, tok_ld);
    return check_launch("gridtd_rel_init");
}

int lrpx_gridtd_rel_step(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, int s, int phase, void* stream) {
    LRPX_TRY(check_rel(tr, rs));
    LRPX_REQUIRE(s >= 0 && s < tr->T && phase >= 0 && phase <= 3, "gridtd_rel_step: bad step/phase");
    const GridRel g = to_rel(tr, rs);
    const dim3 grid(tr->B * tr->T), blk(256);
    if (phase == 0) hipLaunchKernelGGL(gridtd_rel_a_kernel, grid, blk, 0, (hipStream_t)stream, g, s);
    else if (phase == 1) hipLaunchKernelGGL(gridtd_rel_b_kernel, grid, blk, 0, (hipStream_t)stream, g, s);
    else if (phase == 2) hipLaunchKernelGGL(gridtd_rel_c_kernel, grid, blk, 0, (hipStream_t)stream, g, s);
    else hipLaunchKernelGGL(gridtd_rel_ca_kernel, grid, blk, 0, (hipStream_t)stream, g, s);       // 3: phase 2 of s + phase 0 of s + 1
    return check_launch("gridtd_rel_step");
}

int lrpx_gridtd_rel_glob(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, const float* glob_pre,
                         float* a_glob, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_rel_glob", {glob_pre, "glob_pre"}, {a_glob, "a_glob"});
    LRPX_TRY(check_rel(tr, rs));
    LRPX_REQUIRE(glob_pre && a_glob, "gridtd_rel_glob: null pointer");
    hipLaunchKernelGGL(gridtd_rel_glob_kernel, dim3(tr->B * tr->T), dim3(256), 0, (hipStream_t)stream, to_rel(tr, rs),
                       glob_pre, a_glob);
    return check_launch("gridtd_rel_glob");
}

int lrpx_rel_avg_u(const float* r_avg, const float* avg, float* u, int rows, int T, int C, int P, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_rel_avg_u", {r_avg, "r_avg"}, {avg, "avg"}, {u, "u"});
    LRPX_REQUIRE(r_avg && avg && u && rows > 0, "rel_avg_u: bad arguments");
    hipLaunchKernelGGL(gridtd_rel_u_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, r_avg, avg, u, T, C, P);
    return check_launch("rel_avg_u");
}

int lrpx_gridtd_rel_pix_rows(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, const float* Vp,
                             const float* proj_pre, float* a_proj, const int32_t* rows, int n_rows, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_rel_pix_rows", {Vp, "Vp"}, {proj_pre, "proj_pre"}, {a_proj, "a_proj"}, {rows, "rows"});
    LRPX_TRY(check_rel(tr, rs));
    LRPX_REQUIRE(Vp && proj_pre && a_proj, "gridtd_rel_pix: null pointer");
    LRPX_REQUIRE(rows ? (n_rows > 0 && n_rows <= tr->B * tr->T) : n_rows == tr->B * tr->T, "gridtd_rel_pix: bad row list");
    const int kchunk = 28;
    const size_t lds = (size_t)tr->T * kchunk * sizeof(float);
    hipLaunchKernelGGL(gridtd_rel_pix_kernel, dim3(n_rows, (tr->P + kchunk - 1) / kchunk), dim3(256), lds,
                       (hipStream_t)stream, to_rel(tr, rs), Vp, proj_pre, tr->alpha, a_proj, kchunk, rows);
    return check_launch("gridtd_rel_pix");
}

int lrpx_gridtd_rel_pix(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, const float* Vp,
                        const float* proj_pre, float* a_proj, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_rel_pix", {Vp, "Vp"}, {proj_pre, "proj_pre"}, {a_proj, "a_proj"});
    LRPX_REQUIRE(tr, "gridtd_rel_pix: null trace");
    return lrpx_gridtd_rel_pix_rows(tr, rs, Vp, proj_pre, a_proj, nullptr, tr->B * tr->T, stream);
}

int lrpx_rel_words_norm(float* r_words, int rows, int T, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_rel_words_norm", {r_words, "r_words"});
    LRPX_REQUIRE(r_words && rows > 0 && T > 0, "rel_words_norm: bad arguments");
    hipLaunchKernelGGL(rel_words_norm_kernel, dim3((rows + 63) / 64), dim3(64), 0, (hipStream_t)stream, r_words, rows, T);
    return check_launch("rel_words_norm");
}


static GridGrad to_grad(const lrpx_gridtd_trace* t, const lrpx_gridtd_gradstate* r) {
    GridGrad g;
    g.B = t->B; g.T = t->T; g.H = t->H; g.E = t->E; g.P = t->P; g.lens = r->lens;
    g.c1 = t->c1; g.c2 = t->c2; g.g1 = t->g1; g.i1 = t->i1; g.f1 = t->f1; g.o1 = t->o1;
    g.g2 = t->g2; g.i2 = t->i2; g.f2 = t->f2; g.o2 = t->o2; g.sgate = t->sgate; g.beta = t->beta;
    g.d_h2n = r->d_h2n; g.d_c2 = r->d_c2; g.d_c1 = r->d_c1; g.d_ch0 = r->d_ch0; g.d_h2p = r->d_h2p; g.d_glob = r->d_glob;
    g.gates = r->gates; g.dx = r->dx; g.wacc = r->wacc; g.r_words = r->r_words;
    return g;
}

static int check_grad(const lrpx_gridtd_trace* t, const lrpx_gridtd_gradstate* r) {
    LRPX_TRY(check_trace(t));
    LRPX_REQUIRE(t->o1 && t->o2 && t->sgate, "gridtd_grad: the trace must carry o1/o2/sgate (gradient-explainer trace)");
    LRPX_REQUIRE(t->E == t->H, "gridtd_grad: embed_dim must equal hidden_dim");
    LRPX_REQUIRE(r && r->d_h2n && r->d_c2 && r->d_c1 && r->d_ch0 && r->d_h2p && r->d_glob && r->gates && r->dx && r->wacc &&
                     r->r_words, "gridtd_grad: null state tensor");
    return LRPX_OK;
}

int lrpx_gridtd_grad_init(const lrpx_gridtd_trace* tr, const lrpx_gridtd_gradstate* gs, const float* fcw,
                          const long long* tok, int tok_ld, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_grad_init", {fcw, "fcw"}, {tok, "tok"});
    LRPX_TRY(check_grad(tr, gs));
    LRPX_REQUIRE(fcw && tok, "gridtd_grad_init: null pointer");
    hipLaunchKernelGGL(gridtd_grad_init_kernel, dim3(tr->B * tr->T), dim3(256), 0, (hipStream_t)stream, to_grad(tr, gs),
                       fcw, tok, tok_ld);
    return check_launch("gridtd_grad_init");
}

int lrpx_gridtd_grad_step(const lrpx_gridtd_trace* tr, const lrpx_gridtd_gradstate* gs, int s, int phase, void* stream) {
    LRPX_TRY(check_grad(tr, gs));
    LRPX_REQUIRE(s >= 0 && s < tr->T && phase >= 0 && phase <= 2, "gridtd_grad_step: bad step/phase");
    const GridGrad g = to_grad(tr, gs);
    const dim3 grid(tr->B * tr->T), blk(256);
    if (phase == 0) hipLaunchKernelGGL(gridtd_grad_a_kernel, grid, blk, 0, (hipStream_t)stream, g, s);
    else if (phase == 1) hipLaunchKernelGGL(gridtd_grad_b_kernel, grid, blk, 0, (hipStream_t)stream, g, s);
    else hipLaunchKernelGGL(gridtd_grad_c_kernel, grid, blk, 0, (hipStream_t)stream, g, s);
    return check_launch("gridtd_grad_step");
}

int lrpx_spread_pixels_rows(const float* wacc, const float* alpha, const int32_t* lens, float* a_proj, int B, int T, int H,
                            int P, const int32_t* rows, int n_rows, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_spread_pixels_rows", {wacc, "wacc"}, {alpha, "alpha"}, {lens, "lens"}, {a_proj, "a_proj"}, {rows, "rows"});
    LRPX_REQUIRE(wacc && alpha && a_proj && B > 0 && T > 0, "spread_pixels: bad arguments");
    LRPX_REQUIRE(rows ? (n_rows > 0 && n_rows <= B * T) : n_rows == B * T, "spread_pixels: bad row list");
    const int kchunk = 28;
    hipLaunchKernelGGL(spread_pixels_kernel, dim3(n_rows, (P + kchunk - 1) / kchunk), dim3(256),
                       (size_t)T * kchunk * sizeof(float), (hipStream_t)stream, wacc, alpha, lens, a_proj, T, H, P, kchunk, rows);
    return check_launch("spread_pixels");
}

int lrpx_spread_pixels(const float* wacc, const float* alpha, const int32_t* lens, float* a_proj, int B, int T, int H,
                       int P, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_spread_pixels", {wacc, "wacc"}, {alpha, "alpha"}, {lens, "lens"}, {a_proj, "a_proj"});
    return lrpx_spread_pixels_rows(wacc, alpha, lens, a_proj, B, T, H, P, nullptr, B * T, stream);
}

int lrpx_scale(const float* x, float* y, long n, float alpha, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_scale", {x, "x"}, {y, "y"});
    LRPX_REQUIRE(x && y && n > 0, "scale: bad arguments");
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n, alpha);
    return check_launch("scale");
}

int lrpx_positive_mask(const float* x, float* y, long n, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_positive_mask", {x, "x"}, {y, "y"});
    LRPX_REQUIRE(x && y && n > 0, "positive_mask: bad arguments");
    hipLaunchKernelGGL(positive_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, y, n);
    return check_launch("positive_mask");
}

static AoaFwd to_afwd(const lrpx_aoa_trace* t) {
    AoaFwd g;
    g.B = t->B; g.T = t->T; g.H = t->H; g.E = t->E; g.P = t->P; g.NH = t->NH;
    g.xh = t->xh; g.h = t->h; g.c = t->c; g.g = t->g; g.i = t->i; g.f = t->f;
    g.ctx = t->ctx; g.lin = t->lin; g.c_aoa = t->c_aoa; g.hc = t->hc; g.alpha = t->alpha;
    g.o = t->o; g.sg = t->sg;
    return g;
}

static int check_atrace(const lrpx_aoa_trace* t) {
    LRPX_REQUIRE(t && t->B > 0 && t->T > 0 && t->H == 512 && t->E % 4 == 0 && t->P > 0 && t->NH > 0 && t->H % t->NH == 0,
                 "aoa: unsupported trace dims (H must be 512)");
    LRPX_REQUIRE(t->xh && t->h && t->c && t->g && t->i && t->f && t->ctx && t->lin && t->c_aoa && t->hc && t->alpha,
                 "aoa: null trace tensor");
    return LRPX_OK;
}

int lrpx_aoa_fwd_pre(const lrpx_aoa_trace* tr, int t, const float* glob, const float* emb, const long long* tok,
                     int tok_ld, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_fwd_pre", {glob, "glob"}, {emb, "emb"}, {tok, "tok"});
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(glob && emb && tok && t >= 0 && t < tr->T, "aoa_fwd_pre: bad arguments");
    hipLaunchKernelGGL(aoa_fwd_pre_kernel, dim3(tr->B), dim3(256), 0, (hipStream_t)stream, to_afwd(tr), t, glob, emb, tok,
                       tok_ld);
    return check_launch("aoa_fwd_pre");
}

int lrpx_aoa_fwd_lstm(const lrpx_aoa_trace* tr, int t, const float* zz, int ldz, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_fwd_lstm", {zz, "zz"});
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(zz && t >= 0 && t < tr->T, "aoa_fwd_lstm: bad arguments");
    hipLaunchKernelGGL(aoa_fwd_lstm_kernel, dim3(tr->B), dim3(256), 0, (hipStream_t)stream, to_afwd(tr), t, zz, ldz);
    return check_launch("aoa_fwd_lstm");
}

int lrpx_aoa_fwd_attention(const lrpx_aoa_trace* tr, int t, const float* qg, int ldq, const float* key,
                           const float* value, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_fwd_attention", {qg, "qg"}, {key, "key"}, {value, "value"});
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(qg && key && value && t >= 0 && t < tr->T, "aoa_fwd_attention: bad arguments");
    const size_t lds = (size_t)(tr->H / tr->NH + tr->P + 8) * sizeof(float);
    hipLaunchKernelGGL(aoa_fwd_attention_kernel, dim3(tr->B, tr->NH), dim3(256), lds, (hipStream_t)stream, to_afwd(tr), t,
                       qg, ldq, key, value);
    return check_launch("aoa_fwd_attention");
}

int lrpx_aoa_fwd_post(const lrpx_aoa_trace* tr, int t, const float* qg, int ldq, const float* lin, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_fwd_post", {qg, "qg"}, {lin, "lin"});
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(qg && lin && t >= 0 && t < tr->T, "aoa_fwd_post: bad arguments");
    hipLaunchKernelGGL(aoa_fwd_post_kernel, dim3(tr->B), dim3(256), 0, (hipStream_t)stream, to_afwd(tr), t, qg, ldq, lin);
    return check_launch("aoa_fwd_post");
}

static int to_agrad(const lrpx_aoa_trace* t, const lrpx_aoa_gradstate* r, AoaGrad* out) {
    LRPX_TRY(check_atrace(t));
    LRPX_REQUIRE(t->o && t->sg, "aoa gradient: the trace lacks the output gate / aoa gate (allocate it with grad=True)");
    LRPX_REQUIRE(r && r->d_h && r->d_c && r->dA && r->dB && r->gates && r->dx && r->d_glob && r->r_words,
                 "aoa gradient: null state tensor");
    AoaGrad g;
    g.B = t->B; g.T = t->T; g.H = t->H; g.E = t->E; g.P = t->P; g.NH = t->NH; g.lens = r->lens;
    g.c = t->c; g.g = t->g; g.i = t->i; g.f = t->f; g.o = t->o; g.sg = t->sg; g.lin = t->lin; g.alpha = t->alpha;
    g.d_h = r->d_h; g.d_c = r->d_c; g.dA = r->dA; g.dB = r->dB; g.gates = r->gates; g.dx = r->dx; g.d_glob = r->d_glob;
    g.r_words = r->r_words;
    *out = g;
    return LRPX_OK;
}

int lrpx_aoa_grad_init(const lrpx_aoa_trace* tr, const lrpx_aoa_gradstate* gs, const float* fcw, const long long* tok,
                       int tok_ld, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_grad_init", {fcw, "fcw"}, {tok, "tok"});
    AoaGrad g;
    LRPX_TRY(to_agrad(tr, gs, &g));
    LRPX_REQUIRE(fcw && tok, "aoa_grad_init: null pointer");
    hipLaunchKernelGGL(aoa_grad_init_kernel, dim3(g.B * g.T), dim3(256), 0, (hipStream_t)stream, g, fcw, tok, tok_ld);
    return check_launch("aoa_grad_init");
}

int lrpx_aoa_grad_step(const lrpx_aoa_trace* tr, const lrpx_aoa_gradstate* gs, int s, int phase, void* stream) {
    AoaGrad g;
    LRPX_TRY(to_agrad(tr, gs, &g));
    LRPX_REQUIRE(s >= 0 && s < g.T && (phase == 0 || phase == 1), "aoa_grad_step: bad step / phase");
    if (phase == 0) hipLaunchKernelGGL(aoa_grad_step0_kernel, dim3(g.B * g.T), dim3(256), 0, (hipStream_t)stream, g, s);
    else hipLaunchKernelGGL(aoa_grad_step1_kernel, dim3(g.B * g.T), dim3(256), 0, (hipStream_t)stream, g, s);
    return check_launch("aoa_grad_step");
}

int lrpx_aoa_grad_pix_rows(const lrpx_aoa_trace* tr, int head, const float* v1, const float* v2, float* d_feat, int C,
                           const int32_t* rows, int n_rows, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_grad_pix_rows", {v1, "v1"}, {v2, "v2"}, {d_feat, "d_feat"}, {rows, "rows"});
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(v1 && v2 && d_feat && head >= 0 && head < tr->NH && C > 0 && C % 4 == 0, "aoa_grad_pix: bad arguments");
    LRPX_REQUIRE(rows ? (n_rows > 0 && n_rows <= tr->B * tr->T) : n_rows == tr->B * tr->T, "aoa_grad_pix: bad row list");
    const long total = (long)n_rows * tr->P * (C / 4);
    hipLaunchKernelGGL(aoa_grad_pix_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       tr->alpha, tr->T, tr->NH, tr->P, head, v1, v2, d_feat, C / 4, total, rows);
    return check_launch("aoa_grad_pix");
}

int lrpx_aoa_grad_pix(const lrpx_aoa_trace* tr, int head, const float* v1, const float* v2, float* d_feat, int C,
                      void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_grad_pix", {v1, "v1"}, {v2, "v2"}, {d_feat, "d_feat"});
    LRPX_REQUIRE(tr, "aoa_grad_pix: null trace");
    return lrpx_aoa_grad_pix_rows(tr, head, v1, v2, d_feat, C, nullptr, tr->B * tr->T, stream);
}

int lrpx_keep_cols(float* x, long rows, int ncol, int lo, int hi, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_keep_cols", {x, "x"});
    LRPX_REQUIRE(x && rows > 0 && ncol > 0 && lo >= 0 && hi <= ncol && lo < hi, "keep_cols: bad arguments");
    const long total = rows * ncol;
    hipLaunchKernelGGL(keep_cols_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, ncol,
                       lo, hi, total);
    return check_launch("keep_cols");
}

static AoaRel to_arel(const lrpx_aoa_trace* t, const lrpx_aoa_relstate* r) {
    AoaRel g;
    g.B = t->B; g.T = t->T; g.H = t->H; g.E = t->E; g.P = t->P; g.NH = t->NH; g.lens = r->lens;
    g.xh = t->xh; g.h = t->h; g.c = t->c; g.g = t->g; g.i = t->i; g.ctx = t->ctx; g.lin = t->lin; g.c_aoa = t->c_aoa;
    g.hc = t->hc; g.alpha = t->alpha;
    g.r_hn = r->r_hn; g.r_glob = r->r_glob; g.A = r->A; g.rx = r->rx; g.r_words = r->r_words;
    return g;
}

static int check_arel(const lrpx_aoa_trace* t, const lrpx_aoa_relstate* r) {
    LRPX_TRY(check_atrace(t));
    LRPX_REQUIRE(r && r->r_hn && r->r_glob && r->A && r->rx && r->r_words, "aoa: null relevance-state tensor");
    return LRPX_OK;
}

int lrpx_aoa_rel_init(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, const float* fcw, const float* logit,
                      const long long* tok, int tok_ld, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_rel_init", {fcw, "fcw"}, {logit, "logit"}, {tok, "tok"});
    LRPX_TRY(check_arel(tr, rs));
    LRPX_REQUIRE(fcw && logit && tok, "aoa_rel_init: null pointer");
    hipLaunchKernelGGL(aoa_rel_init_kernel, dim3(tr->B * tr->T), dim3(256), 0, (hipStream_t)stream, to_arel(tr, rs), fcw,
                       logit, tok, tok_ld);
    return check_launch("aoa_rel_init");
}

int lrpx_aoa_rel_value_rows(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, const float* r_ctx, const float* value,
                            int head, float* a_val, const int32_t* rows, int n_rows, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_rel_value_rows", {r_ctx, "r_ctx"}, {value, "value"}, {a_val, "a_val"}, {rows, "rows"});
    LRPX_TRY(check_arel(tr, rs));
    LRPX_REQUIRE(r_ctx && value && a_val && head >= 0 && head < tr->NH, "aoa_rel_value: bad arguments");
    LRPX_REQUIRE(rows ? (n_rows > 0 && n_rows <= tr->B * tr->T) : n_rows == tr->B * tr->T, "aoa_rel_value: bad row list");
    hipLaunchKernelGGL(aoa_rel_value_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, to_arel(tr, rs),
                       r_ctx, value, head, a_val, rows, 0);
    return check_launch("aoa_rel_value");
}

int lrpx_aoa_rel_value_head(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, const float* r_ctx, const float* value,
                            int head, float* a_val_head, const int32_t* rows, int n_rows, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_rel_value_head", {r_ctx, "r_ctx"}, {value, "value"}, {a_val_head, "a_val_head"}, {rows, "rows"});
    LRPX_TRY(check_arel(tr, rs));
    LRPX_REQUIRE(r_ctx && value && a_val_head && head >= 0 && head < tr->NH && tr->H % tr->NH == 0, "aoa_rel_value_head: bad arguments");
    LRPX_REQUIRE(rows ? (n_rows > 0 && n_rows <= tr->B * tr->T) : n_rows == tr->B * tr->T, "aoa_rel_value_head: bad row list");
    hipLaunchKernelGGL(aoa_rel_value_kernel, dim3(n_rows), dim3(256), 0, (hipStream_t)stream, to_arel(tr, rs),
                       r_ctx, value, head, a_val_head, rows, 1);
    return check_launch("aoa_rel_value_head");
}

int lrpx_aoa_rel_value(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, const float* r_ctx, const float* value,
                       int head, float* a_val, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_rel_value", {r_ctx, "r_ctx"}, {value, "value"}, {a_val, "a_val"});
    LRPX_REQUIRE(tr, "aoa_rel_value: null trace");
    return lrpx_aoa_rel_value_rows(tr, rs, r_ctx, value, head, a_val, nullptr, tr->B * tr->T, stream);
}

int lrpx_aoa_rel_step(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, int s, int phase, void* stream) {
    LRPX_TRY(check_arel(tr, rs));
    LRPX_REQUIRE(s >= 0 && s < tr->T && (phase == 0 || phase == 1), "aoa_rel_step: bad step/phase");
    const AoaRel g = to_arel(tr, rs);
    if (phase == 0) hipLaunchKernelGGL(aoa_rel_a_kernel, dim3(tr->B * tr->T), dim3(256), 0, (hipStream_t)stream, g, s);
    else hipLaunchKernelGGL(aoa_rel_c_kernel, dim3(tr->B * tr->T), dim3(256), 0, (hipStream_t)stream, g, s);
    return check_launch("aoa_rel_step");
}

// The host loops of the AoA decoder in native code.  Every launch is one of the entry points above; what goes away is the
// interpreter between them: ~4 us per call through ctypes against ~1.5 us for a launch issued from here.  With a CNN stage
// behind it that is noise; the bottom-up path (no CNN, ~330 launches of 4 - 25 us per step of 640 maps) was bound by it.
}  // extern "C"
template <int RT>
static int aoa_fused_steps(const lrpx_aoa_trace* tr, int t0, int t1, const lrpx_aoa_step_args* a, hipStream_t st) {
    const AoaFwd g = to_afwd(tr);
    const int H = tr->H, T = tr->T, B = tr->B;
    for (int t = t0; t < t1; ++t) {
        hipLaunchKernelGGL((aoa_linear_lstm_kernel<RT>), dim3(4 * H / 16), dim3(256), 0, st, g, t, a->w_cat_il, a->b_cat_il);
        LRPX_TRY(check_launch("aoa_linear_lstm"));
        LRPX_TRY(lrpx_linear_small(tr->h + (long)(t + 1) * H, (long)(T + 1) * H, a->w_qg, a->b_qg, a->qg, 2 * H, B, H, 2 * H, 0, st));
        LRPX_TRY(lrpx_aoa_fwd_attention(tr, t, a->qg, 2 * H, a->key, a->value, st));
        hipLaunchKernelGGL((aoa_linear_post_kernel<RT>), dim3(H / 16), dim3(256), 0, st, g, t, a->w_lin, a->b_lin, a->qg, 2 * H, a->glob,
                           a->emb, a->tok, a->tok_ld);
        LRPX_TRY(check_launch("aoa_linear_post"));
    }
    return LRPX_OK;
}

extern "C" {
int lrpx_aoa_fwd_steps(const lrpx_aoa_trace* tr, int t0, int t1, const lrpx_aoa_step_args* a, void* stream) {
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(a && a->glob && a->emb && a->tok && a->w_cat && a->w_qg && a->w_lin && a->key && a->value && a->zz && a->qg &&
                     a->lin && t0 >= 0 && t0 <= t1 && t1 <= tr->T, "aoa_fwd_steps: bad arguments");
    const int B = tr->B, T = tr->T, H = tr->H, W = tr->E + 2 * tr->H;
    // fused steps (4 launches instead of 7): needs the interleaved gate rows, <= 64 images, dims the matrix-core linear takes
    if (a->w_cat_il && B <= 64 && W % 16 == 0 && H % 16 == 0 && t0 < t1) {
        LRPX_TRY(lrpx_aoa_fwd_pre(tr, t0, a->glob, a->emb, a->tok, a->tok_ld, stream));      // (later rows: gathered by the step before)
        hipStream_t st = (hipStream_t)stream;
        if (B <= 16) return aoa_fused_steps<1>(tr, t0, t1, a, st);
        if (B <= 32) return aoa_fused_steps<2>(tr, t0, t1, a, st);
        if (B <= 48) return aoa_fused_steps<3>(tr, t0, t1, a, st);
        return aoa_fused_steps<4>(tr, t0, t1, a, st);
    }
    for (int t = t0; t < t1; ++t) {
        LRPX_TRY(lrpx_aoa_fwd_pre(tr, t, a->glob, a->emb, a->tok, a->tok_ld, stream));
        LRPX_TRY(lrpx_linear_small(tr->xh + (long)t * W, (long)T * W, a->w_cat, a->b_cat, a->zz, 4 * H, B, W, 4 * H, 0, stream));
        LRPX_TRY(lrpx_aoa_fwd_lstm(tr, t, a->zz, 4 * H, stream));
        LRPX_TRY(lrpx_linear_small(tr->h + (long)(t + 1) * H, (long)(T + 1) * H, a->w_qg, a->b_qg, a->qg, 2 * H, B, H, 2 * H, 0, stream));
        LRPX_TRY(lrpx_aoa_fwd_attention(tr, t, a->qg, 2 * H, a->key, a->value, stream));
        LRPX_TRY(lrpx_linear_small(tr->ctx + (long)t * H, (long)T * H, a->w_lin, a->b_lin, a->lin, H, B, H, H, 0, stream));
        LRPX_TRY(lrpx_aoa_fwd_post(tr, t, a->qg, 2 * H, a->lin, stream));
    }
    return LRPX_OK;
}

int lrpx_aoa_fwd_inputs(const lrpx_aoa_trace* tr, const float* glob, const float* emb, const long long* tok, int tok_ld,
                        float* xin, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_fwd_inputs", {glob, "glob"}, {emb, "emb"}, {tok, "tok"}, {xin, "xin"});
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(glob && emb && tok, "aoa_fwd_inputs: null pointer");
    hipLaunchKernelGGL(aoa_fwd_inputs_kernel, dim3(tr->B * tr->T), dim3(256), 0, (hipStream_t)stream, to_afwd(tr), glob, emb, tok,
                       tok_ld, xin);
    return check_launch("aoa_fwd_inputs");
}

static int aoa_recurrence(const lrpx_aoa_trace* tr, const float* w_hh_il, const float* zin, const float* tab, const float* gimg,
                          const long long* tok, int tok_ld, hipStream_t st) {
    if (tr->B > 64) {
        // more than 64 images: slices of at most 64 through the SAME kernels (the recurrence is independent per image), so that an image's
        // trace does not depend on the batch it sits in (B = 65 used to leave the decoupled path altogether)
        const long T = tr->T, H = tr->H, W = tr->E + 2L * tr->H;
        for (int b0 = 0; b0 < tr->B; b0 += 64) {
            lrpx_aoa_trace sub = *tr;
            sub.B = tr->B - b0 < 64 ? tr->B - b0 : 64;
            sub.xh += b0 * T * W;
            sub.h += b0 * (T + 1) * H; sub.c += b0 * (T + 1) * H;
            sub.g += b0 * T * H; sub.i += b0 * T * H; sub.f += b0 * T * H;
            sub.ctx += b0 * T * H; sub.lin += b0 * T * H; sub.c_aoa += b0 * T * H; sub.hc += b0 * T * H;
            sub.alpha += b0 * T * tr->NH * (long)tr->P;
            if (sub.o) sub.o += b0 * T * H;
            if (sub.sg) sub.sg += b0 * T * H;
            LRPX_TRY(aoa_recurrence(&sub, w_hh_il, zin ? zin + b0 * T * 4 * H : nullptr, tab, gimg ? gimg + b0 * 4L * H : nullptr,
                                    tok ? tok + (long)b0 * tok_ld : nullptr, tok_ld, st));
        }
        return LRPX_OK;
    }
    const AoaFwd g = to_afwd(tr);
    const int H = tr->H, B = tr->B;
    for (int t = 0; t < tr->T; ++t) {
        if (B <= 16) hipLaunchKernelGGL((aoa_rec_lstm_kernel<1>), dim3(4 * H / 16), dim3(256), 0, st, g, t, w_hh_il, zin, tab, gimg, tok, tok_ld);
        else if (B <= 32) hipLaunchKernelGGL((aoa_rec_lstm_kernel<2>), dim3(4 * H / 16), dim3(256), 0, st, g, t, w_hh_il, zin, tab, gimg, tok, tok_ld);
        else if (B <= 48) hipLaunchKernelGGL((aoa_rec_lstm_kernel<3>), dim3(4 * H / 16), dim3(256), 0, st, g, t, w_hh_il, zin, tab, gimg, tok, tok_ld);
        else hipLaunchKernelGGL((aoa_rec_lstm_kernel<4>), dim3(4 * H / 16), dim3(256), 0, st, g, t, w_hh_il, zin, tab, gimg, tok, tok_ld);
        LRPX_TRY(check_launch("aoa_rec_lstm"));
    }
    return LRPX_OK;
}

int lrpx_aoa_fwd_recurrence(const lrpx_aoa_trace* tr, const float* w_hh_il, const float* zin, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_fwd_recurrence", {w_hh_il, "w_hh_il"}, {zin, "zin"});
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(w_hh_il && zin && tr->H % 16 == 0, "aoa_fwd_recurrence: bad arguments (H %% 16)");
    return aoa_recurrence(tr, w_hh_il, zin, nullptr, nullptr, nullptr, 0, (hipStream_t)stream);
}

int lrpx_aoa_fwd_recurrence_tab(const lrpx_aoa_trace* tr, const float* w_hh_il, const float* tab, const float* gimg,
                                const long long* tok, int tok_ld, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_fwd_recurrence_tab", {w_hh_il, "w_hh_il"}, {tab, "tab"}, {gimg, "gimg"}, {tok, "tok"});
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(w_hh_il && tab && gimg && tok && tok_ld >= tr->T && tr->H % 16 == 0,
                 "aoa_fwd_recurrence_tab: bad arguments (H %% 16)");
    return aoa_recurrence(tr, w_hh_il, nullptr, tab, gimg, tok, tok_ld, (hipStream_t)stream);
}

int lrpx_aoa_fwd_gather_h(const lrpx_aoa_trace* tr, float* hn, uint32_t* hn_amax, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_fwd_gather_h", {hn, "hn"}, {hn_amax, "hn_amax"});
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(hn, "aoa_fwd_gather_h: null pointer");
    hipLaunchKernelGGL(aoa_fwd_gather_h_kernel, dim3(tr->B * tr->T), dim3(256), 0, (hipStream_t)stream, to_afwd(tr), hn, hn_amax);
    return check_launch("aoa_fwd_gather_h");
}

int lrpx_aoa_fwd_attention_all(const lrpx_aoa_trace* tr, const float* qg, int ldq, const float* key, const float* value,
                               uint32_t* ctx_amax, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_fwd_attention_all", {qg, "qg"}, {key, "key"}, {value, "value"}, {ctx_amax, "ctx_amax"});
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(qg && key && value && tr->H % tr->NH == 0, "aoa_fwd_attention_all: bad arguments");
    const size_t lds = (size_t)(tr->H / tr->NH + tr->P) * sizeof(float);
    hipLaunchKernelGGL(aoa_fwd_attention_all_kernel, dim3(tr->B * tr->T, tr->NH), dim3(64), lds, (hipStream_t)stream, to_afwd(tr), qg,
                       ldq, key, value, ctx_amax);
    return check_launch("aoa_fwd_attention_all");
}

int lrpx_aoa_fwd_post_all(const lrpx_aoa_trace* tr, const float* qg, int ldq, const float* lin, uint32_t* hc_amax, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_fwd_post_all", {qg, "qg"}, {lin, "lin"}, {hc_amax, "hc_amax"});
    LRPX_TRY(check_atrace(tr));
    LRPX_REQUIRE(qg && lin, "aoa_fwd_post_all: null pointer");
    hipLaunchKernelGGL(aoa_fwd_post_all_kernel, dim3(tr->B * tr->T), dim3(256), 0, (hipStream_t)stream, to_afwd(tr), qg, ldq, lin, hc_amax);
    return check_launch("aoa_fwd_post_all");
}

int lrpx_aoa_rel_steps(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, int n_steps, const lrpx_conv_desc* dense,
                       const int32_t* idx, int idx_ld, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_rel_steps", {idx, "idx"});
    LRPX_TRY(check_arel(tr, rs));
    LRPX_REQUIRE(dense && idx && n_steps >= 0 && n_steps <= tr->T && idx_ld >= tr->B * tr->T, "aoa_rel_steps: bad arguments");
    lrpx_conv_desc d = *dense;
    const AoaRel g = to_arel(tr, rs);
    if (n_steps > 0) LRPX_TRY(lrpx_aoa_rel_step(tr, rs, 0, 0, stream));
    for (int s = 0; s < n_steps; ++s) {
        d.map2img = idx + (long)s * idx_ld;          // row -> source row of the multiplicand at lock-step s
        LRPX_TRY(lrpx_conv_mfma(&d, stream));
        if (s + 1 < n_steps) {                       // phase 1 of step s and phase 0 of step s + 1 in one launch
            hipLaunchKernelGGL(aoa_rel_ca_kernel, dim3(tr->B * tr->T), dim3(256), 0, (hipStream_t)stream, g, s);
            LRPX_TRY(check_launch("aoa_rel_ca"));
        } else {
            LRPX_TRY(lrpx_aoa_rel_step(tr, rs, s, 1, stream));
        }
    }
    return LRPX_OK;
}

// ---- the host loops of the gridTD decoder in native code (round 5; as lrpx_aoa_fwd_steps / lrpx_aoa_rel_steps for the AoA decoder): every
// launch is one of the entry points above, what goes away is the interpreter between them (~4 us per call through ctypes against ~1.5 us
// from here): 7 x T + 5 x T calls per explanation - a third of the 5 ms one image costs through the drop-in class
}  // extern "C"
template <int RT>
static int gridtd_fused_steps(const lrpx_gridtd_trace* tr, int t0, int t1, const lrpx_gridtd_step_args* a, hipStream_t st) {
    const GridFwd g = to_fwd(tr);
    const int H = tr->H;
    for (int t = t0; t < t1; ++t) {
        hipLaunchKernelGGL((gridtd_linear_lstm_kernel<RT, 1>), dim3(H / 4 + H / 16), dim3(256), 0, st, g, t, a->w_il1, a->b_il1, a->w_cat1,
                           a->b_cat1, a->zz1, a->glob, a->emb, a->tok, a->tok_ld);
        LRPX_TRY(check_launch("gridtd_linear_lstm1"));
        LRPX_TRY(gridtd_attention(tr, t, a->Vp, a->att_img, a->Wg, a->Ws, a->bs, a->wh, a->att_scratch, a->zz1, st));
        hipLaunchKernelGGL((gridtd_linear_lstm_kernel<RT, 2>), dim3(H / 4), dim3(256), 0, st, g, t, a->w_il2, a->b_il2, a->w_cat2, a->b_cat2,
                           a->zz1, a->glob, a->emb, a->tok, a->tok_ld);
        LRPX_TRY(check_launch("gridtd_linear_lstm2"));
    }
    return LRPX_OK;
}
extern "C" {
int lrpx_gridtd_fwd_steps(const lrpx_gridtd_trace* tr, int t0, int t1, const lrpx_gridtd_step_args* a, void* stream) {
    LRPX_TRY(check_trace(tr));
    LRPX_REQUIRE(a && a->glob && a->emb && a->tok && a->w_cat1 && a->b_cat1 && a->w_cat2 && a->b_cat2 && a->Vp && a->att_img && a->Wg &&
                     a->Ws && a->bs && a->wh && a->zz1 && a->zz2 && a->att_scratch && t0 >= 0 && t0 <= t1 && t1 <= tr->T,
                 "gridtd_fwd_steps: bad arguments");
    const int B = tr->B, T = tr->T, H = tr->H, E = tr->E, W1 = 2 * E + 2 * H;
    // fused steps (4 launches instead of 7): interleaved gate rows given, <= 64 images, dims the matrix-core linear takes
    if (a->w_il1 && a->b_il1 && a->w_il2 && a->b_il2 && B <= 64 && W1 % 16 == 0 && H % 16 == 0 && t0 < t1) {
        LRPX_TRY(lrpx_gridtd_fwd_pre(tr, t0, a->glob, a->emb, a->tok, a->tok_ld, stream));      // (later rows: written by the step before)
        hipStream_t st = (hipStream_t)stream;
        if (B <= 16) return gridtd_fused_steps<1>(tr, t0, t1, a, st);
        if (B <= 32) return gridtd_fused_steps<2>(tr, t0, t1, a, st);
        if (B <= 48) return gridtd_fused_steps<3>(tr, t0, t1, a, st);
        return gridtd_fused_steps<4>(tr, t0, t1, a, st);
    }
    for (int t = t0; t < t1; ++t) {
        LRPX_TRY(lrpx_gridtd_fwd_pre(tr, t, a->glob, a->emb, a->tok, a->tok_ld, stream));
        LRPX_TRY(lrpx_linear_small(tr->xh1 + (long)t * W1, (long)T * W1, a->w_cat1, a->b_cat1, a->zz1, 5 * H, B, W1, 5 * H, 0, stream));
        LRPX_TRY(lrpx_gridtd_fwd_lstm(tr, t, a->zz1, 5 * H, 1, stream));
        LRPX_TRY(lrpx_gridtd_fwd_attention(tr, t, a->Vp, a->att_img, a->Wg, a->Ws, a->bs, a->wh, a->att_scratch, stream));
        LRPX_TRY(lrpx_linear_small(tr->xh2 + (long)t * 3 * H, (long)T * 3 * H, a->w_cat2, a->b_cat2, a->zz2, 4 * H, B, 3 * H, 4 * H, 0, stream));
        LRPX_TRY(lrpx_gridtd_fwd_lstm(tr, t, a->zz2, 4 * H, 2, stream));
    }
    return LRPX_OK;
}

int lrpx_gridtd_rel_steps(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, int n_steps, const lrpx_conv_desc* dense2,
                          const lrpx_conv_desc* dense1, const int32_t* idx, int idx_ld, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_gridtd_rel_steps", {idx, "idx"});
    LRPX_TRY(check_rel(tr, rs));
    LRPX_REQUIRE(dense1 && dense2 && idx && n_steps >= 0 && n_steps <= tr->T && idx_ld >= tr->B * tr->T, "gridtd_rel_steps: bad arguments");
    lrpx_conv_desc d2 = *dense2, d1 = *dense1;
    for (int s = 0; s < n_steps; ++s) {
        d2.map2img = d1.map2img = idx + (long)s * idx_ld;      // row -> source row of the multiplicands at lock-step s
        if (s == 0) LRPX_TRY(lrpx_gridtd_rel_step(tr, rs, s, 0, stream));
        LRPX_TRY(lrpx_conv_mfma(&d2, stream));
        LRPX_TRY(lrpx_gridtd_rel_step(tr, rs, s, 1, stream));
        LRPX_TRY(lrpx_conv_mfma(&d1, stream));
        // the tail of lock-step s and the head of s + 1 in one launch
        LRPX_TRY(lrpx_gridtd_rel_step(tr, rs, s, s + 1 < n_steps ? 3 : 2, stream));
    }
    return LRPX_OK;
}

int lrpx_aoa_rel_steps_fused(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, const lrpx_conv_desc* dense, const int32_t* idx,
                             int idx_ld, float* a_alt, float* wpart, float* coef, void* stream) {
    LRPX_CHECK_PTRS_OPT("lrpx_aoa_rel_steps_fused", {idx, "idx"}, {a_alt, "a_alt"}, {wpart, "wpart"}, {coef, "coef"});
    LRPX_TRY(check_arel(tr, rs));
    const int T = tr->T, rows = tr->B * tr->T, H = tr->H;
    LRPX_REQUIRE(dense && idx && a_alt && wpart && coef && idx_ld >= rows, "aoa_rel_steps_fused: bad arguments");
    LRPX_REQUIRE((dense->f16x3 == 1 || (dense->f16x3 == 0 && !dense->bf16x6)) && dense->taps == 1 && dense->epi == EPI_REL && dense->in == rs->A && dense->x && !dense->u &&
                     dense->cin == H && tr->E == H && H == 512 && dense->n_oc == 3 * H && dense->oc_split == 3 * H && dense->n_maps == rows &&
                     dense->pix_per_map == 1, "aoa_rel_steps_fused: the fused lock-step is built for E = H = 512 on the f16x3 gate rule or (f16x3 = 0: weights "
                     "from lrpx_pack_weights, DENSE_T, kc = 32) on the fp32 MFMA");
    hipStream_t st = (hipStream_t)stream;
    LRPX_TRY(lrpx_aoa_rel_step(tr, rs, 0, 0, stream));                 // A of lock-step 0 (:1116-1120)
    ConvArgs a = {};
    a.wp = dense->wpacked; a.n_maps = rows; a.cin = H; a.n_oc = 3 * H; a.pix_per_map = 1; a.epi = EPI_REL; a.oc_split = 3 * H;
    a.X = dense->x; a.ksplit = 1;
    float *q1 = coef, *dg = coef + (size_t)rows * H;
    int* tmax = reinterpret_cast<int*>(coef + (size_t)2 * rows * H);
    hipLaunchKernelGGL(aoa_rel_coef_kernel, dim3(rows), dim3(256), 0, st, to_arel(tr, rs), q1, dg, tmax);
    LRPX_TRY(check_launch("aoa_rel_coef"));
    AoaStepFuse fz;
    fz.T = T; fz.tmax = tmax; fz.q1 = q1; fz.dg = dg; fz.r_glob = rs->r_glob; fz.wpart = wpart;
    float* buf[2] = {rs->A, a_alt};
    for (int s = 0; s < T; ++s) {
        a.in = buf[s & 1];
        a.map2img = idx + (long)s * idx_ld;        // row -> source row of the multiplicand at lock-step s
        fz.s = s; fz.A_next = buf[(s + 1) & 1];
        if (dense->f16x3) LRPX_TRY(launch_dense_small_f16x3_aoa_step(a, fz, st));
        else LRPX_TRY(launch_dense_ks_aoa_step(a, fz, st));
    }
    hipLaunchKernelGGL(rel_words_norm_parts_kernel, dim3((rows + 63) / 64), dim3(64), 0, st, rs->r_words, wpart, rs->lens, rows, T);
    return check_launch("rel_words_norm_parts");
}

}  // extern "C"
