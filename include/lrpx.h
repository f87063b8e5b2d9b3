/* lrpx — C ABI of the MI355X-native LRP relevance-propagation hot path.
 *
 * Drop-in boundary for SunJiamei/LRP-imagecaptioning-pytorch.  The reference is pure Python; these
 * are the entry points its Python would bind (ctypes) in place of the ATen op
 * clusters of its LRP path.  Each function cites the reference interface it replaces
 * (paths relative to the reference repo).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer to fp32 (or int32 where stated), owned by the caller
 *     (PyTorch's allocator); the library never frees or retains a caller pointer beyond the call
 *     and never allocates device memory (one 4-byte flag word of lrpx_check excepted): workspaces
 *     are caller-provided, sizes come from *_bytes().
 *   - `stream` is a hipStream_t passed as void*; all calls are asynchronous w.r.t. the host.
 *   - internal activation layout is pixel-major NHWC: a [map*H*W][C] row-major matrix.
 *   - return value: 0 = ok, otherwise an LRPX_E* code; lrpx_last_error_string() describes it.
 */
#ifndef LRPX_H
#define LRPX_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum {
    LRPX_OK = 0,
    LRPX_EINVAL = 1,      /* bad shape / null pointer / unsupported configuration */
    LRPX_EARCH = 2,       /* device is not gfx950 */
    LRPX_ELAUNCH = 3,     /* hip launch / runtime failure */
    LRPX_ENONFINITE = 4,  /* NaN/Inf found by lrpx_check_finite (the reference's inline asserts) */
    LRPX_EZERO = 5        /* all-zero relevance (reference: `assert sample.grad.sum()!=0`, lrp_wrapper.py:81) */
};

int lrpx_version(void);
const char* lrpx_last_error_string(void);
/* "" for a release build.  Otherwise the list of timing-experiment / profiling switches the library was compiled with
 * (-DLRPX_EXPERIMENTS with -DLRPXH_EXP=.. / -DLRPX_EPI_EXP=.. / ..., make STAMP=1): such a build computes WRONG results by
 * design (or carries profiling atomics) and must never serve a caller - tests/test_abi.py and smoke() assert "". */
const char* lrpx_build_flags(void);

/* ---- weight packing ------------------------------------------------------------------------- */
enum {
    LRPX_PACK_FWD_DUAL = 0,  /* conv fwd: channels [0,cout) = W, [cout,2cout) = clamp(W,min=0)  (PosNetConv weights, LRPtools/lrp_modules.py:66-67) */
    LRPX_PACK_BWD_POS = 1,   /* relevance pass: transposed conv with clamp(W,min=0)            (lrp_modules.py:136-138 through autograd) */
    LRPX_PACK_BWD_FIRST = 2, /* first layer: out 0..cin-1 = W+ , cin..2cin-1 = W-  (signed inputs, lrp_modules.py:78-84) */
    LRPX_PACK_BWD_PLAIN = 3, /* transposed conv with W (guided backprop, models/gridTDmodel.py:1702-1723) */
    LRPX_PACK_DENSE_T = 4,   /* dense: out = in @ W      W is (k, n)   (epsilon rule W^T contraction, gridTDmodel.py:744-765) */
    LRPX_PACK_DENSE = 5,     /* dense: out = in @ W^T    W is (n, k)   (nn.Linear forward) */
    LRPX_PACK_FWD = 6,       /* conv fwd, plain W only */
    LRPX_PACK_FWD_DUAL_FIRST = 7 /* first conv: the signed input is stored split, channels [0,cin) = x+, [cin,2cin) = x-;
                                  plain part = W on both, Z part = W+ on x+ and W- on x-  (Z = conv(x+,W+)+conv(x-,W-),
                                  lrp_modules.py:81-84) */
};
/* number of floats of the packed image: n_oc/k padded to multiples of 32 / kc */
size_t lrpx_packed_floats(int n_oc, int k, int taps, int kc);
/* w: conv (cout,cin,3,3) or dense matrix; kc = K-chunk the consuming kernel will use (lrpx_conv_kc) */
int lrpx_pack_weights(const float* w, int cout, int cin, int taps, int mode, int kc, float* packed, void* stream);
/* bf16x3 split of the same fragment layout for the fp32-accurate bf16 path (lrpx_conv_desc.bf16x6): every weight is
 * stored as three bf16 planes w = w0 + w1 + w2; modes BWD_POS / BWD_PLAIN / FWD; 3x3 kernels (taps = 9) or - round 6 - a dense
 * (cout, cin) matrix (taps = 1; BWD_PLAIN = the transposed product of the epsilon rules, as lrpx_pack_weights_f16x2) */
size_t lrpx_packed_bf16x3_bytes(int n_oc, int k, int taps);
int lrpx_pack_weights_bf16x3(const float* w, int cout, int cin, int taps, int mode, void* packed, void* stream);
/* f16x2 split for the fp16 matrix-core path (lrpx_conv_desc.f16x3): the layer's weights are scaled by a power of two
 * 2^kW (max|w| * 2^kW in [2^14, 2^15)) and stored as two fp16 planes w * 2^kW ~= hi + lo (22 significand bits); a
 * 64-byte header carries 2^-kW for the consumer's epilogue.  Same modes as lrpx_pack_weights_bf16x3. */
size_t lrpx_packed_f16x2_bytes(int n_oc, int k, int taps);
int lrpx_pack_weights_f16x2(const float* w, int cout, int cin, int taps, int mode, void* packed, void* stream);
/* weights for lrpx_conv_desc.f16x3 = 2 (fp16 hi.hi product + the two cross products on the narrow-format matrix cores;
 * the symbol keeps its round-1 name): scaled into [2^14, 2^15), stored per tap row as fp16 hi planes + block-scaled fp6
 * e2m3 fields of (w - hi)*2^11 and w with one E8M0 exponent per (output channel, tap, 16 channels) in the operand
 * layout of v_mfma_f32_32x32x64_f8f6f4 (K = 3 taps + a zero slot, x 16 channels).  3x3 kernels, modes BWD_POS / BWD_PLAIN */
size_t lrpx_packed_f16f8_bytes(int n_oc, int k);
int lrpx_pack_weights_f16f8(const float* w, int cout, int cin, int mode, void* packed, void* stream);
/* K-chunk used by lrpx_conv_mfma for a given image width / taps / input channels */
int lrpx_conv_kc(int hw, int taps, int cin);

/* ---- the contraction engine ------------------------------------------------------------------ */
enum { LRPX_EPI_FWD_DUAL = 0, LRPX_EPI_REL = 1, LRPX_EPI_FIRST = 2, LRPX_EPI_PLAIN = 3, LRPX_EPI_GUIDED = 4,
       LRPX_EPI_REL_MUL = 5 /* out1 (or out0; exactly one of them) = x * acc: the rule with the next layer's division folded into the
                               multiplicand (x = X for R, x = X / safe(Z_below) for S_next); the f16x3 epilogue */ };
enum { LRPX_STAB_NONE = 0, LRPX_STAB_SAFE = 1, LRPX_STAB_EPS = 2 };

typedef struct lrpx_conv_desc {
    const float* in;      /* [n_maps*pix_per_map][cin], or channel-chunked [cin/kc][n_maps*pix_per_map][kc] if in_chunked */
    const float* wpacked; /* from lrpx_pack_weights with kc = lrpx_conv_kc(hw,taps,cin) */
    int n_maps, hw, cin, n_oc, taps, pix_per_map;
    int epi, stab, oc_split, relu;
    int in_chunked;       /* input stored in K-chunks of lrpx_conv_kc(hw,taps,cin) channels (see lrpx_maxpool2x2_relevance) */
    int bf16x6;           /* 1: contraction on the bf16 matrix cores with exact 3-way operand splits and the 6 leading
                             partial products (fp32 accuracy, 2.67x less matrix-pipe time); wpacked must come from
                             lrpx_pack_weights_bf16x3; 3x3 convs, cin %% 16 == 0.  Epilogues: REL, FWD_DUAL (conv_bf16x6.h) and - round 6,
                             on the conv_f16x3.h tiling - REL_MUL (x = the precomputed multiplicand; with pool_am the input is the
                             relevance at the pool's output and is unpooled while staged).  No operand scales: in_amax / out1_amax unused.
                             taps = 1 (round 6): the (word, pixel) epsilon rules of the decoders (gridTDmodel.py:1125-1128, aoamodel.py:1135-1148)
                             in the same exact arithmetic - REL epilogue with x, map2img and ONE of out0 / out1 (+ zdiv), optional u;
                             cin %% 32 == 0, oc_split %% 4 == 0, 16-byte aligned operands, any number of rows (one kernel: a row's sum
                             never depends on the batch).  What the host layer uses while the conv mode is 0 / 1 */
    const float* bias;
    const float* x;
    const float* u;
    const float* zdiv;
    const int32_t* map2img;
    float* out0;
    float* out1;
    int f16x3;            /* 1: contraction on the fp16 matrix cores: operands scaled into the fp16 range (per map /
                             per layer, powers of two) and split in two halves, 3 partial products, fp32 accumulate
                             (22-bit operands: below the rounding of the fp32 accumulation itself; half the matrix
                             time of bf16x6).  wpacked from lrpx_pack_weights_f16x2; needs in_amax; REL_MUL or FWD_DUAL epilogue,
                             3x3 convs, cin %% 16 == 0.
                             2: as 1, but the two cross products hi*lo + lo*hi (2^-11 of the result) run on the fp6 matrix
                             cores (e2m3, one block exponent per 16-channel slice; rounds 1-2: fp8 e4m3) - error 2^-15 of a
                             product, random sign: the maps move by < 1e-5 of their maximum; wpacked from
                             lrpx_pack_weights_f16f8; REL_MUL, GUIDED or PLAIN epilogue */
    int out_chunk;        /* REL_MUL: > 0 writes the output channel-chunked [C/out_chunk][n_maps*pixels][out_chunk] (16 / 32) */
    const uint32_t* in_amax;  /* f16x3: [n_maps] float bits of max|in| per map (lrpx_amax_maps, or a producer's out1_amax) */
    uint32_t* out1_amax;      /* f16x3 + out1: max|out1| per map is atomicMax-ed into it (zero it first); may be null */
    uint32_t* out0_amax;      /* f16x3 + FWD_DUAL: max of out0 (the activations) per map is atomicMax-ed into it; may be null */
    const uint8_t* pool_am;   /* f16x3, or bf16x6 with REL_MUL: the conv sits under a 2x2 max-pool and `in` is the relevance at the pool's
                                 OUTPUT [n_maps][hw/2*hw/2][cin]; pool_am [n_img][hw/2*hw/2][cin] = window position
                                 (0..3, row-major) of each maximum (lrpx_pool_winner): Pool2d.propagate_relevance
                                 (lrp_modules.py:182-195) is applied while the operand is staged */
    int tile_group;           /* f16x3 / bf16x6 REL_MUL, performance hint only (results do not depend on it): maps [k*g, (k+1)*g) share an
                                 image (the g words of a caption): their tiles of the same image rows are scheduled next to
                                 each other on one XCD, so the per-image multiplicand `x` is fetched from HBM once, not g
                                 times.  0 / 1: plain order.  Needs n_maps %% g == 0 and map-aligned tiles (hw >= 56) */
    int blocked;              /* bit 0 (1): `in`, bit 1 (2): `x`, bit 2 (4): the output are in the BLOCKED layout (lrpx_nhwc_to_blocked
                                 below): [16-channel chunk][32-pixel block][4-channel part][pixel][4] - `in` / out: ONE block set over
                                 all n_maps * pixels, `x`: one block set per image.  f16x3 = 2 with REL_MUL (the relevance chain of conv
                                 mode 3) REQUIRES 7; its 224 x 224 kernel 1 (x and the output stay NHWC); every other kernel 0 */
} lrpx_conv_desc;
/* 3x3/pad-1 convolution (taps=9, square hw x hw maps) or dense GEMM (taps=1) on the fp32 MFMA with
 * the fused epilogues of the relevance rules.  Replaces F.conv2d / conv backward inside
 * LRPtools/utils.py:21-31 `lrp_backward` and torch.matmul/sum inside `lrp_linear_eps`. */
int lrpx_conv_mfma(const lrpx_conv_desc* d, void* stream);

/* ---- elementwise / layout kernels -------------------------------------------------------------- */
/* NHWC <-> BLOCKED (csrc/blocked.h): n_groups tensors of pix_per_group pixels x c channels (c %% 16 == 0), each its own block set
 * of lrpx_blocked_floats(pix_per_group, c) floats: element (pixel p, channel ch) at (ch / 16) * CS + (p / 32) * 512 + ((ch % 16) / 4)
 * * 128 + (p % 32) * 4 + ch % 4, CS = ceil(pix / 32) * 512.  An S tensor of the mode-3 chain is ONE group of n_maps * pixels; the
 * per-image multiplicands are n_img groups. */
size_t lrpx_blocked_floats(long n_pix, int c);
int lrpx_nhwc_to_blocked(const float* src, float* dst, long n_groups, int pix_per_group, int c, void* stream);
int lrpx_blocked_to_nhwc(const float* src, float* dst, long n_groups, int pix_per_group, int c, void* stream);
/* (n,c,h,w) -> [n*h*w][c_pad] with zero padding channels; and back (first c of c_src channels) */
int lrpx_nchw_to_nhwc(const float* src, float* dst, int n, int c, int hw_pix, int c_pad, void* stream);
int lrpx_nhwc_to_nchw(const float* src, float* dst, int n, int c, int hw_pix, int c_src, void* stream);
/* signed image -> [n*h*w][c_pad]: channels [0,c) = max(x,0), [c,2c) = min(x,0), rest 0 */
int lrpx_nchw_to_nhwc_posneg(const float* src, float* dst, int n, int c, int hw_pix, int c_pad, void* stream);
/* MaxPool2d(2,2) forward on NHWC (models/vgg.py:67) */
int lrpx_maxpool2x2_fwd(const float* x, float* y, int n, int h, int w, int c, void* stream);
/* Pool2d.propagate_relevance (LRPtools/lrp_modules.py:182-195) fused with the division by the
 * conv layer below:  r_in = x * [argmax] * (r_out / safe(max));  s_out = r_in / safe(zdiv).
 * x, zdiv: per IMAGE (n_img,2h,2w,c); r_out: per MAP (n_maps,h,w,c); outputs per map at 2h x 2w.
 * r_in / s_out may be null (skip).  zdiv may be null (then s_out = r_in).
 * s_chunk > 0: s_out is written channel-chunked [c/s_chunk][n_maps*2h*2w][s_chunk] (s_chunk = 8), the layout
 * lrpx_conv_mfma reads with in_chunked=1: whole 128-byte lines per chunk instead of 32-byte slices of every pixel. */
int lrpx_maxpool2x2_relevance(const float* x, const float* r_out, const float* zdiv, const int32_t* map2img,
                              float* r_in, float* s_out, int n_maps, int h_out, int w_out, int c, int s_chunk,
                              void* stream);
/* s[n,p,c] = r[n,p,c] / stab(z[img(n),p,c])   (LRPtools/utils.py:16-18 safe_divide with broadcast) */
int lrpx_divide_stab(const float* r, const float* z, const int32_t* map2img, float* s, int n_maps, long pix_c,
                     int stab, void* stream);
/* Per image, for a 2x2 max-pool with input x (n,2h,2w,c) sitting on a conv with Z+ = z (n,2h,2w,c):
 *   am[n,h,w,c]  = window position (0..3, row-major, first maximum wins as in max_pool2d's backward)
 *   xzw[n,h,w,c] = max / safe(z at the winner): the multiplicand that turns the accumulator of the conv ABOVE the pool
 *                  straight into S = R / Z+ of the conv BELOW it at the winner (Pool2d rule + safe_divide fused). */
int lrpx_pool_winner(const float* x, const float* z, float* xzw, uint8_t* am, int n, int h_out, int w_out, int c,
                     void* stream);
/* the Pool2d rule as a scatter for a conv that does not unpool while staging: s_hi (n_maps, 2*h_out, 2*w_out, c) =
 * s_lo (n_maps, h_out, w_out, c) at the winner positions `am` (per image, from lrpx_pool_winner), 0 elsewhere */
int lrpx_unpool_winner(const float* s_lo, const uint8_t* am, const int32_t* map2img, float* s_hi, int n_maps, int h_out,
                       int w_out, int c, void* stream);
/* amax[n] = float bits of max |s[n,:]| (zeroes amax first): the per-map operand scale of the f16x3 convolution */
int lrpx_amax_maps(const float* s, int n_maps, long per, uint32_t* amax, void* stream);
/* running sum over the maps of one image: out[b,t] = sum_{t'<=t} in[b,t']  (the reference's
 * `sample.grad` accumulation, LRPtools/lrp_wrapper.py:64-82); per = floats per map */
int lrpx_cumsum_maps(const float* in, float* out, int n_img, int t_per_img, long per, void* stream);
/* Captions of unequal length (the reference explains whatever length its beam search returns, models/gridTDmodel.py:935-937,
 * 1147-1153): `in` holds only the VALID (image, word) maps, image b's lens[b] maps starting at map offs[b]; `out` is the padded
 * [n_img][t_per_img] layout - the valid maps (accumulate = 1: their running sums, as lrpx_cumsum_maps) and exact zeros behind
 * an image's last word. */
int lrpx_scatter_maps(const float* in, float* out, int n_img, int t_per_img, const int32_t* lens, const int32_t* offs,
                      long per, int accumulate, void* stream);
/* dst[0 .. bytes) = 0 on `stream` (hipMemsetAsync).  The host layer clears its scratch through the library, not through the
 * framework, so that a recorded step (lrp_amd._lib.Recording: the calls of one eager run, replayed without the interpreter's
 * per-launch cost) contains EVERY device operation of the step. */
int lrpx_zero(void* dst, size_t bytes, void* stream);

/* dst[r][0..width) = src[rows[r]][0..width) for 4-byte items (per-row operands of the compacted (word, pixel) rules) */
int lrpx_gather_rows(const void* src, const int32_t* rows, void* dst, int n_rows, int width, void* stream);
/* dst += src  (the `.grad` accumulation of autograd that compute_lrp relies on, lrp_wrapper.py:80-82) */
int lrpx_accumulate(float* dst, const float* src, long n, void* stream);
/* out[row][c] = in[row][c] + in[row][half + c]: joins the x+ / x- halves of a split relevance tensor
 * (R = x+ * convT(S,W+) + x- * convT(S,W-), lrp_modules.py:56-84, for signed layer inputs) */
int lrpx_fold_halves(const float* in, float* out, long rows, int half, void* stream);
/* NaN/Inf + all-zero check of a buffer (the asserts of lrp_modules.py:154-155, lrp_wrapper.py:81);
 * synchronises the stream.  flags: bit0 = fail on non-finite, bit1 = fail on all-zero */
int lrpx_check(const float* buf, long n, int flags, void* stream);

/* THREADING.  Every entry point is thread-compatible: calls on different host threads / streams with different caller
 * buffers never share mutable state.  What the library keeps is (a) the one-time kernel-attribute initialisation of each
 * kernel instantiation (std::call_once), (b) the 4-byte flag word of lrpx_check behind a mutex, (c) the thread-local
 * error string, (d) the PROCESS DEFAULTS set by lrpx_set_conv_mode / lrpx_set_forward_f16 / lrpx_set_bf16x6 (atomics).
 * A caller that wants a mode of its own - or that runs while another thread changes the defaults - passes it per call in
 * an lrpx_vgg16_opts to the *_ex entry points; such a call never reads the defaults. */
typedef struct lrpx_vgg16_opts {
    int conv_mode;     /* 0 fp32 MFMA, 1 bf16x6, 2 f16x3, 3 fp16 + fp6 cross products; < 0: the process default */
    int forward_f16;   /* forward trace on the fp16 split-product kernels: 0 / 1; < 0: the process default */
    float* layer_ms;   /* lrpx_vgg16_relevance_ex only, may be NULL.  HOST array of 17 floats: the call records HIP events of
                          its own around every conv launch, WAITS for them and stores the milliseconds per VGG16 layer index
                          (0 for pools / the first layer); the events are created and destroyed inside the call */
} lrpx_vgg16_opts;
/* the values a call with `opts` (NULL: none) would run with - host logic only, no device access */
int lrpx_vgg16_resolve_opts(const lrpx_vgg16_opts* opts, int* conv_mode, int* forward_f16);

/* Process-wide DEFAULT for lrpx_vgg16_relevance: 1 (default) runs the relevance passes of the 56/28/14-pixel layers on
 * the bf16 matrix cores with exact operand splits (fp32 accuracy, see lrpx_conv_desc.bf16x6), 0 keeps the fp32 MFMA
 * everywhere; a negative value only queries.  Returns the previous setting. */
int lrpx_set_bf16x6(int enable);
/* Matrix-core mode of the fused VGG16 chains: 0 fp32 MFMA, 1 bf16x6 (exact 3-way bf16 splits of both operands, six products, fp32
 * accumulate: 24 significand bits and fp32's exponent range per operand - THE DEFAULT since round 6, arithmetic no narrower than the
 * reference's fp32 convolutions, LRPtools/lrp_modules.py:124-150), 2 f16x3 (fp16 split products behind per-map power-of-two
 * scales: 22 operand bits, fp16's exponent range below the map maximum), 3 = 2 with the cross products of the relevance pass on the fp6
 * matrix cores (lrpx_conv_desc.f16x3 = 2).  Modes 2 / 3 are opt-in speed modes (1.6x / 2.0 - 2.2x the maps/s of mode 1, inside the 1e-4
 * contract on every tested input of natural range).  RANGE CONTRACT of modes 2 / 3: the relevance operand S = R / Z+ of a layer is scaled by
 * ONE power of two per map into the fp16 range; entries more than ~2^29 below the map's maximum flush to zero and entries below ~2^-15 of it
 * lose bits, so a map whose layer activations differ by more than ~2^16 BETWEEN REGIONS of one image (S = R / Z+ then spans that range the
 * other way) is rounded against its largest region: tests/test_gpu_range.py pins 1e-2-grade maps on a 2^33 range, where modes 0 / 1 keep 1e-4.  LRPX_CONV_MODE in the environment sets the initial default (read once at load).
 * Negative: query only.  Returns the previous mode. */
int lrpx_set_conv_mode(int mode);
/* 1 (conv modes 2 / 3 only): the forward trace of conv1_1..conv5_3 also runs on the fp16 split-product kernels (operand scale =
 * per-image maximum of the layer input).  RANGE CONTRACT of that switch: inputs more than ~2^29 below their image's maximum
 * flush to zero; a receptive field made only of such inputs gets Z+ = 0 and the relevance arriving there is dropped, where the
 * reference (LRPtools/utils.py:16-18 stabilises exact zeros only) redistributes it (tests/test_gpu_range.py).
 * 0 (THE DEFAULT since round 6, every conv mode): exact kernels - fp32 MFMA for conv1_1, exact bf16 splits (fp32's exponent
 * range) above; conv mode 0: fp32 MFMA throughout.  LRPX_FORWARD_F16 in the environment sets the initial default.
 * Negative: query.  Returns the previous value. */
int lrpx_set_forward_f16(int enable);

/* ---- device-side consumers of the relevance maps (evaluation.py) -------------------------- */
/* (n,c,hw) -> (n,hw): mode 0 mean over channels (evaluation.py:134), 1 mean of max(x,0) (:410-412), 2 mean of max(-x,0) (:406-408) */
int lrpx_spatial_reduce(const float* maps, int n, int c, long hw, int mode, float* out, void* stream);
/* `_project_maxabs` (evaluation.py:338-343): every map divided by its max |x| in place (all-zero maps stay zero) */
int lrpx_project_maxabs(float* x, int n, long per, void* stream);
/* `block_image` (evaluation.py:57-80): sums over patch x patch squares, mask = 0 on the k squares with the largest sum
 * (ties: lower index first), 1 elsewhere; spatial, mask: (n,h,w) */
int lrpx_patch_mask(const float* spatial, int n, int h, int w, int patch, int k, float* mask, void* stream);
/* `_calculate_overlaped_pixels` (evaluation.py:313-336) for every threshold: boxes [n][4] = x0,y0,x1,y1 (pixels),
 * out[n][nthr] = sum of the relevance > thr inside the box / sum of the relevance > thr (0 if that is 0, capped at 1) */
int lrpx_bbox_ratio(const float* spatial, int n, int h, int w, const int32_t* boxes, const float* thresholds, int nthr,
                    float* out, void* stream);
/* tpfp statistics (evaluation.py:506-513): out[n][4] = mean, mean |x|, mean of the positive entries (0 if none), max */
int lrpx_map_stats(const float* spatial, int n, long per, float* out4, void* stream);

/* np.quantile(map, q) ('linear' method) of every (H*W) map: the 100-point quantiles of the tpfp statistics
 * (evaluation.py:451, :510, :543).  spatial (n, per) floats, q (nq) doubles in [0,1], out (n, nq).
 * `workspace`: lrpx_map_quantiles_workspace(n, per) bytes of device memory (sorted copy + sort scratch; 0 = the
 * shape is unsupported, n*per must stay below 2^31). */
size_t lrpx_map_quantiles_workspace(int n, long per);
int lrpx_map_quantiles(const float* spatial, int n, long per, const double* q, int nq, float* out, void* workspace,
                       size_t workspace_bytes, void* stream);
/* `LRPutil.heatmap(LRPutil.gamma(hm))` (LRPtools/utils.py:67-145) as the explainers' visualize_explanations call it
 * (models/gridTDmodel.py:1196-1198): gamma correction with the map's own max |x|, sum over the channels, projection to
 * [0, 255] with the summed map's max |x|, colour-map lookup.  maps (n,c,hw) -> out (n,hw,3); lut [nlut][3] (the
 * colour map sampled at its nlut = 256 entries); tmp: n*hw floats of scratch. */
int lrpx_heatmap(const float* maps, int n, int c, long hw, float gamma, const float* lut, int nlut, float* tmp, float* out,
                 void* stream);

/* ---- VGG16 encoder: trace + relevance chain ------------------------------------------------------ */
/* bytes of the packed-weight blob / per-batch trace / relevance workspace */
size_t lrpx_vgg16_packed_bytes(void);
size_t lrpx_vgg16_trace_bytes(int n_img);
size_t lrpx_vgg16_workspace_bytes(int n_maps);
/* w[13], b[13]: device pointers to the conv weights (cout,cin,3,3) / biases in layer order */
int lrpx_vgg16_pack(const float* const* w, const float* const* b, void* packed, void* stream);
/* Per-layer timing of lrpx_vgg16_relevance for profiling, per calling THREAD: enable = 1 makes this thread's following
 * plain lrpx_vgg16_relevance calls behave like lrpx_vgg16_relevance_ex with opts.layer_ms (events of their own, the
 * call waits for them); ms17 (may be null) receives the milliseconds per VGG16 layer index of this thread's last such
 * call.  enable < 0: query only.  (Kept for callers of the round-1 ABI; new code passes opts.layer_ms.) */
int lrpx_vgg16_layer_timing(int enable, float* ms17);
/* recompute the trace tensors derived from the saved activations (x / safe(Z+_below), the multiplicand of the fused
 * conv->conv relevance step); lrpx_vgg16_forward calls it, callers that overwrite activations in the trace must too */
int lrpx_vgg16_trace_derive(void* trace, int n_img, void* stream);
/* Encoder.forward (models/gridTDmodel.py:40-43) + the per-layer inputs `save_input_hook` keeps
 * (LRPtools/lrp_wrapper.py:24-25) + Z+ of every conv.  img: (n_img,3,224,224) NCHW.
 * feat_nhwc: (n_img,196,512) encoder output (may be null: it also lives in the trace). */
int lrpx_vgg16_forward(const void* packed, const float* img_nchw, int n_img, void* trace, float* feat_nhwc,
                       void* stream);
/* compute_lrp (LRPtools/lrp_wrapper.py:63-87) for N maps sharing n_img traces: r_feat (N,196,512) NHWC
 * relevance at the encoder output -> out (N,3,224,224) NCHW.  map2img[N] int32 (null: identity, then
 * n_maps == n_img).  `trace` must come from lrpx_vgg16_forward with the same n_img. */
int lrpx_vgg16_relevance(const void* packed, const void* trace, int n_img, const float* r_feat_nhwc,
                         const int32_t* map2img, int n_maps, void* workspace, float* out_nchw, void* stream);
/* offsets (in floats) of the per-layer tensors inside a trace made for n_img images: act_off[18]
 * (act[l] = NHWC input of leaf l in forward order conv,conv,pool,...; act[0] is the image stored split
 * [x+|x-|0 0] with 8 channels; act[17] = encoder output) and zpos_off[17] (Z+ of conv l, 0 for pools).
 * This is what `module.input` is to the reference's hooks (lrp_wrapper.py:24-25). */
int lrpx_vgg16_trace_layout(int n_img, size_t* act_off, size_t* zpos_off);
/* pointer to the encoder output features (n_img,196,512) inside a trace */
const float* lrpx_vgg16_trace_features(const void* trace, int n_img);
/* Channel balance of the Z+ / relevance side: conv layer `layer` (0..16, a conv of cfg 'D') keeps its Z+ in the trace as
 * Z'_c = rs[c] * conv(X, W+)_c and every alpha1beta0 relevance pack carries rs[c] * W+[c,:], rs[c] = 2^(e_max - e_c), e_c = floor(log2
 * max W+[c,:]) (exact powers of two: S'_c (rs_c W+[c,i]) is the reference's product S_c W+[c,i] of LRPtools/lrp_modules.py:124-150 bit
 * for bit).  It keeps the operands of the split-product modes inside fp16's range when a channel's weights are small against its bias
 * (trained weights, models/vgg.py:86-94).  Returns the DEVICE pointer of rs inside `packed` (n_channels floats), null for a pool. */
const float* lrpx_vgg16_channel_scales(const void* packed, int layer, int* n_channels);
/* DEVICE pointer of 17 floats inside `packed` (index = layer; pools: unset): the largest ratio of row maxima max|W[c,:,:,:]| inside one
 * 16-row slice of a conv layer's weights.  The image-gradient chains (guided backprop / plain gradient) multiply with W itself; in conv
 * mode 3 the cross terms of their operands are fp6 fields that share ONE scale per 16-channel slice, so a slice whose rows differ by
 * 2^6 and more loses the cross terms of its small rows (measured on log-normal channel scales: 1.1e-4 of max|gradient|, fp16 split
 * products 4e-6).  A caller that wants the 1e-4 grade on any weights runs those chains in conv mode 2 when a layer exceeds ~64
 * (lrp_amd.ops.Vgg16 does).  The relevance chain is not affected: its rows are balanced (lrpx_vgg16_channel_scales). */
const float* lrpx_vgg16_row_spread(const void* packed);


/* ---- small dense / utility kernels of the decoders ----------------------------------------------- */
/* out[b][n] = act(sum_k x[b][k] w[n][k] + bias[n]); w in nn.Linear layout (N,K); a few rows b (weight-read bound).
 * act: 0 none, 1 relu.  ldx/ldo = row strides in floats.  (nn.Linear / LSTMCell matmuls of
 * models/gridTDmodel.py:773-797, :980-990 at batch size B) */
int lrpx_linear_small(const float* x, long ldx, const float* w, const float* bias, float* out, long ldo, int B, int K,
                      int N, int act, void* stream);
/* avg[b][c] = mean_p f[b][p][c]   (nn.AdaptiveAvgPool2d(1), models/gridTDmodel.py:38,42) */
int lrpx_mean_pixels(const float* f, float* avg, int B, int P, int C, void* stream);
int lrpx_relu(const float* x, float* y, long n, void* stream);
/* first maximum per row -> int64 (torch.argmax / topk(1), models/gridTDmodel.py:499, :849) */
int lrpx_argmax_rows(const float* x, long ld, int rows, int n, long long* out, void* stream);
/* greedy `sample_next_word` (models/gridTDmodel.py:522-526): argmax of log_softmax(x[r]) and its value */
int lrpx_argmax_logprob_rows(const float* x, long ld, int rows, int n, long long* out, float* logprob, void* stream);
/* logit[b*T+t] = fc.weight[tok[b][t+1]] . hc[b][t] + fc.bias[...]  (the one entry of `predictions`
 * that explain_caption_wordt reads, models/gridTDmodel.py:1027,1034) */
int lrpx_target_logit(const float* hc, const float* fcw, const float* fcb, const long long* tok, int tok_ld,
                      float* logit, int B, int T, int H, void* stream);

/* ---- gridTD decoder: trace (get_hidden_parameters, models/gridTDmodel.py:933-1012) ------------------ */
typedef struct lrpx_gridtd_trace {
    int B, T, H, E, P;
    float *xh1, *xh2;                   /* [B][T][2E+2H] = x1t ++ h1t ; [B][T][3H] = x2t ++ h2t  (:1025-1026) */
    float *h1, *c1, *h2, *c2;           /* [B][T+1][H] */
    float *g1, *i1, *f1, *g2, *i2, *f2; /* [B][T][H]  g = pre-activation, i/f = activations (:1002-1009) */
    float *s, *ctx, *ctx_hat, *hc;      /* [B][T][H]  hc = h2[t+1] + ctx_hat[t] (fc input) */
    float *alpha, *beta;                /* [B][T][P], [B][T] */
    float *o1, *o2, *sgate;             /* [B][T][H] output gates + sentinel gate: only the gradient explainers'
                                           trace keeps them (models/gridTDmodel.py:1323-1422); may be null */
} lrpx_gridtd_trace;
/* per step t: build xh1 ; LSTM point-wise (which = 1 AdaLSTM incl. sentinel, 2 LanguageLSTM) ; adaptive attention */
int lrpx_gridtd_fwd_pre(const lrpx_gridtd_trace* tr, int t, const float* glob, const float* emb, const long long* tok,
                        int tok_ld, void* stream);
int lrpx_gridtd_fwd_lstm(const lrpx_gridtd_trace* tr, int t, const float* zz, int ldz, int which, void* stream);
/* LRP-inference decoding, `GridTDModel.sample_lrp` / `forwardlrp_context` (models/gridTDmodel.py:631-702, :579-630).
 * Their forward differs from the model's own in one place: the sentinel gate sees the NEW h1 (:672, :610).
 *   fwd_gate_input: xg[b] = [h2_old | glob | emb | h1_new]  ((B, 2E+2H) floats), the input of x_gate|h_gate;
 *   fwd_sentinel:   s[b,t] = sigmoid(zg[b]) * tanh(c1_new)  with zg = [x_gate|h_gate] xg + biases, (B, ldz);
 *   lrp_reweight:   `get_lrp_weight_step` (:548-577) for step t: k = argmax pred[b]; unless skip[k] (stop words and
 *                   special tokens), R = pred[b][k] goes back through fc (epsilon rule) to h2 + ctx_hat and is split
 *                   between the two; both vectors are normalised to weights x / max|x| + 1 (LRPtools/utils.py:55-64);
 *                   hcw[b] = ctx_hat * w_ctx + w_h2 * h2, the re-weighted fc input (:687).  skip: V bytes. */
int lrpx_gridtd_fwd_gate_input(const lrpx_gridtd_trace* tr, int t, float* xg, void* stream);
int lrpx_gridtd_fwd_sentinel(const lrpx_gridtd_trace* tr, int t, const float* zg, int ldz, void* stream);
int lrpx_gridtd_lrp_reweight(const lrpx_gridtd_trace* tr, int t, const float* pred, long ld, int V, const float* fc_w,
                             const unsigned char* skip, float* hcw, void* stream);
/* the same rule on plain rows: h / ctx = the two fc summands of row r (row strides ldh / ldc, H = 512).  log_softmax = 1
 * is `AOAModel.get_lrp_weight_step` as `AOAModel.sample_lrp` calls it (models/aoamodel.py:597-626, :721-723): the rule
 * sees log_softmax(pred), i.e. the relevance sent back through fc is log p(k). */
int lrpx_lrp_reweight_rows(const float* pred, long ld, int V, const float* h, long ldh, const float* ctx, long ldc,
                           const float* fc_w, const unsigned char* skip, float* hcw, int rows, int H, int log_softmax,
                           void* stream);
/* one beam-search step (GridTDModel.beam_search, models/gridTDmodel.py:437-444; AOAModel.beam_search): the k <= 4 best of
 * cum[r] + log_softmax(x[r,:n])[w] over the n_rows <= 8 live beams; out_idx[j] = r * n + w (int64), out_val[j] the score,
 * best first (ties: lower flat index).  cum may be NULL (zeros). */
int lrpx_beam_topk(const float* x, long ld, int n_rows, int n, const float* cum, int k, long long* out_idx, float* out_val,
                   void* stream);
/* scratch: [B][3*P] floats (scores, W_g h, W_s s) */
int lrpx_gridtd_fwd_attention(const lrpx_gridtd_trace* tr, int t, const float* Vp, const float* att_img,
                              const float* Wg, const float* Ws, const float* bs, const float* wh, float* scratch,
                              void* stream);

/* ---- gridTD decoder: relevance (explain_caption_wordt, models/gridTDmodel.py:1014-1135) -------------- */
typedef struct lrpx_gridtd_relstate {
    const int32_t* lens;                /* [B] words per image (null: T) */
    float *r_h2n, *r_c2, *r_c1, *r_ch0, *r_h2p; /* [B*T][H] */
    float* r_glob;                      /* [B*T][E] */
    float *A, *rx;                      /* dense-rule input [B*T][H] / output [B*T][2E+2H] */
    float* wacc;                        /* [B*T][T][H] */
    float* r_words;                     /* [B*T][T] */
} lrpx_gridtd_relstate;
/* decoder steps t0 <= t < t1 of get_hidden_parameters (models/gridTDmodel.py:952-1012) in ONE call - fwd_pre, the AdaLSTM gate linear,
 * its cell + sentinel, the adaptive attention, the LanguageLSTM linear, its cell - the `for t in range(...)` loop with its host side in
 * native code (the same launches as the per-step entry points: bit-identical) */
typedef struct lrpx_gridtd_step_args {
    const float *glob, *emb;            /* relu(global_img_feature_proj(avg)) [B][E], embedding table */
    const long long* tok; int tok_ld;   /* token ids [B][tok_ld] */
    const float *w_cat1, *b_cat1;       /* [AdaLSTM W_ih | W_hh ; x_gate | h_gate] (5H x (2E+2H)) and bias */
    const float *w_cat2, *b_cat2;       /* LanguageLSTM [W_ih | W_hh] (4H x 3H) and its bias (the explainers' quirk or the model's) */
    const float *Vp, *att_img;          /* relu(img_projector(features)) [B][P][H]; W_v_proj(Vp) + b [B][P][P] */
    const float *Wg, *Ws, *bs, *wh;     /* AdaAttention W_g_proj, W_s_proj (+ bias), w_h */
    float *zz1, *zz2, *att_scratch;     /* scratch [B][5H], [B][4H], [B][3P] */
    /* optional (all four or none): the first 4H rows of w_cat1 / w_cat2 and their biases with the gate rows INTERLEAVED - row 16 j + 4 gate + u =
     * row gate * H + 4 j + u - for the fused step (gate linear + LSTM cell in one launch, the next input row behind the second: 4 launches per
     * time step instead of 7, bit-identical traces; <= 64 images) */
    const float *w_il1, *b_il1, *w_il2, *b_il2;
} lrpx_gridtd_step_args;
int lrpx_gridtd_fwd_steps(const lrpx_gridtd_trace* tr, int t0, int t1, const lrpx_gridtd_step_args* a, void* stream);

int lrpx_gridtd_rel_init(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, const float* fcw,
                         const float* logit, const long long* tok, int tok_ld, void* stream);
/* lock-step s, phase 0: LanguageLSTM cell split (:1061-1069) -> A ; 1: after its dense rule (:1074-1105) -> A ;
 * 2: after the AdaLSTM dense rule (:1110-1115) ; 3: phase 2 of lock-step s and phase 0 of lock-step s + 1 in one launch */
int lrpx_gridtd_rel_step(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, int s, int phase, void* stream);
/* lock-steps 0 <= s < n_steps of explain_caption_wordt's `for i in range(t+1)[::-1]` (models/gridTDmodel.py:1060-1113) in ONE call: phase 0,
 * the LanguageLSTM dense rule `dense2`, phase 1, the AdaLSTM dense rule `dense1`, phase 2 (lrpx_conv_mfma descriptors whose map2img is
 * replaced by idx + s * idx_ld: row -> source row of the multiplicand at lock-step s) */
int lrpx_gridtd_rel_steps(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, int n_steps, const lrpx_conv_desc* dense2,
                          const lrpx_conv_desc* dense1, const int32_t* idx, int idx_ld, void* stream);
int lrpx_gridtd_rel_glob(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, const float* glob_pre,
                         float* a_glob, void* stream);
int lrpx_rel_avg_u(const float* r_avg, const float* avg, float* u, int rows, int T, int C, int P, void* stream);
int lrpx_gridtd_rel_pix(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, const float* Vp,
                        const float* proj_pre, float* a_proj, void* stream);
/* the same for the n_rows (image, word) rows listed in `rows` (row = image * T + word) only, written COMPACTLY: a_proj
 * [n_rows][P][H] - padded words of captions shorter than T then cost nothing in the (word, pixel) rules and in the VGG16
 * chain.  rows = null: all B*T rows (n_rows must be B*T). */
int lrpx_gridtd_rel_pix_rows(const lrpx_gridtd_trace* tr, const lrpx_gridtd_relstate* rs, const float* Vp,
                             const float* proj_pre, float* a_proj, const int32_t* rows, int n_rows, void* stream);
int lrpx_rel_words_norm(float* r_words, int rows, int T, void* stream);

/* ---- AoA decoder: trace (get_hidden_parameters, models/aoamodel.py:990-1062) -------------------------- */
typedef struct lrpx_aoa_trace {
    int B, T, H, E, P, NH;
    float* xh;                     /* [B][T][E+2H] = xt ++ ht[:T]  (:1075) */
    float *h, *c;                  /* [B][T+1][H] */
    float *g, *i, *f;              /* [B][T][H] */
    float *ctx, *lin, *c_aoa, *hc; /* [B][T][H]: context, decoder_aoa_linear(context), gated, fc input */
    float* alpha;                  /* [B][T][NH][P] */
    float *o, *sg;                 /* optional (gradient explainers, :1309-1376): [B][T][H] output gate, sigmoid(aoa gate) */
} lrpx_aoa_trace;
int lrpx_aoa_fwd_pre(const lrpx_aoa_trace* tr, int t, const float* glob, const float* emb, const long long* tok,
                     int tok_ld, void* stream);
int lrpx_aoa_fwd_lstm(const lrpx_aoa_trace* tr, int t, const float* zz, int ldz, void* stream);
/* qg: [B][2H] = [q_proj(h_t) | decoder_aoa_linear_gate(h_t)]; key/value: [B][P][H]  (MultiHeadedDotAttention, :77-108) */
int lrpx_aoa_fwd_attention(const lrpx_aoa_trace* tr, int t, const float* qg, int ldq, const float* key,
                           const float* value, void* stream);
int lrpx_aoa_fwd_post(const lrpx_aoa_trace* tr, int t, const float* qg, int ldq, const float* lin, void* stream);
/* decoder steps t0 <= t < t1 of get_hidden_parameters (models/aoamodel.py:1020-1052) in ONE call: per step fwd_pre, LSTM gate
 * linear, fwd_lstm, [q | aoa gate] linear, fwd_attention, decoder_aoa_linear, fwd_post - the `for t in range(...)` loop of
 * the reference (:1019) with its host side in native code */
typedef struct lrpx_aoa_step_args {
    const float *glob, *emb;            /* global image feature [B][H], embedding table */
    const long long* tok; int tok_ld;   /* token ids [B][tok_ld] (teacher forcing: the caption incl. <start>) */
    const float *w_cat, *b_cat;         /* LSTM [W_ih | W_hh] (4H x (E+2H)) and its bias (the explainers' quirk or the model's) */
    const float *w_cat_il, *b_cat_il;   /* optional: the same with the gate rows interleaved (row 16 j + 4 gate + u = row gate * H + 4 j + u):
                                           the steps then run fused, 4 launches instead of 7 (gate linear + LSTM cell; aoa_linear + gated
                                           sum + the next step's input row); results are bit-identical */
    const float *w_qg, *b_qg;           /* [q_proj ; decoder_aoa_linear_gate] (2H x H) */
    const float *w_lin, *b_lin;         /* decoder_aoa_linear (H x H) */
    const float *key, *value;           /* [B][P][H] */
    float *zz, *qg, *lin;               /* scratch [B][4H], [B][2H], [B][H] */
} lrpx_aoa_step_args;
int lrpx_aoa_fwd_steps(const lrpx_aoa_trace* tr, int t0, int t1, const lrpx_aoa_step_args* a, void* stream);
/* The teacher-forced trace with the recurrence decoupled from what hangs off it: in this model the LSTM reads x_t = [emb(word_t) |
 * global feature] and its own h_{t-1} only (models/aoamodel.py:1030-1033); attention, AoA gate and scores read h_t and feed nothing
 * back.  So: (1) lrpx_aoa_fwd_inputs gathers x_t of every (image, word) row (into xh and, contiguous, into xin [B*T][E+H]); the caller
 * forms zin = xin W_ih^T + b for all rows in one GEMM (columns in the interleaved gate order of w_cat_il above); (2)
 * lrpx_aoa_fwd_recurrence runs the T dependent steps  z = zin[t] + W_hh h_{t-1} -> LSTM cell, ONE launch of K = H each (w_hh_il:
 * (4H, H), rows interleaved); (3) lrpx_aoa_fwd_gather_h writes hn [B*T][H] = h_t rows and the h part of xh; the q / gate linear,
 * lrpx_aoa_fwd_attention_all, decoder_aoa_linear and lrpx_aoa_fwd_post_all then run ONCE over all B*T rows (qg [B*T][ldq] =
 * [q_proj(h_t) | aoa_linear_gate(h_t)], lin [B*T][H]).  Same trace tensors as the stepwise entry points (values equal to rounding:
 * z is (x W_ih^T + b) + W_hh h instead of one dot product over [x | h]). */
int lrpx_aoa_fwd_inputs(const lrpx_aoa_trace* tr, const float* glob, const float* emb, const long long* tok, int tok_ld,
                        float* xin, void* stream);
int lrpx_aoa_fwd_recurrence(const lrpx_aoa_trace* tr, const float* w_hh_il, const float* zin, void* stream);
/* the same with the input part from tables instead of a GEMM per trace: the embedding part of x_t W_ih^T depends on the token alone -
 * tab [V][4H] = embedding W_ie^T (interleaved gate order; once per MODEL), gimg [B][4H] = glob W_ig^T + bias (one small linear per
 * trace): z = (W_hh h + tab[tok[b, t]]) + gimg[b].  xin of lrpx_aoa_fwd_inputs may then be null. */
int lrpx_aoa_fwd_recurrence_tab(const lrpx_aoa_trace* tr, const float* w_hh_il, const float* tab, const float* gimg,
                                const long long* tok, int tok_ld, void* stream);
/* hn_amax / ctx_amax / hc_amax (optional, [B*T] words): float bits of max|row| of hn / ctx / hc, recorded by the kernel that writes
 * the row - the operand scales of the GEMMs that read it (ctx_amax must come zeroed: its 8 heads meet in an atomic maximum) */
int lrpx_aoa_fwd_gather_h(const lrpx_aoa_trace* tr, float* hn, uint32_t* hn_amax, void* stream);
int lrpx_aoa_fwd_attention_all(const lrpx_aoa_trace* tr, const float* qg, int ldq, const float* key, const float* value,
                               uint32_t* ctx_amax, void* stream);
int lrpx_aoa_fwd_post_all(const lrpx_aoa_trace* tr, const float* qg, int ldq, const float* lin, uint32_t* hc_amax, void* stream);

/* ---- AoA decoder: gradient explainers (ExplainAOAGradient.explain_caption_wordt, models/aoamodel.py:1435-1499;
 *      inherited unchanged by the guided / Grad-CAM variants) --------------------------------------------------- */
typedef struct lrpx_aoa_gradstate {
    const int32_t* lens;
    float *d_h, *d_c;              /* [B*T][H] */
    float *dA, *dB;                /* [B*T][H]: gradient into decoder_aoa_linear / decoder_aoa_linear_gate outputs (:1468-1470) */
    float *gates, *dx;             /* [B*T][4H] LSTM gate gradients / [B*T][E+2H] = gates @ [W_ih | W_hh] */
    float *d_glob;                 /* [B*T][H]   (assignment quirk :1487: the i = 0 step survives) */
    float* r_words;                /* [B*T][T] */
} lrpx_aoa_gradstate;
/* seed: d_h = fc.weight[target], dA, dB from the saved gate / linear outputs; needs tr->o and tr->sg */
int lrpx_aoa_grad_init(const lrpx_aoa_trace* tr, const lrpx_aoa_gradstate* gs, const float* fcw, const long long* tok,
                       int tok_ld, void* stream);
/* lock-step s (time index i = t - s): phase 0 LSTM cell backward -> gates; phase 1 after dx = gates @ [W_ih|W_hh] */
int lrpx_aoa_grad_step(const lrpx_aoa_trace* tr, const lrpx_aoa_gradstate* gs, int s, int phase, void* stream);
/* d_feat[row][p][:] = alpha[b,t,head,p] * v1[row][:] + v2[row][:]  (gradient_mha :1415-1433 folded through v_proj and
 * the projector: both are rank-1 in the pixel index) */
int lrpx_aoa_grad_pix(const lrpx_aoa_trace* tr, int head, const float* v1, const float* v2, float* d_feat, int C,
                      void* stream);
int lrpx_aoa_grad_pix_rows(const lrpx_aoa_trace* tr, int head, const float* v1, const float* v2, float* d_feat, int C,
                           const int32_t* rows, int n_rows, void* stream);   /* compact output for the listed rows */
/* x[row][c] = 0 for c outside [lo, hi)  (only one head passes gradient, :1428) */
int lrpx_keep_cols(float* x, long rows, int ncol, int lo, int hi, void* stream);

/* ---- AoA decoder: relevance (explain_caption_wordt :1064-1156, lrp_mha :812-862) ----------------------- */
typedef struct lrpx_aoa_relstate {
    const int32_t* lens;
    float *r_hn, *r_glob;          /* [B*T][H] */
    float *A, *rx;                 /* dense-rule input [B*T][H] / output [B*T][E+2H] */
    float* r_words;                /* [B*T][T] */
} lrpx_aoa_relstate;
int lrpx_aoa_rel_init(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, const float* fcw, const float* logit,
                      const long long* tok, int tok_ld, void* stream);
int lrpx_aoa_rel_value(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, const float* r_ctx, const float* value,
                       int head, float* a_val, void* stream);
int lrpx_aoa_rel_value_rows(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, const float* r_ctx, const float* value,
                            int head, float* a_val, const int32_t* rows, int n_rows, void* stream);   /* compact a_val */
/* the same, only the head's H / NH non-zero columns: a_val_head [n_rows][P][H/NH] (rows = null: all B*T rows).  `lrp_mha` passes
 * relevance through ONE head (models/aoamodel.py:848-860), so the v_proj rule behind it contracts over that head's H/NH rows of W_v
 * (lrpx_pack_weights_f16x2 of the row slice) instead of over H with zeros in 7/8 of the operand: an eighth of the matrix work and of
 * the operand traffic, the same products in the same order (bit-identical). */
int lrpx_aoa_rel_value_head(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, const float* r_ctx, const float* value,
                            int head, float* a_val_head, const int32_t* rows, int n_rows, void* stream);
/* lock-step s: phase 0 g-gate split (:1116-1120) -> A ; phase 1 after the LSTM dense rule (:1129-1133) */
int lrpx_aoa_rel_step(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, int s, int phase, void* stream);
/* lock-steps 0 <= s < n_steps of explain_caption_wordt's `for i in range(t+1)[::-1]` (models/aoamodel.py:1114-1134) in ONE
 * call: phase 0, the LSTM dense rule `dense` (an lrpx_conv_mfma descriptor whose map2img is replaced by idx + s * idx_ld:
 * row -> source row of its multiplicand), phase 1 */
int lrpx_aoa_rel_steps(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, int n_steps, const lrpx_conv_desc* dense,
                       const int32_t* idx, int idx_ld, void* stream);
/* All T lock-steps with ONE launch each: the step's point-wise code (:1116-1120 of the next step, :1129-1133 of this one) runs in the
 * epilogue of the gate rule's GEMM, r_xh is never stored.  `dense` as above (in = rs->A, no addend; E = H = 512) and either on the fp16
 * split products (f16x3 = 1: csrc/dense_f16x3.hip, FUSE) or - round 6, the exact arithmetic of the default mode - on the fp32 MFMA with K split
 * over four waves (f16x3 = 0, wpacked from lrpx_pack_weights(DENSE_T, kc = 32): csrc/dense_small.hip, dense_ks_kernel<REL, FUSE>); a_alt: a second [B*T][H] buffer (the steps ping-pong between rs->A and it);
 * wpart: [B*T][T][4] scratch for the partial sums of r_words; coef: [2 * B*T*H + B*T] floats of scratch (per-trace coefficient tables of
 * :1116-1120 and the rows' last active step).  rs->r_words comes out NORMALISED (lrpx_rel_words_norm included); rs->rx
 * and rs->r_hn are not used.  Values equal to the two-launch steps up to the summation order of r_words. */
int lrpx_aoa_rel_steps_fused(const lrpx_aoa_trace* tr, const lrpx_aoa_relstate* rs, const lrpx_conv_desc* dense, const int32_t* idx,
                             int idx_ld, float* a_alt, float* wpart, float* coef, void* stream);

/* ---- gridTD guided backprop, decoder side (ExplainiGridTDGuidedGradient.explain_caption_wordt,
 *      models/gridTDmodel.py:1588-1675): BPTT with alpha/beta constant ------------------------------------ */
typedef struct lrpx_gridtd_gradstate {
    const int32_t* lens;
    float *d_h2n, *d_c2, *d_c1, *d_ch0, *d_h2p; /* [B*T][H] */
    float* d_glob;                      /* [B*T][E] */
    float *gates, *dx;                  /* gate-gradient rows [B*T][4H] (GEMM input) / GEMM output [B*T][3H] */
    float* wacc;                        /* [B*T][T][H]  d_context per time step */
    float* r_words;                     /* [B*T][T] */
} lrpx_gridtd_gradstate;
int lrpx_gridtd_grad_init(const lrpx_gridtd_trace* tr, const lrpx_gridtd_gradstate* gs, const float* fcw,
                          const long long* tok, int tok_ld, void* stream);
/* phase 0: LanguageLSTM cell backward -> gates ; 1: after gates @ [W_ih|W_hh] -> AdaLSTM cell backward -> gates ;
 * 2: after gates @ W_ih of the AdaLSTM */
int lrpx_gridtd_grad_step(const lrpx_gridtd_trace* tr, const lrpx_gridtd_gradstate* gs, int s, int phase, void* stream);
/* a_proj[row][k][:] = sum_{i<=t} alpha[b][i][k] * wacc[row][i][:]   (:1642-1643) */
int lrpx_spread_pixels(const float* wacc, const float* alpha, const int32_t* lens, float* a_proj, int B, int T, int H,
                       int P, void* stream);
int lrpx_spread_pixels_rows(const float* wacc, const float* alpha, const int32_t* lens, float* a_proj, int B, int T, int H,
                            int P, const int32_t* rows, int n_rows, void* stream);   /* compact a_proj for the listed rows */
int lrpx_scale(const float* x, float* y, long n, float alpha, void* stream);
int lrpx_positive_mask(const float* x, float* y, long n, void* stream);   /* y = [x > 0]  (:1674) */

/* ---- VGG16 guided backprop (explain_cnn, models/gridTDmodel.py:1677-1723) ------------------------------ */
/* d_feat (N,196,512) NHWC gradient at the encoder output -> (N,3,224,224) NCHW image gradient with the guided
 * ReLU rule clamp(g,min=0)*[relu_out>0] at every ReLU.  Same trace / workspace as lrpx_vgg16_relevance. */
int lrpx_vgg16_guided_backprop(const void* packed, const void* trace, int n_img, const float* d_feat_nhwc,
                               const int32_t* map2img, int n_maps, void* workspace, float* out_nchw, void* stream);
/* ExplainGridTDGradient.explain_cnn (models/gridTDmodel.py:1507-1521): the plain autograd gradient of the encoder
 * output w.r.t. the image, `image_feature.backward(d_img_feature)`; same arguments as the guided variant. */
int lrpx_vgg16_gradient(const void* packed, const void* trace, int n_img, const float* d_feat_nhwc,
                        const int32_t* map2img, int n_maps, void* workspace, float* out_nchw, void* stream);
/* ExplainGridTDGradCam.grad_cam (models/gridTDmodel.py:1760-1771): feats (n_img,P,C) NHWC, grads (rows,P,C) ->
 * cam (rows,P) = relu(sum_c feats_c * mean_p grads_c) / (max + 1e-6); P <= 256. */
int lrpx_gradcam(const float* feats, const float* grads, const int32_t* map2img, float* cam, int rows, int P, int C,
                 void* stream);

/* ---- rules of the layers the VGG16 path never reaches (ResNet encoders), csrc/lrpx_rules.hip ---- */
/* Linear.propagate_relevance, epsilon rule (LRPtools/lrp_modules.py:9-37).  x [n_rows][n_in] is the layer's SAVED input
 * and is mutated in place like the reference's (:14): exact zeros become RELEVANCE_RECT = -1e-6.  Z = x W^T; with
 * bias == NULL (ignore_bias, the preset of lrp_wrapper.py:7-12) Z += 0.01 sign(Z), exact zeros -> 0.01; otherwise
 * Z += bias (:20-21).  r_in = x * ((r_out / Z) W).  w [n_out][n_in] as nn.Linear stores it; s_ws: workspace of
 * n_rows * n_out floats (S = r_out / Z).  Any sizes. */
int lrpx_linear_eps_rule(float* x, const float* w, const float* bias, const float* r_out, float* s_ws, float* r_in,
                         int n_rows, int n_in, int n_out, void* stream);
/* BatchNorm2d / BatchNorm1d.propagate_relevance, method != 'identity' (LRPtools/lrp_modules.py:197-246):
 * r_in = safe_divide(|x w|, |x w| + |b|) * r_out with w = gamma / sqrt(var + eps), b = beta - mean gamma / sqrt(var + eps).
 * The result has n_outer * channels * inner elements, element e in channel (e / inner) % channels.  broadcast_x = 0: x and
 * r_out have the same shape (BatchNorm2d on (N,C,H,W): n_outer = N, inner = H*W).  broadcast_x = 1 (n_outer = 1): x and
 * r_out hold `inner` elements and are indexed by e % inner - what the reference's `[:, None, None]` indexing makes of a
 * (N,C) or (1,C,L) input of BatchNorm1d (:236-238): a (C, N, C) / (C, C, L) result. */
int lrpx_batchnorm_rule(const float* x, const float* r_out, const float* gamma, const float* beta, const float* mean,
                        const float* var, float eps, float* r_in, long n_outer, int channels, long inner, int broadcast_x,
                        void* stream);
/* Add.propagate_relevance (LRPtools/lrp_modules.py:256-280): r_k = r_out x_k / (x1 + x2 + 0.01 sign(x1 + x2)), NaN -> 0,
 * plus r_out / 2 each where x1 + x2 == 0.  Non-finite results (x1 == -x2 != 0) are left for lrpx_check, as the reference's
 * asserts (:276-279) catch them. */
int lrpx_add_rule(const float* x1, const float* x2, const float* r_out, float* r1, float* r2, long n, void* stream);
/* Pool2d.propagate_relevance for nn.AvgPool2d (LRPtools/lrp_modules.py:176-177,182-195; table entry :327): Z = avgpool(x),
 * S = r_out / (Z + 1e-7 [Z == 0]) (utils.py:16-18), r_in = x * avgpool^T(S).  x / r_in: (planes, h, w), r_out: (planes, oh, ow) with
 * planes = N * C of the module's NCHW tensors; s_ws: planes * oh * ow floats of scratch.  Window bounds, divisor
 * (count_include_pad, divisor_override > 0, ceil_mode through oh / ow) and summation orders follow ATen's CPU kernels:
 * bit-exact against the reference. */
int lrpx_avgpool_rule(const float* x, const float* r_out, float* s_ws, float* r_in, long planes, int h, int w, int oh, int ow,
                      int kh, int kw, int sh, int sw, int ph, int pw, int count_include_pad, int divisor_override, void* stream);
/* max |a - b| into one device float (Dropout.propagate_relevance's check, LRPtools/lrp_modules.py:251; NaN counts as inf) */
int lrpx_max_abs_diff(const float* a, const float* b, long n, float* out_dev, void* stream);

/* Guided-Grad-CAM (ExplainGridTDGuidedGradCam.explain_cnn, models/gridTDmodel.py:1814-1836; AoA: models/aoamodel.py:1729-1751):
 * out[n][c] = guided[n][c] * E_n with E_n = expand_m cam_n expand_m^T, the (hw x hw) `skimage.transform.pyramid_expand(cam,
 * upscale = hw / p)` of the (p x p) Grad-CAM heat map cam_n.  guided / out: (rows, channels, hw, hw); cam: (rows, p*p);
 * expand_m: (hw, p) row-major, Gaussian smoothing x bilinear resize as ONE matrix (both steps are linear and separable;
 * built on the host by lrp_amd.ops.pyramid_expand_matrix). */
int lrpx_guided_gradcam(const float* guided, const float* cam, const float* expand_m, float* out, int rows, int p, int hw,
                        int channels, void* stream);

/* ---- the VGG16 chains with a per-call context (see THREADING above): same arguments + opts (NULL = process defaults) -- */
int lrpx_vgg16_forward_ex(const void* packed, const float* img_nchw, int n_img, void* trace, float* feat_nhwc,
                          const lrpx_vgg16_opts* opts, void* stream);
int lrpx_vgg16_relevance_ex(const void* packed, const void* trace, int n_img, const float* r_feat_nhwc,
                            const int32_t* map2img, int n_maps, void* workspace, float* out_nchw,
                            const lrpx_vgg16_opts* opts, void* stream);
int lrpx_vgg16_guided_backprop_ex(const void* packed, const void* trace, int n_img, const float* d_feat_nhwc,
                                  const int32_t* map2img, int n_maps, void* workspace, float* out_nchw,
                                  const lrpx_vgg16_opts* opts, void* stream);
int lrpx_vgg16_gradient_ex(const void* packed, const void* trace, int n_img, const float* d_feat_nhwc,
                           const int32_t* map2img, int n_maps, void* workspace, float* out_nchw,
                           const lrpx_vgg16_opts* opts, void* stream);

#ifdef __cplusplus
}
#endif
#endif
