"""Thin torch-tensor wrappers over the lrpx C ABI (include/lrpx.h).

PyTorch is plumbing here: it owns device memory and the current HIP stream; every computation is a
hand-written HIP kernel inside csrc/liblrpx.so.  All tensors are CUDA(=HIP) float32 contiguous;
activations are NHWC ([maps, H*W, C])."""
import ctypes as C
import os

import torch

from . import _lib
from ._lib import (ConvDesc, EPI_FIRST, EPI_FWD_DUAL, EPI_PLAIN, EPI_REL, EPI_REL_MUL, PACK_BWD_FIRST, PACK_BWD_POS,
                   PACK_DENSE, PACK_DENSE_T, PACK_FWD_DUAL, STAB_EPS, STAB_NONE, STAB_SAFE, check, ptr, stream_ptr)


def _dev(t):
    if t is not None and not t.is_cuda:
        raise ValueError("lrpx ops need device tensors (there is no CPU path)")
    return t


def conv_kc(hw, taps, cin):
    return _lib.load().lrpx_conv_kc(hw, taps, cin)


def pack_weights(w, cout, cin, taps, mode, kc):
    """w: conv (cout,cin,3,3) or dense 2-D matrix on device -> packed fragment-major tensor."""
    lib = _lib.load()
    _dev(w)
    if mode in (PACK_FWD_DUAL,):
        n_oc, k = 2 * cout, cin
    elif mode in (PACK_BWD_POS, _lib.PACK_BWD_PLAIN, PACK_DENSE_T):
        n_oc, k = cin, cout
    elif mode == PACK_BWD_FIRST:
        n_oc, k = 2 * cin, cout
    else:
        n_oc, k = cout, cin
    n = lib.lrpx_packed_floats(n_oc, k, taps, kc)
    out = torch.empty(n, dtype=torch.float32, device=w.device)
    check(lib.lrpx_pack_weights(ptr(w.contiguous()), cout, cin, taps, mode, kc, ptr(out), stream_ptr()))
    return out


def pack_weights_bf16x3(w, cout, cin, mode, taps=9):
    """3x3 conv weights (taps = 9) or a dense (cout, cin) matrix (taps = 1; PACK_BWD_PLAIN = the transposed product of the epsilon
    rules, as pack_weights_f16x2) -> three bf16 planes, w == p0 + p1 + p2 exactly, in fragment layout (for conv_mfma(..., bf16x6=1))."""
    lib = _lib.load()
    n_oc, k = (cout, cin) if mode == _lib.PACK_FWD else (cin, cout)
    out = torch.empty(lib.lrpx_packed_bf16x3_bytes(n_oc, k, taps) // 2, dtype=torch.int16, device=w.device)
    check(lib.lrpx_pack_weights_bf16x3(ptr(w.contiguous()), cout, cin, taps, mode, ptr(out), stream_ptr()))
    return out


def pack_weights_f16x2(w, cout, cin, mode, taps=9):
    """3x3 conv weights (taps = 9) or a dense (cout, cin) matrix (taps = 1) -> layer-scaled fp16 hi/lo planes in fragment
    layout (for conv_mfma(..., f16x3=1)).  Dense: PACK_BWD_PLAIN = the transposed product  out[m, i] = sum_o a[m, o] w[o, i]
    of the epsilon rules (what PACK_DENSE_T is for the fp32 kernels)."""
    lib = _lib.load()
    n_oc, k = (cout, cin) if mode == _lib.PACK_FWD else (cin, cout)
    out = torch.empty(lib.lrpx_packed_f16x2_bytes(n_oc, k, taps) // 4, dtype=torch.float32, device=w.device)
    check(lib.lrpx_pack_weights_f16x2(ptr(w.contiguous()), cout, cin, taps, mode, ptr(out), stream_ptr()))
    return out


def pack_weights_f16f8(w, cout, cin, mode):
    """3x3 conv weights for conv_mfma(..., f16x3=2): fp16 hi planes + block-scaled fp6 fields of the cross-product operands."""
    lib = _lib.load()
    n_oc, k = (cin, cout)
    out = torch.empty(lib.lrpx_packed_f16f8_bytes(n_oc, k) // 4, dtype=torch.float32, device=w.device)
    check(lib.lrpx_pack_weights_f16f8(ptr(w.contiguous()), cout, cin, mode, ptr(out), stream_ptr()))
    return out


def zeros_arena(device, shapes):
    """{name: shape} -> {name: zero tensor}, all views of ONE allocation cleared by ONE fill (every view starts on a 256-byte boundary).
    A decoder trace is 11 - 21 small tensors: allocated one by one they cost a fill launch each (~5 us apiece on the stream)."""
    offs, total = {}, 0
    for k, shp in shapes.items():
        n = 1
        for d in shp:
            n *= int(d)
        offs[k] = (total, n)
        total += -(-n // 64) * 64
    flat = zeros(total, device=device)
    return {k: flat[o:o + n].view(*shapes[k]) for k, (o, n) in offs.items()}


def decoder_f16(mode=None):
    """Do the decoders' GEMMs (the (word, pixel) rules, the lock-step gate rules, the plain linears of a trace, the (T,V) scores) run on
    the fp16 split-product kernels (csrc/dense_f16x3.hip: two fp16 halves behind a power-of-two scale per ROW - 22 operand bits, 1.5 - 2.4x
    the decoder throughput)?  They FOLLOW the process's conv mode: in the default, exact mode 1 (and mode 0) every contraction of the
    path is fp32 arithmetic - the decoders on the fp32 MFMA (v_mfma_f32_16x16x4_f32 / 32x32x2) and fp32 VALU kernels, as the reference's
    `lrp_linear_eps` (models/gridTDmodel.py:744-765) and its nn.Linear forwards are - and only the opt-in speed modes 2 / 3 take the
    split products.  LRPX_DECODER_F16=0 / 1 overrides (A/B).  Read at every call of an engine (the packs of both kinds are built
    with the engine; `engine.force_f16 = True / False` pins one engine)."""
    v = os.environ.get("LRPX_DECODER_F16", "")
    if v != "":
        return v != "0"
    return (_lib.load().lrpx_set_conv_mode(-1) if mode is None else int(mode)) >= 2          # (mode: an engine's per-context conv mode)


def zeros(*shape, dtype=torch.float32, device="cuda"):
    """a zero tensor cleared THROUGH THE LIBRARY (lrpx_zero) on the current stream: a recorded step (_lib.Recording) then holds the
    clear like any other launch of the step (a `torch.zeros` would run once, at recording time, and never again)"""
    if torch.device(device).type != "cuda":          # (host-logic tests: layout checks of the arena without a GPU)
        return torch.zeros(*shape, dtype=dtype, device=device)
    t = torch.empty(*shape, dtype=dtype, device=device)
    if t.numel():
        check(_lib.load().lrpx_zero(ptr(t), t.numel() * t.element_size(), stream_ptr()))
    return t


def amax_maps(s, n_maps):
    """Float bits of max|s[n]| per map (int32 tensor): the operand scale an f16x3 convolution needs for its input."""
    s = _dev(s)
    out = torch.empty(n_maps, dtype=torch.int32, device=s.device)
    check(_lib.load().lrpx_amax_maps(ptr(s), n_maps, s.numel() // n_maps, ptr(out), stream_ptr()))
    return out


def conv_desc(inp, wpacked, n_maps, hw, cin, n_oc, taps, epi, *, pix_per_map=0, stab=STAB_NONE, oc_split=0,
              relu=0, bias=None, x=None, u=None, zdiv=None, map2img=None, out0=None, out1=None, bf16x6=0,
              f16x3=0, in_amax=None, out1_amax=None, pool_am=None, out0_amax=None, blocked=0):
    """the lrpx_conv_desc of one contraction (the tensors must outlive its use: the descriptor holds raw pointers)"""
    d = ConvDesc()
    d.in_, d.wpacked = ptr(_dev(inp)), ptr(_dev(wpacked))
    d.n_maps, d.hw, d.cin, d.n_oc, d.taps, d.pix_per_map = n_maps, hw, cin, n_oc, taps, pix_per_map
    d.epi, d.stab, d.oc_split, d.relu = epi, stab, oc_split, relu
    d.bf16x6, d.f16x3 = bf16x6, f16x3
    d.in_amax, d.out1_amax, d.pool_am, d.out0_amax = ptr(in_amax), ptr(out1_amax), ptr(pool_am), ptr(out0_amax)
    d.bias, d.x, d.u, d.zdiv, d.map2img = ptr(bias), ptr(x), ptr(u), ptr(zdiv), ptr(map2img)
    d.out0, d.out1 = ptr(out0), ptr(out1)
    d.blocked = blocked
    return d


def nhwc_to_blocked(src, n_groups, pix_per_group, c):
    """NHWC [n_groups][pix][c] -> the BLOCKED layout of the mode-3 relevance kernels (csrc/blocked.h), one block set per group"""
    lib = _lib.load()
    dst = torch.zeros(n_groups * lib.lrpx_blocked_floats(pix_per_group, c), dtype=torch.float32, device=src.device)
    check(lib.lrpx_nhwc_to_blocked(ptr(_dev(src)), ptr(dst), n_groups, pix_per_group, c, stream_ptr()))
    return dst


def blocked_to_nhwc(src, n_groups, pix_per_group, c, out=None):
    if out is None:
        out = torch.empty(n_groups, pix_per_group, c, dtype=torch.float32, device=src.device)
    check(_lib.load().lrpx_blocked_to_nhwc(ptr(src), ptr(out), n_groups, pix_per_group, c, stream_ptr()))
    return out


def _conv_rel_mul_blocked(inp, wpacked, n_maps, hw, cin, n_oc, taps, epi, **kw):
    """The f16x3 = 2 REL_MUL kernels take `in` BLOCKED and, except the 224 x 224 one, x and the output too (the fused VGG16 chain
    keeps them so end to end).  For callers with NHWC tensors - the single-layer parity tests - convert at this boundary: in as one
    block set over all maps, x as one block set per image, the output back into the caller's tensor."""
    lib = _lib.load()
    ncol = kw.get("oc_split") or n_oc
    pin = (hw // 2) ** 2 if kw.get("pool_am") is not None else hw * hw
    x, out0, out1 = kw.get("x"), kw.get("out0"), kw.get("out1")
    kw = dict(kw)
    inb = nhwc_to_blocked(inp, 1, n_maps * pin, cin)
    if hw == 224:                      # conv1_2's kernel: NHWC multiplicand and output
        kw["blocked"] = 1
        d = conv_desc(inb, wpacked, n_maps, hw, cin, n_oc, taps, epi, **kw)
        check(lib.lrpx_conv_mfma(C.byref(d), stream_ptr()))
        return
    kw["blocked"] = 7
    if x is not None:
        kw["x"] = nhwc_to_blocked(x, x.shape[0], hw * hw, ncol)
    ob = None
    if out0 is not None or out1 is not None:
        ob = torch.zeros(lib.lrpx_blocked_floats(n_maps * hw * hw, ncol), dtype=torch.float32, device=inp.device)
        kw["out0"] = ob if out0 is not None else None
        kw["out1"] = ob if out1 is not None else None
    d = conv_desc(inb, wpacked, n_maps, hw, cin, n_oc, taps, epi, **kw)
    check(lib.lrpx_conv_mfma(C.byref(d), stream_ptr()))
    blocked_to_nhwc(ob, 1, n_maps * hw * hw, ncol, out=out0 if out0 is not None else out1)


def conv_mfma(inp, wpacked, n_maps, hw, cin, n_oc, taps, epi, **kw):
    if kw.get("f16x3") == 2 and epi == EPI_REL_MUL and not kw.get("blocked"):
        return _conv_rel_mul_blocked(inp, wpacked, n_maps, hw, cin, n_oc, taps, epi, **kw)
    d = conv_desc(inp, wpacked, n_maps, hw, cin, n_oc, taps, epi, **kw)
    check(_lib.load().lrpx_conv_mfma(C.byref(d), stream_ptr()))


def nchw_to_nhwc(src, c_pad=None):
    n, c, h, w = src.shape
    c_pad = c_pad or c
    dst = torch.empty(n, h * w, c_pad, dtype=torch.float32, device=src.device)
    check(_lib.load().lrpx_nchw_to_nhwc(ptr(src.contiguous()), ptr(dst), n, c, h * w, c_pad, stream_ptr()))
    return dst


def nhwc_to_nchw(src, c, h, w):
    n, p, c_src = src.shape
    dst = torch.empty(n, c, h, w, dtype=torch.float32, device=src.device)
    check(_lib.load().lrpx_nhwc_to_nchw(ptr(src), ptr(dst), n, c, p, c_src, stream_ptr()))
    return dst


def maxpool2x2_fwd(x, n, h, w, c):
    y = torch.empty(n, (h // 2) * (w // 2), c, dtype=torch.float32, device=x.device)
    check(_lib.load().lrpx_maxpool2x2_fwd(ptr(x), ptr(y), n, h, w, c, stream_ptr()))
    return y


def maxpool2x2_relevance(x, r_out, zdiv, map2img, n_maps, h_out, w_out, c, want_r=True, want_s=False):
    r_in = torch.empty(n_maps, 4 * h_out * w_out, c, dtype=torch.float32, device=x.device) if want_r else None
    s_out = torch.empty(n_maps, 4 * h_out * w_out, c, dtype=torch.float32, device=x.device) if want_s else None
    check(_lib.load().lrpx_maxpool2x2_relevance(ptr(x), ptr(r_out), ptr(zdiv), ptr(map2img), ptr(r_in), ptr(s_out),
                                                n_maps, h_out, w_out, c, 0, stream_ptr()))
    return r_in, s_out


def divide_stab(r, z, map2img, stab):
    n = r.shape[0]
    s = torch.empty_like(r)
    check(_lib.load().lrpx_divide_stab(ptr(r), ptr(z), ptr(map2img), ptr(s), n, r[0].numel(), stab, stream_ptr()))
    return s


def cumsum_maps(maps, n_img, t_per_img, out=None):
    """running sums over the words of each image: what `explain_caption` returns (lrp_wrapper.py:64-82 quirk)"""
    if out is None:
        out = torch.empty_like(maps)
    check(_lib.load().lrpx_cumsum_maps(ptr(maps), ptr(out), n_img, t_per_img, maps[0].numel(), stream_ptr()))
    return out


def gather_rows(src, rows):
    """src (R, ...) 4-byte items, rows int32 (n,) on device -> (n, ...) = src[rows] (lrpx_gather_rows)"""
    src = _dev(src).contiguous()
    n = rows.shape[0]
    out = torch.empty((n,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    if n:
        check(_lib.load().lrpx_gather_rows(ptr(src), ptr(rows), ptr(out), n, src[0].numel(), stream_ptr()))
    return out


def scatter_maps(compact, rg, accumulate=False):
    """compact (n_valid, ...) results of the valid (image, word) rows of `rg` (explainers.ragged.RaggedRows) -> the padded
    (B*T, ...) layout: the rows themselves (accumulate: per-image running sums, the `explain_caption` quirk) and exact zeros
    behind every image's last word."""
    out = torch.empty((rg.B * rg.T,) + tuple(compact.shape[1:]), dtype=torch.float32, device=rg.rows.device)
    per = out[0].numel()
    if rg.n == 0:
        return out.zero_()
    check(_lib.load().lrpx_scatter_maps(ptr(_dev(compact).contiguous()), ptr(out), rg.B, rg.T, ptr(rg.lens), ptr(rg.offs), per,
                                        1 if accumulate else 0, stream_ptr()))
    return out


def check_relevance(buf, finite=True, nonzero=False):
    """The reference's inline asserts (lrp_modules.py:154-155, lrp_wrapper.py:81); syncs the stream."""
    check(_lib.load().lrpx_check(ptr(buf), buf.numel(), (1 if finite else 0) | (2 if nonzero else 0), stream_ptr()))


_EXPAND_CACHE = {}


def pyramid_expand_matrix(p, upscale, device="cuda"):
    """`skimage.transform.pyramid_expand(cam, upscale)` of a (p x p) map as ONE matrix: E = M cam M^T, M (p*upscale, p).
    scikit-image (the reference pins none; 0.16 was current for its PyTorch 1.4) computes
      resize(cam, order=1, mode='reflect', anti_aliasing=False): output row o samples the input at (o + 0.5) / upscale - 0.5,
          linear interpolation, out-of-range neighbours mirrored without repeating the edge (index -1 -> 1, p -> p - 2);
      ndi.gaussian_filter(sigma = 2 * upscale / 6, truncate 4, mode='reflect'): 2 * int(4 sigma + 0.5) + 1 taps, edge-repeating
          reflection (index -1 -> 0).
    Both are linear and separable, so they compose into M = Gauss @ Bilinear (built in float64 on the host, stored fp32)."""
    key = (p, upscale, str(device))
    if key not in _EXPAND_CACHE:
        import math
        import numpy as np
        hw = p * upscale
        Bm = np.zeros((hw, p))
        for o in range(hw):
            r = (o + 0.5) / upscale - 0.5
            r0 = math.floor(r)
            d = r - r0
            for idx, wgt in ((r0, 1.0 - d), (r0 + 1, d)):
                idx = -idx if idx < 0 else (2 * (p - 1) - idx if idx > p - 1 else idx)
                Bm[o, idx] += wgt
        sigma = 2.0 * upscale / 6.0
        lw = int(4.0 * sigma + 0.5)
        w = np.exp(-0.5 * (np.arange(-lw, lw + 1) / sigma) ** 2)
        w /= w.sum()
        G = np.zeros((hw, hw))
        for o in range(hw):
            for k in range(-lw, lw + 1):
                i = o + k
                i = -i - 1 if i < 0 else (2 * hw - 1 - i if i >= hw else i)
                G[o, i] += w[k + lw]
        _EXPAND_CACHE[key] = torch.from_numpy((G @ Bm).astype(np.float32)).to(device).contiguous()
    return _EXPAND_CACHE[key]


def guided_gradcam(guided_maps, cam, p=14):
    """guided_maps (N,C,HW,HW) * pyramid_expand(cam (N, p*p), upscale = HW / p)  (models/gridTDmodel.py:1826-1829)"""
    n, c, hw, _ = guided_maps.shape
    out = torch.empty_like(guided_maps)
    m = pyramid_expand_matrix(p, hw // p, guided_maps.device)
    check(_lib.load().lrpx_guided_gradcam(ptr(_dev(guided_maps).contiguous()), ptr(cam.contiguous()), ptr(m), ptr(out), n, p,
                                          hw, c, stream_ptr()))
    return out


class Vgg16:
    """VGG16 encoder context: packed weights + per-batch trace (device memory owned by torch)."""

    def __init__(self, weights, biases):
        lib = _lib.load()
        dev = weights[0].device
        self.device = dev
        self.packed = torch.empty(lib.lrpx_vgg16_packed_bytes() // 4, dtype=torch.float32, device=dev)
        ws = [w.contiguous() for w in weights]
        bs = [b.contiguous() for b in biases]
        wp = (C.c_void_p * 13)(*[w.data_ptr() for w in ws])
        bp = (C.c_void_p * 13)(*[b.data_ptr() for b in bs])
        check(lib.lrpx_vgg16_pack(wp, bp, ptr(self.packed), stream_ptr()))
        torch.cuda.current_stream().synchronize()   # ws/bs may be temporaries
        # the image-gradient chains in conv mode 3 keep the 1e-4 grade only while the rows of a 16-row weight slice stay within ~2^6 of
        # each other (lrpx_vgg16_row_spread, include/lrpx.h): beyond that this context runs them on the fp16 split products (mode 2)
        off = (lib.lrpx_vgg16_row_spread(ptr(self.packed)) - self.packed.data_ptr()) // 4
        spread = self.packed[off: off + 17].cpu()
        self.row_spread = float(max(spread[l] for l in range(17) if self.IS_CONV[l] and l > 0))
        self.grad_mode2 = self.row_spread > self.GRAD_SPREAD_MAX
        self.trace = None
        self.n_img = 0
        self._ws = None
        # per-context matrix-core mode (None = the process default of lrpx_set_conv_mode / lrpx_set_forward_f16): carried in
        # every call (lrpx_vgg16_opts), so contexts with different modes can be in flight on different streams / threads
        self.conv_mode = None
        self.forward_f16 = None

    # Where the 64 comes from.  The mode-3 gradient kernels carry the two cross products (2^-11 of a product each)
    # as fp6 e2m3 fields behind ONE block scale per 16-row weight slice: a row whose maximum is s times below its slice's has its fields
    # rounded with a half-step of up to s / 60 of their own size (e2m3: half-step 2^-4 at field 1, slice maximum at >= 3.75).  A product
    # of such a row is therefore off by <= 2 x 2^-11 x s / 60 = s x 1.6e-5 of ITSELF, random sign.  An output element sums >= 9 x 16 =
    # 144 products of a slice and the chain has 12 such layers: if it were carried by the smallest rows alone its error would be about
    # s x 1.6e-5 x sqrt(12 / 144) = s x 4.7e-6, i.e. 1e-4 at s ~ 21; elements that large rows feed as well (all of them, with random
    # weights and gradients) dilute that by the rows' own ratio.  Measured on the oracle's activations: 7.6e-6 at s = 63
    # (tests/test_gpu_guided.py::test_gradient_chain_at_the_row_spread_boundary), 8.4e-5 at a slice spread of 2^13 and 1.1e-4 at 2^25
    # (log-normal rows, tests/test_gpu_vgg.py::test_chain_hostile_weights_all_modes): 64 keeps a 4x margin to the contract where the
    # model above would allow ~21 in the worst case and the data allow ~2^13.  Conv mode 1 (the default) has no such rule: exact splits.
    GRAD_SPREAD_MAX = 64.0

    def _opts(self, layer_ms=None, grad=False):
        o = _lib.VggOpts()
        o.conv_mode = -1 if self.conv_mode is None else int(self.conv_mode)
        if grad and self.grad_mode2:
            cm, ff = C.c_int(0), C.c_int(0)
            check(_lib.load().lrpx_vgg16_resolve_opts(C.byref(o), C.byref(cm), C.byref(ff)))
            if cm.value == 3:
                o.conv_mode = 2
        o.forward_f16 = -1 if self.forward_f16 is None else int(self.forward_f16)
        o.layer_ms = layer_ms
        return C.byref(o)

    def replica(self):
        """Same packed weights, own trace / workspace buffers (for a second batch in flight on another stream)."""
        import copy
        r = copy.copy(self)
        r.trace, r.n_img, r._ws = None, 0, None
        if hasattr(r, "_ws_multi"):
            del r._ws_multi
        return r

    def forward(self, img_nchw):
        """Encoder.forward + trace (models/gridTDmodel.py:40-43).  Returns features (B,196,512) NHWC (a view
        into the trace)."""
        lib = _lib.load()
        n = img_nchw.shape[0]
        assert tuple(img_nchw.shape[1:]) == (3, 224, 224)
        need = lib.lrpx_vgg16_trace_bytes(n) // 4
        if self.trace is None or self.trace.numel() < need or self.n_img != n:
            self.trace = torch.empty(need, dtype=torch.float32, device=self.device)
        self.n_img = n
        check(lib.lrpx_vgg16_forward_ex(ptr(self.packed), ptr(img_nchw.contiguous()), n, ptr(self.trace), None,
                                        self._opts(), stream_ptr()))
        off = lib.lrpx_vgg16_trace_features(ptr(self.trace), n) - self.trace.data_ptr()
        return self.trace[off // 4: off // 4 + n * 196 * 512].view(n, 196, 512)

    # (hw, channels) of act[l] for l = 0..17 and of zpos[l]
    ACT_DIMS = [(224, 8), (224, 64), (224, 64), (112, 64), (112, 128), (112, 128), (56, 128), (56, 256), (56, 256),
                (56, 256), (28, 256), (28, 512), (28, 512), (28, 512), (14, 512), (14, 512), (14, 512), (14, 512)]
    IS_CONV = [1, 1, 0, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1, 0, 1, 1, 1]

    def channel_scales(self, layer):
        """rs (cout,) of conv layer `layer` (0..16): the trace keeps Z+ of that layer as rs[c] * conv(X, W+)[c] and the relevance
        packs carry rs[c] * W+[c, :] (exact powers of two; lrpx_vgg16_channel_scales, include/lrpx.h)."""
        n = C.c_int(0)
        p = _lib.load().lrpx_vgg16_channel_scales(ptr(self.packed), int(layer), C.byref(n))
        if not p:
            raise ValueError(f"layer {layer} is not a conv layer of VGG16 cfg 'D'")
        off = (p - self.packed.data_ptr()) // 4
        return self.packed[off: off + n.value]

    def trace_views(self):
        """Views of the saved per-layer tensors (what `module.input` is to the reference's hooks):
        ([act[0..17]] as (n_img, hw*hw, C), [zpos[l] or None]); zpos[l] is Z+ of conv l TIMES `channel_scales(l)` per channel."""
        lib = _lib.load()
        n = self.n_img
        a_off = (C.c_size_t * 18)()
        z_off = (C.c_size_t * 17)()
        check(lib.lrpx_vgg16_trace_layout(n, a_off, z_off))
        acts, zs = [], []
        for l in range(18):
            hw, c = self.ACT_DIMS[l]
            acts.append(self.trace[a_off[l]: a_off[l] + n * hw * hw * c].view(n, hw * hw, c))
        for l in range(17):
            if self.IS_CONV[l]:
                c = self.ACT_DIMS[l + 1][1]
                hw = self.ACT_DIMS[l][0]
                zs.append(self.trace[z_off[l]: z_off[l] + n * hw * hw * c].view(n, hw * hw, c))
            else:
                zs.append(None)
        return acts, zs

    def derive(self):
        """Recompute the trace tensors derived from the activations (call after writing into `trace_views()`)."""
        check(_lib.load().lrpx_vgg16_trace_derive(ptr(self.trace), self.n_img, stream_ptr()))

    def gradient(self, d_feat_nhwc, map2img=None, out=None):
        """explain_cnn of the plain-gradient explainer (models/gridTDmodel.py:1507-1521): the autograd gradient of the
        encoder output w.r.t. the image for (N,196,512) output gradients -> (N,3,224,224)."""
        return self.guided_backprop(d_feat_nhwc, map2img, out, _fn="lrpx_vgg16_gradient_ex")

    def guided_backprop(self, d_feat_nhwc, map2img=None, out=None, _fn="lrpx_vgg16_guided_backprop_ex"):
        """explain_cnn of the guided-backprop explainer (models/gridTDmodel.py:1702-1723): (N,196,512) gradient at
        the encoder output -> (N,3,224,224) image gradient with the guided ReLU rule."""
        lib = _lib.load()
        n_maps = d_feat_nhwc.shape[0]
        need = lib.lrpx_vgg16_workspace_bytes(n_maps) // 4
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.float32, device=self.device)
        if out is None:
            out = torch.empty(n_maps, 3, 224, 224, dtype=torch.float32, device=self.device)
        check(getattr(lib, _fn)(ptr(self.packed), ptr(self.trace), self.n_img, ptr(d_feat_nhwc.contiguous()),
                                ptr(map2img), n_maps, ptr(self._ws), ptr(out), self._opts(grad=True), stream_ptr()))
        return out

    def relevance(self, r_feat_nhwc, map2img=None, out=None, streams=1, layer_ms=None):
        """compute_lrp (LRPtools/lrp_wrapper.py:63-87) for N maps: (N,196,512) -> (N,3,224,224).
        streams=2 splits the maps over two HIP streams (maps are independent): the HBM-bound pool / first-layer
        kernels and the tail wave of each MFMA launch of one half overlap with MFMA work of the other half.
        layer_ms: a ctypes (c_float * 17) array - this call then times its conv launches with HIP events of its own and
        waits for them (profiling; one stream only)."""
        lib = _lib.load()
        n_maps = r_feat_nhwc.shape[0]
        r_feat_nhwc = r_feat_nhwc.contiguous()
        if out is None:
            out = torch.empty(n_maps, 3, 224, 224, dtype=torch.float32, device=self.device)
        if streams <= 1 or n_maps < 2 * streams or map2img is None:
            need = lib.lrpx_vgg16_workspace_bytes(n_maps) // 4
            if self._ws is None or self._ws.numel() < need:
                self._ws = None
                self._ws = torch.empty(need, dtype=torch.float32, device=self.device)
            if layer_ms is None and self.conv_mode is None:
                # plain entry point: honours the thread's legacy lrpx_vgg16_layer_timing switch
                check(lib.lrpx_vgg16_relevance(ptr(self.packed), ptr(self.trace), self.n_img, ptr(r_feat_nhwc),
                                               ptr(map2img), n_maps, ptr(self._ws), ptr(out), stream_ptr()))
            else:
                lm = None if layer_ms is None else C.cast(layer_ms, C.POINTER(C.c_float))
                check(lib.lrpx_vgg16_relevance_ex(ptr(self.packed), ptr(self.trace), self.n_img, ptr(r_feat_nhwc),
                                                  ptr(map2img), n_maps, ptr(self._ws), ptr(out), self._opts(lm),
                                                  stream_ptr()))
            return out
        per = -(-n_maps // streams)
        need = lib.lrpx_vgg16_workspace_bytes(per) // 4
        if getattr(self, "_ws_multi", None) is None or self._ws_multi[0].numel() < need or len(self._ws_multi) < streams:
            self._ws_multi = [torch.empty(need, dtype=torch.float32, device=self.device) for _ in range(streams)]
            self._side = [torch.cuda.Stream() for _ in range(streams)]
        cur = torch.cuda.current_stream()
        for i in range(streams):
            lo, hi = i * per, min((i + 1) * per, n_maps)
            if lo >= hi:
                break
            st = self._side[i]
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                check(lib.lrpx_vgg16_relevance_ex(ptr(self.packed), ptr(self.trace), self.n_img, ptr(r_feat_nhwc[lo:hi]),
                                                  ptr(map2img[lo:hi]), hi - lo, ptr(self._ws_multi[i]), ptr(out[lo:hi]),
                                                  self._opts(), C.c_void_p(st.cuda_stream)))
        for i in range(streams):
            cur.wait_stream(self._side[i])
        return out
