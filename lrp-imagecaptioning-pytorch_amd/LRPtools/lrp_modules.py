"""Per-layer relevance rules — mirror of the reference's LRPtools/lrp_modules.py, computed by HIP kernels.

Same protocol: `get_lrp_module(module)` dispatches on `type(module)` (ValueError for unknown leaves,
lrp_modules.py:321-341) and `propagate_relevance(module, relevance_input, relevance_output, lrp_method,
lrp_params)` returns a tuple with the arity of `relevance_input` (:157-170).  `module.input` must hold the
layer input, as the reference's `save_input_hook` leaves it (lrp_wrapper.py:24-25).  Tensors are NCHW on the
device, as in the reference; layouts are converted at this boundary.

Conv2d 3x3/pad 1, MaxPool2d(2,2) and ReLU are the layers VGG16 exercises (kernels for square maps of 224/112/56/28/14
pixels; any other H x W <= 224 runs zero-embedded in the next larger of those; other kernel sizes / strides raise ValueError).  Linear / BatchNorm2d / BatchNorm1d / Dropout / Add / Flatten / AvgPool2d (SURVEY.md §8(a) M4 and W3's table, only
reached with the reference's ResNet encoders) are HBM-bound streaming kernels (csrc/lrpx_rules.hip), any shape."""
import torch
import torch.nn as nn

from .. import _lib, ops
from .._lib import EPI_FWD_DUAL, EPI_REL, PACK_BWD_FIRST, PACK_FWD_DUAL_FIRST, STAB_SAFE, check, ptr, stream_ptr

_SIZES = (224, 112, 56, 28, 14)


def _require(cond, msg):
    if not cond:
        raise ValueError(msg)


def _pad_to(c, m):
    return -(-c // m) * m


class ReLU:
    def propagate_relevance(self, module, relevance_input, relevance_output, lrp_method, lrp_params=None):
        if lrp_method == 'identity':                        # lrp_modules.py:42-46: pass through
            ops.check_relevance(relevance_output[0])
            return (relevance_output[0],)
        raise NotImplementedError("ReLU is always registered with the 'identity' rule (lrp_wrapper.py:51-52)")


class Conv2d:
    """alpha=1, beta=0, ignore_bias rule (lrp_modules.py:124-150) for signed or non-negative inputs:
    Z = conv(x+,W+) + conv(x-,W-);  S = R/safe(Z);  R_in = x+ * convT(S,W+) + x- * convT(S,W-).
    The input is stored split [x+ | x-] so one MFMA pass serves both terms."""

    def propagate_relevance(self, module, relevance_input, relevance_output, lrp_method, lrp_params=None):
        if lrp_method != "alpha_beta":
            raise NotImplementedError('Only adopt alpha 1 rule for conv layer')       # lrp_modules.py:152
        x = module.input[0].detach()
        r_out = relevance_output[0].detach()
        _require(isinstance(module, nn.Conv2d) and module.kernel_size == (3, 3) and module.padding == (1, 1)
                 and module.stride == (1, 1) and module.groups == 1, "lrpx Conv2d rule: 3x3 / pad 1 / stride 1 only")
        n, cin, h0, w0 = x.shape
        cout = module.out_channels
        _require(r_out.shape == (n, cout, h0, w0), "relevance_output shape mismatch")
        _require(max(h0, w0) <= _SIZES[0], f"lrpx Conv2d rule: maps of at most {_SIZES[0]}x{_SIZES[0]} pixels, got {h0}x{w0}")
        # The kernels are built for the five square map sizes of VGG16.  Any other H x W runs on the next larger one with
        # the map in the top-left corner of a zero canvas: a zero-padded 3x3 conv sees the same zeros beyond the map's
        # edge as beyond the canvas', R_out is zero outside the map, so S = R_out / Z is too, and the rows / columns
        # [0,H) x [0,W) of the result are the rule's output exactly (tests: the 16x16 / 8x8 fixture net of layers.npz).
        h = w = min(sz for sz in _SIZES if sz >= max(h0, w0))
        if (h0, w0) != (h, w):
            xe = torch.zeros(n, cin, h, w, device=x.device, dtype=x.dtype)
            xe[:, :, :h0, :w0] = x
            re = torch.zeros(n, cout, h, w, device=r_out.device, dtype=r_out.dtype)
            re[:, :, :h0, :w0] = r_out
            x, r_out = xe, re
        lib = _lib.load()
        st = stream_ptr()
        dev = x.device
        gran = 16 if h >= 112 else 32
        c2 = 8 if (h == 224 and 2 * cin <= 8) else _pad_to(2 * cin, gran)   # split input channels [x+ | x- | 0..]
        co_p = _pad_to(cout, 32)
        cache = module.__dict__.setdefault("_lrpx_pack", {})
        key = (h, module.weight._version, module.weight.data_ptr())
        if cache.get("key") != key:
            wt = module.weight.detach().to(torch.float32).contiguous()
            if co_p != cout:
                wt = torch.cat([wt, torch.zeros(co_p - cout, cin, 3, 3, device=dev)], 0)
            # forward: k = [x+ | x-] channels, Z part uses W+ on x+ and W- on x-
            kc_f = ops.conv_kc(h, 9, c2)
            n_f = lib.lrpx_packed_floats(2 * co_p, c2, 9, kc_f)
            pf = torch.zeros(n_f, device=dev)
            # pack with cin_eff = c2/2 so that the two halves line up with the split storage
            half = c2 // 2
            wt_h = torch.zeros(co_p, half, 3, 3, device=dev)
            wt_h[:, :cin] = wt
            check(lib.lrpx_pack_weights(ptr(wt_h), co_p, half, 9, PACK_FWD_DUAL_FIRST, kc_f, ptr(pf), st))
            kc_b = ops.conv_kc(h, 9, co_p)
            pb = torch.zeros(lib.lrpx_packed_floats(_pad_to(c2, 32), co_p, 9, kc_b), device=dev)
            check(lib.lrpx_pack_weights(ptr(wt_h), co_p, half, 9, PACK_BWD_FIRST, kc_b, ptr(pb), st))
            cache.update(key=key, pf=pf, pb=pb, half=half)
        half = cache["half"]
        xs = torch.empty(n, h * w, c2, device=dev)
        xsrc = x.to(torch.float32)
        if half != cin:     # x+ occupies channels [0,half), x- starts at `half`: pad the channel axis with zeros
            xsrc = torch.cat([xsrc, torch.zeros(n, half - cin, h, w, device=dev)], 1)
        check(lib.lrpx_nchw_to_nhwc_posneg(ptr(xsrc.contiguous()), ptr(xs), n, half, h * w, c2, st))
        act = torch.empty(n, h * w, co_p, device=dev)
        zpos = torch.empty(n, h * w, co_p, device=dev)
        ops.conv_mfma(xs, cache["pf"], n, h, c2, 2 * co_p, 9, EPI_FWD_DUAL, oc_split=co_p, out0=act, out1=zpos)
        r_nhwc = ops.nchw_to_nhwc(r_out.to(torch.float32), co_p)
        s = ops.divide_stab(r_nhwc, zpos, None, STAB_SAFE)
        n_oc = _pad_to(c2, 32)
        r_split = torch.empty(n, h * w, c2, device=dev)
        ops.conv_mfma(s, cache["pb"], n, h, co_p, n_oc, 9, EPI_REL, oc_split=c2, x=xs, out0=r_split)
        r_half = torch.empty(n, h * w, half, device=dev)
        check(lib.lrpx_fold_halves(ptr(r_split), ptr(r_half), n * h * w, half, st))
        R = ops.nhwc_to_nchw(r_half, cin, h, w)
        if (h0, w0) != (h, w):
            R = R[:, :, :h0, :w0].contiguous()
        ops.check_relevance(R)                                      # lrp_modules.py:154-155
        if relevance_input is not None and len(relevance_input) == 3:
            return R, relevance_input[1], relevance_input[2]
        if relevance_input is not None and len(relevance_input) == 2:
            return R, relevance_input[1]
        return (R,)


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class Pool2d:
    """Pool2d rule (lrp_modules.py:172-195).  MaxPool2d(2,2): winner-take-all routing, first maximum wins.
    AvgPool2d (any kernel / stride / padding / count_include_pad / ceil_mode - the attributes the reference clones at :176-177):
    Z = avgpool(X), R = X * avgpool^T(R_out / safe(Z)).  Quirk reproduced: the reference's clone does NOT carry
    `divisor_override`, so its rule divides by the default window size whatever the module says (fixture `k23_div5`)."""

    def _avgpool(self, module, relevance_output):
        x = _f32c(module.input[0])
        r_out = _f32c(relevance_output[0])
        _require(x.dim() == 4 and r_out.dim() == 4, "lrpx Pool2d rule: AvgPool2d on (N, C, H, W) tensors")
        n, c, h, w = x.shape
        (kh, kw), (ph, pw) = _pair(module.kernel_size), _pair(module.padding)
        sh, sw = _pair(module.stride if module.stride is not None else module.kernel_size)
        oh, ow = r_out.shape[2], r_out.shape[3]
        _require(r_out.shape[:2] == (n, c), "relevance_output shape mismatch")
        R = torch.empty_like(x)
        s_ws = torch.empty_like(r_out)
        check(_lib.load().lrpx_avgpool_rule(ptr(x), ptr(r_out), ptr(s_ws), ptr(R), n * c, h, w, oh, ow, kh, kw, sh, sw, ph, pw,
                                            1 if module.count_include_pad else 0, 0, stream_ptr()))   # (divisor_override: dropped by the reference's clone)
        ops.check_relevance(R)                                      # lrp_modules.py:192-193
        return (R,)

    def propagate_relevance(self, module, relevance_input, relevance_output, lrp_method, lrp_params=None):
        if isinstance(module, nn.AvgPool2d):
            return self._avgpool(module, relevance_output)
        _require(isinstance(module, nn.MaxPool2d), "lrpx Pool2d rule: MaxPool2d / AvgPool2d only")    # lrp_modules.py:179
        ks = module.kernel_size if isinstance(module.kernel_size, tuple) else (module.kernel_size,) * 2
        stq = module.stride if isinstance(module.stride, tuple) else (module.stride,) * 2
        _require(ks == (2, 2) and stq == (2, 2) and module.padding in (0, (0, 0)), "lrpx Pool2d rule: 2x2 / stride 2 only")
        x = module.input[0].detach()
        r_out = relevance_output[0].detach()
        n, c, h, w = x.shape
        _require(h % 2 == 0 and w % 2 == 0, "lrpx Pool2d rule: even spatial size only")
        cp = _pad_to(c, 4)
        xs = ops.nchw_to_nhwc(x.to(torch.float32), cp)
        rs = ops.nchw_to_nhwc(r_out.to(torch.float32), cp)
        r_in, _ = ops.maxpool2x2_relevance(xs, rs, None, None, n, h // 2, w // 2, cp)
        R = ops.nhwc_to_nchw(r_in, c, h, w)
        ops.check_relevance(R)
        return (R,)


class resAdd(nn.Module):
    """The reference's explicit residual-sum module (models/resnet.py:32-37; imported as `resAdd` at lrp_modules.py:5)."""

    def forward(self, x, y):
        return x + y


class resFlatten(nn.Module):
    """models/resnet.py:24-29 (imported as `resFlatten` at lrp_modules.py:6)"""

    def forward(self, x):
        return x.view(x.size(0), -1)


def _f32c(t):
    return t.detach().to(torch.float32).contiguous()


class Linear:
    """Epsilon rule (lrp_modules.py:9-37): the saved input's exact zeros become -1e-6 IN PLACE (:14, quirk h);
    Z = x W^T + 0.01 sign(Z) (exact zeros -> 0.01), or + bias when `ignore_bias` is off; R = x * ((R_out / Z) W)."""

    def propagate_relevance(self, module, relevance_input, relevance_output, lrp_method, lrp_params=None):
        ignore_bias = (lrp_params or {}).get("ignore_bias", True)
        input_ = module.input[0]
        _require(input_.dim() == 2 and input_.shape[1] == module.in_features, "lrpx Linear rule: (N, in_features) input")
        x = input_ if (input_.dtype == torch.float32 and input_.is_contiguous()) else _f32c(input_)
        r_out = _f32c(relevance_output[0])
        n, i = x.shape
        o = module.out_features
        _require(r_out.shape == (n, o), "relevance_output shape mismatch")
        w = _f32c(module.weight)
        b = None if ignore_bias else _f32c(module.bias)
        s_ws = torch.empty(n, o, device=x.device)
        R = torch.empty(n, i, device=x.device)
        check(_lib.load().lrpx_linear_eps_rule(ptr(x.detach()), ptr(w), ptr(b), ptr(r_out), ptr(s_ws), ptr(R), n, i, o,
                                               stream_ptr()))
        if x is not input_:                                 # the mutation must land on the saved input (:14)
            with torch.no_grad():
                input_.copy_(x)
        ops.check_relevance(R)                              # :26-27
        if relevance_input is not None and len(relevance_input) == 3:
            return relevance_input[0], R, relevance_input[2]
        if relevance_input is not None and len(relevance_input) == 2:
            return R, relevance_input[1]
        return (R,)


class _BatchNormRule:
    """R = safe_divide(|x w|, |x w| + |b|) * R_out with the folded scale / shift of the running statistics
    (lrp_modules.py:197-246); 'identity' passes R_out through.  Returns (R, relevance_input[1], relevance_input[2])."""

    def _rule(self, module, relevance_output):
        x = _f32c(module.input[0])
        r_out = _f32c(relevance_output[0])
        c = module.num_features
        args = [_f32c(t) for t in (module.weight, module.bias, module.running_mean, module.running_var)]
        lib = _lib.load()
        if x.dim() == 4:                                    # (N,C,H,W) against w[:, None, None]: the per-channel rule
            _require(x.shape[1] == c and r_out.shape == x.shape, "lrpx BatchNorm rule: (N,C,H,W) input and relevance")
            R = torch.empty_like(x)
            check(lib.lrpx_batchnorm_rule(ptr(x), ptr(r_out), *[ptr(a) for a in args], float(module.eps), ptr(R),
                                          x.shape[0], c, x.shape[2] * x.shape[3], 0, stream_ptr()))
            return R
        # the reference indexes w[:, None, None] in BatchNorm1d too (:236-238): a (N,C) input broadcasts to (C,N,C), a
        # (1,C,L) input to (C,C,L); any other shape fails to broadcast there as well
        if x.dim() == 2:
            _require(x.shape[1] in (c, 1) or c == 1, "lrpx BatchNorm1d rule: shapes do not broadcast")
            shape = (c,) + tuple(x.shape)
        elif x.dim() == 3 and x.shape[0] == 1:
            shape = (c,) + tuple(x.shape[1:])
        else:
            raise RuntimeError("The size of tensor a ({}) must match the size of tensor b ({}) at non-singleton "
                               "dimension 0".format(x.shape[0], c))
        _require(r_out.shape == x.shape, "relevance_output shape mismatch")
        R = torch.empty(shape, device=x.device, dtype=torch.float32)
        check(lib.lrpx_batchnorm_rule(ptr(x), ptr(r_out), *[ptr(a) for a in args], float(module.eps), ptr(R), 1, c,
                                      x.numel(), 1, stream_ptr()))
        return R

    def propagate_relevance(self, module, relevance_input, relevance_output, lrp_method, lrp_params=None):
        R = relevance_output[0] if lrp_method == 'identity' else self._rule(module, relevance_output)
        ops.check_relevance(R, finite=True, nonzero=True)   # :217-219 incl. `assert R.sum() != 0`
        return R, relevance_input[1], relevance_input[2]


class BatchNorm2d(_BatchNormRule):
    pass


class BatchNorm1d(_BatchNormRule):
    pass


class Dropout:
    """(lrp_modules.py:248-254): relevance passes unchanged; asserts |R_out - R_in| < 1e-7 like the reference."""

    def propagate_relevance(self, module, relevance_input, relevance_output, lrp_method, lrp_params=None):
        a, b = _f32c(relevance_output[0]), _f32c(relevance_input[0])
        assert a.shape == b.shape
        m = torch.empty(1, device=a.device, dtype=torch.float32)
        check(_lib.load().lrpx_max_abs_diff(ptr(a), ptr(b), a.numel(), ptr(m), stream_ptr()))
        assert m.item() < 1e-7
        return relevance_input


class Add:
    """`Add` (lrp_modules.py:256-280): proportional split between the two summands; both get R/2 where the sum is zero."""

    def propagate_relevance(self, module, relevance_input, relevance_output, lrp_method, lrp_params=None):
        x1, x2 = _f32c(module.input[0]), _f32c(module.input[1])
        r_out = _f32c(relevance_output[0])
        _require(x1.shape == x2.shape == r_out.shape, "lrpx Add rule: the summands and the relevance share one shape")
        R1, R2 = torch.empty_like(x1), torch.empty_like(x1)
        check(_lib.load().lrpx_add_rule(ptr(x1), ptr(x2), ptr(r_out), ptr(R1), ptr(R2), x1.numel(), stream_ptr()))
        ops.check_relevance(R1)                             # :276-279
        ops.check_relevance(R2)
        return R1, R2


class Flatten:
    """`Flatten` (lrp_modules.py:282-291): the relevance in the shape of the layer input."""

    def propagate_relevance(self, module, relevance_input, relevance_output, lrp_method, lrp_params=None):
        r_out = _f32c(relevance_output[0])
        R = torch.empty(module.input[0].size(), device=r_out.device, dtype=torch.float32)
        check(_lib.load().lrpx_scale(ptr(r_out), ptr(R), r_out.numel(), 1.0, stream_ptr()))
        ops.check_relevance(R)
        return (R,)


_RULES = {nn.ReLU: ReLU, nn.Conv2d: Conv2d, nn.MaxPool2d: Pool2d, nn.AvgPool2d: Pool2d, nn.Linear: Linear, nn.BatchNorm2d: BatchNorm2d,
          nn.BatchNorm1d: BatchNorm1d, nn.Dropout: Dropout, nn.Dropout2d: Dropout, resAdd: Add, resFlatten: Flatten}
# the reference's own `models.resnet.Add` / `Flatten` classes (a user's ResNet is built from those) dispatch by name
_RULES_BY_NAME = {"Add": Add, "Flatten": Flatten}


def get_lrp_module(module):
    """type(module) -> rule object; ValueError for leaves the path does not know (lrp_modules.py:321-341)."""
    cls = _RULES.get(type(module))
    if cls is None and type(module).__module__.endswith("resnet"):
        cls = _RULES_BY_NAME.get(type(module).__name__)
    if cls is None:
        raise ValueError("Layer type {} not known.".format(type(module)))
    return cls()
