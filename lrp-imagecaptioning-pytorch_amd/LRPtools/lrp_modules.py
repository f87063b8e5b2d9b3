"""Per-layer relevance rules — mirror of the reference's LRPtools/lrp_modules.py, computed by HIP kernels.

Same protocol: `get_lrp_module(module)` dispatches on `type(module)` (ValueError for unknown leaves,
lrp_modules.py:321-341) and `propagate_relevance(module, relevance_input, relevance_output, lrp_method,
lrp_params)` returns a tuple with the arity of `relevance_input` (:157-170).  `module.input` must hold the
layer input, as the reference's `save_input_hook` leaves it (lrp_wrapper.py:24-25).  Tensors are NCHW on the
device, as in the reference; layouts are converted at this boundary.

Built for the layers VGG16 exercises (Conv2d 3x3/pad 1, MaxPool2d(2,2), ReLU) on square maps of 224/112/56/28/14
pixels; other shapes raise ValueError.  Linear/BatchNorm/Add/Flatten/Dropout rules (ResNet encoders only) are not
part of the hot path."""
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
        n, cin, h, w = x.shape
        cout = module.out_channels
        _require(h == w and h in _SIZES, f"lrpx Conv2d rule: square maps of {_SIZES} pixels only, got {h}x{w}")
        _require(r_out.shape == (n, cout, h, w), "relevance_output shape mismatch")
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
        ops.check_relevance(R)                                      # lrp_modules.py:154-155
        if relevance_input is not None and len(relevance_input) == 3:
            return R, relevance_input[1], relevance_input[2]
        if relevance_input is not None and len(relevance_input) == 2:
            return R, relevance_input[1]
        return (R,)


class Pool2d:
    """MaxPool2d(2,2) rule (lrp_modules.py:182-195): winner-take-all routing, first maximum wins."""

    def propagate_relevance(self, module, relevance_input, relevance_output, lrp_method, lrp_params=None):
        _require(isinstance(module, nn.MaxPool2d), "lrpx Pool2d rule: MaxPool2d only (AvgPool2d is not on the VGG16 path)")
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


def get_lrp_module(module):
    """type(module) -> rule object; ValueError for leaves the path does not know (lrp_modules.py:321-341)."""
    try:
        cls = {nn.ReLU: ReLU, nn.Conv2d: Conv2d, nn.MaxPool2d: Pool2d}[type(module)]
    except KeyError:
        raise ValueError("Layer type {} not known.".format(type(module)))
    return cls()
