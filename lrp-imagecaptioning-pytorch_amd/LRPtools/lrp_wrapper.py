"""Mirror of the reference's LRPtools/lrp_wrapper.py: `add_lrp(model)` + `model.compute_lrp(sample, target=...)`.

The reference installs a forward hook and a legacy backward hook on every leaf and lets autograd drive the
relevance pass (lrp_wrapper.py:37-87).  Here `add_lrp` validates the same leaf -> rule table, packs the weights
once and attaches a `compute_lrp` with the same signature and return value; the pass itself is the fused HIP
chain `lrpx_vgg16_forward` + `lrpx_vgg16_relevance` (no autograd, no per-word forward with dead wgrad work).
Improvement over the reference: `add_lrp` is idempotent (the reference stacks hooks on every call, which
multiplies its cost without changing the result)."""
import torch
import torch.nn as nn

from . import lrp_modules
from .. import _lib, ops
from .._lib import check, ptr, stream_ptr


class SequentialPresetA(object):
    def __init__(self):
        self.lrp_params = {"alpha": 1., "beta": 0., "ignore_bias": True}      # lrp_wrapper.py:7-12


VGG16_FEATURES = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]


def _leaves(model):
    return [m for m in model.modules() if len(list(m.children())) == 0]


def _match_vgg16(leaves):
    """True if the leaves are conv3x3+ReLU / MaxPool2d(2,2) in the VGG16 'D' order without the last pool
    (models/vgg.py:62-81, models/gridTDmodel.py:34)."""
    i, cin = 0, 3
    for v in VGG16_FEATURES:
        if i >= len(leaves):
            return False
        m = leaves[i]
        if v == 'M':
            if not isinstance(m, nn.MaxPool2d):
                return False
            i += 1
        else:
            if not (isinstance(m, nn.Conv2d) and m.in_channels == cin and m.out_channels == v and
                    m.kernel_size == (3, 3) and m.padding == (1, 1) and i + 1 < len(leaves) and
                    isinstance(leaves[i + 1], nn.ReLU)):
                return False
            cin = v
            i += 2
    return i == len(leaves)


def add_lrp(model):
    """Attach `model.compute_lrp`.  Leaf -> rule as in lrp_wrapper.py:42-56 (Conv2d/MaxPool2d: alpha_beta,
    ReLU: identity); unknown leaves raise ValueError like `get_lrp_module`."""
    leaves = _leaves(model)
    for m in leaves:
        lrp_modules.get_lrp_module(m)                     # ValueError("Layer type ... not known.")
    if not _match_vgg16(leaves):
        raise ValueError("lrpx add_lrp: only the VGG16 encoder (features[0:-1]) is built as a fused chain; "
                         "use the per-layer rules in lrp_modules for other stacks")
    convs = [m for m in leaves if isinstance(m, nn.Conv2d)]
    dev = convs[0].weight.device
    if dev.type != "cuda":
        raise _lib.LrpxError("add_lrp: the model must live on the GPU (no CPU path)")
    zeros = lambda c: torch.zeros(c, device=dev)
    ctx = ops.Vgg16([c.weight.detach().float() for c in convs],
                    [c.bias.detach().float() if c.bias is not None else zeros(c.out_channels) for c in convs])
    model._lrpx_ctx = ctx
    model.compute_lrp = lambda sample, **kwargs: compute_lrp(model, sample, **kwargs)


def compute_lrp(model, sample, target=None, return_output=False, rectify_logits=False, explain_diff=False):
    """lrp_wrapper.compute_lrp (:63-87): relevance of `target` (N,512,14,14) propagated to `sample` (N,3,224,224).
    Like the reference, the result ACCUMULATES in `sample.grad` across calls on the same tensor (autograd's
    `.grad` semantics, :66-82) and the returned tensor is a clone of that running sum."""
    ctx = model._lrpx_ctx
    lib = _lib.load()
    if sample.requires_grad is False:
        sample.requires_grad = True
    x = sample.detach().to(torch.float32).contiguous()
    feats = ctx.forward(x)                                       # (N,196,512) NHWC
    n = x.shape[0]
    if target is None:
        raise ValueError("compute_lrp needs `target` (the reference passes the anchor to backward(), :80)")
    t_nhwc = ops.nchw_to_nhwc(target.detach().to(torch.float32))
    r = ctx.relevance(t_nhwc, None)
    if sample.grad is None:
        sample.grad = r
    else:
        check(lib.lrpx_accumulate(ptr(sample.grad), ptr(r), r.numel(), stream_ptr()))
    ops.check_relevance(sample.grad, finite=True, nonzero=True)  # `assert sample.grad.sum()!=0` (:81)
    output = sample.grad.clone().detach()
    if return_output:
        logits = ops.nhwc_to_nchw(feats.contiguous(), 512, 14, 14)
        return output, logits
    return output
