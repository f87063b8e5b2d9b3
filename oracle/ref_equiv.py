"""Reference-equivalent CPU mode  —  TEST / BASELINE INFRASTRUCTURE ONLY (see oracle/lrp_oracle.py).

A CPU PyTorch program that replays the reference's OP SEQUENCE for the LRP hot path, so that its
wall time is a fair stand-in for the reference on the GPU box's host cores (the reference's Python
cannot travel there).  Unlike `lrp_oracle` (closed forms, vectorised), this module deliberately
keeps the reference's structure:

  * CNN: for every word one full VGG16 forward + one autograd backward with a relevance hook on
    every leaf (LRPtools/lrp_wrapper.py:37-87).  The Conv2d hook builds positive / negative copies
    of the layer and runs two forward+backward passes through them — 8 conv-equivalents per layer
    per word, the beta branch included although beta = 0 (LRPtools/lrp_modules.py:124-150).
  * decoder: `lrp_linear_eps` materialises the (out,in) attribution matrix for every call, identity
    weights included, and is called per pixel / per time step (models/gridTDmodel.py:744-765,
    :1060-1128).

`bench.py` times `explain_words()` on a bounded sample and reports it as `cpu_baseline`
(kind "port").  `tests/test_ref_equiv.py` checks that it reproduces the oracle's numbers.
"""
import time

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import lrp_oracle as O

EPS = O.EPSILON


# ----------------------------------------------------------------------------------------------
# CNN side: hooks + autograd, as LRPtools does it
# ----------------------------------------------------------------------------------------------
def _keep_input(module, inp, out):
    module.saved_input = inp


def _divide_safely(num, den):
    return num / (den + O.Z_EPSILON * (den == 0).float())


class _SignedConvPair(nn.Module):
    """two bias-free copies of a conv, one with the positive and one with the negative weights;
    `flip=False`: pos(x+) + neg(x-)   `flip=True`: pos(x-) + neg(x+)   (lrp_modules.py:56-114)"""

    def __init__(self, conv, flip):
        super().__init__()
        kw = dict(stride=conv.stride, padding=conv.padding, dilation=conv.dilation, groups=conv.groups)
        self.pos = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, **kw)
        self.neg = nn.Conv2d(conv.in_channels, conv.out_channels, conv.kernel_size, **kw)
        self.pos.weight = nn.Parameter(conv.weight.data.clone().clamp(min=0), requires_grad=False)
        self.neg.weight = nn.Parameter(conv.weight.data.clone().clamp(max=0), requires_grad=False)
        self.pos.bias = None
        self.neg.bias = None
        self.flip = flip

    def forward(self, x):
        lo, hi = torch.clamp(x, max=0), torch.clamp(x, min=0)
        return self.pos(lo) + self.neg(hi) if self.flip else self.pos(hi) + self.neg(lo)


def _forward_backward_relevance(x, layer, r_out):
    """utils.lrp_backward (LRPtools/utils.py:21-31): z = layer(x); s = r/z; x * d(z.s)/dx"""
    r_out = r_out.clone().detach()
    with torch.enable_grad():
        z = layer(x)
        s = _divide_safely(r_out, z)
        z.backward(s)
        return x * x.grad


def _conv_hook(module, grad_in, grad_out):
    x0 = module.saved_input[0]
    a_net, b_net = _SignedConvPair(module, False), _SignedConvPair(module, True)
    with torch.enable_grad():
        x = x0.clone().detach().requires_grad_(True)
        r_a = 1.0 * _forward_backward_relevance(x, a_net, grad_out[0])
        x.grad.detach_()
        x.grad.zero_()
        r_b = 0.0 * _forward_backward_relevance(x, b_net, grad_out[0])     # beta = 0, still computed
        r = r_a - r_b
    assert not torch.isnan(r.sum()) and not torch.isinf(r.sum())
    return (r,) + tuple(grad_in[1:])


def _pool_hook(module, grad_in, grad_out):
    x0 = module.saved_input[0]
    clone = nn.MaxPool2d(module.kernel_size, stride=module.stride, padding=module.padding)
    with torch.enable_grad():
        x = x0.clone().detach().requires_grad_(True)
        z = clone(x)
        s = _divide_safely(grad_out[0].clone().detach(), z)
        z.backward(s)
        r = x * x.grad
    assert not torch.isnan(r.sum())
    return (r,)


def _relu_hook(module, grad_in, grad_out):
    r = grad_out[0].clone().detach()
    assert not torch.isnan(r.sum())
    return (grad_out[0],)


def build_vgg(sd, prefix="img_encoder.encoder."):
    """nn.Sequential with the reference's leaf structure (conv, ReLU(inplace), pool) and hooks installed."""
    mods = []
    for kind, idx, cin, cout in O.vgg_layers():
        if kind == "conv":
            c = nn.Conv2d(cin, cout, 3, padding=1)
            c.weight.data = sd[f"{prefix}{idx}.weight"].clone()
            c.bias.data = sd[f"{prefix}{idx}.bias"].clone()
            mods += [c, nn.ReLU(inplace=True)]
        else:
            mods.append(nn.MaxPool2d(2, 2))
    net = nn.Sequential(*mods).eval()
    import warnings
    warnings.filterwarnings("ignore", message=".*non-full backward hook.*")
    for m in net:
        m.register_forward_hook(_keep_input)
        hook = _conv_hook if isinstance(m, nn.Conv2d) else (_pool_hook if isinstance(m, nn.MaxPool2d) else _relu_hook)
        m.register_backward_hook(hook)      # the legacy hook API, as lrp_wrapper.py:55 uses
    return net


def compute_lrp(net, sample, target):
    """lrp_wrapper.compute_lrp (LRPtools/lrp_wrapper.py:63-87); note `sample.grad` accumulates over calls."""
    if not sample.requires_grad:
        sample.requires_grad = True
    out = net(sample)
    net.zero_grad()
    sample.retain_grad()
    out.backward(target, retain_graph=True)
    assert sample.grad.sum() != 0
    res = sample.grad.clone().detach()
    net.zero_grad()
    return res


# ----------------------------------------------------------------------------------------------
# decoder side: materialising epsilon rule + per-pixel loops
# ----------------------------------------------------------------------------------------------
def eps_rule_materialised(r_out, x, z, w):
    """lrp_linear_eps (models/gridTDmodel.py:744-765): builds the (out,in) attribution, transposes, divides,
    multiplies and reduces — also when `w` is an identity matrix."""
    attribution = w * x
    zs = EPS * z.sign() + z
    zs.masked_fill_(zs == 0, EPS)
    norm = attribution.transpose(0, 1) / zs
    return torch.sum(norm * r_out, dim=1)


def gridtd_wordt_loops(sd, tr, t):
    """explain_caption_wordt (models/gridTDmodel.py:1014-1135) with the reference's loops and identity matrices."""
    Hd = tr["h1"].shape[1]
    E = (tr["x1"].shape[1] - Hd) // 2
    P, C = tr["F_pix"].shape
    k = tr["caption"][t + 1]
    eye = lambda n: torch.eye(n)
    wg1 = torch.cat([sd["AdaLSTM.lstm_cell.weight_ih"].chunk(4, 0)[2], sd["AdaLSTM.lstm_cell.weight_hh"].chunk(4, 0)[2]], 1)
    wg2 = torch.cat([sd["LanguageLSTM.weight_ih"].chunk(4, 0)[2], sd["LanguageLSTM.weight_hh"].chunk(4, 0)[2]], 1)
    n = t + 1
    xh1 = torch.cat([tr["x1"][:n], tr["h1"][:n]], 1)
    xh2 = torch.cat([tr["x2"][:n], tr["h2"][:n]], 1)
    word_rel = torch.zeros(1, sd["fc.weight"].shape[0])
    word_rel[0, k] = tr["pred"][t][k]
    r_h1, r_c1 = torch.zeros(n + 1, Hd), torch.zeros(n + 1, Hd)
    r_h2, r_c2 = torch.zeros(n + 1, Hd), torch.zeros(n + 1, Hd)
    r_ch = torch.zeros(n, Hd)
    r_glob, r_emb = torch.zeros(E), torch.zeros(n, E)
    r_feat, r_proj = torch.zeros(P, C), torch.zeros(P, Hd)
    hc = tr["h2"][t + 1] + tr["ctx_hat"][t]
    r_hc = eps_rule_materialised(word_rel, hc, tr["pred"][t], sd["fc.weight"])
    r_h2[t + 1] = eps_rule_materialised(r_hc, tr["h2"][t + 1], hc, eye(Hd))
    r_ch[t] = eps_rule_materialised(r_hc, tr["ctx_hat"][t], hc, eye(Hd))
    for i in range(t, -1, -1):
        r_c2[i + 1] = r_c2[i + 1] + r_h2[i + 1]
        r_g2 = eps_rule_materialised(r_c2[i + 1], tr["i2"][i] * torch.tanh(tr["g2"][i]), tr["c2"][i + 1], eye(Hd))
        r_c2[i] = eps_rule_materialised(r_c2[i + 1], tr["f2"][i] * tr["c2"][i], tr["c2"][i + 1], eye(Hd))
        r_x = eps_rule_materialised(r_g2, xh2[i], tr["g2"][i], wg2)
        r_h2[i], r_h1[i + 1] = r_x[2 * Hd:], r_x[Hd:2 * Hd]
        r_ch[i] += r_x[:Hd]
        r_s = eps_rule_materialised(r_ch[i], tr["beta"][i] * tr["s"][i], tr["ctx_hat"][i], eye(Hd))
        r_cx = eps_rule_materialised(r_ch[i], tr["ctx"][i] * (1 - tr["beta"][i]), tr["ctx_hat"][i], eye(Hd))
        for p in range(P):
            r_proj[p] += eps_rule_materialised(r_cx, tr["Vp"][p] * tr["alpha"][i][p], tr["ctx"][i], eye(Hd))
        r_c1[i + 1] += r_s
        r_c1[i + 1] += r_h1[i + 1]
        r_g1 = eps_rule_materialised(r_c1[i + 1], tr["i1"][i] * torch.tanh(tr["g1"][i]), tr["c1"][i + 1], eye(Hd))
        r_c1[i] = eps_rule_materialised(r_c1[i + 1], tr["f1"][i] * tr["c1"][i], tr["c1"][i + 1], eye(Hd))
        r_x1 = eps_rule_materialised(r_g1, xh1[i], tr["g1"][i], wg1)
        r_h1[i] = r_x1[2 * E + Hd:]
        r_h2[i] += r_x1[:Hd]
        r_glob = r_glob + r_x1[Hd:Hd + E]
        r_emb[i] = r_x1[Hd + E:Hd + 2 * E]
    r_avg = eps_rule_materialised(r_glob, tr["avg"], tr["glob_pre"], sd["global_img_feature_proj.weight"])
    w_proj = sd["img_projector.weight"].reshape(Hd, C)
    for p in range(P):
        r_feat[p] = eps_rule_materialised(r_avg, tr["F_pix"][p] / P, tr["avg"], eye(C))
        r_feat[p] = r_feat[p] + eps_rule_materialised(r_proj[p], tr["F_pix"][p], tr["proj_pre"][p], w_proj)
    r_words = r_emb.sum(-1)
    m = r_words.abs().max()
    if m > 0:
        r_words = r_words / m
    return r_feat, r_words


def explain_words(sd, img, caption, words):
    """explain_caption (models/gridTDmodel.py:1141-1156) restricted to `words`; returns
    (maps, r_words, seconds_trace, seconds_words)."""
    t0 = time.time()
    net = build_vgg(sd)
    with torch.no_grad():
        feats, avg, _ = O.vgg_forward(sd, img)
        tr = O.gridtd_trace(sd, feats[0], avg[0], caption)
    t_trace = time.time() - t0
    sample = img.clone()
    maps, rws = [], []
    t0 = time.time()
    for t in words:
        with torch.no_grad():
            r_feat, r_w = gridtd_wordt_loops(sd, tr, t)
        maps.append(compute_lrp(net, sample, O.pix_to_nchw(r_feat, feats.shape[-2:])))
        rws.append(r_w)
    return maps, rws, t_trace, time.time() - t0
