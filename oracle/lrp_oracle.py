"""CPU oracle for the LRP hot path  —  TEST INFRASTRUCTURE ONLY.

A plain PyTorch-CPU fp32 restatement of the reference's relevance-propagation algorithm
(SunJiamei/LRP-imagecaptioning-pytorch), written from the reference's formulas with every
function citing the reference file:line it follows.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import this module; the product path
(`lrp-imagecaptioning-pytorch_amd/`) never does and fails loudly without its HIP library.

Pinning: the reference ships no tests or golden vectors, so this oracle is pinned
against outputs of the reference itself, generated in the build container by
`tests/golden/make_golden.py` (imports /root/reference read-only) and committed under
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks every one of them.

All tensors are torch CPU float32.  Shapes follow the reference (batch-1 explainers); the VGG
part additionally accepts N relevance maps that share one image's activations.
"""
import math

import torch
import torch.nn.functional as F

EPSILON = 0.01        # LRPtools/utils.py:10
Z_EPSILON = 1e-7      # LRPtools/utils.py:11

# models/vgg.py:81 cfgs['D'] minus the last pool (models/gridTDmodel.py:34)
VGG16_CFG = [64, 64, 'M', 128, 128, 'M', 256, 256, 256, 'M', 512, 512, 512, 'M', 512, 512, 512]


def _t(x):
    return x if isinstance(x, torch.Tensor) else torch.from_numpy(x)


def state_to_torch(sd):
    return {k: _t(v).float() for k, v in sd.items()}


# ----------------------------------------------------------------------------------------------
# LRPtools numerics
# ----------------------------------------------------------------------------------------------
def safe_divide(num, den):
    """LRPtools/utils.py:16-18 — only exact zeros are stabilised."""
    return num / (den + Z_EPSILON * (den == 0).float())


def eps_stabilise(z):
    """models/gridTDmodel.py:757-759 — z + eps*sign(z); exact zeros -> eps."""
    zt = EPSILON * z.sign() + z
    return torch.where(zt == 0, torch.full_like(zt, EPSILON), zt)


def eps_identity(r, x, z):
    """`lrp_linear_eps` with weight=eye (models/gridTDmodel.py:744-765 as used at :1051-1069):
    R_in = (x / z~) * r, elementwise; broadcasting allowed."""
    return (x / eps_stabilise(z)) * r


def eps_dense(r, x, z, w):
    """`lrp_linear_eps` with a real weight (models/gridTDmodel.py:744-765):
    R_in[i] = sum_o (W[o,i]*x[i] / z~[o]) * r[o]  ==  x * (W^T (r / z~)).
    r,z: (..., out); x: (..., in); w: (out, in)."""
    return x * ((r / eps_stabilise(z)) @ w)


def conv_alpha1beta0(x, w, r_out):
    """LRPtools/lrp_modules.py:124-150 + utils.lrp_backward :21-31 with alpha=1, beta=0,
    ignore_bias=True (lrp_wrapper.py:7-12): Z = conv(x+,w+) + conv(x-,w-);
    S = safe_divide(R, Z); R_in = x+ * convT(S, w+) + x- * convT(S, w-).
    x: (1 or N, Cin, H, W) activations; r_out: (N, Cout, H, W)."""
    wp, wn = w.clamp(min=0), w.clamp(max=0)
    xp, xn = x.clamp(min=0), x.clamp(max=0)
    z = F.conv2d(xp, wp, padding=1) + F.conv2d(xn, wn, padding=1)
    s = safe_divide(r_out, z)
    cp = F.conv_transpose2d(s, wp, padding=1)
    cn = F.conv_transpose2d(s, wn, padding=1)
    return xp * cp + xn * cn


def maxpool_rule(x, r_out):
    """LRPtools/lrp_modules.py:182-195 for MaxPool2d(2,2): Z = maxpool(x); S = safe_divide(R, Z);
    R_in = x * dZ/dx(S)  (winner-take-all; first maximum in row-major window order wins)."""
    z, idx = F.max_pool2d(x, 2, 2, return_indices=True)
    s = safe_divide(r_out, z)
    if idx.shape[0] != s.shape[0]:
        idx = idx.expand(s.shape[0], -1, -1, -1)
    g = F.max_unpool2d(s, idx, 2, 2, output_size=x.shape[-2:])
    return x * g


def avgpool_rule(x, r_out, kernel_size, stride=None, padding=0, ceil_mode=False, count_include_pad=True,
                 divisor_override=None):
    """LRPtools/lrp_modules.py:172-195 `Pool2d` on an `nn.AvgPool2d` (clone at :176-177, table entry :327):
    Z = avgpool(x); S = safe_divide(R, Z); R_in = x * dZ/dx(S).  Written out as explicit window loops (no autograd): every
    output distributes S / divisor to the inputs of its (clipped) window; divisor = padded window size clipped at H + pad
    (count_include_pad), the clipped window size, or divisor_override - ATen's avg_pool2d definition.  (The reference's clone
    copies stride / padding / count_include_pad / ceil_mode only: a module's divisor_override never reaches its rule.)"""
    kh, kw = (kernel_size, kernel_size) if isinstance(kernel_size, int) else kernel_size
    stride = kernel_size if stride is None else stride
    sh, sw = (stride, stride) if isinstance(stride, int) else stride
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    n, c, h, w = x.shape
    oh, ow = r_out.shape[2], r_out.shape[3]
    z = torch.zeros_like(r_out)
    wins = []
    for i in range(oh):
        for j in range(ow):
            h0, w0 = i * sh - ph, j * sw - pw
            h1, w1 = min(h0 + kh, h + ph), min(w0 + kw, w + pw)
            pool = (h1 - h0) * (w1 - w0)
            h0, w0, h1, w1 = max(h0, 0), max(w0, 0), min(h1, h), min(w1, w)
            div = divisor_override or (pool if count_include_pad else (h1 - h0) * (w1 - w0))
            wins.append((i, j, h0, h1, w0, w1, float(div)))
            acc = torch.zeros_like(z[:, :, i, j])
            for hh in range(h0, h1):              # row-major, one addend at a time: the order of ATen's window loop
                for ww in range(w0, w1):
                    acc = acc + x[:, :, hh, ww]
            z[:, :, i, j] = acc / div
    s = safe_divide(r_out, z)
    g = torch.zeros_like(x)
    for i, j, h0, h1, w0, w1, div in wins:
        g[:, :, h0:h1, w0:w1] += (s[:, :, i, j] / div)[:, :, None, None]
    return x * g


# ----------------------------------------------------------------------------------------------
# rules of the layers VGG16 never reaches (ResNet encoders)
# ----------------------------------------------------------------------------------------------
RELEVANCE_RECT = -1e-6   # LRPtools/utils.py:14


def linear_eps_rule(x, w, r_out, bias=None):
    """LRPtools/lrp_modules.py:9-37 `Linear.propagate_relevance` (epsilon rule): exact-zero inputs are nudged to -1e-6
    IN PLACE on the saved input (:14, quirk h); Z = x W^T; ignore_bias (the preset, lrp_wrapper.py:7-12): Z += eps sign Z,
    exact zeros -> eps; otherwise Z += bias; R = x * ((R_out / Z) W).  Returns (R, x_after)."""
    x = x.clone()
    x[x == 0] = RELEVANCE_RECT
    z = x @ w.t()
    if bias is None:
        z = z + EPSILON * z.sign()
        z = torch.where(z == 0, torch.full_like(z, EPSILON), z)
    else:
        z = z + bias
    return x * ((r_out / z) @ w), x


def batchnorm_rule(x, r_out, gamma, beta, mean, var, eps):
    """LRPtools/lrp_modules.py:197-246 `BatchNorm2d` / `BatchNorm1d` (method != 'identity'):
    w = gamma / sqrt(var + eps), b = beta - mean gamma / sqrt(var + eps), both indexed [:, None, None];
    R = safe_divide(|x w|, |x w| + |b|) * R_out.  The indexing is the 2-d one in BOTH classes, so a (N,C) input of
    BatchNorm1d broadcasts to a (C,N,C) result: torch's broadcasting reproduces it here as it does in the reference."""
    w = (gamma / torch.sqrt(var + eps))[:, None, None]
    b = (beta - (mean * gamma) / torch.sqrt(var + eps))[:, None, None]
    xw = x * w
    return safe_divide(xw.abs(), xw.abs() + b.abs()) * r_out


def add_rule(x1, x2, r_out):
    """LRPtools/lrp_modules.py:256-280 `Add`: proportional split R_k = R x_k / (x1 + x2 + eps sign(x1 + x2)); NaNs (0/0) -> 0;
    entries whose sum is exactly zero additionally get R/2 each."""
    out = x1 + x2
    half = torch.zeros_like(out).masked_fill_(out == 0, 0.5)
    out = out + EPSILON * out.sign()
    r1, r2 = r_out * x1 / out, r_out * x2 / out
    r1[r1 != r1] = 0
    r2[r2 != r2] = 0
    return r1 + r_out * half, r2 + r_out * half


# ----------------------------------------------------------------------------------------------
# VGG16 encoder: forward trace and LRP (lrp_wrapper.compute_lrp, LRPtools/lrp_wrapper.py:63-87)
# ----------------------------------------------------------------------------------------------
def vgg_layers():
    layers, idx, cin = [], 0, 3
    for v in VGG16_CFG:
        if v == 'M':
            layers.append(('pool', idx, cin, cin))
            idx += 1
        else:
            layers.append(('conv', idx, cin, v))
            idx += 2
            cin = v
    return layers


def vgg_forward(sd, img, prefix="img_encoder.encoder."):
    """models/gridTDmodel.py:40-43 `Encoder.forward` for vgg16 (`features[0:-1]`, :33-34).
    Returns (features (B,512,h,w), avg (B,512), [input of every conv/pool layer])."""
    x = img
    saved = []
    for kind, idx, cin, cout in vgg_layers():
        saved.append(x)
        if kind == 'conv':
            x = F.relu(F.conv2d(x, sd[f"{prefix}{idx}.weight"], sd[f"{prefix}{idx}.bias"], padding=1))
        else:
            x = F.max_pool2d(x, 2, 2)
    return x, x.mean(dim=(2, 3)), saved


def vgg_lrp(sd, saved, r_feat, prefix="img_encoder.encoder."):
    """Relevance from the encoder output back to the pixels: reverse walk over the leaf modules
    (LRPtools/lrp_wrapper.py:37-59: Conv2d -> alpha_beta, ReLU -> identity, MaxPool2d ->
    Pool2d rule).  `saved[l]` holds ONE image's activations (1,C,H,W); r_feat is (N,512,h,w)."""
    r = r_feat
    for (kind, idx, cin, cout), x in zip(reversed(vgg_layers()), reversed(saved)):
        if kind == 'conv':
            r = conv_alpha1beta0(x, sd[f"{prefix}{idx}.weight"], r)
        else:
            r = maxpool_rule(x, r)
    return r


def vgg_guided_backprop(sd, saved, g_feat, prefix="img_encoder.encoder."):
    """models/gridTDmodel.py:1677-1723: gradient of the encoder output w.r.t. the image with the
    guided ReLU rule g_in = clamp(g_out, min=0) * [relu_out > 0] at every ReLU."""
    layers = vgg_layers()
    g = g_feat
    for li in range(len(layers) - 1, -1, -1):
        kind, idx, cin, cout = layers[li]
        x = saved[li]
        if kind == 'conv':
            w, b = sd[f"{prefix}{idx}.weight"], sd[f"{prefix}{idx}.bias"]
            y = F.relu(F.conv2d(x, w, b, padding=1))
            g = g.clamp(min=0) * (y > 0).float()
            g = F.conv_transpose2d(g, w, padding=1)
        else:
            z, pidx = F.max_pool2d(x, 2, 2, return_indices=True)
            if pidx.shape[0] != g.shape[0]:
                pidx = pidx.expand(g.shape[0], -1, -1, -1)
            g = F.max_unpool2d(g, pidx, 2, 2, output_size=x.shape[-2:])
    return g


def vgg_gradient(sd, saved, g_feat, prefix="img_encoder.encoder."):
    """models/gridTDmodel.py:1507-1521 `ExplainGridTDGradient.explain_cnn`: the plain autograd gradient of the encoder
    output w.r.t. the image (`image_feature.backward(d_img_feature)`): g_in = g_out * [relu_out > 0] at every ReLU,
    arg-max routing through the pools."""
    layers = vgg_layers()
    g = g_feat
    for li in range(len(layers) - 1, -1, -1):
        kind, idx, cin, cout = layers[li]
        x = saved[li]
        if kind == 'conv':
            w, b = sd[f"{prefix}{idx}.weight"], sd[f"{prefix}{idx}.bias"]
            y = F.relu(F.conv2d(x, w, b, padding=1))
            g = g * (y > 0).to(g.dtype)
            g = F.conv_transpose2d(g, w, padding=1)
        else:
            z, pidx = F.max_pool2d(x, 2, 2, return_indices=True)
            if pidx.shape[0] != g.shape[0]:
                pidx = pidx.expand(g.shape[0], -1, -1, -1)
            g = F.max_unpool2d(g, pidx, 2, 2, output_size=x.shape[-2:])
    return g


def grad_cam(features, grads):
    """models/gridTDmodel.py:1760-1771 `ExplainGridTDGradCam.grad_cam`: features, grads (1,C,h,w) ->
    relu(sum_c features_c * mean_hw(grads_c)) / (max|.| + 1e-6), flattened to (h*w,)."""
    weights = grads.mean(dim=(2, 3), keepdim=True)
    cam = (features * weights).sum(dim=(0, 1)).clamp(min=0)
    return (cam / (cam.abs().max() + 1e-6)).reshape(-1)


# ----------------------------------------------------------------------------------------------
# gridTD decoder (adaptive attention + two LSTMs)
# ----------------------------------------------------------------------------------------------
def pyramid_expand(cam, upscale=16):
    """`skimage.transform.pyramid_expand(cam, upscale, multichannel=False)` as the Guided-Grad-CAM explainers call it
    (models/gridTDmodel.py:1826, models/aoamodel.py:1741).  scikit-image is an un-vendored dependency of the reference
    (no pinned version; 0.16.x matches its PyTorch 1.4) and is NOT installed here: this restates its published algorithm -
    `resize(order=1, mode='reflect', anti_aliasing=False)` = sampling at (o + 0.5) / upscale - 0.5 with linear interpolation
    and numpy-style 'reflect' borders, then `_smooth` = scipy.ndimage.gaussian_filter(sigma = 2 upscale / 6, mode='reflect')
    (scipy IS here and is called as skimage calls it).  PARITY UNPINNED for this one function: no output of skimage itself
    could be generated.  cam: (h, w) tensor -> (h*upscale, w*upscale) float32 tensor."""
    import numpy as np
    from scipy import ndimage as ndi
    a = np.asarray(cam, dtype=np.float64)
    h, w = a.shape
    pad = np.pad(a, 1, mode="reflect")                      # index -1 -> 1, h -> h - 2
    rr = (np.arange(h * upscale) + 0.5) / upscale - 0.5
    cc = (np.arange(w * upscale) + 0.5) / upscale - 0.5
    r0, c0 = np.floor(rr).astype(int), np.floor(cc).astype(int)
    dr, dc = (rr - r0)[:, None], (cc - c0)[None, :]
    R0, C0 = r0[:, None] + 1, c0[None, :] + 1               # indices into the padded image
    top = (1 - dc) * pad[R0, C0] + dc * pad[R0, C0 + 1]
    bot = (1 - dc) * pad[R0 + 1, C0] + dc * pad[R0 + 1, C0 + 1]
    resized = (1 - dr) * top + dr * bot
    out = ndi.gaussian_filter(resized, 2 * upscale / 6.0, mode="reflect", cval=0)
    return torch.from_numpy(out).float()


def guided_grad_cam(features, grads, guided_map, upscale=16):
    """ExplainGridTDGuidedGradCam.explain_cnn (models/gridTDmodel.py:1814-1836) after the guided backward: the guided
    gradient times the expanded Grad-CAM map of the same decoder gradient (:1825-1829).  `grad_cam` there returns the
    (h, w) map (:1799-1810).  features, grads: (1,C,h,w); guided_map: (1,3,H,W)."""
    h, w = features.shape[-2:]
    cam = grad_cam(features, grads).view(h, w)
    return guided_map * pyramid_expand(cam, upscale).expand_as(guided_map)


def _lstm_cell(x, h, c, w_ih, w_hh, bias):
    """models/gridTDmodel.py:773-797: returns (h', c', z_g, sigmoid(z_i), sigmoid(z_f))."""
    z = w_ih @ x + w_hh @ h + bias
    zi, zf, zg, zo = z.chunk(4)
    i, f = torch.sigmoid(zi), torch.sigmoid(zf)
    c2 = f * c + i * torch.tanh(zg)
    h2 = torch.sigmoid(zo) * torch.tanh(c2)
    return h2, c2, zg, i, f


def _adaptive_attention(sd, V, h, s):
    """models/gridTDmodel.py:71-103 for one sample.  V: (P,H) projected pixels, h,s: (H,)."""
    wv, bv = sd["AdaAttention.W_v_proj.weight"], sd["AdaAttention.W_v_proj.bias"]
    ws, bs = sd["AdaAttention.W_s_proj.weight"], sd["AdaAttention.W_s_proj.bias"]
    wg, wh = sd["AdaAttention.W_g_proj.weight"], sd["AdaAttention.w_h.weight"]
    img_proj = V @ wv.t() + bv                         # (P,P)
    h_proj = wg @ h                                    # (P,)
    z = torch.tanh(img_proj + h_proj.unsqueeze(1)) @ wh.t()   # (P,1): ht_proj expanded over columns (:82)
    alpha = torch.softmax(z, dim=0)                    # (P,1)
    ctx = (V * alpha).sum(0)
    att_s = wh @ torch.tanh(ws @ s + bs + h_proj)      # (1,)
    alpha_hat = torch.softmax(torch.cat([z.squeeze(1), att_s]), dim=0)
    beta = alpha_hat[-1]
    ctx_hat = beta * s + (1 - beta) * ctx
    return ctx_hat, ctx, alpha.squeeze(1), beta


def gridtd_trace(sd, features, avg, caption, model_bias=False, gate_h_new=False):
    """models/gridTDmodel.py:933-1012 `get_hidden_parameters` for one image with a given caption
    (`caption[0]` = <start>; T = len(caption)-1 words).  features (512,h,w), avg (512,).
    Keeps the reference's quirks: LanguageLSTM adds bias_ih twice (:789); the sentinel gate uses
    h_{t-1} (:982).  `model_bias=True` gives the MODEL's own forward instead
    (`predict_next_word` :137-144 through nn.LSTMCell: bias_ih + bias_hh), used for decoding.
    `gate_h_new=True` feeds the sentinel gate the NEW h1 as `sample_lrp`/`forwardlrp_context` do (:610, :672)."""
    Hd = sd["fc.weight"].shape[1]
    C, hh, ww = features.shape
    P = hh * ww
    T = len(caption) - 1
    F_pix = features.reshape(C, P).t().contiguous()                     # (P,C)
    w_proj = sd["img_projector.weight"].reshape(Hd, C)
    proj_pre = F_pix @ w_proj.t() + sd["img_projector.bias"]            # (P,H)
    Vp = F.relu(proj_pre)
    glob_pre = sd["global_img_feature_proj.weight"] @ avg + sd["global_img_feature_proj.bias"]
    glob = F.relu(glob_pre)
    a_wi, a_wh = sd["AdaLSTM.lstm_cell.weight_ih"], sd["AdaLSTM.lstm_cell.weight_hh"]
    a_b = sd["AdaLSTM.lstm_cell.bias_hh"] + sd["AdaLSTM.lstm_cell.bias_ih"]
    l_wi, l_wh = sd["LanguageLSTM.weight_ih"], sd["LanguageLSTM.weight_hh"]
    l_b = sd["LanguageLSTM.bias_ih"] + sd["LanguageLSTM.bias_ih"]       # quirk (a)
    if model_bias:
        l_b = sd["LanguageLSTM.bias_ih"] + sd["LanguageLSTM.bias_hh"]
    E = sd["embedding.weight"].shape[1]
    tr = dict(T=T, P=P, F_pix=F_pix, avg=avg, proj_pre=proj_pre, Vp=Vp, glob_pre=glob_pre, glob=glob,
              caption=list(int(c) for c in caption))
    z = lambda *s: torch.zeros(*s)
    for k in ("h1", "c1", "h2", "c2"):
        tr[k] = z(T + 1, Hd)
    for k in ("g1", "i1", "f1", "g2", "i2", "f2", "s", "ctx", "ctx_hat"):
        tr[k] = z(T, Hd)
    tr["x1"], tr["x2"] = z(T, 2 * E + Hd), z(T, 2 * Hd)
    tr["alpha"], tr["beta"] = z(T, P), z(T)
    tr["pred"] = z(T, sd["fc.weight"].shape[0])
    for t in range(T):
        emb = sd["embedding.weight"][caption[t]]
        x1 = torch.cat([tr["h2"][t], glob, emb])
        h1, c1, g1, i1, f1 = _lstm_cell(x1, tr["h1"][t], tr["c1"][t], a_wi, a_wh, a_b)
        gate = torch.sigmoid(sd["AdaLSTM.x_gate.weight"] @ x1 + sd["AdaLSTM.x_gate.bias"]
                             + sd["AdaLSTM.h_gate.weight"] @ (h1 if gate_h_new else tr["h1"][t])
                             + sd["AdaLSTM.h_gate.bias"])
        s = gate * torch.tanh(c1)
        ctx_hat, ctx, alpha, beta = _adaptive_attention(sd, Vp, h1, s)
        x2 = torch.cat([ctx_hat, h1])
        h2, c2, g2, i2, f2 = _lstm_cell(x2, tr["h2"][t], tr["c2"][t], l_wi, l_wh, l_b)
        tr["pred"][t] = sd["fc.weight"] @ (ctx_hat + h2) + sd["fc.bias"]
        tr["x1"][t], tr["x2"][t] = x1, x2
        tr["h1"][t + 1], tr["c1"][t + 1], tr["g1"][t], tr["i1"][t], tr["f1"][t] = h1, c1, g1, i1, f1
        tr["h2"][t + 1], tr["c2"][t + 1], tr["g2"][t], tr["i2"][t], tr["f2"][t] = h2, c2, g2, i2, f2
        tr["s"][t], tr["ctx"][t], tr["ctx_hat"][t] = s, ctx, ctx_hat
        tr["alpha"][t], tr["beta"][t] = alpha, beta
    return tr


def gridtd_greedy_caption(sd, features, avg, max_words, start_id, end_id):
    """Greedy decoding with the explainer's forward (models/gridTDmodel.py:799-853
    `forward_greedy`): argmax of the raw logits, stop at <end>.  Returns token ids incl. <start>."""
    cap = [start_id]
    for _ in range(max_words):
        tr = gridtd_trace(sd, features, avg, cap + [0])
        nxt = int(torch.argmax(tr["pred"][-1]))
        if nxt == end_id:
            break
        cap.append(nxt)
    return cap


def gridtd_model_greedy(sd, img, max_cap_length, start_id, end_id):
    """models/gridTDmodel.py:480-520 `greedy_search` for one image: argmax of log_softmax per step,
    tokens after the first <end> are forced to 0 (:503-504); returns `seqs_temp[0]`
    (max_cap_length ids incl. <start>)."""
    feats, avg, _ = vgg_forward(sd, img)
    seq, unfinished = [start_id], True
    for step in range(max_cap_length - 1):
        tr = gridtd_trace(sd, feats[0], avg[0], seq + [0], model_bias=True)
        top = int(torch.argmax(torch.log_softmax(tr["pred"][-1], dim=-1)))
        unfinished = unfinished and (top != end_id)
        seq.append(top if unfinished else 0)
    return seq


def beam_search(step_logits, vocab, beam_size, max_cap_length, start_id, end_id):
    """models/gridTDmodel.py:400-478 `beam_search` (AoA: the same code on its own step), host logic only.
    step_logits(seqs) -> (len(seqs), vocab) scores of the NEXT word for every live sequence (teacher-forced re-run of the
    model forward on each prefix: same values as carrying the LSTM states along the beams, :460-465).
    `beam_idx = top_words / vocab_size` (:444) is the integer division of the pinned PyTorch 1.4.
    Returns `seq` (:469-472): the best complete sequence, else the first live one cut at 20 tokens."""
    seqs = [[start_id] for _ in range(beam_size)]
    cum = torch.zeros(beam_size)
    complete, complete_scores = [], []
    n_live = beam_size
    for step in range(max_cap_length):
        scores = cum[:n_live].unsqueeze(1) + torch.log_softmax(step_logits(seqs), dim=-1)
        if step == 0:
            top_sc, top = scores[0].topk(beam_size, -1, True, True)
        else:
            top_sc, top = scores.reshape(-1).topk(n_live, -1, True, True)
        beam_idx, nxt = (top // vocab).tolist(), (top % vocab).tolist()
        seqs = [seqs[b] + [w] for b, w in zip(beam_idx, nxt)]
        inc = [i for i, w in enumerate(nxt) if w != end_id]
        for i in sorted(set(range(len(nxt))) - set(inc)):
            complete.append(seqs[i])
            complete_scores.append(float(top_sc[i]))
        n_live -= len(nxt) - len(inc)
        if n_live == 0:
            break
        seqs = [seqs[i] for i in inc]
        cum = top_sc[inc]
    if complete:
        return complete[complete_scores.index(max(complete_scores))]
    return seqs[0][:20]


def gridtd_beam_caption(sd, img, beam_size, max_cap_length, word_map):
    """what `get_hidden_parameters` (models/gridTDmodel.py:935-937) explains: [<start>] + sen_idx of beam_search (:474)."""
    feats, avg, _ = vgg_forward(sd, img)
    start, end = word_map['<start>'], word_map['<end>']

    def step_logits(seqs):
        return torch.stack([gridtd_trace(sd, feats[0], avg[0], s + [0], model_bias=True)["pred"][-1] for s in seqs])
    seq = beam_search(step_logits, sd["fc.weight"].shape[0], beam_size, max_cap_length, start, end)
    drop = {word_map[k] for k in ('<start>', '<end>', '<unk>', '<pad>')}
    return [start] + [w for w in seq if w not in drop], seq


def aoa_beam_caption(sd, img, beam_size, max_cap_length, word_map, num_head=8):
    feats, _, _ = vgg_forward(sd, img)
    F_pix = feats[0].reshape(feats.shape[1], -1).t().contiguous()
    start, end = word_map['<start>'], word_map['<end>']

    def step_logits(seqs):
        return torch.stack([aoa_trace(sd, F_pix, s + [0], num_head=num_head, grad=True)["pred"][-1] for s in seqs])
    seq = beam_search(step_logits, sd["fc.weight"].shape[0], beam_size, max_cap_length, start, end)
    drop = {word_map[k] for k in ('<start>', '<end>', '<unk>', '<pad>')}
    return [start] + [w for w in seq if w not in drop], seq


def normalize_relevance(x):
    """LRPtools/utils.py:55-64 with temperature = 1: x / max|x| + 1 (an all-zero row stays 0 -> weight 1)."""
    v = x.abs().max()
    v = v if v != 0 else torch.ones(())
    return x / v + 1


def gridtd_lrp_weights(sd, pred, h2, ctx_hat, skip_ids):
    """models/gridTDmodel.py:548-577 `get_lrp_weight_step` for one row: the predicted word's logit is redistributed
    to h2 + ctx_hat through fc (epsilon rule, one-hot relevance) and split between the two summands; both relevance
    vectors are normalised to weights around 1.  Words in `skip_ids` (stop words and specials) get weights of 1."""
    k = int(torch.argmax(pred))
    Hd = h2.shape[0]
    if k in skip_ids:
        return torch.ones(Hd), torch.ones(Hd)
    hc = h2 + ctx_hat
    r_hc = (sd["fc.weight"][k] * hc / eps_stabilise(pred[k])) * pred[k]
    r_h2 = eps_identity(r_hc, h2, hc)
    r_ctx = eps_identity(r_hc, ctx_hat, hc)
    return normalize_relevance(r_ctx), normalize_relevance(r_h2)


def gridtd_sample_lrp(sd, img, max_length, start_id, end_id, skip_ids):
    """models/gridTDmodel.py:631-702 `sample_lrp` (greedy) for one image: every step's logits are recomputed from the
    LRP-reweighted fc input `ctx_hat * w_ctx + w_h2 * h2` before the next word is taken.
    Returns (seq (max_length,) int64, seq_logprobs (max_length,))."""
    feats, avg, _ = vgg_forward(sd, img)
    toks, unfinished = [start_id], True
    seq, lps = [], []
    for t in range(max_length):
        tr = gridtd_trace(sd, feats[0], avg[0], toks + [0], model_bias=True, gate_h_new=True)
        h2, ctx_hat = tr["h2"][t + 1], tr["ctx_hat"][t]
        w_ctx, w_h2 = gridtd_lrp_weights(sd, tr["pred"][t], h2, ctx_hat, skip_ids)
        wp = sd["fc.weight"] @ (ctx_hat * w_ctx + w_h2 * h2) + sd["fc.bias"]
        lsm = torch.log_softmax(wp, dim=-1)
        it = int(torch.argmax(lsm))
        lps.append(float(lsm[it]))
        unfinished = unfinished and (it != end_id)
        it = it if unfinished else 0
        seq.append(it)
        toks.append(it)
    return seq, lps


def gridtd_forwardlrp_context(sd, img, caption, length, skip_ids):
    """models/gridTDmodel.py:579-630 `forwardlrp_context` for one image, forward values only: teacher-forced model forward
    (bias_ih + bias_hh, sentinel gate on the new h1 :617); per step the raw scores and the scores recomputed from the
    LRP-reweighted fc input (:625-627).  caption: ids incl. <start>; length = caption_length - 1 steps.
    Returns (predictions (L,V), weighted_predictions (L,V))."""
    feats, avg, _ = vgg_forward(sd, img)
    tr = gridtd_trace(sd, feats[0], avg[0], list(caption[:length]) + [0], model_bias=True, gate_h_new=True)
    preds, wpreds = [], []
    for t in range(length):
        h2, ctx_hat = tr["h2"][t + 1], tr["ctx_hat"][t]
        w_ctx, w_h2 = gridtd_lrp_weights(sd, tr["pred"][t], h2, ctx_hat, skip_ids)
        preds.append(tr["pred"][t])
        wpreds.append(sd["fc.weight"] @ (ctx_hat * w_ctx + w_h2 * h2) + sd["fc.bias"])
    return torch.stack(preds), torch.stack(wpreds)


def aoa_forwardlrp_context(sd, img, caption, length, skip_ids, num_head=8):
    """models/aoamodel.py:628-677 `forwardlrp_context` for one image, forward values only; `get_lrp_weight_step`
    (:597-626) sees the RAW scores here (`sample_lrp` hands it their log-softmax).  Dropout = identity (eval)."""
    feats, _, _ = vgg_forward(sd, img)
    C, hh, ww = feats.shape[1:]
    F_pix = feats[0].reshape(C, hh * ww).t().contiguous()
    tr = aoa_trace(sd, F_pix, list(caption[:length]) + [0], num_head=num_head, grad=True)   # model forward: bias_ih + bias_hh
    Hd = sd["fc.weight"].shape[1]
    preds, wpreds = [], []
    for t in range(length):
        h, c_aoa, pred = tr["h"][t + 1], tr["c_aoa"][t], tr["pred"][t]
        k = int(torch.argmax(pred))
        if k in skip_ids:
            w_c, w_h = torch.ones(Hd), torch.ones(Hd)
        else:
            hc = h + c_aoa
            r_hc = (sd["fc.weight"][k] * hc / eps_stabilise(pred[k])) * pred[k]
            w_h = normalize_relevance(eps_identity(r_hc, h, hc))
            w_c = normalize_relevance(eps_identity(r_hc, c_aoa, hc))
        preds.append(pred)
        wpreds.append(sd["fc.weight"] @ (w_c * c_aoa + h * w_h) + sd["fc.bias"])
    return torch.stack(preds), torch.stack(wpreds)


def gridtd_explain_wordt(sd, tr, t):
    """models/gridTDmodel.py:1014-1135 `explain_caption_wordt`.  Returns
    (r_feat (P,C) relevance of the encoder features, r_words (t+1,))."""
    Hd = tr["h1"].shape[1]
    E = (tr["x1"].shape[1] - Hd) // 2
    P = tr["P"]
    k = tr["caption"][t + 1]
    wg1 = torch.cat([sd["AdaLSTM.lstm_cell.weight_ih"].chunk(4, 0)[2],
                     sd["AdaLSTM.lstm_cell.weight_hh"].chunk(4, 0)[2]], dim=1)      # (H, 2E+2H)
    wg2 = torch.cat([sd["LanguageLSTM.weight_ih"].chunk(4, 0)[2],
                     sd["LanguageLSTM.weight_hh"].chunk(4, 0)[2]], dim=1)           # (H, 3H)
    xh1 = torch.cat([tr["x1"], tr["h1"][:-1]], dim=1)
    xh2 = torch.cat([tr["x2"], tr["h2"][:-1]], dim=1)
    n = t + 1
    r_h1, r_c1 = torch.zeros(n + 1, Hd), torch.zeros(n + 1, Hd)
    r_h2, r_c2 = torch.zeros(n + 1, Hd), torch.zeros(n + 1, Hd)
    r_ctx_hat = torch.zeros(n, Hd)
    r_glob = torch.zeros(E)
    r_emb = torch.zeros(n, E)
    r_proj = torch.zeros(P, Hd)
    # fc: one-hot relevance = the target logit (:1033-1034, :1047-1050)
    hc = tr["h2"][t + 1] + tr["ctx_hat"][t]
    logit = tr["pred"][t][k]
    r_hc = (sd["fc.weight"][k] * hc / eps_stabilise(logit)) * logit
    r_h2[t + 1] = eps_identity(r_hc, tr["h2"][t + 1], hc)                 # :1051-1054
    r_ctx_hat[t] = eps_identity(r_hc, tr["ctx_hat"][t], hc)               # :1056-1059
    for i in range(t, -1, -1):
        r_c2[i + 1] = r_c2[i + 1] + r_h2[i + 1]                           # :1061
        r_g2 = eps_identity(r_c2[i + 1], tr["i2"][i] * torch.tanh(tr["g2"][i]), tr["c2"][i + 1])
        r_c2[i] = eps_identity(r_c2[i + 1], tr["f2"][i] * tr["c2"][i], tr["c2"][i + 1])
        r_xh2 = eps_dense(r_g2, xh2[i], tr["g2"][i], wg2)                 # :1070-1073
        r_h2[i] = r_xh2[2 * Hd:]
        r_h1[i + 1] = r_xh2[Hd:2 * Hd]
        r_ctx_hat[i] = r_ctx_hat[i] + r_xh2[:Hd]
        r_s = eps_identity(r_ctx_hat[i], tr["beta"][i] * tr["s"][i], tr["ctx_hat"][i])
        r_ctx = eps_identity(r_ctx_hat[i], tr["ctx"][i] * (1 - tr["beta"][i]), tr["ctx_hat"][i])
        # pixel spread (:1091-1095), all pixels at once
        r_proj = r_proj + eps_identity(r_ctx.unsqueeze(0), tr["Vp"] * tr["alpha"][i].unsqueeze(1),
                                       tr["ctx"][i].unsqueeze(0))
        r_c1[i + 1] = r_c1[i + 1] + r_s
        r_c1[i + 1] = r_c1[i + 1] + r_h1[i + 1]
        r_g1 = eps_identity(r_c1[i + 1], tr["i1"][i] * torch.tanh(tr["g1"][i]), tr["c1"][i + 1])
        r_c1[i] = eps_identity(r_c1[i + 1], tr["f1"][i] * tr["c1"][i], tr["c1"][i + 1])
        r_xh1 = eps_dense(r_g1, xh1[i], tr["g1"][i], wg1)                 # :1106-1109
        r_h1[i] = r_xh1[2 * E + Hd:]            # dead: overwritten by :1075 next iteration (quirk c)
        r_h2[i] = r_h2[i] + r_xh1[:Hd]
        r_glob = r_glob + r_xh1[Hd:Hd + E]
        r_emb[i] = r_xh1[Hd + E:Hd + 2 * E]
    r_avg = eps_dense(r_glob, tr["avg"], tr["glob_pre"], sd["global_img_feature_proj.weight"])
    w_proj = sd["img_projector.weight"].reshape(Hd, -1)
    r_feat = eps_identity(r_avg.unsqueeze(0), tr["F_pix"] / P, tr["avg"].unsqueeze(0)) \
        + eps_dense(r_proj, tr["F_pix"], tr["proj_pre"], w_proj)          # :1120-1128
    r_words = r_emb.sum(-1)
    m = r_words.abs().max()
    if m > 0:
        r_words = r_words / m
    return r_feat, r_words


# ----------------------------------------------------------------------------------------------
# gridTD guided-backprop decoder (models/gridTDmodel.py:1214-1422 trace, :1588-1675 backward)
# ----------------------------------------------------------------------------------------------
def _lstm_cell_full(x, h, c, w_ih, w_hh, bias):
    """models/gridTDmodel.py:1248-1274: LSTM cell returning all four gate activations."""
    z = w_ih @ x + w_hh @ h + bias
    zi, zf, zg, zo = z.chunk(4)
    i, f, g, o = torch.sigmoid(zi), torch.sigmoid(zf), torch.tanh(zg), torch.sigmoid(zo)
    c2 = f * c + i * g
    return o * torch.tanh(c2), c2, i, f, g, o


def gridtd_grad_trace(sd, features, avg, caption):
    """models/gridTDmodel.py:1323-1422: the gradient explainers' trace.  Unlike the LRP trace the
    LanguageLSTM bias is correct here (bias_ih + bias_hh, :1262-1274) and all four gate
    activations plus the sentinel gate are kept."""
    Hd = sd["fc.weight"].shape[1]
    C, hh, ww = features.shape
    P = hh * ww
    T = len(caption) - 1
    F_pix = features.reshape(C, P).t().contiguous()
    w_proj = sd["img_projector.weight"].reshape(Hd, C)
    Vp = F.relu(F_pix @ w_proj.t() + sd["img_projector.bias"])
    glob = F.relu(sd["global_img_feature_proj.weight"] @ avg + sd["global_img_feature_proj.bias"])
    a_wi, a_wh = sd["AdaLSTM.lstm_cell.weight_ih"], sd["AdaLSTM.lstm_cell.weight_hh"]
    a_b = sd["AdaLSTM.lstm_cell.bias_hh"] + sd["AdaLSTM.lstm_cell.bias_ih"]
    l_wi, l_wh = sd["LanguageLSTM.weight_ih"], sd["LanguageLSTM.weight_hh"]
    l_b = sd["LanguageLSTM.bias_hh"] + sd["LanguageLSTM.bias_ih"]
    tr = dict(T=T, P=P, F_pix=F_pix, Vp=Vp, glob=glob, caption=list(int(c) for c in caption))
    z = lambda *s: torch.zeros(*s)
    for k in ("h1", "c1", "h2", "c2"):
        tr[k] = z(T + 1, Hd)
    for k in ("i1", "f1", "g1", "o1", "i2", "f2", "g2", "o2", "s", "sen_gate", "ctx", "ctx_hat"):
        tr[k] = z(T, Hd)
    tr["alpha"], tr["beta"] = z(T, P), z(T)
    tr["pred"] = z(T, sd["fc.weight"].shape[0])
    for t in range(T):
        emb = sd["embedding.weight"][caption[t]]
        x1 = torch.cat([tr["h2"][t], glob, emb])
        h1, c1, i1, f1, g1, o1 = _lstm_cell_full(x1, tr["h1"][t], tr["c1"][t], a_wi, a_wh, a_b)
        gate = torch.sigmoid(sd["AdaLSTM.x_gate.weight"] @ x1 + sd["AdaLSTM.x_gate.bias"]
                             + sd["AdaLSTM.h_gate.weight"] @ tr["h1"][t] + sd["AdaLSTM.h_gate.bias"])
        s = gate * torch.tanh(c1)
        ctx_hat, ctx, alpha, beta = _adaptive_attention(sd, Vp, h1, s)
        x2 = torch.cat([ctx_hat, h1])
        h2, c2, i2, f2, g2, o2 = _lstm_cell_full(x2, tr["h2"][t], tr["c2"][t], l_wi, l_wh, l_b)
        tr["pred"][t] = sd["fc.weight"] @ (ctx_hat + h2) + sd["fc.bias"]
        tr["h1"][t + 1], tr["c1"][t + 1], tr["h2"][t + 1], tr["c2"][t + 1] = h1, c1, h2, c2
        for k, v in dict(i1=i1, f1=f1, g1=g1, o1=o1, i2=i2, f2=f2, g2=g2, o2=o2, s=s, sen_gate=gate, ctx=ctx,
                         ctx_hat=ctx_hat, alpha=alpha).items():
            tr[k][t] = v
        tr["beta"][t] = beta
    return tr


def gridtd_guided_wordt(sd, tr, t, mask_features=True):
    """models/gridTDmodel.py:1588-1675 `ExplainiGridTDGuidedGradient.explain_caption_wordt`: hand-written BPTT with
    alpha / beta treated as constants.  Quirks kept: `d_h1[i]` from the AdaLSTM recurrence is overwritten by
    :1646 (no h1 recurrence), the sentinel gate's own inputs get no gradient, and the two projector 'ReLU gates'
    test `< 0` on post-ReLU tensors (:1663, :1665), i.e. never fire; only `features <= 0` (:1674) does.
    mask_features=False gives the parent class's `ExplainGridTDGradient.explain_caption_wordt` (:1424-1505): the same
    BPTT (same quirks) without any of the three gates.
    Returns (d_feat (P,C), r_words (t+1,))."""
    Hd = tr["h1"].shape[1]
    P = tr["P"]
    k = tr["caption"][t + 1]
    E = sd["embedding.weight"].shape[1]
    a_wi, a_wh = sd["AdaLSTM.lstm_cell.weight_ih"], sd["AdaLSTM.lstm_cell.weight_hh"]
    l_wi, l_wh = sd["LanguageLSTM.weight_ih"], sd["LanguageLSTM.weight_hh"]
    n = t + 1
    d_h1, d_c1 = torch.zeros(n + 1, Hd), torch.zeros(n + 1, Hd)
    d_h2, d_c2 = torch.zeros(n + 1, Hd), torch.zeros(n + 1, Hd)
    d_ch = torch.zeros(n, Hd)
    d_glob, d_emb, d_proj = torch.zeros(E), torch.zeros(n, E), torch.zeros(P, Hd)
    d_hc = sd["fc.weight"][k].clone()
    d_ch[t] = d_hc
    d_h2[t + 1] = d_hc
    for i in range(t, -1, -1):
        tc2 = torch.tanh(tr["c2"][i + 1])
        d_o2a = d_h2[i + 1] * tc2
        d_c2[i + 1] = d_c2[i + 1] + d_h2[i + 1] * tr["o2"][i] * (1 - tc2 ** 2)
        d_f2a = d_c2[i + 1] * tr["c2"][i]
        d_c2[i] = d_c2[i + 1] * tr["f2"][i]
        d_i2a, d_g2a = d_c2[i + 1] * tr["g2"][i], d_c2[i + 1] * tr["i2"][i]
        gates2 = torch.cat([d_i2a * tr["i2"][i] * (1 - tr["i2"][i]), d_f2a * tr["f2"][i] * (1 - tr["f2"][i]),
                            d_g2a * (1 - tr["g2"][i] ** 2), d_o2a * tr["o2"][i] * (1 - tr["o2"][i])])
        d_h2[i] = gates2 @ l_wh
        d_x2 = gates2 @ l_wi
        d_ch[i] = d_ch[i] + d_x2[:Hd]
        d_ctx = d_ch[i] * (1 - tr["beta"][i])
        d_proj = d_proj + d_ctx.unsqueeze(0) * tr["alpha"][i].unsqueeze(1)
        d_s = d_ch[i] * tr["beta"][i]
        tc1 = torch.tanh(tr["c1"][i + 1])
        d_c1[i + 1] = d_c1[i + 1] + d_s * tr["sen_gate"][i] * (1 - tc1 ** 2)
        d_h1[i + 1] = d_x2[Hd:]
        d_o1a = d_h1[i + 1] * tc1
        d_c1[i + 1] = d_c1[i + 1] + d_h1[i + 1] * tr["o1"][i] * (1 - tc1 ** 2)
        d_f1a = d_c1[i + 1] * tr["c1"][i]
        d_c1[i] = d_c1[i + 1] * tr["f1"][i]
        d_i1a, d_g1a = d_c1[i + 1] * tr["g1"][i], d_c1[i + 1] * tr["i1"][i]
        gates1 = torch.cat([d_i1a * tr["i1"][i] * (1 - tr["i1"][i]), d_f1a * tr["f1"][i] * (1 - tr["f1"][i]),
                            d_g1a * (1 - tr["g1"][i] ** 2), d_o1a * tr["o1"][i] * (1 - tr["o1"][i])])
        d_h1[i] = gates1 @ a_wh                 # dead: overwritten by :1646 in the next iteration
        d_x1 = gates1 @ a_wi
        d_glob = d_glob + d_x1[Hd:Hd + E]
        d_emb[i] = d_x1[Hd + E:]
        d_h2[i] = d_h2[i] + d_x1[:Hd]
    d_avg = d_glob @ sd["global_img_feature_proj.weight"]
    w_proj = sd["img_projector.weight"].reshape(Hd, -1)
    d_feat = d_avg.unsqueeze(0) / P + d_proj @ w_proj
    if mask_features:
        d_feat = d_feat * (tr["F_pix"] > 0).float()      # :1674
    r_words = d_emb.sum(-1)
    m = r_words.abs().max()
    if m > 0:
        r_words = r_words / m
    return d_feat, r_words


def gridtd_guided_explain_caption(sd, img, caption, words=None, return_feat=False):
    """`ExplainiGridTDGuidedGradient.explain_caption` (:1525-1539 inherited + explain_cnn :1702-1723); the image
    gradient is zeroed after every word (:1717), so these maps are NOT running sums."""
    feats, avg, saved = vgg_forward(sd, img)
    tr = gridtd_grad_trace(sd, feats[0], avg[0], caption)
    maps, rws, dfs = [], [], []
    for t in (range(tr["T"]) if words is None else words):
        d_feat, r_words = gridtd_guided_wordt(sd, tr, t)
        d_feat = pix_to_nchw(d_feat, feats.shape[-2:])
        dfs.append(d_feat)
        maps.append(vgg_guided_backprop(sd, saved, d_feat))
        rws.append(r_words)
    if return_feat:
        return maps, rws, dfs, tr
    return maps, rws


def gridtd_gradient_explain_caption(sd, img, caption, words=None, return_feat=False, cam=False):
    """`ExplainGridTDGradient.explain_caption` (:1523-1537): plain decoder gradient + autograd through the encoder;
    cam=True: `ExplainGridTDGradCam` (:1752-1771), the per-word result is the (1, h*w) Grad-CAM heat map instead."""
    feats, avg, saved = vgg_forward(sd, img)
    tr = gridtd_grad_trace(sd, feats[0], avg[0], caption)
    maps, rws, dfs = [], [], []
    for t in (range(tr["T"]) if words is None else words):
        d_feat, r_words = gridtd_guided_wordt(sd, tr, t, mask_features=False)
        d_feat = pix_to_nchw(d_feat, feats.shape[-2:])
        dfs.append(d_feat)
        maps.append(grad_cam(feats, d_feat).unsqueeze(0) if cam else vgg_gradient(sd, saved, d_feat))
        rws.append(r_words)
    if return_feat:
        return maps, rws, dfs, tr
    return maps, rws


# ----------------------------------------------------------------------------------------------
# AoA decoder (multi-head attention on attention)
# ----------------------------------------------------------------------------------------------
def aoa_trace(sd, F_pix, caption, num_head=8, grad=False):
    """models/aoamodel.py:990-1062 `get_hidden_parameters` for one image.
    F_pix: (P,C) encoder features, pixel-major.  Quirk: bias_ih added twice (:873).
    grad=True: the trace of the gradient explainers (:1309-1376): the LSTM uses bias_ih + bias_hh (:1298) and the
    output gate is kept (tr["o"])."""
    Hd = sd["fc.weight"].shape[1]
    P, C = F_pix.shape
    T = len(caption) - 1
    dk = Hd // num_head
    w_proj = sd["img_projector.weight"].reshape(Hd, C)
    proj_pre = F_pix @ w_proj.t() + sd["img_projector.bias"]
    Vp = F.relu(proj_pre)
    glob = Vp.mean(0)
    key = Vp @ sd["decoder_k_proj.weight"].t() + sd["decoder_k_proj.bias"]
    value = Vp @ sd["decoder_v_proj.weight"].t() + sd["decoder_v_proj.bias"]
    l_wi, l_wh = sd["LanguageLSTM.weight_ih"], sd["LanguageLSTM.weight_hh"]
    l_b = sd["LanguageLSTM.bias_ih"] + (sd["LanguageLSTM.bias_hh"] if grad else sd["LanguageLSTM.bias_ih"])
    E = sd["embedding.weight"].shape[1]
    z = lambda *s: torch.zeros(*s)
    tr = dict(T=T, P=P, F_pix=F_pix, proj_pre=proj_pre, Vp=Vp, glob=glob, key=key, value=value, o=z(T, Hd),
              caption=list(int(c) for c in caption), num_head=num_head)
    tr["h"], tr["c"] = z(T + 1, Hd), z(T + 1, Hd)
    for k in ("g", "i", "f", "ctx", "c_aoa", "c_aoa_lin", "c_aoa_gate"):
        tr[k] = z(T, Hd)
    tr["x"] = z(T, E + Hd)
    tr["alpha"] = z(T, num_head, P)
    tr["pred"] = z(T, sd["fc.weight"].shape[0])
    kh = key.view(P, num_head, dk).transpose(0, 1)        # (heads,P,dk)
    vh = value.view(P, num_head, dk).transpose(0, 1)
    for t in range(T):
        emb = sd["embedding.weight"][caption[t]]
        x = torch.cat([emb, glob])
        h, c, g, i, f = _lstm_cell(x, tr["h"][t], tr["c"][t], l_wi, l_wh, l_b)
        tr["o"][t] = torch.sigmoid((l_wi @ x + l_wh @ tr["h"][t] + l_b).chunk(4)[3])
        q = sd["decoder_multihead_attention.q_proj.weight"] @ h + sd["decoder_multihead_attention.q_proj.bias"]
        qh = q.view(num_head, 1, dk)
        scores = (qh @ kh.transpose(1, 2)) / math.sqrt(dk)           # (heads,1,P)  aoamodel.py:77-85
        alpha = torch.softmax(scores, dim=-1)
        ctx = (alpha @ vh).reshape(Hd)
        gate = sd["decoder_aoa_linear_gate.weight"] @ h + sd["decoder_aoa_linear_gate.bias"]
        lin = sd["decoder_aoa_linear.weight"] @ ctx + sd["decoder_aoa_linear.bias"]
        c_aoa = torch.sigmoid(gate) * lin
        tr["pred"][t] = sd["fc.weight"] @ (c_aoa + h) + sd["fc.bias"]
        tr["x"][t] = x
        tr["h"][t + 1], tr["c"][t + 1], tr["g"][t], tr["i"][t], tr["f"][t] = h, c, g, i, f
        tr["ctx"][t], tr["c_aoa"][t], tr["c_aoa_lin"][t], tr["c_aoa_gate"][t] = ctx, c_aoa, lin, gate
        tr["alpha"][t] = alpha.squeeze(1)
    return tr


def aoa_sample_lrp(sd, img, max_length, start_id, end_id, skip_ids, num_head=8):
    """models/aoamodel.py:679-745 `AOAModel.sample_lrp` (greedy) for one image.  Unlike the gridTD version (:548-577 of
    gridTDmodel.py) `get_lrp_weight_step` (:597-626) is handed the LOG-SOFTMAX of the scores (:721-723): the relevance
    that goes back through fc is log p(k), stabilised with log p(k) itself.
    Returns (seq (max_length,), seq_logprobs (max_length,))."""
    feats, _, _ = vgg_forward(sd, img)
    C, hh, ww = feats.shape[1:]
    F_pix = feats[0].reshape(C, hh * ww).t().contiguous()
    Hd = sd["fc.weight"].shape[1]
    toks, unfinished = [start_id], True
    seq, lps = [], []
    for t in range(max_length):
        tr = aoa_trace(sd, F_pix, toks + [0], num_head=num_head, grad=True)       # model forward: bias_ih + bias_hh
        h, c_aoa = tr["h"][t + 1], tr["c_aoa"][t]
        lsm0 = torch.log_softmax(tr["pred"][t], dim=-1)
        k = int(torch.argmax(lsm0))
        if k in skip_ids:
            w_c, w_h = torch.ones(Hd), torch.ones(Hd)
        else:
            hc = h + c_aoa
            r_hc = (sd["fc.weight"][k] * hc / eps_stabilise(lsm0[k])) * lsm0[k]
            w_h = normalize_relevance(eps_identity(r_hc, h, hc))
            w_c = normalize_relevance(eps_identity(r_hc, c_aoa, hc))
        lsm = torch.log_softmax(sd["fc.weight"] @ (c_aoa * w_c + w_h * h) + sd["fc.bias"], dim=-1)
        it = int(torch.argmax(lsm))
        lps.append(float(lsm[it]))
        unfinished = unfinished and (it != end_id)
        it = it if unfinished else 0
        seq.append(it)
        toks.append(it)
    return seq, lps


def aoa_explain_wordt(sd, tr, t, head_idx):
    """models/aoamodel.py:1064-1156 `explain_caption_wordt` (+ `lrp_mha` :812-862).
    Returns (r_feat (P,C), r_words (t+1,))."""
    Hd = tr["h"].shape[1]
    E = tr["x"].shape[1] - Hd
    P, nh = tr["P"], tr["num_head"]
    dk = Hd // nh
    k = tr["caption"][t + 1]
    wg = torch.cat([sd["LanguageLSTM.weight_ih"].chunk(4, 0)[2],
                    sd["LanguageLSTM.weight_hh"].chunk(4, 0)[2]], dim=1)            # (H, E+2H)
    xh = torch.cat([tr["x"], tr["h"][:-1]], dim=1)
    n = t + 1
    r_h = torch.zeros(n + 1, Hd)
    r_emb = torch.zeros(n, E)
    r_glob = torch.zeros(Hd)
    hc = tr["h"][t + 1] + tr["c_aoa"][t]
    logit = tr["pred"][t][k]
    r_hc = (sd["fc.weight"][k] * hc / eps_stabilise(logit)) * logit                 # :1092-1095
    r_h[t + 1] = eps_identity(r_hc, tr["h"][t + 1], hc)
    r_caoa = eps_identity(r_hc, tr["c_aoa"][t], hc)
    r_ctx = eps_dense(r_caoa, tr["ctx"][t], tr["c_aoa_lin"][t], sd["decoder_aoa_linear.weight"])  # :1107
    # lrp_mha: only head `head_idx` receives relevance (:848-860)
    r_value = torch.zeros(P, Hd)
    sl = slice(head_idx * dk, (head_idx + 1) * dk)
    r_value[:, sl] = eps_identity(r_ctx[sl].unsqueeze(0),
                                  tr["value"][:, sl] * tr["alpha"][t][head_idx].unsqueeze(1),
                                  tr["ctx"][t][sl].unsqueeze(0))
    for i in range(t, -1, -1):
        r_c = r_h[i + 1]                                                            # :1116 (assignment)
        r_g = eps_identity(r_c, tr["i"][i] * torch.tanh(tr["g"][i]), tr["c"][i + 1])
        r_xh = eps_dense(r_g, xh[i], tr["g"][i], wg)
        r_h[i] = r_xh[Hd + E:]
        r_emb[i] = r_xh[:E]
        r_glob = r_glob + r_xh[E:E + Hd]
    r_proj = eps_identity(r_glob.unsqueeze(0), tr["Vp"] / P, tr["glob"].unsqueeze(0)) \
        + eps_dense(r_value, tr["Vp"], tr["value"], sd["decoder_v_proj.weight"])    # :1136-1144
    w_proj = sd["img_projector.weight"].reshape(Hd, -1)
    r_feat = eps_dense(r_proj, tr["F_pix"], tr["proj_pre"], w_proj)                 # :1145-1148
    r_words = r_emb.sum(-1)
    m = r_words.abs().max()
    if m > 0:
        r_words = r_words / m
    return r_feat, r_words


# ----------------------------------------------------------------------------------------------
# whole-image drivers (models/gridTDmodel.py:1141-1156, models/aoamodel.py:1165-1181)
# ----------------------------------------------------------------------------------------------
def pix_to_nchw(r_feat, hw):
    """(P,C) -> (1,C,h,w): the reference's `.unsqueeze(0).transpose(1,2).view(...)` (:1133)."""
    P, C = r_feat.shape
    return r_feat.t().reshape(1, C, hw[0], hw[1])


def _accumulate(maps):
    """Quirk (i), LRPtools/lrp_wrapper.py:64-82: `compute_lrp` reads `sample.grad` of the SAME leaf
    tensor (`self.img`) on every word and never clears it (`model.zero_grad()` only touches
    parameters), so the map the reference returns for word t is the running sum of the per-word
    relevance maps of words 0..t.  Verified against the reference (tests/golden)."""
    out, run = [], None
    for m in maps:
        run = m if run is None else run + m
        out.append(run)
    return out


def gridtd_explain_caption(sd, img, caption, words=None, return_feat=False, accumulate=True):
    """One image (1,3,H,W) + teacher-forced caption -> ([T] x (1,3,H,W), [T] x (t+1,)).
    accumulate=True reproduces the reference's returned maps (running sums, see `_accumulate`);
    False gives the per-word maps."""
    feats, avg, saved = vgg_forward(sd, img)
    tr = gridtd_trace(sd, feats[0], avg[0], caption)
    maps, rws, rfs = [], [], []
    for t in (range(tr["T"]) if words is None else words):
        r_feat, r_words = gridtd_explain_wordt(sd, tr, t)
        r_feat = pix_to_nchw(r_feat, feats.shape[-2:])
        rfs.append(r_feat)
        maps.append(vgg_lrp(sd, saved, r_feat))
        rws.append(r_words)
    if accumulate:
        maps = _accumulate(maps)
    if return_feat:
        return maps, rws, rfs, tr
    return maps, rws


def aoa_explain_caption(sd, img, caption, head_idx, words=None, return_feat=False, accumulate=True):
    feats, avg, saved = vgg_forward(sd, img)
    C = feats.shape[1]
    F_pix = feats[0].reshape(C, -1).t().contiguous()
    tr = aoa_trace(sd, F_pix, caption)
    maps, rws, rfs = [], [], []
    for t in (range(tr["T"]) if words is None else words):
        r_feat, r_words = aoa_explain_wordt(sd, tr, t, head_idx)
        r_feat = pix_to_nchw(r_feat, feats.shape[-2:])
        rfs.append(r_feat)
        maps.append(vgg_lrp(sd, saved, r_feat))
        rws.append(r_words)
    if accumulate:
        maps = _accumulate(maps)
    if return_feat:
        return maps, rws, rfs, tr
    return maps, rws


# ----------------------------------------------------------------------------------------------
# AoA gradient family: models/aoamodel.py:1257-1776
# ----------------------------------------------------------------------------------------------
def aoa_gradient_wordt(sd, tr, t, head_idx):
    """models/aoamodel.py:1435-1499 `ExplainAOAGradient.explain_caption_wordt` (+ `gradient_mha` :1415-1433): BPTT
    through the language LSTM with the attention weights constant and only head `head_idx` passing gradient.
    Quirks kept: `d_global_img_feature = d_xt[i][E:]` is an assignment (:1487), so only the i = 0 step survives;
    the projector ReLU gets no derivative; `ExplainAOAGuidedGradient` inherits this method unchanged.
    `tr` must be a grad=True trace.  Returns (d_feat (P,C), r_words (t+1,))."""
    Hd = tr["h"].shape[1]
    E = tr["x"].shape[1] - Hd
    P, nh = tr["P"], tr["num_head"]
    dk = Hd // nh
    k = tr["caption"][t + 1]
    l_wi, l_wh = sd["LanguageLSTM.weight_ih"], sd["LanguageLSTM.weight_hh"]
    n = t + 1
    d_h, d_c = torch.zeros(n + 1, Hd), torch.zeros(n + 1, Hd)
    d_emb = torch.zeros(n, E)
    d_hc = sd["fc.weight"][k].clone()
    d_h[t + 1] = d_hc
    sg = torch.sigmoid(tr["c_aoa_gate"][t])
    d_A = d_hc * sg
    d_B = d_hc * tr["c_aoa_lin"][t] * (1 - sg) * sg
    d_ctx = d_A @ sd["decoder_aoa_linear.weight"]
    d_h[t + 1] = d_h[t + 1] + d_B @ sd["decoder_aoa_linear_gate.weight"]
    d_value = torch.zeros(P, Hd)
    sl = slice(head_idx * dk, (head_idx + 1) * dk)
    d_value[:, sl] = tr["alpha"][t, head_idx].unsqueeze(1) * d_ctx[sl].unsqueeze(0)
    d_glob = torch.zeros(Hd)
    for i in range(t, -1, -1):
        tc = torch.tanh(tr["c"][i + 1])
        d_oa = d_h[i + 1] * tc
        d_c[i + 1] = d_c[i + 1] + d_h[i + 1] * tr["o"][i] * (1 - tc ** 2)
        d_fa = d_c[i + 1] * tr["c"][i]
        d_c[i] = d_c[i + 1] * tr["f"][i]
        d_ia, d_ga = d_c[i + 1] * torch.tanh(tr["g"][i]), d_c[i + 1] * tr["i"][i]
        gt = torch.tanh(tr["g"][i])
        gates = torch.cat([d_ia * tr["i"][i] * (1 - tr["i"][i]), d_fa * tr["f"][i] * (1 - tr["f"][i]),
                           d_ga * (1 - gt ** 2), d_oa * tr["o"][i] * (1 - tr["o"][i])])
        d_h[i] = gates @ l_wh
        d_x = gates @ l_wi
        d_glob = d_x[E:]                      # assignment, :1487
        d_emb[i] = d_x[:E]
    d_proj = d_value @ sd["decoder_v_proj.weight"] + d_glob.unsqueeze(0) / P
    w_proj = sd["img_projector.weight"].reshape(Hd, -1)
    d_feat = d_proj @ w_proj
    r_words = d_emb.sum(-1)
    m = r_words.abs().max()
    if m > 0:
        r_words = r_words / m
    return d_feat, r_words


def aoa_gradient_explain_caption(sd, img, caption, head_idx, kind="gradient", num_head=8, return_feat=False):
    """`explain_caption` of ExplainAOAGradient (kind="gradient", :1501-1534), ExplainAOAGuidedGradient ("guided",
    :1621-1640: guided backprop through the encoder) and ExplainAOAGradCam ("gradcam", :1669-1689)."""
    feats, avg, saved = vgg_forward(sd, img)
    F_pix = feats[0].reshape(feats.shape[1], -1).t().contiguous()
    tr = aoa_trace(sd, F_pix, caption, num_head, grad=True)
    maps, rws, dfs = [], [], []
    for t in range(tr["T"]):
        d_feat, r_words = aoa_gradient_wordt(sd, tr, t, head_idx)
        d_feat = pix_to_nchw(d_feat, feats.shape[-2:])
        dfs.append(d_feat)
        if kind == "gradcam":
            maps.append(grad_cam(feats, d_feat).unsqueeze(0))
        elif kind == "guided":
            maps.append(vgg_guided_backprop(sd, saved, d_feat))
        else:
            maps.append(vgg_gradient(sd, saved, d_feat))
        rws.append(r_words)
    if return_feat:
        return maps, rws, dfs, tr
    return maps, rws


# ----------------------------------------------------------------------------------------------
# teacherforce_forward of the explainers (the evaluation experiments call it after every explanation: evaluation.py:266,437,702,767)
# ----------------------------------------------------------------------------------------------
def gridtd_teacherforce(sd, img, caption_with_start, gradient_family=False):
    """models/gridTDmodel.py:892-931 (LRP explainer; LanguageLSTM adds bias_ih twice, :789) / :1282-1321 (gradient family: correct
    bias, :1265): scores (len(caption_with_start), V) - step t reads token t, the last word included."""
    feats, avg, _ = vgg_forward(sd, img)
    cap = [int(c) for c in caption_with_start] + [0]          # one more step than words: the step that reads the last word
    return gridtd_trace(sd, feats[0], avg[0], cap, model_bias=gradient_family)["pred"]


def aoa_teacherforce(sd, img, caption_with_start, gradient_family=False, num_head=8):
    """models/aoamodel.py:952-988 (LRP explainer, bias quirk :873) / :1377-1413 (gradient family, :1298)"""
    feats, _, _ = vgg_forward(sd, img)
    F_pix = feats[0].reshape(feats.shape[1], -1).t().contiguous()
    cap = [int(c) for c in caption_with_start] + [0]
    return aoa_trace(sd, F_pix, cap, num_head, grad=gradient_family)["pred"]
