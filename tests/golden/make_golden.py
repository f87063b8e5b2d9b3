#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation (read-only import from
/root/reference) on seeded synthetic inputs.  Run in the build container only:

    python tests/golden/make_golden.py [--only layers,gridtd,aoa,greedy,guided]

The reference never travels to the GPU box; the .npz files written next to this script do.
Harness-only shims (never shipped): inert stubs for modules the reference imports at module
level but that are absent here (torchvision, skimage, nltk), `.cuda()` mapped to identity
(the reference hard-codes it), and file/PIL/beam-search steps replaced by in-memory equivalents.
`vgg16(pretrained=True)` would download weights; it is rebound to the random-init constructor and
the weights are then overwritten by this repo's seeded generator (weights.py).
"""
import argparse
import os
import sys
import tempfile
import types
import warnings

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
warnings.filterwarnings("ignore")


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Compose:
        def __init__(self, *a, **k):
            pass

    tv = mod("torchvision")
    tv.models = mod("torchvision.models")
    tv.models.utils = mod("torchvision.models.utils", load_state_dict_from_url=lambda *a, **k: {})
    tv.transforms = mod("torchvision.transforms", Compose=_Compose, Resize=_Compose, ToTensor=_Compose,
                        Normalize=_Compose)
    sk = mod("skimage")
    sk.transform = mod("skimage.transform")
    nl = mod("nltk")
    nl.corpus = mod("nltk.corpus", stopwords=types.SimpleNamespace(words=lambda lang: []))
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.empty_cache = lambda: None
    sys.path.insert(0, REF)


def load_pkg():
    import lrp_amd  # noqa: F401  (the shim mounts the package)
    from lrp_amd import weights
    return weights


def ref_vgg_patch():
    import models.vgg as vgg
    vgg.vgg16 = lambda pretrained=False, progress=True, **kw: vgg._vgg('vgg16', 'D', False, False, progress, **kw)


def to_torch_sd(sd):
    return {k: torch.from_numpy(v.copy()) for k, v in sd.items()}


def make_args(tmp, **kw):
    d = dict(embed_dim=512, hidden_dim=512, encoder='vgg16', weight='', save_path=tmp, dataset='synthetic',
             height=224, width=224, num_head=8)
    d.update(kw)
    return types.SimpleNamespace(**d)


def sub4(x):
    """strided subsample of a (1,3,H,W) map (every 4th pixel) for compact fixtures"""
    return x[..., ::4, ::4].contiguous()


def stats(x):
    x = x.double()
    return np.array([x.sum().item(), x.abs().max().item(), x.pow(2).sum().sqrt().item()], np.float64)


# ------------------------------------------------------------------------------------------------
def gen_layers(out):
    """Layer-level rules through the reference's own hook machinery (LRPtools/)."""
    from LRPtools import lrp_wrapper, lrp_modules
    from LRPtools import utils as lutil
    rs = np.random.RandomState(123)
    g = {}
    # --- Conv2d alpha1beta0 + ReLU identity + MaxPool2d through add_lrp/compute_lrp
    net = nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.ReLU(inplace=True),
                        nn.Conv2d(8, 8, 3, padding=1), nn.ReLU(inplace=True),
                        nn.MaxPool2d(2, 2),
                        nn.Conv2d(8, 16, 3, padding=1), nn.ReLU(inplace=True))
    for m in net:
        if isinstance(m, nn.Conv2d):
            m.weight.data = torch.from_numpy(rs.standard_normal(m.weight.shape).astype(np.float32) * 0.2)
            m.bias.data = torch.from_numpy(rs.standard_normal(m.bias.shape).astype(np.float32) * 0.1)
    net.eval()
    lrp_wrapper.add_lrp(net)
    x = torch.from_numpy(rs.standard_normal((1, 3, 16, 16)).astype(np.float32))
    x[0, :, 3, 4] = 0.0           # exact-zero input pixels
    target = torch.from_numpy(rs.standard_normal((1, 16, 8, 8)).astype(np.float32))
    target[0, 2] = 0.0            # a channel with zero relevance
    r = net.compute_lrp(x.clone(), target=target.clone())
    g["mini_w0"], g["mini_b0"] = net[0].weight.data.numpy(), net[0].bias.data.numpy()
    g["mini_w2"], g["mini_b2"] = net[2].weight.data.numpy(), net[2].bias.data.numpy()
    g["mini_w5"], g["mini_b5"] = net[5].weight.data.numpy(), net[5].bias.data.numpy()
    g["mini_x"], g["mini_target"], g["mini_r"] = x.numpy(), target.numpy(), r.numpy()
    # --- single rules called directly on crafted inputs
    conv = nn.Conv2d(4, 6, 3, padding=1)
    conv.weight.data = torch.from_numpy(rs.standard_normal(conv.weight.shape).astype(np.float32) * 0.3)
    xin = torch.from_numpy(rs.standard_normal((2, 4, 6, 6)).astype(np.float32))
    xin[0, :, :2] = 0.0           # a zero region: Z == 0 there -> safe_divide path
    conv.input = (xin,)
    rout = torch.from_numpy(rs.standard_normal((2, 6, 6, 6)).astype(np.float32))
    params = lrp_wrapper.SequentialPresetA().lrp_params
    rin = lrp_modules.Conv2d().propagate_relevance(conv, (xin, conv.weight), (rout,), 'alpha_beta', params)[0]
    g["conv_w"], g["conv_x"], g["conv_rout"], g["conv_rin"] = conv.weight.data.numpy(), xin.numpy(), rout.numpy(), rin.detach().numpy()
    pool = nn.MaxPool2d(2, 2)
    xp = torch.relu(torch.from_numpy(rs.standard_normal((2, 3, 8, 8)).astype(np.float32)))
    xp[0, 0, 0:2, 0:2] = 0.7      # a tie: all four equal
    xp[0, 1, 2:4, 2:4] = 0.0      # an all-zero window
    pool.input = (xp,)
    rp = torch.from_numpy(rs.standard_normal((2, 3, 4, 4)).astype(np.float32))
    rpin = lrp_modules.Pool2d().propagate_relevance(pool, None, (rp,), 'alpha_beta', params)[0]
    g["pool_x"], g["pool_rout"], g["pool_rin"] = xp.numpy(), rp.numpy(), rpin.detach().numpy()
    lin = nn.Linear(10, 7)
    rs_lin = np.random.RandomState(124)      # own stream: the draws below keep their values
    lin.weight.data = torch.from_numpy(rs_lin.uniform(-0.3, 0.3, (7, 10)).astype(np.float32))
    lin.bias.data = torch.from_numpy(rs_lin.uniform(-0.3, 0.3, (7,)).astype(np.float32))
    xl = torch.from_numpy(rs.standard_normal((3, 10)).astype(np.float32))
    xl[1, 4] = 0.0
    lin.input = (xl.clone(),)
    rl = torch.from_numpy(rs.standard_normal((3, 7)).astype(np.float32))
    rlin = lrp_modules.Linear().propagate_relevance(lin, (torch.zeros(3, 10), torch.zeros(10, 7)), (rl,), 'epsilon', params)[0]
    g["lin_w"], g["lin_x"], g["lin_rout"], g["lin_rin"] = lin.weight.data.numpy(), xl.numpy(), rl.numpy(), rlin.detach().numpy()
    # --- lrp_linear_eps (explainer helper), dense and eye, with exact zeros in z
    import models.gridTDmodel as gtd
    ex = gtd.ExplainGridTDAttention.__new__(gtd.ExplainGridTDAttention)
    w = torch.from_numpy(rs.standard_normal((12, 20)).astype(np.float32))
    xi = torch.from_numpy(rs.standard_normal((20,)).astype(np.float32))
    z = w @ xi
    z[3] = 0.0
    ro = torch.from_numpy(rs.standard_normal((12,)).astype(np.float32))
    g["eps_w"], g["eps_x"], g["eps_z"], g["eps_r"] = w.numpy(), xi.numpy(), z.numpy(), ro.numpy()
    g["eps_dense_out"] = ex.lrp_linear_eps(ro, xi, z, w).numpy()
    xe = torch.from_numpy(rs.standard_normal((12,)).astype(np.float32))
    g["eps_eye_x"] = xe.numpy()
    g["eps_eye_out"] = ex.lrp_linear_eps(ro, xe, z, torch.eye(12)).numpy()
    g["safe_div"] = lutil.safe_divide(ro, z).numpy()
    np.savez(os.path.join(out, "layers.npz"), **g)
    print("layers.npz:", {k: v.shape for k, v in g.items()})


# ------------------------------------------------------------------------------------------------
def _patch_explainer(ex, img, caption):
    ex.preprocess_img = lambda path: torch.from_numpy(img.copy())
    words = " ".join(f"w{c}" for c in caption[1:])
    ex.model.beam_search = lambda *a, **k: ([words], list(int(c) for c in caption[1:]))
    ex.visualize_explanations = lambda *a, **k: None
    ex.save_linguistic_explanation = lambda *a, **k: None


def gen_gridtd(out, weights, T=3, V=9586, seed=0):
    import models.gridTDmodel as gtd
    sd = weights.make_gridtd_state(seed=seed, vocab_size=V)
    model = gtd.GridTDModel(512, 512, V, 'vgg16')
    model.load_state_dict(to_torch_sd(sd))
    wm = weights.make_word_map(V)
    img = weights.make_images(seed, 1)
    cap = weights.make_captions(seed + 1, 1, T, V)[0]
    with tempfile.TemporaryDirectory() as tmp:
        ex = gtd.ExplainGridTDAttention(make_args(tmp), wm, model=model)
        _patch_explainer(ex, img, cap)
        feats = []
        orig = ex.explain_caption_wordt

        def wrapped(t):
            rf, rw = orig(t)
            feats.append(rf.clone())
            return rf, rw
        ex.explain_caption_wordt = wrapped
        maps, rws = ex.explain_caption("synthetic.jpg")
    g = dict(seed=np.int64(seed), V=np.int64(V), caption=cap)
    g["features"] = ex.image_features.detach().numpy()
    for k in ("predictions", "alphas", "betas", "h1t", "c1t", "h2t", "c2t", "g1t", "g2t", "i1t_act", "f1t_act",
              "i2t_act", "f2t_act", "st", "context", "context_hat"):
        v = getattr(ex, k).detach()
        g["tr_" + k] = (v[:, ::97] if k == "predictions" else v).numpy()
    for t in range(T):
        g[f"r_feat_{t}"] = feats[t].detach().numpy()
        g[f"r_words_{t}"] = rws[t].detach().numpy()
        g[f"map_stats_{t}"] = stats(maps[t])
        g[f"map_sub4_{t}"] = sub4(maps[t]).numpy()
    g[f"map_full_{T - 1}"] = maps[T - 1].numpy()
    np.savez(os.path.join(out, "gridtd_T3.npz"), **g)
    print("gridtd_T3.npz written; map absmax:", [float(m.abs().max()) for m in maps])


def gen_aoa(out, weights, T=3, V=11027, seed=0, heads=(0, 5)):
    import models.aoamodel as aoa
    sd = weights.make_aoa_state(seed=seed, vocab_size=V)
    model = aoa.AOAModel(512, 512, 8, V, 'vgg16')
    model.load_state_dict(to_torch_sd(sd))
    wm = weights.make_word_map(V)
    img = weights.make_images(seed, 1)
    cap = weights.make_captions(seed + 1, 1, T, V)[0]
    g = dict(seed=np.int64(seed), V=np.int64(V), caption=cap, heads=np.array(heads))
    with tempfile.TemporaryDirectory() as tmp:
        ex = aoa.ExplainAOAAttention(make_args(tmp), wm, model=model)
        _patch_explainer(ex, img, cap)
        for hd in heads:
            feats = []
            orig = aoa.ExplainAOAAttention.explain_caption_wordt

            def wrapped(t, head_idx, _orig=orig, _feats=feats):
                rf, rw = _orig(ex, t, head_idx)
                _feats.append(rf.clone())
                return rf, rw
            ex.explain_caption_wordt = wrapped
            maps, rws = ex.explain_caption("synthetic.jpg", hd)
            for t in range(T):
                g[f"h{hd}_r_feat_{t}"] = feats[t].detach().numpy()
                g[f"h{hd}_r_words_{t}"] = rws[t].detach().numpy()
                g[f"h{hd}_map_stats_{t}"] = stats(maps[t])
                g[f"h{hd}_map_sub4_{t}"] = sub4(maps[t]).numpy()
            if hd == heads[0]:
                g[f"h{hd}_map_full_{T - 1}"] = maps[T - 1].numpy()
        g["tr_predictions"] = ex.predictions.detach()[:, ::97].numpy()
        g["tr_alphas"] = ex.alphas.detach().numpy()
        g["tr_ht"] = ex.ht.detach().numpy()
        g["tr_context"] = ex.context.detach().numpy()
    np.savez(os.path.join(out, "aoa_T3.npz"), **g)
    print("aoa_T3.npz written")


def gen_aoa_bu(out, weights, T=3, V=11027, seed=0, head=0):
    """Config 5 (SURVEY §8(a) row A-BU): no BU explainer exists in the reference, so the oracle
    construction is ExplainAOAAttention on an AOAModel whose encoder is a stub returning the 36x2048
    region features as (1,2048,6,6) and whose img_projector carries the Linear(2048,H) weights."""
    import models.aoamodel as aoa
    sd = weights.make_aoa_state(seed=seed, vocab_size=V, feat_dim=2048, with_encoder=False)
    feats_np = weights.make_bu_features(seed, 1)[0]                       # (36,2048)

    class StubEnc(nn.Module):
        feat_dim = 2048

        def forward(self, img):
            f = torch.from_numpy(feats_np.T.copy()).reshape(1, 2048, 6, 6)
            return f, f.mean(dim=(2, 3)).squeeze()
    model = aoa.AOAModel(512, 512, 8, V, 'vgg16')
    model.img_encoder = StubEnc()
    model.encoder_raw_dim = 2048
    model.img_projector = nn.Conv2d(2048, 512, 1)
    model.load_state_dict(to_torch_sd(sd))
    wm = weights.make_word_map(V)
    cap = weights.make_captions(seed + 1, 1, T, V)[0]
    g = dict(seed=np.int64(seed), V=np.int64(V), caption=cap, head=np.int64(head))
    with tempfile.TemporaryDirectory() as tmp:
        ex = aoa.ExplainAOAAttention(make_args(tmp), wm, model=model)
        _patch_explainer(ex, np.zeros((1, 3, 8, 8), np.float32), cap)
        ex.get_hidden_parameters("synthetic")
        for t in range(T):
            with torch.no_grad():
                rf, rw = ex.explain_caption_wordt(t, head)
            g[f"r_feat_{t}"] = rf.detach().reshape(2048, 36).t().contiguous().numpy()   # (36,2048)
            g[f"r_words_{t}"] = rw.detach().numpy()
    np.savez(os.path.join(out, "aoa_bu_T3.npz"), **g)
    print("aoa_bu_T3.npz written")


def gen_guided(out, weights, T=3, V=9586, seed=0):
    """ExplainiGridTDGuidedGradient (models/gridTDmodel.py:1585-1723): guided backprop maps + word scores."""
    import models.gridTDmodel as gtd
    sd = weights.make_gridtd_state(seed=seed, vocab_size=V)
    wm = weights.make_word_map(V)
    img = weights.make_images(seed, 1)
    cap = weights.make_captions(seed + 1, 1, T, V)[0]
    with tempfile.TemporaryDirectory() as tmp:
        args = make_args(tmp)
        # the class builds its own model and loads args.weight: feed it through a patched torch.load
        real_load = torch.load
        torch.load = lambda *a, **k: {"state_dict": to_torch_sd(sd)}
        try:
            ex = gtd.ExplainiGridTDGuidedGradient(args, wm)
        finally:
            torch.load = real_load
        _patch_explainer(ex, img, cap)
        feats = []
        orig = ex.explain_caption_wordt

        def wrapped(t):
            rf, rw = orig(t)
            feats.append(rf.clone())
            return rf, rw
        ex.explain_caption_wordt = wrapped
        maps, rws = ex.explain_caption("synthetic.jpg")
    g = dict(seed=np.int64(seed), V=np.int64(V), caption=cap)
    for t in range(T):
        g[f"d_feat_{t}"] = feats[t].detach().numpy()
        g[f"r_words_{t}"] = rws[t].detach().numpy()
        g[f"map_stats_{t}"] = stats(maps[t])
        g[f"map_sub4_{t}"] = sub4(maps[t]).numpy()
    g[f"map_full_{T - 1}"] = maps[T - 1].numpy()
    g["tr_sen_gate"] = ex.sen_gate.detach().numpy()
    g["tr_o2t_act"] = ex.o2t_act.detach().numpy()
    g["tr_h2t"] = ex.h2t.detach().numpy()
    np.savez(os.path.join(out, "guided_T3.npz"), **g)
    print("guided_T3.npz written; map absmax:", [float(m.abs().max()) for m in maps])


def _gen_grad_family(cls_name, out, weights, T, V, seed):
    """Shared driver for the gradient-family explainers that build their own model from args.weight."""
    import models.gridTDmodel as gtd
    sd = weights.make_gridtd_state(seed=seed, vocab_size=V)
    wm = weights.make_word_map(V)
    img = weights.make_images(seed, 1)
    cap = weights.make_captions(seed + 1, 1, T, V)[0]
    with tempfile.TemporaryDirectory() as tmp:
        args = make_args(tmp)
        real_load = torch.load
        torch.load = lambda *a, **k: {"state_dict": to_torch_sd(sd)}
        try:
            ex = getattr(gtd, cls_name)(args, wm)
        finally:
            torch.load = real_load
        _patch_explainer(ex, img, cap)
        feats = []
        orig = ex.explain_caption_wordt

        def wrapped(t):
            rf, rw = orig(t)
            feats.append(rf.clone())
            return rf, rw
        ex.explain_caption_wordt = wrapped
        maps, rws = ex.explain_caption("synthetic.jpg")
    return cap, feats, maps, rws


def gen_gradient(out, weights, T=3, V=9586, seed=0):
    """ExplainGridTDGradient (models/gridTDmodel.py:1214-1539): plain-gradient maps (autograd through the encoder,
    :1507-1521) + the hand-written decoder BPTT (:1424-1505) + word scores."""
    cap, feats, maps, rws = _gen_grad_family("ExplainGridTDGradient", out, weights, T, V, seed)
    g = dict(seed=np.int64(seed), V=np.int64(V), caption=cap)
    for t in range(T):
        g[f"d_feat_{t}"] = feats[t].detach().numpy()
        g[f"r_words_{t}"] = rws[t].detach().numpy()
        g[f"map_stats_{t}"] = stats(maps[t])
        g[f"map_sub4_{t}"] = sub4(maps[t]).numpy()
    g[f"map_full_{T - 1}"] = maps[T - 1].numpy()
    np.savez(os.path.join(out, "gradient_T3.npz"), **g)
    print("gradient_T3.npz written; map absmax:", [float(m.abs().max()) for m in maps])


def gen_gradcam(out, weights, T=3, V=9586, seed=0):
    """ExplainGridTDGradCam (models/gridTDmodel.py:1752-1771): per word the (1,196) Grad-CAM heat map
    relu(sum_c features_c * mean_hw(grad_c)) / (max + 1e-6) of the plain decoder gradient."""
    cap, feats, maps, rws = _gen_grad_family("ExplainGridTDGradCam", out, weights, T, V, seed)
    g = dict(seed=np.int64(seed), V=np.int64(V), caption=cap)
    for t in range(T):
        g[f"cam_{t}"] = maps[t].detach().numpy()
        g[f"r_words_{t}"] = rws[t].detach().numpy()
    np.savez(os.path.join(out, "gradcam_T3.npz"), **g)
    print("gradcam_T3.npz written; cam shapes:", [tuple(m.shape) for m in maps], "max:", [float(m.max()) for m in maps])


def gen_aoa_gradient(out, weights, T=3, V=11027, seed=0, head=5):
    """ExplainAOAGradient / ExplainAOAGuidedGradient / ExplainAOAGradCam (models/aoamodel.py:1257-1711), head `head`."""
    import models.aoamodel as aoa
    sd = weights.make_aoa_state(seed=seed, vocab_size=V)
    wm = weights.make_word_map(V)
    img = weights.make_images(seed, 1)
    cap = weights.make_captions(seed + 1, 1, T, V)[0]
    g = dict(seed=np.int64(seed), V=np.int64(V), caption=cap, head=np.int64(head))
    for cls_name, tag in (("ExplainAOAGradient", "grad"), ("ExplainAOAGuidedGradient", "guided"),
                          ("ExplainAOAGradCam", "cam")):
        with tempfile.TemporaryDirectory() as tmp:
            args = make_args(tmp)
            real_load = torch.load
            torch.load = lambda *a, **k: {"state_dict": to_torch_sd(sd)}
            try:
                ex = getattr(aoa, cls_name)(args, wm)
            finally:
                torch.load = real_load
            _patch_explainer(ex, img, cap)
            feats = []
            orig = ex.explain_caption_wordt

            def wrapped(t, head_idx, _orig=orig, _feats=feats):
                rf, rw = _orig(t, head_idx)
                _feats.append(rf.clone())
                return rf, rw
            ex.explain_caption_wordt = wrapped
            maps, rws = ex.explain_caption("synthetic.jpg", head)
        for t in range(T):
            if tag == "cam":
                g[f"cam_{t}"] = maps[t].detach().numpy()
            else:
                g[f"{tag}_map_stats_{t}"] = stats(maps[t])
                g[f"{tag}_map_sub4_{t}"] = sub4(maps[t]).numpy()
            if tag == "grad":
                g[f"d_feat_{t}"] = feats[t].detach().numpy()
                g[f"r_words_{t}"] = rws[t].detach().numpy()
        if tag == "grad":
            g["tr_ot_act"] = ex.ot_act.detach().numpy()
            g["tr_ht"] = ex.ht.detach().numpy()
    np.savez(os.path.join(out, "aoa_gradient_T3.npz"), **g)
    print("aoa_gradient_T3.npz written")


def gen_eval(out, weights, seed=0):
    """Relevance-map consumers of evaluation.py called on seeded maps: `block_image` (:57-80),
    `_calculate_overlaped_pixels` (:313-336), `_project_maxabs` (:338-343).  (2,3,224,224) maps with the statistics
    of real relevance maps: heavy-tailed, both signs, one of them with an exactly-zero border."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    plt.show = lambda *a, **k: None
    import evaluation
    rs = np.random.RandomState(seed)
    maps = (rs.standard_normal((2, 3, 224, 224)) * np.exp(2 * rs.standard_normal((2, 3, 224, 224)))).astype(np.float32)
    maps[1, :, :16, :] = 0
    maps[1, :, :, -24:] = 0
    ex = types.SimpleNamespace(model=types.SimpleNamespace(eval=lambda: None), word_map={"<start>": 0})
    ev = evaluation.EvaluationExperiments(ex)
    g = dict(seed=np.int64(seed))        # the maps are regenerated from the seed by the tests (same two lines)
    boxes = np.array([[30, 40, 150, 200], [0, 0, 224, 100]], np.int64)
    g["boxes"] = boxes
    thresholds = [0, 0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9]
    g["thresholds"] = np.array(thresholds, np.float32)
    for i in range(2):
        spatial = torch.from_numpy(maps[i:i + 1]).mean(dim=(0, 1))                    # evaluation.py:134
        g[f"mask_{i}"] = ev.block_image(spatial).numpy().astype(np.uint8)
        rel = np.mean(np.maximum(maps[i:i + 1], 0), axis=(0, 1))                      # evaluation.py:410-411
        rel = ev._project_maxabs(rel)
        g[f"proj_{i}"] = rel.astype(np.float32)
        work = rel.copy()
        g[f"ratio_{i}"] = np.array([ev._calculate_overlaped_pixels(list(boxes[i]), work, t) for t in thresholds], np.float64)
    # heat-map rendering of visualize_explanations (models/gridTDmodel.py:1196-1198): LRPutil.gamma + LRPutil.heatmap
    from LRPtools import utils as LRPutil
    for i in range(2):
        hm = maps[i:i + 1].transpose(0, 2, 3, 1).copy()
        hm = LRPutil.heatmap(LRPutil.gamma(hm))[0]                                    # (224,224,3) float32
        g[f"heatmap_sub2_{i}"] = hm[::2, ::2].astype(np.float32)
    g["lut"] = np.asarray(plt.cm.get_cmap("seismic")(np.arange(256)))[:, :3].astype(np.float32)
    plt.close("all")
    np.savez(os.path.join(out, "eval_consumers.npz"), **g)
    print("eval_consumers.npz written; ratios:", g["ratio_0"][:3], g["ratio_1"][:3])


def gen_sample_lrp(out, weights, V=9586, seed=0, max_len=8, batch=2):
    """LRP-inference decoding, `GridTDModel.sample_lrp` (models/gridTDmodel.py:631-702, greedy): token ids (bit-exact
    target) and their log-probabilities.  STOP_WORDS is empty here (nltk is a stub): only the special tokens are
    exempt from the re-weighting, the tests use the same exemption list."""
    import models.gridTDmodel as gtd
    sd = weights.make_gridtd_state(seed=seed, vocab_size=V)
    model = gtd.GridTDModel(512, 512, V, 'vgg16')
    model.load_state_dict(to_torch_sd(sd))
    model.eval()
    wm = weights.make_word_map(V)
    rev = {v: k for k, v in wm.items()}
    imgs = torch.from_numpy(weights.make_images(seed + 3, batch))
    with torch.no_grad():
        seq, lps, ml = model.sample_lrp(imgs, rev, wm, [max_len + 1] * batch, {})
        # second case: the most frequent word of the first run becomes a stop word (exempt from re-weighting) and
        # the last word of image 0 plays <end>, so that the skip branch and the zero padding after <end> are hit
        ids, counts = np.unique(seq.numpy(), return_counts=True)
        stop_id = int(ids[np.argmax(counts)])
        gtd.STOP_WORDS = [rev[stop_id]]
        wm2 = dict(wm)
        wm2['<end>'] = end2 = int(seq[0, -1])
        seq2, lps2, _ = model.sample_lrp(imgs, rev, wm2, [max_len + 1] * batch, {})
        gtd.STOP_WORDS = []
    special = [wm[k] for k in ('<start>', '<end>', '<pad>', '<unk>')]
    np.savez(os.path.join(out, "sample_lrp.npz"), seed=np.int64(seed), V=np.int64(V), max_len=np.int64(ml),
             seq=seq.numpy().astype(np.int64), logprobs=lps.numpy().astype(np.float32),
             skip=np.array(special, np.int64),
             seq2=seq2.numpy().astype(np.int64), logprobs2=lps2.numpy().astype(np.float32),
             skip2=np.array(special + [stop_id], np.int64), end2=np.int64(end2))
    print("sample_lrp tokens:", seq.tolist(), seq2.tolist(), "stop", stop_id, "end2", end2)


def gen_aoa_sample_lrp(out, weights, V=11027, seed=0, max_len=8, batch=2):
    """`AOAModel.sample_lrp` (models/aoamodel.py:679-745, greedy): same two cases as gen_sample_lrp."""
    import models.aoamodel as aoa
    sd = weights.make_aoa_state(seed=seed, vocab_size=V)
    model = aoa.AOAModel(512, 512, 8, V, 'vgg16')
    model.load_state_dict(to_torch_sd(sd))
    model.eval()
    wm = weights.make_word_map(V)
    rev = {v: k for k, v in wm.items()}
    imgs = torch.from_numpy(weights.make_images(seed + 3, batch))
    with torch.no_grad():
        seq, lps, ml = model.sample_lrp(imgs, rev, wm, [max_len + 1] * batch, {})
        ids, counts = np.unique(seq.numpy(), return_counts=True)
        stop_id = int(ids[np.argmax(counts)])
        aoa.STOP_WORDS = [rev[stop_id]]
        wm2 = dict(wm)
        others = [int(x) for x in seq[0] if int(x) != stop_id]
        wm2['<end>'] = end2 = others[-1] if others else int(seq[0, -1])
        seq2, lps2, _ = model.sample_lrp(imgs, rev, wm2, [max_len + 1] * batch, {})
        aoa.STOP_WORDS = []
    special = [wm[k] for k in ('<start>', '<end>', '<pad>', '<unk>')]
    np.savez(os.path.join(out, "aoa_sample_lrp.npz"), seed=np.int64(seed), V=np.int64(V), max_len=np.int64(ml),
             seq=seq.numpy().astype(np.int64), logprobs=lps.numpy().astype(np.float32),
             skip=np.array(special, np.int64),
             seq2=seq2.numpy().astype(np.int64), logprobs2=lps2.numpy().astype(np.float32),
             skip2=np.array(special + [stop_id], np.int64), end2=np.int64(end2))
    print("aoa_sample_lrp tokens:", seq.tolist(), seq2.tolist(), "stop", stop_id, "end2", end2)


def stats4(x):
    x = x.double()
    return np.array([x.sum().item(), x.abs().max().item(), x.pow(2).sum().sqrt().item(), x.abs().sum().item()], np.float64)


def gen_t20(out, weights, T=20, seed=0, n_img=2):
    """Decoder relevance at the HEADLINE caption length (T=20; BASELINE configs 2, 3 and 5), where the lock-step
    row machinery of the engines runs its full depth: `explain_caption_wordt` (models/gridTDmodel.py:1014-1135,
    models/aoamodel.py:1064-1156) of the reference for every word of `n_img` images per model.  Stored per (image,
    word): statistics of r_feat [sum, absmax, L2, L1], a channel subsample ((t % s)::s, s = 32 of 512 / 64 of 2048
    channels), the full r_words; full r_feat for two rows per model; for gridTD image 0 also the pixel maps
    (running sums, every 8th pixel) of the whole `explain_caption`."""
    import models.gridTDmodel as gtd
    import models.aoamodel as aoa
    g = dict(seed=np.int64(seed), T=np.int64(T), n_img=np.int64(n_img), img_seed=np.int64(50))
    imgs = weights.make_images(50, n_img)
    # ---------------- gridTD, V = 9586
    V = 9586
    sd = weights.make_gridtd_state(seed=seed, vocab_size=V)
    model = gtd.GridTDModel(512, 512, V, 'vgg16')
    model.load_state_dict(to_torch_sd(sd))
    wm = weights.make_word_map(V)
    caps = weights.make_captions(51, n_img, T, V)
    g["grid_V"], g["grid_caption"] = np.int64(V), caps
    for b in range(n_img):
        with tempfile.TemporaryDirectory() as tmp:
            ex = gtd.ExplainGridTDAttention(make_args(tmp), wm, model=model)
            _patch_explainer(ex, imgs[b:b + 1], caps[b])
            if b == 0:
                feats = []
                orig = ex.explain_caption_wordt

                def wrapped(t, _orig=orig, _feats=feats):
                    rf, rw = _orig(t)
                    _feats.append(rf.clone())
                    return rf, rw
                ex.explain_caption_wordt = wrapped
                maps, rws = ex.explain_caption("synthetic.jpg")
                for t in range(T):
                    g[f"grid0_map_stats_{t}"] = stats4(maps[t])
                    g[f"grid0_map_sub8_{t}"] = maps[t][..., ::8, ::8].contiguous().numpy()
            else:
                ex.get_hidden_parameters("synthetic.jpg")
                feats, rws = [], []
                for t in range(T):
                    with torch.no_grad():
                        rf, rw = ex.explain_caption_wordt(t)
                    feats.append(rf.clone()); rws.append(rw.clone())
        for t in range(T):
            rf = feats[t].detach()[0]                                   # (512,14,14)
            g[f"grid{b}_r_feat_stats_{t}"] = stats4(rf)
            g[f"grid{b}_r_feat_sub_{t}"] = rf[(t % 32)::32].contiguous().numpy()
            g[f"grid{b}_r_words_{t}"] = rws[t].detach().numpy()
        g[f"grid{b}_r_feat_full_{T - 1 - 9 * b}"] = feats[T - 1 - 9 * b].detach()[0].numpy()
        print("t20 gridTD image", b, "done", flush=True)
    # ---------------- AoA, V = 11027, head 0 for both images and head 3 for image 1
    V = 11027
    sd = weights.make_aoa_state(seed=seed, vocab_size=V)
    model = aoa.AOAModel(512, 512, 8, V, 'vgg16')
    model.load_state_dict(to_torch_sd(sd))
    wm = weights.make_word_map(V)
    caps = weights.make_captions(52, n_img, T, V)
    g["aoa_V"], g["aoa_caption"] = np.int64(V), caps
    for b in range(n_img):
        with tempfile.TemporaryDirectory() as tmp:
            ex = aoa.ExplainAOAAttention(make_args(tmp), wm, model=model)
            _patch_explainer(ex, imgs[b:b + 1], caps[b])
            ex.get_hidden_parameters("synthetic.jpg")
            for hd in ((0,) if b == 0 else (0, 3)):
                for t in range(T):
                    with torch.no_grad():
                        rf, rw = ex.explain_caption_wordt(t, hd)
                    rf = rf.detach()[0]
                    g[f"aoa{b}_h{hd}_r_feat_stats_{t}"] = stats4(rf)
                    g[f"aoa{b}_h{hd}_r_feat_sub_{t}"] = rf[(t % 32)::32].contiguous().numpy()
                    g[f"aoa{b}_h{hd}_r_words_{t}"] = rw.detach().numpy()
                    if hd == 0 and t == T - 1 - 9 * b:
                        g[f"aoa{b}_h0_r_feat_full_{t}"] = rf.numpy()
        print("t20 AoA image", b, "done", flush=True)
    # ---------------- AoA bottom-up (config 5), 36 x 2048 region features, head 0
    sd = weights.make_aoa_state(seed=seed, vocab_size=V, feat_dim=2048, with_encoder=False)
    feats_np = weights.make_bu_features(50, n_img)
    caps = weights.make_captions(53, n_img, T, V)
    g["bu_caption"] = caps
    for b in range(n_img):
        class StubEnc(nn.Module):
            feat_dim = 2048

            def forward(self, img, _f=feats_np[b]):
                f = torch.from_numpy(_f.T.copy()).reshape(1, 2048, 6, 6)
                return f, f.mean(dim=(2, 3)).squeeze()
        model = aoa.AOAModel(512, 512, 8, V, 'vgg16')
        model.img_encoder = StubEnc()
        model.encoder_raw_dim = 2048
        model.img_projector = nn.Conv2d(2048, 512, 1)
        model.load_state_dict(to_torch_sd(sd))
        with tempfile.TemporaryDirectory() as tmp:
            ex = aoa.ExplainAOAAttention(make_args(tmp), wm, model=model)
            _patch_explainer(ex, np.zeros((1, 3, 8, 8), np.float32), caps[b])
            ex.get_hidden_parameters("synthetic")
            for t in range(T):
                with torch.no_grad():
                    rf, rw = ex.explain_caption_wordt(t, 0)
                rf = rf.detach().reshape(2048, 36).t().contiguous()                     # (36,2048)
                g[f"bu{b}_r_feat_stats_{t}"] = stats4(rf)
                g[f"bu{b}_r_feat_sub_{t}"] = rf[:, (t % 64)::64].contiguous().numpy()
                g[f"bu{b}_r_words_{t}"] = rw.detach().numpy()
                if t == T - 1 - 9 * b:
                    g[f"bu{b}_r_feat_full_{t}"] = rf.numpy()
        print("t20 BU image", b, "done", flush=True)
    np.savez(os.path.join(out, "t20.npz"), **g)
    print("t20.npz written:", sum(v.nbytes for v in g.values() if hasattr(v, "nbytes")) / 1e6, "MB")


def gen_t20_f64(out, weights, T=20, seed=0, n_img=2):
    """The rows of t20.npz once more with the reference's classes in DOUBLE precision (weights, image, every tensor the
    explainers allocate: `torch.set_default_dtype(float64)` for the call): the fp64 value of the reference's own formula,
    forward included.  |ref32 - fp64| of a row is the reference's own rounding noise there; tests/test_gpu_t20.py bounds
    |GPU - fp64| by 3 x that (floor 1e-5) instead of a flat tolerance.  Stored: r_words (float64) and
    r_feat statistics [sum, absmax, L2, L1] per (model, image, head, word), and the fp64 VGG16 features of both images
    (every 4th channel, as float32) so the GPU forward can be placed against fp64 on the GPU box."""
    import models.gridTDmodel as gtd
    import models.aoamodel as aoa
    g32 = np.load(os.path.join(out, "t20.npz"))
    g = dict(seed=np.int64(seed), T=np.int64(T), n_img=np.int64(n_img), img_seed=np.int64(50))
    imgs = weights.make_images(50, n_img).astype(np.float64)
    old = torch.get_default_dtype()
    torch.set_default_dtype(torch.float64)
    try:
        V = 9586
        sd = weights.make_gridtd_state(seed=seed, vocab_size=V)
        model = gtd.GridTDModel(512, 512, V, 'vgg16')
        model.load_state_dict(to_torch_sd(sd))
        model = model.double()
        wm = weights.make_word_map(V)
        caps = g32["grid_caption"]
        for b in range(n_img):
            with tempfile.TemporaryDirectory() as tmp:
                ex = gtd.ExplainGridTDAttention(make_args(tmp), wm, model=model)
                _patch_explainer(ex, imgs[b:b + 1], caps[b])
                ex.get_hidden_parameters("synthetic.jpg")
                assert ex.image_features.dtype == torch.float64
                g[f"features64_{b}"] = ex.image_features.detach()[0, ::4].reshape(128, 196).float().numpy()
                g[f"features64_absmax_{b}"] = np.float64(ex.image_features.abs().max().item())
                for t in range(T):
                    with torch.no_grad():
                        rf, rw = ex.explain_caption_wordt(t)
                    assert rw.dtype == torch.float64
                    g[f"grid{b}_r_feat_stats64_{t}"] = stats4(rf.detach()[0])
                    g[f"grid{b}_r_words64_{t}"] = rw.detach().numpy()
            print("t20_f64 gridTD image", b, "done", flush=True)
        V = 11027
        sd = weights.make_aoa_state(seed=seed, vocab_size=V)
        model = aoa.AOAModel(512, 512, 8, V, 'vgg16')
        model.load_state_dict(to_torch_sd(sd))
        model = model.double()
        wm = weights.make_word_map(V)
        caps = g32["aoa_caption"]
        for b in range(n_img):
            with tempfile.TemporaryDirectory() as tmp:
                ex = aoa.ExplainAOAAttention(make_args(tmp), wm, model=model)
                _patch_explainer(ex, imgs[b:b + 1], caps[b])
                ex.get_hidden_parameters("synthetic.jpg")
                for hd in ((0,) if b == 0 else (0, 3)):
                    for t in range(T):
                        with torch.no_grad():
                            rf, rw = ex.explain_caption_wordt(t, hd)
                        g[f"aoa{b}_h{hd}_r_feat_stats64_{t}"] = stats4(rf.detach()[0])
                        g[f"aoa{b}_h{hd}_r_words64_{t}"] = rw.detach().numpy()
            print("t20_f64 AoA image", b, "done", flush=True)
        sd = weights.make_aoa_state(seed=seed, vocab_size=V, feat_dim=2048, with_encoder=False)
        feats_np = weights.make_bu_features(50, n_img).astype(np.float64)
        caps = g32["bu_caption"]
        for b in range(n_img):
            class StubEnc(nn.Module):
                feat_dim = 2048

                def forward(self, img, _f=feats_np[b]):
                    f = torch.from_numpy(_f.T.copy()).reshape(1, 2048, 6, 6)
                    return f, f.mean(dim=(2, 3)).squeeze()
            model = aoa.AOAModel(512, 512, 8, V, 'vgg16')
            model.img_encoder = StubEnc()
            model.encoder_raw_dim = 2048
            model.img_projector = nn.Conv2d(2048, 512, 1)
            model.load_state_dict(to_torch_sd(sd))
            model = model.double()
            with tempfile.TemporaryDirectory() as tmp:
                ex = aoa.ExplainAOAAttention(make_args(tmp), wm, model=model)
                _patch_explainer(ex, np.zeros((1, 3, 8, 8), np.float64), caps[b])
                ex.get_hidden_parameters("synthetic")
                for t in range(T):
                    with torch.no_grad():
                        rf, rw = ex.explain_caption_wordt(t, 0)
                    g[f"bu{b}_r_feat_stats64_{t}"] = stats4(rf.detach().reshape(2048, 36).t())
                    g[f"bu{b}_r_words64_{t}"] = rw.detach().numpy()
            print("t20_f64 BU image", b, "done", flush=True)
    finally:
        torch.set_default_dtype(old)
    np.savez(os.path.join(out, "t20_f64.npz"), **g)
    print("t20_f64.npz written:", sum(v.nbytes for v in g.values() if hasattr(v, "nbytes")) / 1e6, "MB")

def gen_t20_guided(out, weights, T=20, seed=0, n_img=2):
    """Guided-Backprop decoder BPTT at the headline caption length (BASELINE config 4 is LRP + Guided-Backprop at T = 20):
    `ExplainiGridTDGuidedGradient.explain_caption_wordt` (models/gridTDmodel.py:1588-1675) for every word of the two
    images / captions of t20.npz.  Stored per (image, word): statistics of d_feat [sum, absmax, L2, L1], the channel
    subsample (t % 32)::32, r_words; two full d_feat rows; for image 0 the pixel maps of the whole `explain_caption`
    (every 8th pixel + statistics)."""
    import models.gridTDmodel as gtd
    g32 = np.load(os.path.join(out, "t20.npz"))
    V = int(g32["grid_V"])
    g = dict(seed=np.int64(seed), T=np.int64(T), n_img=np.int64(n_img), img_seed=np.int64(50), V=np.int64(V))
    imgs = weights.make_images(50, n_img)
    caps = g32["grid_caption"]
    g["caption"] = caps
    sd = weights.make_gridtd_state(seed=seed, vocab_size=V)
    wm = weights.make_word_map(V)
    for b in range(n_img):
        with tempfile.TemporaryDirectory() as tmp:
            real_load = torch.load
            torch.load = lambda *a, **k: {"state_dict": to_torch_sd(sd)}
            try:
                ex = gtd.ExplainiGridTDGuidedGradient(make_args(tmp), wm)
            finally:
                torch.load = real_load
            _patch_explainer(ex, imgs[b:b + 1], caps[b])
            feats, rws = [], []
            if b == 0:
                orig = ex.explain_caption_wordt

                def wrapped(t, _orig=orig, _feats=feats):
                    rf, rw = _orig(t)
                    _feats.append(rf.clone())
                    return rf, rw
                ex.explain_caption_wordt = wrapped
                maps, rws = ex.explain_caption("synthetic.jpg")
                for t in range(T):
                    g[f"gb0_map_stats_{t}"] = stats4(maps[t])
                    g[f"gb0_map_sub8_{t}"] = maps[t][..., ::8, ::8].contiguous().numpy()
            else:
                ex.get_hidden_parameters("synthetic.jpg")
                ex.image_feature_proj = ex.image_feature_proj.transpose(1, 2)       # as explain_caption does (:1528)
                for t in range(T):
                    with torch.no_grad():
                        rf, rw = ex.explain_caption_wordt(t)
                    feats.append(rf.clone()); rws.append(rw.clone())
        for t in range(T):
            rf = feats[t].detach()[0]                                   # (512,14,14)
            g[f"gb{b}_d_feat_stats_{t}"] = stats4(rf)
            g[f"gb{b}_d_feat_sub_{t}"] = rf[(t % 32)::32].contiguous().numpy()
            g[f"gb{b}_r_words_{t}"] = rws[t].detach().numpy()
        g[f"gb{b}_d_feat_full_{T - 1 - 9 * b}"] = feats[T - 1 - 9 * b].detach()[0].numpy()
        print("t20_guided image", b, "done", flush=True)
    np.savez(os.path.join(out, "t20_guided.npz"), **g)
    print("t20_guided.npz written:", sum(v.nbytes for v in g.values() if hasattr(v, "nbytes")) / 1e6, "MB")

def toy_resnet(rs, add_cls, flatten_cls):
    """Conv-BN-ReLU -> [Conv-BN-ReLU] + skip (explicit Add module) -> MaxPool -> Flatten -> Linear, weights drawn from `rs`.
    Shared by make_golden.py (reference side: models.resnet.Add / Flatten) and tests/test_gpu_hooks.py (this repo's classes)."""
    class Toy(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(3, 8, 3, padding=1); self.bn1 = nn.BatchNorm2d(8); self.relu1 = nn.ReLU()
            self.conv2 = nn.Conv2d(8, 8, 3, padding=1, bias=False); self.bn2 = nn.BatchNorm2d(8); self.relu2 = nn.ReLU()
            self.add = add_cls(); self.pool = nn.MaxPool2d(2, 2); self.flat = flatten_cls()
            self.fc = nn.Linear(8 * 7 * 7, 10)

        def forward(self, x):
            x = self.relu1(self.bn1(self.conv1(x)))
            y = self.relu2(self.bn2(self.conv2(x)))
            z = self.pool(self.add(x, y))
            return self.fc(self.flat(z))
    net = Toy()
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            m.weight.data = torch.from_numpy(rs.standard_normal(m.weight.shape).astype(np.float32) * 0.2)
            if m.bias is not None:
                m.bias.data = torch.from_numpy(rs.standard_normal(m.bias.shape).astype(np.float32) * 0.1)
        if isinstance(m, nn.BatchNorm2d):
            m.weight.data = torch.from_numpy(rs.uniform(0.5, 1.5, m.weight.shape).astype(np.float32))
            m.bias.data = torch.from_numpy(rs.standard_normal(m.bias.shape).astype(np.float32) * 0.2)
            m.running_mean = torch.from_numpy(rs.standard_normal(m.bias.shape).astype(np.float32) * 0.2)
            m.running_var = torch.from_numpy(rs.uniform(0.5, 1.5, m.bias.shape).astype(np.float32))
    return net.eval()


def gen_toy(out):
    """The reference's `add_lrp` / `compute_lrp` (LRPtools/lrp_wrapper.py:37-87) on a NON-VGG leaf sequence with a residual
    branch: Conv2d alpha1beta0, BatchNorm2d, ReLU, the explicit Add / Flatten modules of models/resnet.py:25-38, MaxPool2d
    and the Linear epsilon rule, driven by autograd + the reference's hooks.  Two calls on the same sample tensor: the second
    result carries the `.grad` running sum (:64-82).  (nn.Dropout in eval mode is an alias of its input under this PyTorch:
    the reference's own hook then fires on the Flatten view's node and its assert fails - not part of the fixture.)"""
    from LRPtools import lrp_wrapper
    import models.resnet as rn
    rs = np.random.RandomState(5)
    net = toy_resnet(rs, rn.Add, rn.Flatten)
    lrp_wrapper.add_lrp(net)
    x = torch.from_numpy(rs.standard_normal((2, 3, 14, 14)).astype(np.float32))
    x[1, :, 5, 6] = 0.0
    target = torch.from_numpy(rs.standard_normal((2, 10)).astype(np.float32))
    target2 = torch.from_numpy(rs.standard_normal((2, 10)).astype(np.float32))
    xs = x.clone()
    r1, logits = net.compute_lrp(xs, target=target.clone(), return_output=True)
    r2 = net.compute_lrp(xs, target=target2.clone())
    g = dict(x=x.numpy(), target=target.numpy(), target2=target2.numpy(), r1=r1.numpy(), r2=r2.numpy(),
             logits=logits.detach().numpy(), seed=np.int64(5))
    np.savez(os.path.join(out, "toy_resnet.npz"), **g)
    print("toy_resnet.npz: r1 absmax %.4f sum %.4f; r2 - r1 absmax %.4f" % (r1.abs().max(), r1.sum(), (r2 - r1).abs().max()))


AVGPOOL_CASES = [   # (name, input shape, AvgPool2d kwargs)
    ("k2", (2, 5, 8, 6), dict(kernel_size=2, stride=2)),
    ("k3s2p1", (1, 3, 7, 9), dict(kernel_size=3, stride=2, padding=1)),
    ("k3s2p1_nopad", (1, 3, 7, 9), dict(kernel_size=3, stride=2, padding=1, count_include_pad=False)),
    ("k3s2_ceil", (1, 2, 8, 8), dict(kernel_size=3, stride=2, ceil_mode=True)),
    ("k23_div5", (2, 2, 6, 7), dict(kernel_size=(2, 3), stride=(1, 2), padding=(1, 0), divisor_override=5)),   # (dropped by the reference's clone)
    ("global7", (2, 4, 7, 7), dict(kernel_size=7)),
]


def avgpool_net(rs, Flatten):
    """Conv - ReLU - AvgPool2d(2,2) - Conv - ReLU - AvgPool2d(7) - Flatten - Linear: the ResNet-style head the reference's table
    entry exists for (lrp_modules.py:327), small enough for a fixture"""
    import collections
    f32 = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    net = nn.Sequential(collections.OrderedDict([
        ("conv1", nn.Conv2d(3, 8, 3, padding=1)), ("relu1", nn.ReLU()), ("pool1", nn.AvgPool2d(2, 2)),
        ("conv2", nn.Conv2d(8, 16, 3, padding=1)), ("relu2", nn.ReLU()), ("pool2", nn.AvgPool2d(7)),
        ("flat", Flatten()), ("fc", nn.Linear(16, 5))]))
    for m in net.modules():
        if isinstance(m, (nn.Conv2d, nn.Linear)):
            m.weight.data = f32(rs.standard_normal(tuple(m.weight.shape)) * 0.2)
            m.bias.data = f32(rs.standard_normal(tuple(m.bias.shape)) * 0.05)
    return net.eval()


def gen_avgpool(out):
    """Pool2d.propagate_relevance on nn.AvgPool2d modules (LRPtools/lrp_modules.py:172-195; the table entry :327): rule-level
    cases (non-overlapping 2x2 incl. an all-zero window, overlapping windows with padding counted or not, ceil_mode,
    divisor_override, a global pool) and a small Conv / AvgPool / Linear net through the reference's own add_lrp."""
    from LRPtools import lrp_wrapper, lrp_modules
    params = lrp_wrapper.SequentialPresetA().lrp_params
    rs = np.random.RandomState(77)
    g = {}
    for name, shape, kw in AVGPOOL_CASES:
        m = nn.AvgPool2d(**kw)
        x = torch.from_numpy(rs.standard_normal(shape).astype(np.float32))
        if name == "k2":
            x[0, 1, 2:4, 2:4] = 0.0          # an all-zero window: Z == 0 -> safe_divide's 1e-7
            x[1, 0, 0, 0] = 0.0
        m.input = (x,)
        r = torch.from_numpy(rs.standard_normal(tuple(m(x).shape)).astype(np.float32))
        res = lrp_modules.Pool2d().propagate_relevance(m, None, (r,), 'alpha_beta', params)
        g[name + "_x"], g[name + "_rout"], g[name + "_rin"] = x.numpy(), r.numpy(), res[0].detach().numpy()
    import models.resnet as rn
    net = avgpool_net(np.random.RandomState(78), rn.Flatten)
    lrp_wrapper.add_lrp(net)
    x = torch.from_numpy(rs.standard_normal((2, 3, 14, 14)).astype(np.float32))
    target = torch.from_numpy(rs.standard_normal((2, 5)).astype(np.float32))
    rn, logits = net.compute_lrp(x.clone(), target=target.clone(), return_output=True)
    g["net_x"], g["net_target"], g["net_r"], g["net_logits"] = x.numpy(), target.numpy(), rn.numpy(), logits.detach().numpy()
    np.savez(os.path.join(out, "avgpool.npz"), **g)
    print("avgpool.npz:", {k: float(np.abs(v).max()) for k, v in g.items() if k.endswith("_rin") or k == "net_r"})


def gen_m4(out):
    """The rule classes the VGG16 path never reaches (ResNet encoders): Linear epsilon rule with the
    in-place zero nudge (LRPtools/lrp_modules.py:9-37), BatchNorm2d / BatchNorm1d (:197-246), Dropout (:248-254),
    Add (:256-280), Flatten (:282-291), each called directly on crafted inputs incl. the edge cases (exact-zero inputs,
    Z == 0, |xw| + |b| == 0, zero sums)."""
    from LRPtools import lrp_wrapper, lrp_modules
    import models.resnet as resnet
    rs = np.random.RandomState(321)
    f32 = lambda a: torch.from_numpy(np.asarray(a, np.float32))
    params = lrp_wrapper.SequentialPresetA().lrp_params
    g = {}
    # ---- Linear: N=5 rows, 70 -> 41 features (no multiple of anything), zeros in x, a zero weight row (Z == 0 -> 0.01)
    lin = nn.Linear(70, 41)
    lin.weight.data = f32(rs.uniform(-0.2, 0.2, (41, 70)))
    lin.bias.data = f32(rs.uniform(-0.2, 0.2, (41,)))
    lin.weight.data[7] = 0.0
    x = f32(rs.standard_normal((5, 70)))
    x[1, 4] = 0.0
    x[3, :] = 0.0                       # a whole row of zeros -> all -1e-6
    g["lin_x"] = x.numpy().copy()
    lin.input = (x,)
    r = f32(rs.standard_normal((5, 41)))
    res = lrp_modules.Linear().propagate_relevance(lin, (torch.zeros(41), torch.zeros(5, 70), torch.zeros(70, 41)), (r,),
                                                   'epsilon', params)
    g["lin_w"], g["lin_b"], g["lin_rout"], g["lin_rin"] = lin.weight.data.numpy(), lin.bias.data.numpy(), r.numpy(), res[1].detach().numpy()
    g["lin_x_after"] = lin.input[0].numpy().copy()          # quirk (h): the saved input is mutated in place
    # ---- BatchNorm2d, eval statistics; channel 2 has b == 0 and zero inputs (|xw| + |b| == 0 -> safe_divide guard)
    bn = nn.BatchNorm2d(6).eval()
    bn.weight.data, bn.bias.data = f32(rs.uniform(0.5, 1.5, 6)), f32(rs.uniform(-0.5, 0.5, 6))
    bn.running_mean.data, bn.running_var.data = f32(rs.uniform(-0.5, 0.5, 6)), f32(rs.uniform(0.5, 2.0, 6))
    bn.bias.data[2] = 0.0
    bn.running_mean.data[2] = 0.0
    bn.weight.data[4] = -0.7             # a negative gamma
    xb = f32(rs.standard_normal((2, 6, 5, 7)))
    xb[:, 2, 1:3] = 0.0
    xb[0, 0, 0, 0] = 0.0
    bn.input = (xb,)
    rb = f32(rs.standard_normal((2, 6, 5, 7)))
    res = lrp_modules.BatchNorm2d().propagate_relevance(bn, (None, 1, 2), (rb,), 'epsilon', params)
    for k, v in (("gamma", bn.weight), ("beta", bn.bias), ("mean", bn.running_mean), ("var", bn.running_var)):
        g["bn2_" + k] = v.detach().numpy()
    g["bn2_eps"] = np.float64(bn.eps)
    g["bn2_x"], g["bn2_rout"], g["bn2_rin"] = xb.numpy(), rb.numpy(), res[0].detach().numpy()
    # ---- BatchNorm1d: the rule indexes w[:, None, None] like the 2-d one (:236-238), so a (N,C) input broadcasts to a
    # (C,N,C) result and a (1,C,L) input to (C,C,L); reproduced as it is
    b1 = nn.BatchNorm1d(6).eval()
    b1.weight.data, b1.bias.data = f32(rs.uniform(0.5, 1.5, 6)), f32(rs.uniform(-0.5, 0.5, 6))
    b1.running_mean.data, b1.running_var.data = f32(rs.uniform(-0.5, 0.5, 6)), f32(rs.uniform(0.5, 2.0, 6))
    for k, v in (("gamma", b1.weight), ("beta", b1.bias), ("mean", b1.running_mean), ("var", b1.running_var)):
        g["bn1_" + k] = v.detach().numpy()
    x1 = f32(rs.standard_normal((4, 6)))
    x1[2, 3] = 0.0
    b1.input = (x1,)
    r1 = f32(rs.standard_normal((4, 6)))
    res = lrp_modules.BatchNorm1d().propagate_relevance(b1, (None, 1, 2), (r1,), 'epsilon', params)
    g["bn1_x"], g["bn1_rout"], g["bn1_rin"] = x1.numpy(), r1.numpy(), res[0].detach().numpy()
    x3 = f32(rs.standard_normal((1, 6, 5)))
    b1.input = (x3,)
    r3 = f32(rs.standard_normal((1, 6, 5)))
    res = lrp_modules.BatchNorm1d().propagate_relevance(b1, (None, 1, 2), (r3,), 'epsilon', params)
    g["bn1_x3"], g["bn1_rout3"], g["bn1_rin3"] = x3.numpy(), r3.numpy(), res[0].detach().numpy()
    # ---- Add: proportional split; both-zero entries get 0.5 / 0.5 (:262-272)
    add = resnet.Add()
    a1, a2 = f32(rs.standard_normal((2, 4, 3, 3))), f32(rs.standard_normal((2, 4, 3, 3)))
    a1[0, 1], a2[0, 1] = 0.0, 0.0        # zero sums from zero inputs
    a1[1, 2, 0, 0], a2[1, 2, 0, 0] = 0.0, 0.7
    add.input = (a1, a2)
    ra = f32(rs.standard_normal((2, 4, 3, 3)))
    R1, R2 = lrp_modules.Add().propagate_relevance(add, None, (ra,), 'alpha_beta', params)
    g["add_x1"], g["add_x2"], g["add_rout"], g["add_r1"], g["add_r2"] = a1.numpy(), a2.numpy(), ra.numpy(), R1.numpy(), R2.numpy()
    # ---- Flatten / Dropout
    fl = resnet.Flatten()
    xf = f32(rs.standard_normal((3, 4, 2, 2)))
    fl.input = (xf,)
    rf = f32(rs.standard_normal((3, 16)))
    g["flat_rout"], g["flat_rin"] = rf.numpy(), lrp_modules.Flatten().propagate_relevance(fl, None, (rf,), 'alpha_beta', params)[0].numpy()
    dr = nn.Dropout(0.5).eval()
    rd = f32(rs.standard_normal((3, 16)))
    got = lrp_modules.Dropout().propagate_relevance(dr, (rd.clone(),), (rd,), 'alpha_beta', params)
    g["drop_r"], g["drop_rin"] = rd.numpy(), got[0].numpy()
    np.savez(os.path.join(out, "m4.npz"), **g)
    print("m4.npz:", {k: getattr(v, "shape", None) for k, v in g.items()})



def gen_forwardlrp(out, weights, seed=0, T=8, batch=2):
    """The forward half of LRP-inference fine-tuning: `GridTDModel.forwardlrp_context` (models/gridTDmodel.py:579-630) and
    `AOAModel.forwardlrp_context` (models/aoamodel.py:628-677) in evaluation mode (dropout = identity) on a teacher-forced
    batch: raw and LRP-reweighted scores of every step.  Stored: every 13th vocabulary column, the arg-max of both per step,
    two full rows.  A second gridTD case makes the most frequent arg-max word a stop word (weights of 1 -> both scores equal)."""
    import models.gridTDmodel as gtd
    import models.aoamodel as aoa
    g = dict(seed=np.int64(seed), T=np.int64(T), batch=np.int64(batch))
    imgs = torch.from_numpy(weights.make_images(seed + 5, batch))
    for tag, mod, V, mk, ctor in (("grid", gtd, 9586, weights.make_gridtd_state, lambda V: gtd.GridTDModel(512, 512, V, 'vgg16')),
                                  ("aoa", aoa, 11027, weights.make_aoa_state, lambda V: aoa.AOAModel(512, 512, 8, V, 'vgg16'))):
        model = ctor(V)
        model.load_state_dict(to_torch_sd(mk(seed=seed, vocab_size=V)))
        model.eval()
        wm = weights.make_word_map(V)
        rev = {v: k for k, v in wm.items()}
        caps = torch.from_numpy(weights.make_captions(seed + 6, batch, T, V))
        lengths = [T + 1, T - 1]                       # (the reference only uses max(lengths) - 1)
        special = [wm[k] for k in ('<start>', '<end>', '<pad>', '<unk>')]
        with torch.no_grad():
            p, wp, L = model.forwardlrp_context(imgs, caps, lengths, rev)
            cases = [("", p, wp, special)]
            if tag == "grid":
                ids, counts = np.unique(p.argmax(-1).numpy(), return_counts=True)
                stop_id = int(ids[np.argmax(counts)])
                mod.STOP_WORDS = [rev[stop_id]]
                p2, wp2, _ = model.forwardlrp_context(imgs, caps, lengths, rev)
                mod.STOP_WORDS = []
                cases.append(("2", p2, wp2, special + [stop_id]))
        g[f"{tag}_V"], g[f"{tag}_caption"], g[f"{tag}_lengths"], g[f"{tag}_L"] = np.int64(V), caps.numpy(), np.array(lengths), np.int64(L)
        for sfx, p_, wp_, skip in cases:
            g[f"{tag}_pred_sub{sfx}"] = p_[:, :, ::13].numpy()
            g[f"{tag}_wpred_sub{sfx}"] = wp_[:, :, ::13].numpy()
            g[f"{tag}_pred_argmax{sfx}"] = p_.argmax(-1).numpy()
            g[f"{tag}_wpred_argmax{sfx}"] = wp_.argmax(-1).numpy()
            g[f"{tag}_pred_row{sfx}"] = p_[1, L - 1].numpy()
            g[f"{tag}_wpred_row{sfx}"] = wp_[1, L - 1].numpy()
            g[f"{tag}_skip{sfx}"] = np.array(skip, np.int64)
        print(tag, "forwardlrp_context L =", L, "argmax", g[f"{tag}_pred_argmax"].tolist(), g[f"{tag}_wpred_argmax"].tolist())
    np.savez(os.path.join(out, "forwardlrp.npz"), **g)
    print("forwardlrp.npz written:", sum(v.nbytes for v in g.values() if hasattr(v, "nbytes")) / 1e6, "MB")



def gen_guided_gradcam(out, weights, T=3, seed=0, head=5):
    """ExplainGridTDGuidedGradCam (models/gridTDmodel.py:1796-1836) and ExplainAOAGuidedGradCam (models/aoamodel.py:1714-1751):
    guided backprop x the 16x expanded Grad-CAM map.  scikit-image is not installed here, so the ONE call into it,
    `skimage.transform.pyramid_expand(cam, upscale=16, multichannel=False)`, is served by the restatement of its published
    algorithm in oracle/lrp_oracle.py (scipy.ndimage.gaussian_filter called as skimage calls it): these vectors pin
    everything around that call with the reference's own code; the expansion itself stays "parity unpinned"."""
    from oracle import lrp_oracle as O
    import models.gridTDmodel as gtd
    import models.aoamodel as aoa
    expand = lambda a, upscale=2, multichannel=False, **kw: O.pyramid_expand(torch.from_numpy(np.asarray(a)), upscale).double().numpy()
    sys.modules["skimage.transform"].pyramid_expand = expand
    sys.modules["skimage"].transform = sys.modules["skimage.transform"]
    g = dict(seed=np.int64(seed), T=np.int64(T), head=np.int64(head))
    img = weights.make_images(seed, 1)
    for tag, mod, cls, V, mk in (("grid", gtd, "ExplainGridTDGuidedGradCam", 9586, weights.make_gridtd_state),
                                 ("aoa", aoa, "ExplainAOAGuidedGradCam", 11027, weights.make_aoa_state)):
        sd = mk(seed=seed, vocab_size=V)
        wm = weights.make_word_map(V)
        cap = weights.make_captions(seed + 1, 1, T, V)[0]
        with tempfile.TemporaryDirectory() as tmp:
            real_load = torch.load
            torch.load = lambda *a, **k: {"state_dict": to_torch_sd(sd)}
            try:
                ex = getattr(mod, cls)(make_args(tmp), wm)
            finally:
                torch.load = real_load
            _patch_explainer(ex, img, cap)
            maps, rws = ex.explain_caption("synthetic.jpg") if tag == "grid" else ex.explain_caption("synthetic.jpg", head)
        g[f"{tag}_V"], g[f"{tag}_caption"] = np.int64(V), cap
        for t in range(T):
            g[f"{tag}_map_stats_{t}"] = stats(maps[t])
            g[f"{tag}_map_sub4_{t}"] = sub4(maps[t]).numpy()
            g[f"{tag}_r_words_{t}"] = rws[t].detach().numpy()
        g[f"{tag}_map_full_{T - 1}"] = maps[T - 1].numpy()
        print(tag, "guided grad-cam absmax:", [float(m.abs().max()) for m in maps])
    np.savez(os.path.join(out, "guided_gradcam_T3.npz"), **g)
    print("guided_gradcam_T3.npz written")



def gen_beam(out, weights, seed=0):
    """The captions the explainers explain when none is given: `GridTDModel.beam_search(beam_size=2, max_cap_length=50)`
    (models/gridTDmodel.py:400-478, called at :935) and `AOAModel.beam_search(beam_size=3, max_cap_length=20)`
    (models/aoamodel.py, called at :992).  gridTD's `beam_idx = top_words / vocab_size` (:444) is integer division in the
    PyTorch 1.4 the reference pins and true division today (a float index then raises): the harness restores the 1.4
    meaning of `/` on integer tensors for the duration of the call.  Three cases per model: the natural run (random-init
    weights never emit <end>: the cut at 20 tokens, :472), a word map whose <end> is a word the best beam produces (complete
    sequences, :449-453, :469-470) and one whose <unk> is such a word (dropped from the encoded caption, :474)."""
    import models.gridTDmodel as gtd
    import models.aoamodel as aoa
    g = dict(seed=np.int64(seed))
    img = torch.from_numpy(weights.make_images(seed + 7, 1))
    orig_div = torch.Tensor.__truediv__

    def div14(a, b):
        if torch.is_tensor(a) and not a.is_floating_point() and isinstance(b, int):
            return torch.div(a, b, rounding_mode="floor")
        return orig_div(a, b)
    for tag, V, mk, ctor, beam, steps in (("grid", 9586, weights.make_gridtd_state, lambda V: gtd.GridTDModel(512, 512, V, 'vgg16'), 2, 50),
                                          ("aoa", 11027, weights.make_aoa_state, lambda V: aoa.AOAModel(512, 512, 8, V, 'vgg16'), 3, 20)):
        model = ctor(V)
        model.load_state_dict(to_torch_sd(mk(seed=seed, vocab_size=V)))
        model.eval()
        wm = weights.make_word_map(V)
        torch.Tensor.__truediv__ = div14
        try:
            _, sen = model.beam_search(img, wm, beam_size=beam, max_cap_length=steps)
            late = [w for w in sen[3:] if w not in sen[:2]][0]          # a word the best beam reaches after a few steps
            wm_end = dict(wm); wm_end['<end>'] = int(late)
            _, sen_end = model.beam_search(img, wm_end, beam_size=beam, max_cap_length=steps)
            wm_end0 = dict(wm); wm_end0['<end>'] = int(sen[0])         # <end> as the very first word: an empty caption
            _, sen_end0 = model.beam_search(img, wm_end0, beam_size=beam, max_cap_length=steps)
            wm_unk = dict(wm); wm_unk['<unk>'] = int(sen[1])
            _, sen_unk = model.beam_search(img, wm_unk, beam_size=beam, max_cap_length=steps)
        finally:
            torch.Tensor.__truediv__ = orig_div
        g[f"{tag}_V"], g[f"{tag}_beam"], g[f"{tag}_steps"] = np.int64(V), np.int64(beam), np.int64(steps)
        g[f"{tag}_sen"], g[f"{tag}_sen_end"], g[f"{tag}_sen_unk"] = np.array(sen, np.int64), np.array(sen_end, np.int64), np.array(sen_unk, np.int64)
        g[f"{tag}_end2"], g[f"{tag}_unk2"] = np.int64(wm_end['<end>']), np.int64(wm_unk['<unk>'])
        g[f"{tag}_end0"], g[f"{tag}_sen_end0"] = np.int64(wm_end0['<end>']), np.array(sen_end0, np.int64)
        print(tag, "beam:", sen, "| <end> :=", wm_end['<end>'], sen_end, "| <end> first:", sen_end0, "| <unk> :=", wm_unk['<unk>'], sen_unk)
    np.savez(os.path.join(out, "beam.npz"), **g)



def gen_teacherforce(out, weights, T=3, seed=0):
    """`teacherforce_forward(img, beam_caption_encode)` of the four explainer families, called as evaluation.py:266,437,702,767 call
    it (after `get_hidden_parameters`, with the explainer's own image and the caption INCLUDING <start>): models/gridTDmodel.py
    :892-931 (LRP explainer: LanguageLSTM adds bias_ih twice) and :1282-1321 (gradient family: correct bias), models/aoamodel.py
    :952-988 and :1377-1413.  Stored: every 97th score of every step, the arg-max ids, the full last row."""
    import models.aoamodel as aoa
    import models.gridTDmodel as gtd
    g = dict(seed=np.int64(seed), T=np.int64(T))
    for tag, mod, cls_name, V, make_state in (("grid_lrp", gtd, "ExplainGridTDAttention", 9586, weights.make_gridtd_state),
                                              ("grid_grad", gtd, "ExplainGridTDGradient", 9586, weights.make_gridtd_state),
                                              ("aoa_lrp", aoa, "ExplainAOAAttention", 11027, weights.make_aoa_state),
                                              ("aoa_grad", aoa, "ExplainAOAGradient", 11027, weights.make_aoa_state)):
        sd = make_state(seed=seed, vocab_size=V)
        wm = weights.make_word_map(V)
        img = weights.make_images(seed, 1)
        cap = weights.make_captions(seed + 1, 1, T, V)[0]
        with tempfile.TemporaryDirectory() as tmp:
            real_load = torch.load
            torch.load = lambda *a, **k: {"state_dict": to_torch_sd(sd)}
            try:
                ex = getattr(mod, cls_name)(make_args(tmp, weight="synthetic.pth"), wm)
            finally:
                torch.load = real_load
            _patch_explainer(ex, img, cap)
            with torch.no_grad():
                ex.get_hidden_parameters("synthetic.jpg")
                assert list(ex.beam_caption_encode) == [int(c) for c in cap]
                pred = ex.teacherforce_forward(ex.img.detach().clone(), ex.beam_caption_encode)
        assert tuple(pred.shape) == (T + 1, V)
        g[f"{tag}_V"] = np.int64(V)
        g[f"{tag}_caption"] = cap
        g[f"{tag}_pred_sub"] = pred[:, ::97].detach().numpy()
        g[f"{tag}_argmax"] = pred.argmax(-1).numpy().astype(np.int64)
        g[f"{tag}_pred_last"] = pred[-1].detach().numpy()
        g[f"{tag}_absmax"] = np.float64(pred.abs().max().item())
    np.savez(os.path.join(out, "teacherforce.npz"), **g)
    print("teacherforce.npz written:", {k: getattr(v, "shape", v) for k, v in g.items() if k.endswith("argmax")})


def gen_greedy(out, weights, V=9586, seed=0, max_len=11):
    """Config 1: greedy token ids from the reference model's own `greedy_search`
    (models/gridTDmodel.py:480-520), int64, bit-exact target."""
    import models.gridTDmodel as gtd
    sd = weights.make_gridtd_state(seed=seed, vocab_size=V)
    model = gtd.GridTDModel(512, 512, V, 'vgg16')
    model.load_state_dict(to_torch_sd(sd))
    model.eval()
    wm = weights.make_word_map(V)
    img = torch.from_numpy(weights.make_images(seed, 1))
    _, seqs = model.greedy_search(img, wm, max_cap_length=max_len)
    np.savez(os.path.join(out, "greedy_cfg1.npz"), seed=np.int64(seed), V=np.int64(V),
             tokens=np.array(seqs[0], np.int64))
    print("greedy tokens:", seqs[0])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="layers,gridtd,aoa,aoa_bu,greedy,guided,gradient,gradcam,aoa_gradient,eval,sample_lrp,aoa_sample_lrp,t20,t20_f64,t20_guided,toy,m4,avgpool,forwardlrp,guided_gradcam,beam,teacherforce")
    ap.add_argument("--threads", type=int, default=1)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    install_stubs()
    weights = load_pkg()
    ref_vgg_patch()
    todo = set(a.only.split(","))
    if "layers" in todo:
        gen_layers(HERE)
    if "gridtd" in todo:
        gen_gridtd(HERE, weights)
    if "aoa" in todo:
        gen_aoa(HERE, weights)
    if "aoa_bu" in todo:
        gen_aoa_bu(HERE, weights)
    if "greedy" in todo:
        gen_greedy(HERE, weights)
    if "t20" in todo:
        gen_t20(HERE, weights)
    if "t20_f64" in todo:
        gen_t20_f64(HERE, weights)
    if "t20_guided" in todo:
        gen_t20_guided(HERE, weights)
    if "toy" in todo:
        gen_toy(HERE)
    if "m4" in todo:
        gen_m4(HERE)
    if "avgpool" in todo:
        gen_avgpool(HERE)
    if "forwardlrp" in todo:
        gen_forwardlrp(HERE, weights)
    if "guided_gradcam" in todo:
        gen_guided_gradcam(HERE, weights)
    if "beam" in todo:
        gen_beam(HERE, weights)
    if "teacherforce" in todo:
        gen_teacherforce(HERE, weights)
    if "guided" in todo:
        gen_guided(HERE, weights)
    if "gradient" in todo:
        gen_gradient(HERE, weights)
    if "gradcam" in todo:
        gen_gradcam(HERE, weights)
    if "aoa_gradient" in todo:
        gen_aoa_gradient(HERE, weights)
    if "eval" in todo:
        gen_eval(HERE, weights)
    if "sample_lrp" in todo:
        gen_sample_lrp(HERE, weights)
    if "aoa_sample_lrp" in todo:
        gen_aoa_sample_lrp(HERE, weights)


if __name__ == "__main__":
    main()
