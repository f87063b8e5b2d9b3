"""GPU parity of the guided-backprop path (BASELINE config 4 runs it side by side with LRP): decoder BPTT kernels
and the VGG16 guided backward chain against the reference's golden (tests/golden/guided_T3.npz) and the oracle."""
import os

import numpy as np
import pytest
import torch

import lrp_amd  # noqa: F401
from conftest import GOLDEN, rel_err, cosine, assert_close_modulo_pool_ties

pytestmark = pytest.mark.gpu
# End-to-end guided-backprop maps: besides the max-pool winners, every ReLU gate [a > 0] / clamp(g, min=0) is a
# discrete decision on values that are often within rounding of zero, so more pixels move by > 1e-4 of the maximum
# than for LRP when the forward comes from a different conv implementation (5 % on the golden image); cosine and
# the relative L2 error (< 2e-3) are unaffected.  The strict 1e-4 check runs on identical activations below.
# Bounds = 3x the worst observation of this file's 13 comparisons (round 2: 2.2 % of the pixels, max 3.1e-3, rel. L2 2.7e-4).
E2E = dict(frac=0.07, hard=1e-2, l2=8e-4)


@pytest.fixture(scope="module")
def case():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    g = np.load(os.path.join(GOLDEN, "guided_T3.npz"))
    sd = weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["V"]))
    eng = GridTDEngine(sd)
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    cap = torch.from_numpy(g["caption"]).view(1, -1)
    return g, sd, eng, img, cap


def test_guided_decoder_and_maps_vs_reference(case):
    g, sd, eng, img, cap = case
    maps, r_words, d_feat, tr, enc = eng.explain_batch_guided(img, cap, return_features=True)
    assert rel_err(tr["sgate"][0].cpu(), g["tr_sen_gate"]) < 1e-4 and rel_err(tr["o2"][0].cpu(), g["tr_o2t_act"]) < 1e-4
    assert rel_err(tr["h2"][0].cpu(), g["tr_h2t"]) < 1e-4
    for t in range(3):
        want = torch.from_numpy(g[f"d_feat_{t}"])[0].reshape(512, 196).t()
        assert rel_err(d_feat[0, t].cpu(), want) < 1e-4, t
        assert np.abs(r_words[0, t, :t + 1].cpu().numpy() - g[f"r_words_{t}"]).max() < 5e-5
        assert_close_modulo_pool_ties(maps[0, t][None, :, ::4, ::4].cpu(), g[f"map_sub4_{t}"], what=t, **E2E)
    assert_close_modulo_pool_ties(maps[0, 2].cpu(), g["map_full_2"][0], what="full", **E2E)


def test_guided_cnn_chain_strict_on_identical_activations(case):
    """the guided VGG16 backward on the oracle's own activations (same pool winners, same ReLU masks): 1e-4"""
    from test_gpu_vgg import _inject_oracle_trace, to_nhwc
    from oracle import lrp_oracle as O
    g, sd, eng, img, cap = case
    eng.vgg.forward(img.cuda())
    _inject_oracle_trace(eng.vgg, sd, img)
    d = torch.cat([torch.from_numpy(g[f"d_feat_{t}"]) for t in range(3)])
    maps = eng.vgg.guided_backprop(to_nhwc(d).cuda(), torch.zeros(3, dtype=torch.int32, device="cuda")).cpu()
    sdt = O.state_to_torch(sd)
    _, _, saved = O.vgg_forward(sdt, img)
    want = O.vgg_guided_backprop(sdt, saved, d)
    assert rel_err(maps, want) < 1e-4
    assert rel_err(maps[2:3], g["map_full_2"]) < 1e-4


def test_lrp_and_guided_side_by_side_batch(case):
    """config 4 shape: LRP and guided backprop on the same batch (B=2, T=3), guided vs the oracle"""
    from lrp_amd import weights
    from oracle import lrp_oracle as O
    g, sd, eng, img, cap = case
    V = int(g["V"])
    imgs = torch.from_numpy(weights.make_images(21, 2))
    caps = torch.from_numpy(weights.make_captions(22, 2, 3, V))
    lrp_maps, _ = eng.explain_batch(imgs, caps)
    gb_maps, gb_words, d_feat, _, enc = eng.explain_batch_guided(imgs, caps, return_features=True)
    assert lrp_maps.shape == gb_maps.shape == (2, 3, 3, 224, 224)
    sdt = O.state_to_torch(sd)
    for b in range(2):
        w_maps, w_rw, w_df, w_tr = O.gridtd_guided_explain_caption(sdt, imgs[b:b + 1], caps[b].numpy(), return_feat=True)
        # d_feat carries the gate [features > 0] (:1674): encoder outputs within rounding of zero may be gated
        # differently by the GPU and the CPU forward, so compare where both gates agree (and they must almost always)
        same = (enc["feats"][b].cpu() > 0) == (w_tr["F_pix"] > 0)
        assert (~same).float().mean().item() < 1e-4
        for t in range(3):
            want = w_df[t][0].reshape(512, 196).t()
            assert rel_err(d_feat[b, t].cpu() * same, want * same) < 2e-4, (b, t)
            assert np.abs(gb_words[b, t, :t + 1].cpu().numpy() - w_rw[t].numpy()).max() < 1e-4
            assert_close_modulo_pool_ties(gb_maps[b, t].cpu(), w_maps[t][0], what=(b, t), **E2E)


def test_guided_explainer_class(case):
    import types
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import ExplainiGridTDGuidedGradient
    g, sd, eng, img, cap = case
    V = int(g["V"])
    args = types.SimpleNamespace(embed_dim=512, hidden_dim=512, encoder='vgg16', weight='', save_path='/tmp',
                                 dataset='synthetic', height=224, width=224)
    ex = ExplainiGridTDGuidedGradient(args, weights.make_word_map(V), model={k: torch.from_numpy(v) for k, v in sd.items()})
    maps, rws = ex.explain_caption(img, caption_encode=[int(c) for c in g["caption"]])
    assert len(maps) == 3 and maps[1].shape == (1, 3, 224, 224)
    for t in range(3):
        assert np.abs(rws[t].cpu().numpy() - g[f"r_words_{t}"]).max() < 5e-5
        assert_close_modulo_pool_ties(maps[t][..., ::4, ::4].cpu(), g[f"map_sub4_{t}"], what=t, **E2E)


def test_guided_gradcam_kernel_is_the_expansion_operator():
    """`lrpx_guided_gradcam` alone: guided (N,3,224,224) x expand(cam (N,196)) against the oracle's direct evaluation of
    pyramid_expand (bilinear resize + scipy gaussian_filter) - strict, elementwise"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import ops
    from oracle import lrp_oracle as O
    g = torch.Generator().manual_seed(2)
    guided = torch.randn(3, 3, 224, 224, generator=g)
    cam = torch.relu(torch.randn(3, 196, generator=g))
    cam[1] = 0
    got = ops.guided_gradcam(guided.cuda(), cam.cuda(), 14).cpu()
    for n in range(3):
        want = guided[n] * O.pyramid_expand(cam[n].view(14, 14), 16)
        assert (got[n] - want).abs().max().item() <= 2e-6 * max(want.abs().max().item(), 1e-30) or want.abs().max() == 0
    assert got[1].abs().max().item() == 0


def test_guided_gradcam_vs_reference_fixture():
    """ExplainGridTDGuidedGradCam (models/gridTDmodel.py:1796-1836): batched engine and drop-in class against the fixture
    made by the reference's class (tests/golden/guided_gradcam_T3.npz; its skimage call served by the restated
    pyramid_expand, see make_golden.py).  Word 1 is the all-negative Grad-CAM case: a zero map."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import types
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine, ExplainGridTDGuidedGradCam
    g = np.load(os.path.join(GOLDEN, "guided_gradcam_T3.npz"))
    V = int(g["grid_V"])
    sd = weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=V)
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    cap = torch.from_numpy(g["grid_caption"]).view(1, -1)
    eng = GridTDEngine(sd)
    maps, r_words = eng.explain_batch_guided(img, cap, gradcam=True)
    ex = ExplainGridTDGuidedGradCam(types.SimpleNamespace(height=224, width=224), weights.make_word_map(V), model=sd)
    dmaps, drw = ex.explain_caption(img, caption_encode=g["grid_caption"].tolist())
    for t in range(3):
        assert torch.equal(dmaps[t][0], maps[0, t])                       # drop-in == batched engine
        scale = g[f"grid_map_stats_{t}"][1]
        if scale == 0:
            assert maps[0, t].abs().max().item() == 0
            continue
        assert_close_modulo_pool_ties(maps[0, t][None, :, ::4, ::4].cpu(), g[f"grid_map_sub4_{t}"], what=("ggc", t), **E2E)
        assert np.abs(r_words[0, t, :t + 1].cpu().numpy() - g[f"grid_r_words_{t}"]).max() < 5e-5
    assert_close_modulo_pool_ties(maps[0, 2].cpu(), g["grid_map_full_2"][0], what="ggc full", **E2E)
    # explain_cnn of the drop-in: one word's decoder gradient -> the same map
    d_feat, _ = ex.explain_caption_wordt(2)
    assert torch.equal(ex.explain_cnn(d_feat)[0], maps[0, 2])


@pytest.mark.parametrize("plain", [False, True])
def test_image_gradient_chain_agrees_across_conv_modes(case, plain):
    """guided backprop / plain gradient through VGG16 on the same trace in the bf16x6 kernels (mode 1, fp32-grade), the
    f16x3 kernels (mode 2: pool-backward kernels, operand maxima recorded by the producers) and the default fp16 + fp8
    kernels (mode 3: pooled-input staging under the pools, 8-wave workgroups, matrix-core first layer): 1e-4 per map, on 5
    maps of 2 images (map counts that do not divide the per-XCD tile ranges) with scales 1e6 apart"""
    from test_gpu_vgg import to_nhwc
    from lrp_amd import weights
    g, sd, eng, img, cap = case
    vgg = eng.vgg
    imgs = torch.from_numpy(weights.make_images(77, 2)).cuda()
    gen = torch.Generator().manual_seed(5)
    d = torch.randn(5, 512, 14, 14, generator=gen) * torch.exp(2 * torch.randn(5, 512, 14, 14, generator=gen))
    d = (d * torch.logspace(0, -6, 5).view(-1, 1, 1, 1)).contiguous()
    m2i = torch.tensor([0, 1, 1, 0, 1], dtype=torch.int32, device="cuda")
    fn = vgg.gradient if plain else vgg.guided_backprop
    keep = vgg.conv_mode
    outs = {}
    try:
        for mode in (1, 2, 3):
            vgg.conv_mode = mode
            vgg.forward(imgs)
            if mode == 1:
                trace1 = vgg.trace.clone()
            else:
                vgg.trace.copy_(trace1)          # the same activations / pool winners / ReLU masks for every mode
            outs[mode] = fn(to_nhwc(d).cuda(), m2i).cpu()
    finally:
        vgg.conv_mode = keep
    for mode in (2, 3):
        for i in range(5):
            assert rel_err(outs[mode][i], outs[1][i]) < 1e-4, (mode, i, rel_err(outs[mode][i], outs[1][i]))


@pytest.mark.parametrize("target", [63.0, 65.0])
def test_gradient_chain_at_the_row_spread_boundary(target):
    """the opt-in mode-3 image-gradient chains share one fp6 block scale over a 16-row weight slice; `ops.Vgg16`
    moves them to mode 2 when the rows of a slice differ by more than GRAD_SPREAD_MAX (64).  Both sides of the boundary against the
    CPU oracle (`O.vgg_guided_backprop`, models/gridTDmodel.py:1677-1723) on the oracle's own activations: one row of every slice is
    scaled so that the largest in-slice ratio of row maxima is 63 (mode 3 runs, must hold 1e-4) or 65 (falls back to mode 2)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import test_gpu_vgg as TV
    from lrp_amd import ops, weights
    from oracle import lrp_oracle as O
    sd = weights.make_gridtd_state(seed=9, vocab_size=64)
    for k in [k for k in sd if k.startswith("img_encoder.encoder.") and k.endswith(".weight")][1:]:
        w = sd[k]
        rmax = np.abs(w).reshape(w.shape[0], -1).max(1)
        for s0 in range(0, w.shape[0], 16):
            sl = slice(s0, s0 + 16)
            # row s0 becomes the slice's smallest: slice maximum / target (the other rows of a kaiming slice lie within 2x of it)
            w[s0] *= (rmax[sl].max() / target) / rmax[s0]
    img = torch.from_numpy(weights.make_images(10, 1))
    vgg = TV._vgg(ops, sd)
    assert abs(vgg.row_spread / target - 1.0) < 1e-3 and vgg.grad_mode2 == (target > vgg.GRAD_SPREAD_MAX), (vgg.row_spread, target)
    vgg.conv_mode = 3
    vgg.forward(img.cuda())
    TV._inject_oracle_trace(vgg, sd, img)
    sdt = O.state_to_torch(sd)
    _, _, saved = O.vgg_forward(sdt, img)
    gen = torch.Generator().manual_seed(11)
    d = (torch.randn(2, 512, 14, 14, generator=gen) * torch.exp(torch.randn(2, 512, 14, 14, generator=gen))).contiguous()
    got = vgg.guided_backprop(TV.to_nhwc(d).cuda(), torch.zeros(2, dtype=torch.int32, device="cuda")).cpu()
    for i in range(2):
        want = O.vgg_guided_backprop(sdt, saved, d[i:i + 1])
        e = rel_err(got[i:i + 1], want)
        print(f"in-slice row spread {target:.0f} ({'mode 2 fallback' if vgg.grad_mode2 else 'mode 3'}): guided-backprop map {i} vs the oracle {e:.2e}")
        assert e < 1e-4, (target, i, e)
