"""End-to-end parity of BOTH explainers against the reference's own golden vectors in EVERY matrix-core mode of the VGG16
chains (parity tests parametrised over the headline mode).  Mode 1 - exact bf16 splits, the process
default and the mode `bench.py` quotes - runs the round-6 B6 kernels (fused multiplicands, pooled-input staging); modes 2 / 3
are the opt-in speed modes; mode 0 the fp32 MFMA.  Reference: LRPtools/lrp_wrapper.py:63-87 (compute_lrp),
lrp_modules.py:124-195 (Conv2d alpha1beta0, Pool2d), models/gridTDmodel.py:1014-1156, models/aoamodel.py:1064-1181.
Tolerances as in SURVEY 8(d); pixel maps modulo max-pool tie flips (conftest.assert_close_modulo_pool_ties)."""
import os

import numpy as np
import pytest
import torch

import lrp_amd  # noqa: F401
from conftest import GOLDEN, rel_err, cosine, assert_close_modulo_pool_ties, forward_flips

pytestmark = pytest.mark.gpu
MODES = [1, 0, 2, 3]          # the headline mode first


@pytest.fixture(scope="module")
def gridtd():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    g = np.load(os.path.join(GOLDEN, "gridtd_T3.npz"))
    sd = weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["V"]))
    return g, GridTDEngine(sd), torch.from_numpy(weights.make_images(int(g["seed"]), 1)), torch.from_numpy(g["caption"]).view(1, -1), sd


@pytest.fixture(scope="module")
def aoa():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    g = np.load(os.path.join(GOLDEN, "aoa_T3.npz"))
    sd = weights.make_aoa_state(seed=int(g["seed"]), vocab_size=int(g["V"]))
    return g, AOAEngine(sd), torch.from_numpy(weights.make_images(int(g["seed"]), 1)), torch.from_numpy(g["caption"]).view(1, -1)


def test_process_default_is_the_exact_split_mode():
    """the library's default arithmetic is no narrower than the reference's fp32 (include/lrpx.h, lrpx_set_conv_mode)"""
    from lrp_amd import _lib
    want = int(os.environ.get("LRPX_CONV_MODE", "1"))
    assert _lib.load().lrpx_set_conv_mode(-1) == want


def test_decoder_arithmetic_follows_the_conv_mode(gridtd, aoa):
    """VERDICT r5 item 1 (nothing on the headline path narrower than fp32): in the exact modes 0 / 1 the decoders' GEMMs take the fp32
    kernels, the fp16 split products (csrc/dense_f16x3.hip: 22 operand bits behind a per-row scale) only with the opt-in modes 2 / 3
    (ops.decoder_f16); the two agree to ~1e-6 of a row's maximum on the golden caption."""
    if "LRPX_DECODER_F16" in os.environ:
        pytest.skip("LRPX_DECODER_F16 overrides the rule under test")
    from lrp_amd import _lib
    for eng in (gridtd[1], aoa[1]):
        keep = eng.vgg.conv_mode
        try:
            for mode, want in ((0, False), (1, False), (2, True), (3, True)):
                eng.vgg.conv_mode = mode
                assert eng._f16() is want, (type(eng).__name__, mode)
            eng.vgg.conv_mode = None
            assert eng._f16() is (_lib.load().lrpx_set_conv_mode(-1) >= 2)
        finally:
            eng.vgg.conv_mode = keep
    g, eng, img, cap, sd = gridtd
    enc = eng.encode(img)
    got = {}
    for f16 in (False, True):
        eng.force_f16 = f16
        try:
            tr = eng.trace(enc, cap.cuda(), predictions=False)
            r_feat, r_words, _ = eng.relevance(enc, tr)
            got[f16] = (r_feat.clone(), r_words.clone())
        finally:
            eng.force_f16 = None
    d = (got[True][0] - got[False][0]).flatten(1).abs().amax(1) / got[False][0].flatten(1).abs().amax(1)
    print(f"gridTD decoder fp32 vs f16x3 GEMMs: r_feat worst row {d.max().item():.2e} of max|R|, r_words {(got[True][1] - got[False][1]).abs().max().item():.2e}")
    assert d.max().item() < 5e-6 and (got[True][1] - got[False][1]).abs().max().item() < 5e-6
    assert d.max().item() > 0           # (they ARE different kernels)


@pytest.mark.parametrize("mode", MODES)
def test_gridtd_explain_vs_reference_in_every_mode(gridtd, mode):
    g, eng, img, cap, sd = gridtd
    keep = eng.vgg.conv_mode
    eng.vgg.conv_mode = mode
    try:
        maps, r_words, r_feat, tr, enc = eng.explain_batch(img, cap, accumulate=True, return_features=True)
        torch.cuda.synchronize()
        # the discrete decisions of THIS forward trace that differ from the reference's (oneDNN) forward: what the end-to-end deviation of
        # the pixel maps is made of; printed per run so that the caveat stays quantified
        flips = forward_flips(eng.vgg, sd, img)
    finally:
        eng.vgg.conv_mode = keep
    n_relu = sum(f for _, kind, _, f in flips if kind == "conv")
    n_pool = sum(f for _, kind, _, f in flips if kind == "pool")
    d_full = (maps[0, 2].cpu() - torch.from_numpy(g["map_full_2"][0])).abs() / np.abs(g["map_full_2"][0]).max()
    print(f"conv mode {mode}, golden image: {n_relu} ReLU sign flips, {n_pool} live pool-winner flips against the oneDNN forward "
          f"(per pool: {[f for _, kind, _, f in flips if kind == 'pool']}); full map: worst pixel {d_full.max().item():.2e}, "
          f"{(d_full > 1e-4).float().mean().item():.2e} of the pixels above 1e-4 of max|R|")
    maps, r_words, r_feat = maps.cpu(), r_words.cpu(), r_feat.cpu()
    for t in range(3):
        want = torch.from_numpy(g[f"r_feat_{t}"])[0].reshape(512, 196).t()
        assert rel_err(r_feat[0, t], want) < 1e-4, (mode, t)
        assert cosine(r_feat[0, t], want) > 0.99999
        assert np.abs(r_words[0, t, :t + 1].numpy() - g[f"r_words_{t}"]).max() < 1e-5
        assert_close_modulo_pool_ties(maps[0, t][None, :, ::4, ::4], g[f"map_sub4_{t}"], what=(mode, t),
                                      frac=1e-2, hard=8e-3, l2=1.5e-3)
    assert_close_modulo_pool_ties(maps[0, 2], g["map_full_2"][0], what=(mode, "full"), frac=1e-2, hard=8e-3, l2=1.5e-3)
    # BASELINE's absolute bound (per-pixel relevance is O(1e-6) with random-init weights: SURVEY 7)
    assert (maps[0, 2] - torch.from_numpy(g["map_full_2"][0])).abs().max() < 1e-4


@pytest.mark.parametrize("mode", MODES)
def test_aoa_explain_vs_reference_in_every_mode(aoa, mode):
    g, eng, img, cap = aoa
    head = 5
    keep = eng.vgg.conv_mode
    eng.vgg.conv_mode = mode
    try:
        maps, r_words, r_feat, tr, enc = eng.explain_batch(cap, head, images=img, accumulate=True, return_features=True)
        torch.cuda.synchronize()
    finally:
        eng.vgg.conv_mode = keep
    maps, r_words, r_feat = maps.cpu(), r_words.cpu(), r_feat.cpu()
    for t in range(3):
        want = torch.from_numpy(g[f"h{head}_r_feat_{t}"])[0].reshape(512, 196).t()
        assert rel_err(r_feat[0, t], want) < 1e-4, (mode, t)
        assert np.abs(r_words[0, t, :t + 1].numpy() - g[f"h{head}_r_words_{t}"]).max() < 1e-5
        assert_close_modulo_pool_ties(maps[0, t][None, :, ::4, ::4], g[f"h{head}_map_sub4_{t}"], what=(mode, head, t),
                                      frac=1e-2, hard=8e-3, l2=1.5e-3)


@pytest.mark.parametrize("mode", MODES)
def test_chain_on_the_reference_activations_strict_in_every_mode(mode):
    """the strict 1e-4 bound of SURVEY 8(d) on IDENTICAL activations: the oracle's fp32 forward (activations, Z+) is injected into
    the trace, so no pool winner can differ; the reference decoder's own r_feat (golden) -> pixel maps against the reference's maps
    (running sums: lrp_wrapper.py:64-82), per-map sums against the reference's (conservation), in the given conv mode"""
    import test_gpu_vgg as TV
    from lrp_amd import ops, weights
    g = np.load(os.path.join(GOLDEN, "gridtd_T3.npz"))
    sd = weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["V"]))
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    vgg = TV._vgg(ops, sd)
    vgg.conv_mode = mode
    vgg.forward(img.cuda())
    TV._inject_oracle_trace(vgg, sd, img)
    r_feat = torch.cat([torch.from_numpy(g[f"r_feat_{t}"]) for t in range(3)])
    maps = vgg.relevance(TV.to_nhwc(r_feat).cuda(), torch.zeros(3, dtype=torch.int32, device="cuda"))
    ops.check_relevance(maps, finite=True, nonzero=True)
    cum = ops.cumsum_maps(maps, 1, 3).cpu()
    for t in range(3):
        scale = g[f"map_stats_{t}"][1]
        assert np.abs(cum[t:t + 1, :, ::4, ::4].numpy() - g[f"map_sub4_{t}"]).max() / scale < 1e-4, (mode, t)
        assert abs(cum[t].double().sum().item() - g[f"map_stats_{t}"][0]) <= 1e-3 * abs(g[f"map_stats_{t}"][0])
    err = rel_err(cum[2:3], g["map_full_2"])
    print(f"conv mode {mode}: full map vs the reference on its own activations {err:.2e}")
    assert err < (2e-5 if mode <= 2 else 1e-4)          # the exact-split / fp32 / f16x3 chains sit at ~1e-6
    assert cosine(cum[2:3], g["map_full_2"]) > 0.99999
