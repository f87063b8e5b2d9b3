"""Decoder relevance at the HEADLINE caption length (T = 20) inside full-size batches.

The lock-step row machinery of the engines (`idx[s]` row remapping, `wacc` of shape (rows, T, 512), r_words with its
~100x cancellation) runs its full depth only at T = 20; the T = 3 goldens never reach it.  tests/golden/t20.npz holds
the reference's own `explain_caption_wordt` outputs (models/gridTDmodel.py:1014-1135, models/aoamodel.py:1064-1156)
for every word of two images per model; here those two images sit at positions 3 and 11 of a B = 16 batch of other
images and captions (BASELINE configs 2 / 3; B = 32 for the bottom-up config 5), and their 2 x 20 rows of the batched
engines are compared with the reference: r_feat <= 1e-4 of its maximum (channel subsample of every word + two full
rows + L2 / sum statistics of all channels), r_words <= 1e-5 for gridTD.  AoA r_words at T = 20: two of the 40 golden
rows (image 0, words 17 and 19) are ill-conditioned - their normalising entry is a 512-term sum that cancels to ~1/200 of
its terms: the reference's own fp32 value sits 1.0e-5 / 4.3e-5 from the fp64 evaluation of its formula on the same trace,
the CPU oracle (same forward, another summation order) 2.4e-5 from the reference (tests/test_oracle_golden.py), and the
GPU forward (fp32-grade, but a sequential K-chain of up to 288 MFMA accumulations: features 3e-6 of max from fp64 against
7e-7 for oneDNN) moves them by 0.4e-4 / 1.5e-4 (tools/t20_probe.py; the same with the exact-split bf16x6 forward).  So
for the AoA models: every row <= 5e-4, at most 10 % of the rows above 1e-5 (observed: 2 of 40; the other 38 <= 2.5e-6)."""
import os

import numpy as np
import pytest
import torch

import lrp_amd  # noqa: F401
from conftest import GOLDEN, rel_err, cosine, assert_close_modulo_pool_ties

pytestmark = pytest.mark.gpu
TOL = 1e-4
POS = (3, 11)          # batch positions of the two golden images


@pytest.fixture(scope="module")
def g20():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return np.load(os.path.join(GOLDEN, "t20.npz"))


def _batch(g, B, cap_key, V, cap_seed):
    from lrp_amd import weights
    T = int(g["T"])
    caps = weights.make_captions(cap_seed, B, T, V)
    for k, p in enumerate(POS):
        caps[p] = g[cap_key][k]
    return T, torch.from_numpy(caps)


def _images(g, B):
    from lrp_amd import weights
    imgs = weights.make_images(77, B)
    gold = weights.make_images(int(g["img_seed"]), int(g["n_img"]))
    for k, p in enumerate(POS):
        imgs[p] = gold[k]
    return torch.from_numpy(imgs)


def _check_rows(g, prefix, r_feat, r_words, T, C, stride, tol_words, layout):
    """r_feat (B,T,P,C) / r_words (B,T,T) of the engine vs the golden rows `prefix{k}_*`.
    layout 'chw': golden sub = (C/stride, 14, 14); 'pc': golden sub = (P, C/stride)."""
    worst_f, worst_w, n_soft = 0.0, 0.0, 0
    if not isinstance(tol_words, tuple):
        tol_words = (tol_words, tol_words)          # (soft bound that <= 10 % of the rows may exceed, hard bound)
    for k, p in enumerate(POS):
        for t in range(T):
            st = g[f"{prefix}{k}_r_feat_stats_{t}"]                       # sum, absmax, L2, L1 over ALL channels
            got = r_feat[p, t].double()                                   # (P,C)
            sub = torch.from_numpy(g[f"{prefix}{k}_r_feat_sub_{t}"]).double()
            want = sub.reshape(sub.shape[0], -1).t() if layout == "chw" else sub
            off = t % stride
            e = ((got[:, off::stride] - want).abs().max() / st[1]).item()
            worst_f = max(worst_f, e)
            assert e < TOL, (prefix, k, t, e)
            assert abs(got.norm().item() - st[2]) <= 1e-4 * st[2], (prefix, k, t, "L2")
            assert abs(got.sum().item() - st[0]) <= 1e-4 * st[3], (prefix, k, t, "sum")
            assert abs(got.abs().max().item() - st[1]) <= 1e-4 * st[1], (prefix, k, t, "absmax")
            w = np.abs(r_words[p, t, :t + 1].numpy() - g[f"{prefix}{k}_r_words_{t}"]).max()
            worst_w = max(worst_w, float(w))
            n_soft += int(w >= tol_words[0])
            assert w < tol_words[1], (prefix, k, t, w)
            if t + 1 < T:
                assert r_words[p, t, t + 1:].abs().max().item() == 0      # nothing beyond the word's own prefix
        tf = T - 1 - 9 * k
        full = torch.from_numpy(g[f"{prefix}{k}_r_feat_full_{tf}"])
        want = full.reshape(full.shape[0], -1).t() if layout == "chw" else full
        assert rel_err(r_feat[p, tf], want) < TOL, (prefix, k, "full")
        assert cosine(r_feat[p, tf], want) > 0.99999
    assert n_soft <= 0.1 * len(POS) * T, (prefix, n_soft)
    print(f"T=20 {prefix}: worst r_feat error {worst_f:.2e} of max|R|, worst r_words error {worst_w:.2e} "
          f"({n_soft} of {len(POS) * T} rows above {tol_words[0]:.0e})")


def test_gridtd_t20_rows_inside_b16_batch(g20):
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    g = g20
    V, B = int(g["grid_V"]), 16
    T, caps = _batch(g, B, "grid_caption", V, 61)
    eng = GridTDEngine(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=V))
    maps, r_words, r_feat, tr, enc = eng.explain_batch(_images(g, B), caps, accumulate=True, return_features=True)
    torch.cuda.synchronize()
    _check_rows(g, "grid", r_feat.cpu(), r_words.cpu(), T, 512, 32, 1e-5, "chw")
    # pixel maps of golden image 0: the reference's running sums over all 20 words (lrp_wrapper.py:64-82 quirk), GPU forward
    # included -> modulo max-pool tie flips (conftest)
    m = maps[POS[0]].cpu()
    for t in range(T):
        assert_close_modulo_pool_ties(m[t][None, :, ::8, ::8], g[f"grid0_map_sub8_{t}"], what=("map", t))
        st = g[f"grid0_map_stats_{t}"]
        assert abs(m[t].double().norm().item() - st[2]) <= 2e-3 * st[2]


@pytest.mark.parametrize("head", [0, 3])
def test_aoa_t20_rows_inside_b16_batch(g20, head):
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    g = g20
    V, B = int(g["aoa_V"]), 16
    T, caps = _batch(g, B, "aoa_caption", V, 62)
    eng = AOAEngine(weights.make_aoa_state(seed=int(g["seed"]), vocab_size=V))
    enc = eng.encode(_images(g, B).cuda())
    tr = eng.trace(enc, caps.cuda(), predictions=False)
    r_feat, r_words, _ = eng.relevance(enc, tr, head)
    torch.cuda.synchronize()
    r_feat, r_words = r_feat.view(B, T, 196, 512).cpu(), r_words.view(B, T, T).cpu()
    if head == 0:
        _check_rows(_Prefixed(g, "_h0"), "aoa", r_feat, r_words, T, 512, 32, (1e-5, 5e-4), "chw")
    else:      # head 3 was generated for golden image 1 only
        gg = _Prefixed(g, "_h3")
        k, p = 1, POS[1]
        for t in range(T):
            st = gg[f"aoa{k}_r_feat_stats_{t}"]
            sub = torch.from_numpy(gg[f"aoa{k}_r_feat_sub_{t}"]).double()
            want = sub.reshape(sub.shape[0], -1).t()
            e = ((r_feat[p, t].double()[:, (t % 32)::32] - want).abs().max() / st[1]).item()
            assert e < TOL, (t, e)
            assert np.abs(r_words[p, t, :t + 1].numpy() - gg[f"aoa{k}_r_words_{t}"]).max() < 1e-5   # image 1: well conditioned


class _Prefixed:
    """view of the golden file whose keys `aoa{k}_x` resolve to `aoa{k}{tag}_x` (the AoA rows are stored per head)"""

    def __init__(self, g, tag):
        self.g, self.tag = g, tag

    def __getitem__(self, key):
        head, rest = key.split("_", 1)
        return self.g[f"{head}{self.tag}_{rest}"]


def test_aoa_bottom_up_t20_rows_inside_b32_batch(g20):
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    g = g20
    V, B = int(g["aoa_V"]), 32
    T, caps = _batch(g, B, "bu_caption", V, 63)
    feats = weights.make_bu_features(78, B)
    gold = weights.make_bu_features(int(g["img_seed"]), int(g["n_img"]))
    for k, p in enumerate(POS):
        feats[p] = gold[k]
    eng = AOAEngine(weights.make_aoa_state(seed=int(g["seed"]), vocab_size=V, feat_dim=2048, with_encoder=False))
    r_feat, r_words = eng.explain_batch(caps, 0, features=torch.from_numpy(feats))
    torch.cuda.synchronize()
    _check_rows(g, "bu", r_feat.cpu(), r_words.cpu(), T, 2048, 64, (1e-5, 5e-4), "pc")
