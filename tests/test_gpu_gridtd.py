"""GPU parity of the gridTD decoder half (trace + relevance) and of the whole explain_batch pipeline,
against the reference's golden vectors and the CPU oracle.  Tolerances as in SURVEY §8(d)."""
import os

import numpy as np
import pytest
import torch

import lrp_amd  # noqa: F401
from conftest import GOLDEN, rel_err, cosine, assert_close_modulo_pool_ties

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def case():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    g = np.load(os.path.join(GOLDEN, "gridtd_T3.npz"))
    sd = weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=int(g["V"]))
    eng = GridTDEngine(sd)
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    cap = torch.from_numpy(g["caption"]).view(1, -1)
    maps, r_words, r_feat, tr, enc = eng.explain_batch(img, cap, accumulate=True, return_features=True)
    torch.cuda.synchronize()
    return g, sd, eng, img, cap, maps.cpu(), r_words.cpu(), r_feat.cpu(), tr, enc


def test_trace_vs_reference(case):
    g, sd, eng, img, cap, maps, r_words, r_feat, tr, enc = case
    pairs = dict(h1t="h1", c1t="c1", h2t="h2", c2t="c2", g1t="g1", g2t="g2", i1t_act="i1", f1t_act="f1",
                 i2t_act="i2", f2t_act="f2", st="s", context="ctx", context_hat="ctx_hat", alphas="alpha", betas="beta")
    for ref_name, mine in pairs.items():
        assert rel_err(tr[mine][0].cpu(), g["tr_" + ref_name]) < 1e-4, ref_name
    pred = eng.logits(tr["hc"].view(-1, 512)).cpu()
    assert rel_err(pred[:, ::97], g["tr_predictions"]) < 1e-4


def test_feature_relevance_and_word_relevance_vs_reference(case):
    g, sd, eng, img, cap, maps, r_words, r_feat, tr, enc = case
    for t in range(3):
        want = torch.from_numpy(g[f"r_feat_{t}"])[0].reshape(512, 196).t()      # (P,C)
        assert rel_err(r_feat[0, t], want) < TOL, t
        assert cosine(r_feat[0, t], want) > 0.99999
        assert np.abs(r_words[0, t, :t + 1].numpy() - g[f"r_words_{t}"]).max() < 1e-5


def test_maps_end_to_end_vs_reference(case):
    """whole pipeline on the GPU; compared modulo max-pool tie flips (conftest.assert_close_modulo_pool_ties)"""
    g, sd, eng, img, cap, maps, r_words, r_feat, tr, enc = case
    for t in range(3):
        assert_close_modulo_pool_ties(maps[0, t][None, :, ::4, ::4], g[f"map_sub4_{t}"], what=t)
    assert_close_modulo_pool_ties(maps[0, 2], g["map_full_2"][0], what="full")
    assert (maps[0, 2] - torch.from_numpy(g["map_full_2"][0])).abs().max() < 1e-4


def test_batch_of_images_vs_oracle():
    """B=2 images with different captions, T=4: every (image, word) row against the per-image oracle;
    decoder relevance strict (1e-4), pixel maps end-to-end (1e-3)."""
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    from oracle import lrp_oracle as O
    V = 503
    sd = weights.make_gridtd_state(seed=7, vocab_size=V, vgg_bias_std=0.02)
    sdt = O.state_to_torch(sd)
    eng = GridTDEngine(sd)
    img = torch.from_numpy(weights.make_images(11, 2))
    cap = torch.from_numpy(weights.make_captions(12, 2, 4, V))
    maps, r_words, r_feat, tr, enc = eng.explain_batch(img, cap, return_features=True)
    maps, r_words, r_feat = maps.cpu(), r_words.cpu(), r_feat.cpu()
    for b in range(2):
        w_maps, w_rw, w_rf, _ = O.gridtd_explain_caption(sdt, img[b:b + 1], cap[b].numpy(), return_feat=True,
                                                         accumulate=False)
        for t in range(4):
            want_rf = w_rf[t][0].reshape(512, 196).t()
            assert rel_err(r_feat[b, t], want_rf) < 2e-4, (b, t)   # forward differences (GPU vs CPU conv) included
            assert np.abs(r_words[b, t, :t + 1].numpy() - w_rw[t].numpy()).max() < 1e-4
            assert_close_modulo_pool_ties(maps[b, t], w_maps[t][0], what=(b, t))


def test_greedy_tokens_bit_exact(case):
    """config 1: greedy decoding with the model's own forward; integer token ids must match the
    reference's `greedy_search` exactly (tests/golden/greedy_cfg1.npz)."""
    g, sd, eng, img, cap, *_ = case
    gg = np.load(os.path.join(GOLDEN, "greedy_cfg1.npz"))
    V = int(gg["V"])
    enc = eng.encode(img.cuda())
    toks = eng.greedy(enc, len(gg["tokens"]), V - 2, V - 1)
    assert toks[0].cpu().tolist() == [int(x) for x in gg["tokens"]]


def test_sample_lrp_tokens_bit_exact():
    """LRP-inference decoding (GridTDEngine.sample_lrp = `GridTDModel.sample_lrp` greedy, models/gridTDmodel.py:631-702)
    against the reference's own output (tests/golden/sample_lrp.npz): token ids bit-exact, log-probabilities to 1e-4
    absolute; case 2 has a stop word (exempt from the re-weighting) and an <end> that is hit (zero padding)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    g = np.load(os.path.join(GOLDEN, "sample_lrp.npz"))
    V, L = int(g["V"]), int(g["max_len"])
    eng = GridTDEngine(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=V))
    imgs = torch.from_numpy(weights.make_images(int(g["seed"]) + 3, g["seq"].shape[0]))
    wm = weights.make_word_map(V)
    enc = eng.encode(imgs.cuda())
    for seq_k, lp_k, skip_k, end_id in (("seq", "logprobs", "skip", wm['<end>']),
                                        ("seq2", "logprobs2", "skip2", int(g["end2"]))):
        seq, lps = eng.sample_lrp(enc, L, wm['<start>'], end_id, g[skip_k].tolist())
        assert seq.cpu().tolist() == g[seq_k].tolist()
        assert np.abs(lps.cpu().numpy() - g[lp_k]).max() < 1e-4


def test_sample_lrp_stops_writing_once_all_finished():
    """the reference leaves the loop when every sequence has produced <end> (:699-700): later columns stay 0"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    g = np.load(os.path.join(GOLDEN, "sample_lrp.npz"))
    V = int(g["V"])
    eng = GridTDEngine(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=V))
    imgs = torch.from_numpy(weights.make_images(int(g["seed"]) + 3, 2))
    wm = weights.make_word_map(V)
    first = int(g["seq"][0, 0])                         # both images start with this word: let it play <end>
    assert int(g["seq"][1, 0]) == first
    seq, lps = eng.sample_lrp(eng.encode(imgs.cuda()), 5, wm['<start>'], first, g["skip"].tolist())
    assert seq.abs().sum().item() == 0
    assert (lps[:, 0] < 0).all() and lps[:, 1:].abs().sum().item() == 0


def test_explain_stream_matches_serial():
    """independent batches in flight on separate HIP streams (GridTDEngine.explain_stream, shared weights, own
    buffers per stream) give bit-identical maps and word relevances to explaining them one after the other"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    eng = GridTDEngine(weights.make_gridtd_state(seed=3, vocab_size=50))
    batches = [(torch.from_numpy(weights.make_images(10 + i, 2)), torch.from_numpy(weights.make_captions(20 + i, 2, 3, 50)))
               for i in range(5)]
    serial = [tuple(t.clone() for t in eng.explain_batch(im, cp)) for im, cp in batches]
    torch.cuda.synchronize()
    piped = list(eng.explain_stream(batches, depth=3))
    assert len(piped) == len(serial)
    for (m0, w0), (m1, w1) in zip(serial, piped):
        assert torch.equal(m0, m1) and torch.equal(w0, w1)
    again = eng.explain_batch(*batches[0])          # the engine is usable serially afterwards
    torch.cuda.synchronize()
    assert torch.equal(again[0], serial[0][0])


def _check_forwardlrp(g, tag, sfx, preds, wpreds, L):
    preds, wpreds = preds.cpu(), wpreds.cpu()
    assert tuple(preds.shape) == tuple(wpreds.shape) == (int(g["batch"]), L, int(g[f"{tag}_V"]))
    assert rel_err(preds[:, :, ::13], g[f"{tag}_pred_sub{sfx}"]) < 1e-4
    assert rel_err(wpreds[:, :, ::13], g[f"{tag}_wpred_sub{sfx}"]) < 1e-4
    assert preds.argmax(-1).tolist() == g[f"{tag}_pred_argmax{sfx}"].tolist()          # token ids: bit-exact
    assert wpreds.argmax(-1).tolist() == g[f"{tag}_wpred_argmax{sfx}"].tolist()
    assert rel_err(preds[1, L - 1], g[f"{tag}_pred_row{sfx}"]) < 1e-4
    assert rel_err(wpreds[1, L - 1], g[f"{tag}_wpred_row{sfx}"]) < 1e-4


def test_forwardlrp_context_vs_reference():
    """the forward half of LRP-inference fine-tuning (`GridTDModel.forwardlrp_context`, models/gridTDmodel.py:579-630):
    raw and LRP-reweighted scores of every teacher-forced step against the reference's own outputs
    (tests/golden/forwardlrp.npz), incl. the case with a stop word (exempt rows: weights of 1)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    g = np.load(os.path.join(GOLDEN, "forwardlrp.npz"))
    V = int(g["grid_V"])
    eng = GridTDEngine(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=V))
    imgs = torch.from_numpy(weights.make_images(int(g["seed"]) + 5, int(g["batch"])))
    enc = eng.encode(imgs.cuda())
    caps = torch.from_numpy(g["grid_caption"])
    for sfx in ("", "2"):
        preds, wpreds, L = eng.forwardlrp_context(enc, caps, g["grid_lengths"].tolist(), g[f"grid_skip{sfx}"].tolist())
        assert L == int(g["grid_L"])
        _check_forwardlrp(g, "grid", sfx, preds, wpreds, L)


def test_beam_search_caption_bit_exact_and_drop_in_explains_it():
    """the caption the reference explains when none is given: `beam_search(beam_size=2, max_cap_length=50)`
    (models/gridTDmodel.py:935).  Token ids against the reference's own beam_search (tests/golden/beam.npz), bit-exact, for
    the natural run (cut at 20 tokens), an <end> that is reached, an <end> as first word (empty caption) and a dropped
    <unk>; the drop-in `explain_caption(img)` then explains exactly that caption."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import types
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine, ExplainGridTDAttention
    from lrp_amd.explainers.beam import caption_from_sequence
    from test_oracle_golden import _beam_cases
    g = np.load(os.path.join(GOLDEN, "beam.npz"))
    V, cases = _beam_cases(g, "grid")
    sd = weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=V)
    eng = GridTDEngine(sd)
    img = torch.from_numpy(weights.make_images(int(g["seed"]) + 7, 1)).cuda()
    enc = eng.encode(img)
    for key, wm in cases:
        seq = eng.beam_search(enc, int(g["grid_beam"]), int(g["grid_steps"]), wm['<start>'], wm['<end>'])
        assert caption_from_sequence(seq, wm)[1:] == g[f"grid_{key}"].tolist(), key
    ex = ExplainGridTDAttention(types.SimpleNamespace(height=224, width=224), cases[1][1], model=sd)   # <end> is reached
    maps, rw = ex.explain_caption(img)
    assert ex.beam_caption_encode[1:] == g["grid_sen_end"].tolist() and len(maps) == len(rw) == len(g["grid_sen_end"])
    want, _ = eng.explain_batch(img, torch.tensor([ex.beam_caption_encode], dtype=torch.int64), accumulate=True)
    assert torch.equal(torch.cat(maps), want[0])
    ex0 = ExplainGridTDAttention(types.SimpleNamespace(height=224, width=224), cases[2][1], model=sd)  # empty caption
    assert ex0.explain_caption(img) == ([], [])


def test_beam_topk_kernel_vs_host_reference():
    """`lrpx_beam_topk` (one step of the reference's beam search, models/gridTDmodel.py:437-444: the k best of cum[r] + log_softmax(x[r])
    over the live beams, flat index r * V + w) against the same selection on the host in float64: indices bit-exact incl. exact ties
    (lower flat index first), values to 1e-5; 1 - 4 live rows, k = 1 - 4, a vocabulary that is no multiple of the block."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import ctypes as C
    from lrp_amd import _lib
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    for n_rows, k, V in ((1, 2, 9586), (2, 2, 9586), (3, 3, 11027), (4, 4, 1001), (2, 1, 257)):
        x = torch.randn(n_rows, V, generator=g) * 3
        x[:, 7] = x[:, 11]                                   # exact ties inside a row ...
        if n_rows > 1:
            x[1] = x[0]                                      # ... and across rows (equal cumulative scores below)
        cum = torch.zeros(n_rows) if n_rows > 1 else None
        xd = x.cuda()
        idx = torch.zeros(4, dtype=torch.int64, device="cuda")
        val = torch.zeros(4, device="cuda")
        _lib.check(lib.lrpx_beam_topk(_lib.ptr(xd), V, n_rows, V, _lib.ptr(cum.cuda()) if cum is not None else None, k, _lib.ptr(idx),
                                      _lib.ptr(val), _lib.stream_ptr()))
        lp = torch.log_softmax(xd.cpu().double(), dim=1).flatten()
        order = sorted(range(n_rows * V), key=lambda f: (-lp[f].item(), f))[:k]
        # float32 scores may order two float64-distinct candidates differently only when they are equal in float32: compare values, and
        # indices where the float64 gap to the next candidate is resolvable
        got_i, got_v = idx[:k].cpu().tolist(), val[:k].cpu()
        for o in range(k):
            assert abs(got_v[o].item() - lp[order[o]].item()) < 1e-5, (n_rows, k, V, o)
        lp32 = (xd - torch.logsumexp(xd, dim=1, keepdim=True)).cpu().flatten()
        want32 = sorted(range(n_rows * V), key=lambda f: (-lp32[f].item(), f))[:k]
        assert got_i == want32 or got_i == order, (n_rows, k, V, got_i, want32, order)


@pytest.mark.parametrize("B", [3, 20, 50])
def test_fused_trace_steps_are_bit_identical_to_the_seven_launch_step(B):
    """The teacher-forced trace with the gate linears and their LSTM cells in one launch each and the next input row behind the second
    (lrpx_gridtd_fwd_steps with interleaved gate rows: 4 launches per time step) against the 7-launch step that follows the reference's
    loop statement by statement (models/gridTDmodel.py:952-1012): the same dot products in the same order and the same point-wise
    expressions, so EVERY trace tensor is bit-identical - both LSTM biases (the explainers' quirk and the model's), with and without
    the gradient explainers' extras, 1 .. 4 row tiles of 16 images."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    V, T = 467, 6
    eng = GridTDEngine(weights.make_gridtd_state(seed=41, vocab_size=V))
    imgs = torch.from_numpy(weights.make_images(42, B))
    with pytest.raises(TypeError, match="no CPU path"):          # a host tensor never reaches a kernel (the engines move their inputs)
        eng.vgg.forward(imgs)
    enc = eng.encode(imgs)
    cap = torch.from_numpy(weights.make_captions(43, B, T, V)).cuda()
    assert eng.fused_steps
    for grad in (False, True):
        a = eng.trace(enc, cap, predictions=True, grad=grad)
        eng.fused_steps = False
        try:
            b = eng.trace(enc, cap, predictions=True, grad=grad)
        finally:
            eng.fused_steps = True
        keys = [k for k, v in a.items() if torch.is_tensor(v) and not k.startswith("_") and k != "captions"]
        assert {"xh1", "xh2", "h1", "c1", "h2", "c2", "g1", "i1", "f1", "g2", "i2", "f2", "s", "ctx", "ctx_hat", "hc", "alpha", "beta",
                "logit", "pred"} <= set(keys), keys
        if grad:
            assert {"o1", "o2", "sgate"} <= set(keys), keys
        for k in keys:
            assert torch.equal(a[k], b[k]), (k, grad, B, (a[k] - b[k]).abs().max().item())


def test_one_image_alone_equals_the_same_image_inside_a_batch_of_16():
    """the drop-in's calling pattern is ONE image per call (evaluation.py:806-838); its results must be the bits the
    batched engine gives the same image inside a B = 16 batch - every kernel decision on the path depends on the LAYER or on the row,
    never on the batch: forward K splits per layer, decoder GEMMs with row-local operand scales, tile order hints that do not touch a
    result.  Trace, decoder relevance, r_words and the pixel maps of two images, bit for bit, in the default conv mode."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    V, B, T = 401, 16, 5
    eng = GridTDEngine(weights.make_gridtd_state(seed=6, vocab_size=V))
    imgs = torch.from_numpy(weights.make_images(51, B)).cuda()
    caps = torch.from_numpy(weights.make_captions(52, B, T, V)).cuda()
    maps, r_words, r_feat, tr, enc = eng.explain_batch(imgs, caps, accumulate=True, return_features=True)
    maps, r_words, r_feat = maps.clone(), r_words.clone(), r_feat.clone()
    feats, hc = enc["feats"].clone(), tr["hc"].clone()
    for b in (0, 11):
        m1, w1, f1, tr1, enc1 = eng.explain_batch(imgs[b:b + 1].contiguous(), caps[b:b + 1].contiguous(), accumulate=True, return_features=True)
        torch.cuda.synchronize()
        assert torch.equal(enc1["feats"][0], feats[b]), ("encoder features", b)
        assert torch.equal(tr1["hc"][0], hc[b]), ("decoder trace", b)
        assert torch.equal(f1[0], r_feat[b]), ("decoder relevance", b, (f1[0] - r_feat[b]).abs().max().item())
        assert torch.equal(w1[0], r_words[b]), ("r_words", b)
        assert torch.equal(m1[0], maps[b]), ("pixel maps", b, (m1[0] - maps[b]).abs().max().item())
