"""GPU parity of the AoA decoder (trace + relevance, heads 0 and 5), the whole AoA pipeline, and the bottom-up
(36x2048 region features) variant, against the reference's golden vectors."""
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
    from lrp_amd.explainers.aoa import AOAEngine
    g = np.load(os.path.join(GOLDEN, "aoa_T3.npz"))
    sd = weights.make_aoa_state(seed=int(g["seed"]), vocab_size=int(g["V"]))
    eng = AOAEngine(sd)
    img = torch.from_numpy(weights.make_images(int(g["seed"]), 1))
    cap = torch.from_numpy(g["caption"]).view(1, -1)
    return g, sd, eng, img, cap


@pytest.mark.parametrize("head", [0, 5])
def test_aoa_vs_reference(case, head):
    g, sd, eng, img, cap = case
    maps, r_words, r_feat, tr, enc = eng.explain_batch(cap, head, images=img, accumulate=True, return_features=True)
    maps, r_words, r_feat = maps.cpu(), r_words.cpu(), r_feat.cpu()
    assert rel_err(tr["alpha"][0].cpu(), g["tr_alphas"]) < 1e-4
    assert rel_err(tr["h"][0].cpu(), g["tr_ht"]) < 1e-4
    assert rel_err(tr["ctx"][0].cpu(), g["tr_context"]) < 1e-4
    for t in range(3):
        want = torch.from_numpy(g[f"h{head}_r_feat_{t}"])[0].reshape(512, 196).t()
        assert rel_err(r_feat[0, t], want) < TOL, t
        assert cosine(r_feat[0, t], want) > 0.99999
        assert np.abs(r_words[0, t, :t + 1].numpy() - g[f"h{head}_r_words_{t}"]).max() < 1e-5
        assert_close_modulo_pool_ties(maps[0, t][None, :, ::4, ::4], g[f"h{head}_map_sub4_{t}"], what=(head, t))
    if f"h{head}_map_full_2" in g:
        assert_close_modulo_pool_ties(maps[0, 2], g[f"h{head}_map_full_2"][0], what="full")


def test_aoa_bottom_up_regions_vs_reference():
    """config 5: (36,2048) bottom-up features, relevance back to the region features (no CNN stage)."""
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    g = np.load(os.path.join(GOLDEN, "aoa_bu_T3.npz"))
    sd = weights.make_aoa_state(seed=int(g["seed"]), vocab_size=int(g["V"]), feat_dim=2048, with_encoder=False)
    eng = AOAEngine(sd)
    feats = torch.from_numpy(weights.make_bu_features(int(g["seed"]), 1))
    cap = torch.from_numpy(g["caption"]).view(1, -1)
    r_feat, r_words = eng.explain_batch(cap, int(g["head"]), features=feats)
    for t in range(3):
        assert rel_err(r_feat[0, t].cpu(), g[f"r_feat_{t}"]) < TOL
        # r_words = sum of 512 signed embedding relevances / max (cancellation ~1e2): device expf/tanhf vs glibc
        # rounding shows up at 1.4e-5 here, so the GPU bound is 5e-5 (CPU oracle vs reference stays < 1e-5)
        assert np.abs(r_words[0, t, :t + 1].cpu().numpy() - g[f"r_words_{t}"]).max() < 5e-5
    with pytest.raises(ValueError):
        eng.encode(images=torch.zeros(1, 3, 224, 224))


def test_aoa_batch_vs_oracle():
    """B=3 images, T=4, head 2: decoder relevance of every row against the per-image oracle."""
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    from oracle import lrp_oracle as O
    V = 401
    sd = weights.make_aoa_state(seed=9, vocab_size=V)
    sdt = O.state_to_torch(sd)
    eng = AOAEngine(sd)
    img = torch.from_numpy(weights.make_images(13, 3))
    cap = torch.from_numpy(weights.make_captions(14, 3, 4, V))
    maps, r_words, r_feat, tr, enc = eng.explain_batch(cap, 2, images=img, return_features=True)
    for b in range(3):
        w_maps, w_rw, w_rf, _ = O.aoa_explain_caption(sdt, img[b:b + 1], cap[b].numpy(), 2, return_feat=True, accumulate=False)
        for t in range(4):
            want = w_rf[t][0].reshape(512, 196).t()
            assert rel_err(r_feat[b, t].cpu(), want) < 2e-4, (b, t)
            assert np.abs(r_words[b, t, :t + 1].cpu().numpy() - w_rw[t].numpy()).max() < 1e-4
            assert_close_modulo_pool_ties(maps[b, t].cpu(), w_maps[t][0], what=(b, t))


def test_aoa_sample_lrp_tokens_bit_exact():
    """LRP-inference decoding of the AoA model (AOAEngine.sample_lrp = `AOAModel.sample_lrp` greedy,
    models/aoamodel.py:679-745) against the reference's own output (tests/golden/aoa_sample_lrp.npz): token ids
    bit-exact, log-probabilities to 1e-4 absolute; case 2 has a stop word and an <end> that is hit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    g = np.load(os.path.join(GOLDEN, "aoa_sample_lrp.npz"))
    V, L = int(g["V"]), int(g["max_len"])
    eng = AOAEngine(weights.make_aoa_state(seed=int(g["seed"]), vocab_size=V))
    imgs = torch.from_numpy(weights.make_images(int(g["seed"]) + 3, g["seq"].shape[0]))
    wm = weights.make_word_map(V)
    enc = eng.encode(images=imgs.cuda())
    for seq_k, lp_k, skip_k, end_id in (("seq", "logprobs", "skip", wm['<end>']),
                                        ("seq2", "logprobs2", "skip2", int(g["end2"]))):
        seq, lps = eng.sample_lrp(enc, L, wm['<start>'], end_id, g[skip_k].tolist())
        assert seq.cpu().tolist() == g[seq_k].tolist()
        assert np.abs(lps.cpu().numpy() - g[lp_k]).max() < 1e-4


def test_aoa_explain_stream_matches_serial():
    """independent batches in flight on separate HIP streams (AOAEngine.explain_stream) give bit-identical maps and word
    relevances to explaining them one after the other"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    eng = AOAEngine(weights.make_aoa_state(seed=4, vocab_size=60))
    batches = [(torch.from_numpy(weights.make_images(30 + i, 2)), torch.from_numpy(weights.make_captions(40 + i, 2, 3, 60)))
               for i in range(5)]
    serial = [tuple(t.clone() for t in eng.explain_batch(cp, 5, images=im)) for im, cp in batches]
    torch.cuda.synchronize()
    piped = list(eng.explain_stream(batches, 5, depth=3))
    assert len(piped) == len(serial)
    for (m0, w0), (m1, w1) in zip(serial, piped):
        assert torch.equal(m0, m1) and torch.equal(w0, w1)


def test_aoa_forwardlrp_context_vs_reference():
    """`AOAModel.forwardlrp_context` (models/aoamodel.py:628-677), forward values, against the reference's own outputs"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    from test_gpu_gridtd import _check_forwardlrp
    g = np.load(os.path.join(GOLDEN, "forwardlrp.npz"))
    V = int(g["aoa_V"])
    eng = AOAEngine(weights.make_aoa_state(seed=int(g["seed"]), vocab_size=V))
    imgs = torch.from_numpy(weights.make_images(int(g["seed"]) + 5, int(g["batch"])))
    enc = eng.encode(images=imgs.cuda())
    preds, wpreds, L = eng.forwardlrp_context(enc, torch.from_numpy(g["aoa_caption"]), g["aoa_lengths"].tolist(),
                                              g["aoa_skip"].tolist())
    assert L == int(g["aoa_L"])
    _check_forwardlrp(g, "aoa", "", preds, wpreds, L)


def test_aoa_beam_search_caption_bit_exact():
    """`AOAModel.beam_search(beam_size=3, max_cap_length=20)` as `get_hidden_parameters` calls it (models/aoamodel.py:992):
    token ids against the reference's own output (tests/golden/beam.npz), and the drop-in explains that caption"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    import types
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine, ExplainAOAAttention
    from lrp_amd.explainers.beam import caption_from_sequence
    from test_oracle_golden import _beam_cases
    g = np.load(os.path.join(GOLDEN, "beam.npz"))
    V, cases = _beam_cases(g, "aoa")
    sd = weights.make_aoa_state(seed=int(g["seed"]), vocab_size=V)
    eng = AOAEngine(sd)
    img = torch.from_numpy(weights.make_images(int(g["seed"]) + 7, 1)).cuda()
    enc = eng.encode(images=img)
    for key, wm in cases:
        seq = eng.beam_search(enc, int(g["aoa_beam"]), int(g["aoa_steps"]), wm['<start>'], wm['<end>'])
        assert caption_from_sequence(seq, wm)[1:] == g[f"aoa_{key}"].tolist(), key
    ex = ExplainAOAAttention(types.SimpleNamespace(num_head=8), cases[1][1], model=sd)
    maps, rw = ex.explain_caption(img, 0)
    assert ex.beam_caption_encode[1:] == g["aoa_sen_end"].tolist() and len(maps) == len(g["aoa_sen_end"])


@pytest.mark.parametrize("bu", [False, True])
def test_aoa_decoupled_trace_matches_the_stepwise_trace(bu):
    """The teacher-forced trace with the recurrence decoupled (lrpx_aoa_fwd_recurrence: one GEMM for the input part of all gate
    pre-activations, T launches of K = H, the attention / AoA half once over all rows) against the stepwise kernels that follow the
    reference's loop statement by statement (models/aoamodel.py:1019-1052): every trace tensor to 1e-5 of its maximum (z is formed
    as (x W_ih^T + b) + W_hh h instead of one dot product), both LSTM biases, with and without the gradient explainers' extras."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    V, B, T = 523, 20, 7
    if bu:
        eng = AOAEngine(weights.make_aoa_state(seed=31, vocab_size=V, feat_dim=2048, with_encoder=False))
        enc = eng.encode(features=torch.from_numpy(weights.make_bu_features(32, B)))
    else:
        eng = AOAEngine(weights.make_aoa_state(seed=31, vocab_size=V))
        enc = eng.encode(torch.from_numpy(weights.make_images(32, B)))
    cap = torch.from_numpy(weights.make_captions(33, B, T, V)).cuda()
    assert eng.decoupled
    for grad in (False, True):
        a = eng.trace(enc, cap, predictions=True, grad=grad)
        eng.decoupled = False
        try:
            b = eng.trace(enc, cap, predictions=True, grad=grad)
        finally:
            eng.decoupled = True
        for k in ["xh", "h", "c", "g", "i", "f", "ctx", "lin", "c_aoa", "hc", "alpha", "logit", "pred"] + (["o", "sg"] if grad else []):
            e = rel_err(a[k].cpu(), b[k].cpu())
            assert e < 1e-5, (k, grad, e)


@pytest.mark.parametrize("f16", [False, True], ids=["decoder-fp32", "decoder-f16x3"])
@pytest.mark.parametrize("bu", [False, True])
def test_aoa_fused_lock_steps_match_the_two_launch_steps(bu, f16):
    """The relevance lock-steps with the step's point-wise code inside the gate rule's GEMM (lrpx_aoa_rel_steps_fused: one launch per
    step, r_xh never stored) against GEMM + point-wise kernel: the same expressions in the same order, so r_feat - everything behind
    r_h and r_glob - is bit-identical; r_words sums its 512 embedding columns in another order (four 128-column partial sums): equal
    to 1e-5 (the rows are normalised to a largest entry of 1; ~100x cancellation).  Rows that do not fill a 32-row tile, captions of unequal length, two heads."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    V, B, T = 523, 5, 9
    if bu:
        eng = AOAEngine(weights.make_aoa_state(seed=35, vocab_size=V, feat_dim=2048, with_encoder=False))
        enc = eng.encode(features=torch.from_numpy(weights.make_bu_features(36, B)))
    else:
        eng = AOAEngine(weights.make_aoa_state(seed=35, vocab_size=V))
        enc = eng.encode(torch.from_numpy(weights.make_images(36, B)))
    cap = torch.from_numpy(weights.make_captions(37, B, T, V)).cuda()
    tr = eng.trace(enc, cap, predictions=False)
    eng.force_f16 = f16           # the fusion exists in both arithmetics: the fp16 split-product kernel (speed modes) and the fp32 K-split kernel (default)
    assert eng.fused_rel
    for head, lens in ((0, None), (6, [9, 2, 5, 0, 9])):
        a_feat, a_words, _ = eng.relevance(enc, tr, head, lens, compact=False)
        eng.fused_rel = False
        try:
            b_feat, b_words, _ = eng.relevance(enc, tr, head, lens, compact=False)
        finally:
            eng.fused_rel = True
        assert torch.equal(a_feat, b_feat), (head, (a_feat - b_feat).abs().max().item())
        assert (a_words - b_words).abs().max().item() < 1e-5, (head, (a_words - b_words).abs().max().item())
        assert a_words.abs().max().item() == 1.0


@pytest.mark.parametrize("bu", [False, True])
def test_aoa_head_slice_of_the_v_proj_rule_is_bit_identical(bu):
    """`lrp_mha` passes relevance through one head (models/aoamodel.py:848-860): the v_proj dense rule behind it (:1141-1144) contracts over
    that head's 64 rows of W_v (lrpx_aoa_rel_value_head + a pack of the row slice) instead of over all 512 with 448 zero columns in the
    operand - the same products in the same order: r_feat bit for bit, with captions of unequal length and every head position (first /
    middle / last)."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    V, B, T = 523, 6, 5
    if bu:
        eng = AOAEngine(weights.make_aoa_state(seed=38, vocab_size=V, feat_dim=2048, with_encoder=False))
        enc = eng.encode(features=torch.from_numpy(weights.make_bu_features(39, B)))
    else:
        eng = AOAEngine(weights.make_aoa_state(seed=38, vocab_size=V))
        enc = eng.encode(torch.from_numpy(weights.make_images(39, B)))
    cap = torch.from_numpy(weights.make_captions(40, B, T, V)).cuda()
    tr = eng.trace(enc, cap, predictions=False)
    eng.force_f16 = True          # (the head slice is a pack of the fp16 split-product kernel)
    assert eng.head_only and eng.p_v_rel_head is not None
    for head, lens in ((0, None), (4, [5, 1, 0, 3, 5, 2]), (7, None)):
        a_feat, a_words, _ = eng.relevance(enc, tr, head, lens)
        eng.head_only = False
        try:
            b_feat, b_words, _ = eng.relevance(enc, tr, head, lens)
        finally:
            eng.head_only = True
        assert torch.equal(a_feat, b_feat) and torch.equal(a_words, b_words), (head, (a_feat - b_feat).abs().max().item())


def test_trace_is_the_same_in_every_batch():
    """an image's trace must not depend on the batch it sits in.  The decoupled trace (one table lookup + T recurrence launches
    + the attention half over all rows, models/aoamodel.py:1019-1052) now takes ANY batch size - the recurrence in slices of at most 64
    images inside the library - where B = 65 used to fall back to the stepwise kernels: images at B = 1, 64 and 65, bit for bit."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    V, T = 211, 4
    eng = AOAEngine(weights.make_aoa_state(seed=4, vocab_size=V, feat_dim=2048, with_encoder=False))
    feats = torch.from_numpy(weights.make_bu_features(31, 65)).cuda()
    caps = torch.from_numpy(weights.make_captions(32, 65, T, V)).cuda()
    keys = ("h", "c", "g", "i", "f", "ctx", "lin", "c_aoa", "hc", "alpha")

    def run(sel):
        enc = eng.encode(features=feats[sel].contiguous())
        tr = eng.trace(enc, caps[sel].contiguous(), predictions=False)
        torch.cuda.synchronize()
        return {k: tr[k].clone() for k in keys}
    t65 = run(slice(0, 65))
    t64 = run(slice(0, 64))
    for b in (0, 37, 63):
        one = run(slice(b, b + 1))
        for k in keys:
            assert torch.equal(one[k][0], t64[k][b]), ("B = 1 vs B = 64", b, k, (one[k][0] - t64[k][b]).abs().max().item())
            assert torch.equal(one[k][0], t65[k][b]), ("B = 1 vs B = 65", b, k, (one[k][0] - t65[k][b]).abs().max().item())
    one = run(slice(64, 65))
    for k in keys:
        assert torch.equal(one[k][0], t65[k][64]), k
    # and the relevance of those rows: the same maps whatever the batch (rows are independent in every kernel behind `relevance`)
    r65, w65 = eng.explain_batch(caps, 3, features=feats)
    r1, w1 = eng.explain_batch(caps[64:65].contiguous(), 3, features=feats[64:65].contiguous())
    assert torch.equal(w1[0], w65[64]) and torch.equal(r1[0], r65[64])
