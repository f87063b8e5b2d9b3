"""Recorded steps (lrp_amd._lib.Recording, `explain_batch_replay` of both engines): the library calls of one eager run of a step,
issued again without the interpreter's per-launch cost.  A replay runs the same kernels in the same order on the same buffers, so its
results must be BIT-IDENTICAL to the eager step on the same inputs - for the inputs of the recording and for new ones."""
import pytest
import torch

import lrp_amd  # noqa: F401

pytestmark = pytest.mark.gpu


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")


def test_aoa_bottom_up_replay_is_bit_identical():
    """config 5's step (36 x 2048 region features, relevance back to the features; models/aoamodel.py:1064-1156 with P = 36)"""
    _need_gpu()
    from lrp_amd import weights
    from lrp_amd.explainers.aoa import AOAEngine
    V, B, T = 503, 4, 5
    eng = AOAEngine(weights.make_aoa_state(seed=3, vocab_size=V, feat_dim=2048, with_encoder=False))
    outs = []
    for k in range(3):          # call 0 records, calls 1 and 2 replay on new inputs
        feats = torch.from_numpy(weights.make_bu_features(10 + k, B)).cuda()
        caps = torch.from_numpy(weights.make_captions(20 + k, B, T, V)).cuda()
        r_feat, r_words = eng.explain_batch_replay(caps, 2, features=feats, predictions=True)
        got = (r_feat.clone(), r_words.clone())
        want = eng.explain_batch(caps, 2, features=feats, predictions=True)
        torch.cuda.synchronize()
        assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1]), k
        outs.append(got[0])
    assert not torch.equal(outs[0], outs[1])          # (the inputs really differed)
    assert len(eng._recordings) == 1
    rec = next(iter(eng._recordings.values()))
    assert 10 <= len(rec.calls) <= 120, len(rec.calls)      # a bottom-up step: ~25 library calls (the native step loops fold ~60 launches)
    # another head is another recording; a replica has none of its own
    eng.explain_batch_replay(caps, 5, features=feats, predictions=True)
    assert len(eng._recordings) == 2 and not hasattr(eng.replica(), "_recordings")


def test_gridtd_replay_is_bit_identical_and_bound_to_its_stream():
    """the image path: VGG16 forward trace, gridTD decoder trace and relevance, VGG16 relevance chain, running sums"""
    _need_gpu()
    from lrp_amd import _lib, weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    V, B, T = 307, 2, 3
    eng = GridTDEngine(weights.make_gridtd_state(seed=1, vocab_size=V))
    for k in range(2):
        img = torch.from_numpy(weights.make_images(30 + k, B)).cuda()
        caps = torch.from_numpy(weights.make_captions(40 + k, B, T, V)).cuda()
        maps, r_words, pred = eng.explain_batch_replay(img, caps, accumulate=True, predictions=True)
        got = (maps.clone(), r_words.clone(), pred.clone())
        want = eng.explain_batch(img, caps, accumulate=True, predictions=True)
        torch.cuda.synchronize()
        for a, b in zip(got, want):
            assert torch.equal(a, b), k
    # one image, as the drop-in calls it: its own recording (key: shapes)
    m1, _ = eng.explain_batch_replay(img[:1], caps[:1])
    m1 = m1.clone()
    m1b, _ = eng.explain_batch_replay(img[:1], caps[:1])
    torch.cuda.synchronize()
    assert torch.equal(m1, m1b) and torch.equal(m1, eng.explain_batch(img[:1], caps[:1])[0])
    # a recording belongs to the stream it was made on: another stream gets its own
    n_before = len(eng._recordings)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        m2, _ = eng.explain_batch_replay(img[:1], caps[:1])
        m2 = m2.clone()
    side.synchronize()
    assert len(eng._recordings) == n_before + 1 and torch.equal(m1, m2)
    rec = next(iter(eng._recordings.values()))
    with torch.cuda.stream(side):
        if rec.stream != _lib.stream_ptr().value:
            with pytest.raises(_lib.LrpxError):
                rec.replay()
