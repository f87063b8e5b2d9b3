"""The reference-equivalent CPU mode (the cpu_baseline of bench.py) must give the oracle's numbers:
it replays the reference's op sequence (hooks + autograd, materialised epsilon rules, per-pixel loops)."""
import numpy as np
import torch

import lrp_amd  # noqa: F401
from lrp_amd import weights
from conftest import rel_err
from oracle import lrp_oracle as O
from oracle import ref_equiv as RE


def test_ref_equiv_matches_oracle_and_accumulates():
    V = 211
    sd = O.state_to_torch(weights.make_gridtd_state(seed=2, vocab_size=V))
    img = torch.from_numpy(weights.make_images(3, 1))
    cap = weights.make_captions(4, 1, 2, V)[0]
    maps, rws, t_trace, t_words = RE.explain_words(sd, img, cap, [0, 1])
    w_maps, w_rws = O.gridtd_explain_caption(sd, img, cap, accumulate=True)
    for t in range(2):
        assert rel_err(maps[t], w_maps[t]) < 1e-5          # incl. the running-sum quirk of sample.grad
        assert np.abs(rws[t].numpy() - w_rws[t].numpy()).max() < 1e-5


def test_ref_equiv_timing_record_against_the_reference():
    """the CPU baseline that travels (oracle/ref_equiv.py) is a fair stand-in for the reference - the record of
    tools/ref_timing.py (build container: the imported reference and ref_equiv on the same image, weights, 20-word caption and
    thread count) must show the same maps and wall times within +-10 %."""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r04_ref_equiv_vs_reference.json")
    rec = json.load(open(path))
    assert rec["words"] == 20 and rec["vocab"] == 9586
    assert 0.9 <= rec["ratio_ref_equiv_over_reference_time"] <= 1.1, rec
    assert rec["max_rel_map_difference"] < 1e-5 and rec["max_r_words_difference"] < 1e-5
