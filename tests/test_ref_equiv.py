"""The reference-equivalent CPU mode (the cpu_baseline of bench.py) must give the oracle's numbers:
it replays the reference's op sequence (hooks + autograd, materialised epsilon rules, per-pixel loops)."""
import numpy as np
import pytest
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


def test_ref_equiv_against_the_imported_reference_when_it_is_here():
    """the CPU baseline that travels (oracle/ref_equiv.py) is a stand-in for the reference.  Where the reference is
    present (the build container) it is RE-RUN against the imported `ExplainGridTDAttention.explain_caption` on a 3-word caption
    (tools/ref_timing.py --words 3: the same maps and r_words to 1e-5 incl. the running-sum quirk, wall time within a factor of 1.5 at this size - the
    20-word figure, 1.04x, is the evidence file profiles/r04_ref_equiv_vs_reference.json, written by the same tool).  Elsewhere
    (the GPU box has no /root/reference) the test is skipped: nothing is asserted on a committed file."""
    import json
    import os
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/models"):
        pytest.skip("the reference is not present on this machine")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "ref_timing.py"), "--words", "3", "--tag", "tmp_test", "--threads", "8"],
                       capture_output=True, text=True, timeout=900)
    path = os.path.join(root, "profiles", "tmp_test_ref_equiv_vs_reference.json")
    try:
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        rec = json.load(open(path))
    finally:
        if os.path.exists(path):
            os.remove(path)
    assert rec["words"] == 3 and rec["vocab"] == 9586
    # ref_equiv replays the reference's fp32 op sequence: the same bounds as the 20-word record (measured here at 3 words:
    # maps 4.2e-7 of their maximum, r_words 5.7e-7)
    assert rec["max_rel_map_difference"] < 1e-5 and rec["max_r_words_difference"] < 1e-5, rec
    # wall time: only this window is wider than the record's +-10 % - a 3-word run is 5 s of which the one-off trace is a third, and the
    # container's 8 shared cores move it by +-20 % between repetitions (measured 0.93; the 20-word record: 1.04)
    assert 0.5 <= rec["ratio_ref_equiv_over_reference_time"] <= 1.5, rec
