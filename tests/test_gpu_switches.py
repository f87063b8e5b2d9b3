"""Every A/B switch of the shipped dispatch paths (csrc/common.h `Switches`, read once per process) at its NON-default
value: the kernels behind it are reachable from an inherited environment, so they are held to the same results as the
defaults.  One child process per value (the switches latch on first use): a small gridTD explanation
(LRP + guided backprop, 2 images x 3 words) whose outputs must agree with the default process within the bounds two
valid kernel choices differ by (another summation order / another, equally exact tiling)."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, sys, torch
sys.path.insert(0, %r)
import lrp_amd
from lrp_amd import weights
from lrp_amd.explainers.gridtd import GridTDEngine
import os
from lrp_amd import _lib
if os.environ.get("LRPX_CHILD_MODE"):
    _lib.load().lrpx_set_conv_mode(int(os.environ["LRPX_CHILD_MODE"]))
V = 307
eng = GridTDEngine(weights.make_gridtd_state(seed=3, vocab_size=V))
img = torch.from_numpy(weights.make_images(4, 2))
cap = torch.from_numpy(weights.make_captions(5, 2, 3, V))
maps, r_words, r_feat, tr, enc = eng.explain_batch(img, cap, return_features=True, accumulate=False)
gmaps, g_words = eng.explain_batch_guided(img, cap)
torch.cuda.synchronize()
torch.save({"maps": maps.cpu(), "r_words": r_words.cpu(), "r_feat": r_feat.cpu(), "feats": enc["feats"].cpu(),
            "gmaps": gmaps.cpu(), "g_words": g_words.cpu()}, sys.argv[1])
''' % ROOT


def _run(tmp_path, name, env):
    out = tmp_path / f"{name}.pt"
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, "-c", CHILD, str(out)], env=e, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    return torch.load(out)


_DEFAULTS = {}


@pytest.fixture(scope="module")
def default_run(tmp_path_factory):
    """the same explanation with every switch at its default, per conv mode (one child process each, on first use)"""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")

    def get(mode):
        if mode not in _DEFAULTS:
            _DEFAULTS[mode] = _run(tmp_path_factory.mktemp("sw"), f"default{mode}", {"LRPX_CHILD_MODE": mode} if mode else {})
        return _DEFAULTS[mode]
    return get


# (switch, value, conv mode of the child: "" = the process default (mode 1 since round 6; mode 3 before); LRPX_FIRST_VALU / LRPX_S21_NHWC choose kernels of the NHWC
# chain of mode 2 - the mode-3 chain keeps its tensors in the blocked layout and has no such alternatives)
SWITCHES = [("LRPX_FWD_WIDE", "15", ""), ("LRPX_CONV11_F16", "0", ""), ("LRPX_WIDE", "0", ""), ("LRPX_FWD_KSPLIT", "1", ""),
            ("LRPX_FWD_KSPLIT28", "4", ""), ("LRPX_FIRST_VALU", "1", "2"), ("LRPX_S21_NHWC", "1", "2"), ("LRPX_GUIDED_POOLBWD", "1", ""),
            ("LRPX_DENSE_1WAVE", "1", ""), ("LRPX_LINEAR_VALU", "1", ""),
            # round 6: the lock-step rule of the exact modes on the one-wave kernel; the decoders' GEMMs on / off the fp16 split products
            # against the conv mode's rule (lrp_amd.ops.decoder_f16)
            ("LRPX_DENSE_KS_REL", "0", ""), ("LRPX_DECODER_F16", "1", ""), ("LRPX_DECODER_F16", "0", "3"),
            ("LRPX_B6_FWD_KSPLIT28", "1", ""), ("LRPX_B6_FWD_KSPLIT56", "1", ""), ("LRPX_B6_REL_KSPLIT14", "1", ""), ("LRPX_X6_LEGACY", "1", "")]


@pytest.mark.parametrize("name,value,mode", SWITCHES)
def test_non_default_switch_gives_the_same_results(default_run, tmp_path, name, value, mode):
    from conftest import rel_err
    got = _run(tmp_path, name, {name: value, **({"LRPX_CHILD_MODE": mode} if mode else {})})
    ref = default_run(mode)
    fwd = name.startswith("LRPX_FWD") or name.startswith("LRPX_B6_FWD") or name in ("LRPX_CONV11_F16", "LRPX_X6_LEGACY")
    # forward switches change the summation order of the trace (features move at the 1e-6 level, a pool winner may flip);
    # everything else runs the same trace through another kernel of the same arithmetic
    assert rel_err(got["feats"], ref["feats"]) < (1e-5 if fwd else 1e-7), name
    assert rel_err(got["r_feat"], ref["r_feat"]) < 1e-4 and (got["r_words"] - ref["r_words"]).abs().max().item() < 1e-4
    assert (got["g_words"] - ref["g_words"]).abs().max().item() < 1e-4
    for key, frac in (("maps", 1e-2), ("gmaps", 7e-2)):
        a, b = got[key].double(), ref[key].double()
        scale = b.abs().amax(dim=(2, 3, 4), keepdim=True)
        d = (a - b).abs() / scale
        if fwd:
            assert (d > 1e-4).double().mean().item() < frac and d.max().item() < 1e-2, (name, key, d.max().item())
        else:
            assert d.max().item() < 1e-4, (name, key, d.max().item())
