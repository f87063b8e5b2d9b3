"""DIAGNOSTIC (not collected by pytest): where does the distance of the worst T = 20 gridTD r_words row (golden image 1, word 17) to
the fp64 evaluation come from?  The engine's encoder features and decoder trace of that image (inside the B = 16 batch of
tests/test_gpu_t20.py) are fed, tensor group by tensor group, into the oracle's fp64 evaluation of explain_caption_wordt
(oracle/lrp_oracle.py:gridtd_trace / gridtd_explain_wordt, models/gridTDmodel.py:933-1135).
usage (GPU box): python tests/diag_t20_words.py [word]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.dirname(__file__))
import lrp_amd  # noqa: E402,F401
from lrp_amd import weights  # noqa: E402
from lrp_amd.explainers.gridtd import GridTDEngine  # noqa: E402
from oracle import lrp_oracle as orc  # noqa: E402
import test_gpu_t20 as t20  # noqa: E402

word = int(sys.argv[1]) if len(sys.argv) > 1 else 17
k_img = 1
g = np.load(os.path.join(t20.GOLDEN, "t20.npz"))
g64 = np.load(os.path.join(t20.GOLDEN, "t20_f64.npz"))
V, B = int(g["grid_V"]), 16
T, caps = t20._batch(g, B, "grid_caption", V, 61)
sd = weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=V)
eng = GridTDEngine(sd)
imgs = t20._images(g, B)
p = t20.POS[k_img]
enc = eng.encode(imgs)
tr = eng.trace(enc, caps.cuda(), predictions=False)
r_feat, r_words, _ = eng.relevance(enc, tr)
torch.cuda.synchronize()
got = r_words.view(B, T, T)[p, word, :word + 1].cpu().double()
ref64 = torch.from_numpy(g64[f"grid{k_img}_r_words64_{word}"]).double()
ref32 = torch.from_numpy(g[f"grid{k_img}_r_words_{word}"]).double()
print(f"engine row vs fp64 golden: {(got - ref64).abs().max():.2e}; reference fp32 row vs fp64: {(ref32 - ref64).abs().max():.2e}")

torch.set_default_dtype(torch.float64)
sd64 = {k: torch.as_tensor(v).double() for k, v in sd.items() if torch.as_tensor(v).is_floating_point()}
feats = enc["feats"][p].cpu().double()                     # (P,C)
C = feats.shape[1]
features = feats.t().reshape(C, 14, 14).contiguous()
avg = enc["avg"][p].cpu().double()
caption = [int(c) for c in caps[p]]
tr64 = orc.gridtd_trace(sd64, features, avg, caption)
rw64 = orc.gridtd_explain_wordt(sd64, tr64, word)[1]
print(f"fp64 decoder on the ENGINE's features vs fp64 golden (the VGG16 forward's share): {(rw64 - ref64).abs().max():.2e}")
# the same in fp32 on the CPU (what the reference's arithmetic does with these features)
torch.set_default_dtype(torch.float32)
sd32 = {k: v.float() for k, v in sd64.items()}
tr32 = orc.gridtd_trace(sd32, features.float(), avg.float(), caption)
rw32 = orc.gridtd_explain_wordt(sd32, tr32, word)[1].double()
print(f"fp32 CPU decoder (trace + relevance) on the engine's features vs the fp64 decoder on them: {(rw32 - rw64).abs().max():.2e}")
torch.set_default_dtype(torch.float64)
E, Hd = eng.E, eng.H
G = {k: tr[k][p].cpu().double() for k in ("h1", "c1", "h2", "c2", "g1", "i1", "f1", "g2", "i2", "f2", "s", "ctx", "ctx_hat", "alpha", "beta", "xh1", "xh2")}
G["x1"], G["x2"] = G["xh1"][:, :2 * E + Hd], G["xh2"][:, :2 * Hd]
for k in ("h1", "c1", "h2", "c2", "g1", "i1", "f1", "g2", "i2", "f2", "s", "ctx", "ctx_hat", "alpha", "beta", "x1", "x2"):
    a, b = G[k], tr64[k]
    print(f"  trace tensor {k:8s}: engine vs fp64 max|d| / max|.| = {((a - b).abs().max() / b.abs().max()).item():.2e}   "
          f"(fp32 CPU trace: {((tr32[k].double() - b).abs().max() / b.abs().max()).item():.2e})")


def hybrid(keys, src):
    h = dict(tr64)
    for k in keys:
        h[k] = src[k].double()
    return orc.gridtd_explain_wordt(sd64, h, word)[1]


groups = {"all": ("h1", "c1", "h2", "c2", "g1", "i1", "f1", "g2", "i2", "f2", "s", "ctx", "ctx_hat", "alpha", "beta", "x1", "x2"),
          "g1 g2 (gate pre-activations: the rules' denominators)": ("g1", "g2"), "c1 c2": ("c1", "c2"), "i f": ("i1", "f1", "i2", "f2"),
          "h1 h2 x1 x2": ("h1", "h2", "x1", "x2"), "s ctx ctx_hat alpha beta": ("s", "ctx", "ctx_hat", "alpha", "beta")}
for name, keys in groups.items():
    e_g = (hybrid(keys, G) - rw64).abs().max().item()
    e_c = (hybrid(keys, tr32) - rw64).abs().max().item()
    print(f"fp64 relevance on the fp64 trace with [{name}] from the ENGINE's trace: {e_g:.2e}   (from the fp32 CPU trace: {e_c:.2e})")
print(f"engine relevance vs fp64 relevance on the engine's own trace: {(got - hybrid(groups['all'], G)).abs().max():.2e}")
