#!/usr/bin/env python3
"""host-side profile of ONE ExplainGridTDAttention.explain_caption call on a resident image (B = 1, T = 20)"""
import cProfile, os, pstats, sys, time, types
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrp_amd  # noqa
from lrp_amd import weights
from lrp_amd.explainers.gridtd import ExplainGridTDAttention
V, T = 9586, 20
sd = {k: torch.from_numpy(v) for k, v in weights.make_gridtd_state(seed=0, vocab_size=V).items()}
wm = weights.make_word_map(V)
args = types.SimpleNamespace(embed_dim=512, hidden_dim=512, encoder="vgg16", weight="", save_path="/tmp", dataset="synthetic", height=224, width=224)
img = torch.from_numpy(weights.make_images(100, 1)).cuda()
cap = [int(c) for c in weights.make_captions(200, 1, T, V)[0]]
ex = ExplainGridTDAttention(args, wm, model=sd)
for _ in range(5):
    ex.explain_caption(img, caption_encode=cap)
torch.cuda.synchronize()
# host time of the call itself (no sync inside?) vs wall
ts = []
for _ in range(10):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ex.explain_caption(img, caption_encode=cap); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e3, (t2 - t0) * 1e3))
print("host-return / synced ms:", " ".join(f"{a:.2f}/{b:.2f}" for a, b in ts))
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    ex.explain_caption(img, caption_encode=cap)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(28)
