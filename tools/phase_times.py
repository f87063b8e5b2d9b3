#!/usr/bin/env python3
"""Serial step of config 2 split into its phases by HIP events (median of 10 steps)."""
import os, sys, statistics
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import weights, ops
from lrp_amd.explainers.gridtd import GridTDEngine
B, T, V = 16, 20, 9586
eng = GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=V))
if "--lockstep-fp32" in sys.argv:
    eng.lockstep_f16 = False
images = torch.from_numpy(weights.make_images(100, B)).cuda()
caps = torch.from_numpy(weights.make_captions(200, B, T, V)).cuda()
names = ["encode", "trace", "relevance", "chain", "cumsum"]
acc = {n: [] for n in names}
tot = []
out = torch.empty(B * T, 3, 224, 224, device="cuda")
out2 = torch.empty_like(out)
for it in range(13):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    enc = eng.encode(images); ev[1].record()
    tr = eng.trace(enc, caps, predictions=True); ev[2].record()
    r_feat, r_words, row2img = eng.relevance(enc, tr); ev[3].record()
    maps = eng.vgg.relevance(r_feat, row2img, out=out); ev[4].record()
    ops.cumsum_maps(maps, B, T, out=out2); ev[5].record()
    torch.cuda.synchronize()
    if it >= 3:
        for i, n in enumerate(names):
            acc[n].append(ev[i].elapsed_time(ev[i + 1]))
        tot.append(ev[0].elapsed_time(ev[5]))
print("serial step %.2f ms: " % statistics.median(tot) + "  ".join("%s %.2f" % (n, statistics.median(acc[n])) for n in names))
