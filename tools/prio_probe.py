#!/usr/bin/env python3
"""Does a high-priority stream for the forward trace + decoder phases (small, latency-bound kernels) improve the overlap
with the CNN relevance chains of the other batches in flight?  Same step as bench.py, n batches in flight."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import weights
from lrp_amd.explainers.gridtd import GridTDEngine
B, T, V = 16, 20, 9586
eng = GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=V))
images = torch.from_numpy(weights.make_images(100, B)).cuda()
caps = torch.from_numpy(weights.make_captions(200, B, T, V)).cuda()
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range (least, greatest):", lo, hi)
for n_pipe in (3, 2, 4):
    engines = [eng] + [eng.replica() for _ in range(n_pipe - 1)]
    for mode in ("one stream per batch", "high-priority stream for forward+decoder", "high-priority stream for the chain"):
        if mode == "one stream per batch":
            s_small = [torch.cuda.Stream() for _ in range(n_pipe)]; s_chain = s_small
        elif mode.startswith("high-priority stream for forward"):
            s_small = [torch.cuda.Stream(priority=hi) for _ in range(n_pipe)]; s_chain = [torch.cuda.Stream(priority=lo) for _ in range(n_pipe)]
        else:
            s_small = [torch.cuda.Stream(priority=lo) for _ in range(n_pipe)]; s_chain = [torch.cuda.Stream(priority=hi) for _ in range(n_pipe)]
        outs = [torch.empty(B * T, 3, 224, 224, device="cuda") for _ in range(n_pipe)]
        def step(i):
            k = i % n_pipe
            e = engines[k]
            s_small[k].wait_stream(s_chain[k])               # the previous chain of this engine still reads its trace
            with torch.cuda.stream(s_small[k]):
                enc = e.encode(images)
                tr = e.trace(enc, caps, predictions=False)
                r_feat, r_words, row2img = e.relevance(enc, tr)
            s_chain[k].wait_stream(s_small[k])
            with torch.cuda.stream(s_chain[k]):
                e.vgg.relevance(r_feat, row2img, out=outs[k])
                r_feat.record_stream(s_chain[k])
        for i in range(2 * n_pipe): step(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(18): step(i)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 18
        print(f"in flight {n_pipe}  {mode:45s} {dt*1e3:7.2f} ms/step  {B*T/dt:8.1f} maps/s", flush=True)
