#!/usr/bin/env python3
"""Where does a pipelined step's time go?  Two batches in flight (as bench.py), with parts of the step replaced by
results kept from an earlier pass: full step / no VGG forward / no forward + decoder trace / CNN relevance chain only."""
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
reps = [eng, eng.replica()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
kept = []
for r, st in zip(reps, streams):
    with torch.cuda.stream(st):
        enc = r.encode(images)
        tr = r.trace(enc, caps, predictions=False)
        r_feat, r_words, row2img = r.relevance(enc, tr, None)
        kept.append((enc, tr, r_feat, row2img))
torch.cuda.synchronize()

def step(k, variant):
    r = reps[k]
    enc, tr, r_feat, row2img = kept[k]
    if variant == "full":
        enc = r.encode(images)
    if variant in ("full", "no_fwd"):
        tr = r.trace(enc, caps, predictions=False)
    if variant in ("full", "no_fwd", "no_fwd_trace"):
        r_feat, _, row2img = r.relevance(enc, tr, None)
    return r.vgg.relevance(r_feat, row2img)

for depth in (2, 1):
    for variant in ("full", "no_fwd", "no_fwd_trace", "chain_only"):
        def run(n):
            for i in range(n):
                k = i % depth
                with torch.cuda.stream(streams[k]):
                    step(k, variant)
        run(4); torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(12)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 12
        print(f"in flight {depth}  {variant:14s} {dt*1e3:7.2f} ms/step", flush=True)
