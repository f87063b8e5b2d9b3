#!/usr/bin/env python3
"""the drop-in's one-image pattern (B = 1, 20 words, gridTD): phase table by HIP events (VGG16 forward trace / decoder trace / decoder
relevance / VGG16 relevance chain + running sums), median of 20 calls, per conv mode; run under rocprofv3 --kernel-trace --stats for
the kernel breakdown (profiles/r06_dropin_b1_kernel_stats.txt): python tools/dbg/b1_phases.py [mode ...]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrp_amd  # noqa
from lrp_amd import ops, weights
from lrp_amd.explainers.gridtd import GridTDEngine
V, T = 9586, 20
modes = [int(m) for m in sys.argv[1:]] or [1]
eng = GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=V))
img = torch.from_numpy(weights.make_images(100, 1)).cuda()
cap = torch.from_numpy(weights.make_captions(200, 1, T, V)).cuda()
for mode in modes:
    eng.vgg.conv_mode = mode
    rows = []
    for it in range(24):
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ev[0].record()
        enc = eng.encode(img); ev[1].record()
        tr = eng.trace(enc, cap, predictions=True); ev[2].record()
        r_feat, r_words, row2img = eng.relevance(enc, tr); ev[3].record()
        maps = ops.cumsum_maps(eng.vgg.relevance(r_feat, row2img), 1, T); ev[4].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        if it >= 4:
            rows.append([ev[i].elapsed_time(ev[i + 1]) for i in range(4)] + [wall])
    med = [sorted(r[i] for r in rows)[len(rows) // 2] for i in range(5)]
    print(f"conv mode {mode}, B = 1, T = {T}: VGG16 forward trace {med[0]:.2f} ms | decoder trace (+ (T,V) predictions) {med[1]:.2f} ms | decoder relevance {med[2]:.2f} ms | "
          f"VGG16 relevance chain of {T} maps + running sums {med[3]:.2f} ms | wall {med[4]:.2f} ms per image = {T / med[4] * 1e3:.0f} maps/s", flush=True)
