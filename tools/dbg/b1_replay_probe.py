#!/usr/bin/env python3
"""one image, 20 words: GridTDEngine.explain_batch eager against explain_batch_replay (host cost of ~350 launches), per conv mode"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrp_amd  # noqa
from lrp_amd import weights
from lrp_amd.explainers.gridtd import GridTDEngine
V, T = 9586, 20
eng = GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=V))
img = torch.from_numpy(weights.make_images(100, 1)).cuda()
cap = torch.from_numpy(weights.make_captions(200, 1, T, V)).cuda()


def timed(fn, n=10):
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


for mode in (1, 3):
    eng.vgg.conv_mode = mode
    for _ in range(3):
        eng.explain_batch(img, cap, accumulate=True, predictions=True)
        eng.explain_batch_replay(img, cap, accumulate=True, predictions=True)
    e = timed(lambda: eng.explain_batch(img, cap, accumulate=True, predictions=True))
    r = timed(lambda: eng.explain_batch_replay(img, cap, accumulate=True, predictions=True))
    # host time of issuing alone (no sync inside)
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.explain_batch(img, cap, accumulate=True, predictions=True); h_e = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize(); t0 = time.perf_counter(); eng.explain_batch_replay(img, cap, accumulate=True, predictions=True); h_r = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    for _ in range(3):
        eng.explain_batch_graph(img, cap, accumulate=True, predictions=True)
    g = timed(lambda: eng.explain_batch_graph(img, cap, accumulate=True, predictions=True))
    print(f"mode {mode}: HIP graph replay {g:.2f} ms", flush=True)
    n = len(next(iter(eng._recordings.values())).calls)
    print(f"mode {mode}: eager {e:.2f} ms (host issue {h_e:.2f} ms), replay {r:.2f} ms (host issue {h_r:.2f} ms), {n} recorded calls", flush=True)
