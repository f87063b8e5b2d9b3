#!/usr/bin/env python3
"""Throughput of the other explainers of gridTD at BASELINE config-2 size (16 images x 20 words): guided backprop,
plain gradient, Grad-CAM - maps/s next to the LRP number of bench.py (same engine, same synthetic inputs)."""
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
runs = {"LRP (explain_batch)": lambda: eng.explain_batch(images, caps),
        "guided backprop (explain_batch_guided)": lambda: eng.explain_batch_guided(images, caps),
        "plain gradient (explain_batch_gradient)": lambda: eng.explain_batch_gradient(images, caps),
        "Grad-CAM (explain_batch_gradient cam=True)": lambda: eng.explain_batch_gradient(images, caps, cam=True)}
for name, fn in runs.items():
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:45s} {dt*1e3:8.2f} ms/step  {B*T/dt:9.1f} maps/s")
