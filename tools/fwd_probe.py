#!/usr/bin/env python3
"""VGG16 forward trace time for 16 and 4 images, serial (fixed per-layer latency vs per-image work)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import weights
from lrp_amd.explainers.gridtd import GridTDEngine
eng = GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=50))
for B in (16, 4, 32, 48, 64):
    images = torch.from_numpy(weights.make_images(100, B)).cuda()
    for _ in range(3): f = eng.vgg.forward(images)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): f = eng.vgg.forward(images)
    torch.cuda.synchronize()
    print(f"B={B} forward {(time.perf_counter()-t0)/20*1e3:.3f} ms  checksum {f.double().sum().item():.6f}")
