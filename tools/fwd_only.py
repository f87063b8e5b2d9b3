#!/usr/bin/env python3
"""VGG16 forward trace (lrpx_vgg16_forward incl. the derived tensors) alone, B images, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import weights
from lrp_amd.explainers.gridtd import GridTDEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 30
eng = GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=50))
images = torch.from_numpy(weights.make_images(100, B)).cuda()
for _ in range(3):
    f = eng.vgg.forward(images)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    f = eng.vgg.forward(images)
torch.cuda.synchronize()
print(f"B={B} forward {(time.perf_counter() - t0) / N * 1e3:.3f} ms")
