#!/usr/bin/env python3
"""Micro-benchmark of the VGG16 relevance chain (the dominant stage): maps/s and MFMA TFLOP/s."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa: E402,F401
from lrp_amd import ops, weights  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=4)
ap.add_argument("--maps", type=int, default=80)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--streams", type=int, default=1)
ap.add_argument("--fp32", action="store_true", help="same as --mode 0")
ap.add_argument("--mode", type=int, default=None, help="0 fp32 MFMA, 1 bf16x6 exact splits, 2 f16x3, 3 f16x3 with fp6 cross products in the relevance pass (default: the library's process default = 1)")
a = ap.parse_args()

from lrp_amd import _lib
if a.fp32 or a.mode is not None:
    _lib.load().lrpx_set_conv_mode(0 if a.fp32 else a.mode)
sd = weights.make_gridtd_state(seed=0, vocab_size=64)
names = [k for k in sd if k.startswith("img_encoder.encoder.") and k.endswith(".weight")]
vgg = ops.Vgg16([torch.from_numpy(sd[k]).cuda() for k in names],
                [torch.from_numpy(sd[k.replace(".weight", ".bias")]).cuda() for k in names])
img = torch.from_numpy(weights.make_images(0, a.images)).cuda()
feats = vgg.forward(img)
torch.manual_seed(0)
r_feat = torch.randn(a.maps, 196, 512, device="cuda")
m2i = (torch.arange(a.maps, device="cuda") * a.images // a.maps).to(torch.int32)
out = vgg.relevance(r_feat, m2i, streams=a.streams)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.iters):
    vgg.forward(img)
torch.cuda.synchronize()
tf = (time.time() - t0) / a.iters
t0 = time.time()
for _ in range(a.iters):
    vgg.relevance(r_feat, m2i, out=out, streams=a.streams)
torch.cuda.synchronize()
tr = (time.time() - t0) / a.iters
print(f"forward(+Z+): {tf*1e3:.2f} ms for {a.images} images = {a.images*61.4e9/tf/1e12:.1f} TFLOP/s")
print(f"relevance   : {tr*1e3:.2f} ms for {a.maps} maps = {a.maps/tr:.1f} maps/s = {a.maps*30.69e9/tr/1e12:.1f} TFLOP/s")
