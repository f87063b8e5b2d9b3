#!/usr/bin/env python3
"""Forward trace on the fp16 split-product kernels vs bf16x6: accuracy of activations / Z+ and time."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import _lib, ops, weights
lib = _lib.load()
sd = weights.make_gridtd_state(seed=0, vocab_size=64)
names = [k for k in sd if k.startswith("img_encoder.encoder.") and k.endswith(".weight")]
vgg = ops.Vgg16([torch.from_numpy(sd[k]).cuda() for k in names],
                [torch.from_numpy(sd[k.replace(".weight", ".bias")]).cuda() for k in names])
img = torch.from_numpy(weights.make_images(0, 16)).cuda()
res = {}
for mode in (0, 1):
    lib.lrpx_set_forward_f16(mode)
    vgg.forward(img); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(5): vgg.forward(img)
    torch.cuda.synchronize()
    acts, zs = vgg.trace_views()
    res[mode] = ([a.clone() for a in acts], [z.clone() if z is not None else None for z in zs], (time.time() - t0) / 5)
print(f"forward: bf16x6 {res[0][2]*1e3:.2f} ms, f16x3 {res[1][2]*1e3:.2f} ms")
for l in range(18):
    a0, a1 = res[0][0][l], res[1][0][l]
    e = ((a0 - a1).abs().max() / a0.abs().max()).item()
    flips = ((a0 > 0) != (a1 > 0)).sum().item()
    ez = ""
    if l < 17 and res[0][1][l] is not None:
        z0, z1 = res[0][1][l], res[1][1][l]
        ez = f"  Z+ rel {((z0 - z1).abs().max() / z0.abs().max()).item():.2e}  worst pointwise rel {(((z0 - z1).abs()) / z0.abs().clamp_min(1e-30))[z0 > 1e-6 * z0.max()].max().item():.2e}"
    print(f"act[{l:2d}] max-rel diff {e:.2e}  relu flips {flips}{ez}")
