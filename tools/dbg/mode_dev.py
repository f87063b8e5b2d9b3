#!/usr/bin/env python3
"""worst-map deviation of the VGG16 relevance chain in conv mode <m> from mode 0 (fp32 MFMA) on ONE trace (forward pass in mode 0),
+ per-layer times: python tools/dbg/mode_dev.py <mode> [images] [maps]"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrp_amd  # noqa
from lrp_amd import _lib, ops, weights
mode = int(sys.argv[1]); images = int(sys.argv[2]) if len(sys.argv) > 2 else 4; maps = int(sys.argv[3]) if len(sys.argv) > 3 else 40
lib = _lib.load()
lib.lrpx_set_conv_mode(0)
sd = weights.make_gridtd_state(seed=0, vocab_size=64)
names = [k for k in sd if k.startswith("img_encoder.encoder.") and k.endswith(".weight")]
vgg = ops.Vgg16([torch.from_numpy(sd[k]).cuda() for k in names], [torch.from_numpy(sd[k.replace(".weight", ".bias")]).cuda() for k in names])
img = torch.from_numpy(weights.make_images(0, images)).cuda()
f0 = vgg.forward(img).clone()
lib.lrpx_set_conv_mode(mode)
import time
f1 = vgg.forward(img).clone()
torch.cuda.synchronize(); t0 = time.time()
for _ in range(3): vgg.forward(img)
torch.cuda.synchronize(); t_f = (time.time() - t0) / 3
print(f"forward features mode {mode} vs mode 0: max diff {((f1 - f0).abs().max() / f0.abs().max()).item():.3e} of the maximum; forward {t_f * 1e3:.2f} ms for {images} images", flush=True)
lib.lrpx_set_conv_mode(0)
vgg.forward(img)
torch.manual_seed(0)
r_feat = torch.randn(maps, 196, 512, device="cuda")
m2i = (torch.arange(maps, device="cuda") * images // maps).to(torch.int32)
ref = vgg.relevance(r_feat, m2i).clone()
lib.lrpx_set_conv_mode(mode)
got = vgg.relevance(r_feat, m2i).clone()
torch.cuda.synchronize()
err = (got - ref).flatten(1).abs().amax(1) / ref.flatten(1).abs().amax(1)
print(f"mode {mode} vs mode 0 on one trace, {maps} maps of {images} images: worst map {err.max().item():.3e}, mean {err.mean().item():.3e}, "
      f"sum conservation {abs((got.sum() / ref.sum()).item() - 1):.2e}, finite {bool(torch.isfinite(got).all())}", flush=True)
ms = (C.c_float * 17)()
best = None
for _ in range(4):
    vgg.relevance(r_feat, m2i, out=got, layer_ms=ms)
    torch.cuda.synchronize()
    cur = [ms[i] for i in range(17)]
    best = cur if best is None else [min(a, b) for a, b in zip(best, cur)]
print(f"mode {mode}: chain {sum(best):.3f} ms  " + " ".join(f"{l}:{v:.3f}" for l, v in enumerate(best) if v > 0), flush=True)
