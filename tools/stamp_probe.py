#!/usr/bin/env python3
"""Phase breakdown of the bf16x6 conv kernels of conv_inst_x6.hip (56/28/14 relevance) from s_memtime stamps.
Needs a profiling build: make -C lrp-imagecaptioning-pytorch_amd/csrc clean && make ... STAMP=1"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa: E402,F401
from lrp_amd import _lib, ops, weights  # noqa: E402

lib = _lib.load()
if os.environ.get("LRPX_CONV_MODE"):
    lib.lrpx_set_conv_mode(int(os.environ["LRPX_CONV_MODE"]))
sd = weights.make_gridtd_state(seed=0, vocab_size=64)
names = [k for k in sd if k.startswith("img_encoder.encoder.") and k.endswith(".weight")]
vgg = ops.Vgg16([torch.from_numpy(sd[k]).cuda() for k in names],
                [torch.from_numpy(sd[k.replace(".weight", ".bias")]).cuda() for k in names])
img = torch.from_numpy(weights.make_images(0, 16)).cuda()
vgg.forward(img)
r_feat = torch.randn(320, 196, 512, device="cuda")
m2i = (torch.arange(320, device="cuda") * 16 // 320).to(torch.int32)
out = vgg.relevance(r_feat, m2i)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 12)()
raw = C.CDLL(_lib.LIB_PATH)
fn = getattr(raw, sys.argv[1] if len(sys.argv) > 1 else "lrpx_debug_stamps")   # _h3 (56/28/14) / _h3b (224/112) / bf16x6
fn(buf, 1)
vgg.relevance(r_feat, m2i, out=out)
torch.cuda.synchronize()
fn(buf, 0)
v = list(buf)
n = max(v[7], 1)
names = ["prologue", "issue(next chunk loads)", "mfma phase", "commit(split+ds_write)", "barrier", "epilogue", "total"]
print(f"waves: {v[7]}   ({sys.argv[1:]} one chain pass of 320 maps)")
for k, x in zip(names, v[:7]):
    print(f"  {k:28s} {x / n:12.0f} clk/wave  {100.0 * x / max(v[6], 1):6.1f} %")
if v[8]:
    for k, x in zip(["mfma taps 0-2", "mfma taps 3-5", "mfma taps 6-8"], v[8:11]):
        print(f"    {k:26s} {x / n:12.0f} clk/wave  {100.0 * x / max(v[6], 1):6.1f} %")
