#!/usr/bin/env python3
"""per-layer times of the VGG16 relevance chain by its own HIP events: python tools/dbg/chain_time.py <mode> [images] [maps]
(A/B of two builds on one box: LRPX_LIB_PATH=<other liblrpx.so>); prints a sha of the maps too"""
import ctypes as C, hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrp_amd  # noqa
from lrp_amd import _lib, ops, weights
mode = int(sys.argv[1]); images = int(sys.argv[2]) if len(sys.argv) > 2 else 16; maps = int(sys.argv[3]) if len(sys.argv) > 3 else 320
_lib.load().lrpx_set_conv_mode(mode)
sd = weights.make_gridtd_state(seed=0, vocab_size=64)
names = [k for k in sd if k.startswith("img_encoder.encoder.") and k.endswith(".weight")]
vgg = ops.Vgg16([torch.from_numpy(sd[k]).cuda() for k in names], [torch.from_numpy(sd[k.replace(".weight", ".bias")]).cuda() for k in names])
img = torch.from_numpy(weights.make_images(0, images)).cuda()
vgg.forward(img)
torch.manual_seed(0)
r_feat = torch.randn(maps, 196, 512, device="cuda")
m2i = (torch.arange(maps, device="cuda") * images // maps).to(torch.int32)
res = vgg.relevance(r_feat, m2i)
ms = (C.c_float * 17)()
best = None
for _ in range(6):
    vgg.relevance(r_feat, m2i, out=res, layer_ms=ms)
    torch.cuda.synchronize()
    cur = [ms[i] for i in range(17)]
    best = cur if best is None else [min(a, b) for a, b in zip(best, cur)]
sha = hashlib.sha1(res.cpu().numpy().tobytes()).hexdigest()[:12]
print(f"mode {mode} lib {os.path.basename(os.environ.get('LRPX_LIB_PATH', 'in-tree'))}: chain {sum(best):.3f} ms  " + " ".join(f"{l}:{v:.3f}" for l, v in enumerate(best) if v > 0) + f"  sha {sha}", flush=True)
