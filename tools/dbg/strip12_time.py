#!/usr/bin/env python3
"""(needs tools/experiments/r5_conv12_strip_kernel.patch applied and `make`: the strip kernel is not in the shipped library)
per-layer times of the relevance chain with the current LRPX_STRIP12 / LRPX_S12_DBG environment (diagnosis)"""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrp_amd  # noqa
from lrp_amd import ops, weights
images, maps = int(sys.argv[1]), int(sys.argv[2])
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
for _ in range(4):
    vgg.relevance(r_feat, m2i, out=res, layer_ms=ms)
    torch.cuda.synchronize()
    cur = [ms[i] for i in range(17)]
    best = cur if best is None else [min(a, b) for a, b in zip(best, cur)]
print(f"STRIP12={os.environ.get('LRPX_STRIP12')} DBG={os.environ.get('LRPX_S12_DBG')}: conv1_2 {best[1]:.3f} ms", flush=True)
