#!/usr/bin/env python3
"""ReLU-mask and pool-winner flips of the GPU forward trace, per layer, against (a) the fp32 CPU forward of the oracle (oneDNN:
what the reference runs) and (b) an fp64 forward of the same weights.  A flip = a discrete decision (ReLU sign, 2x2 arg-max) taken
differently: every one moves relevance / gradient by a finite amount, so this is what the end-to-end deviations are made of.
Run once per forward variant (the switches latch per process):  LRPX_FWD_KSPLIT28=1 python tools/flip_probe.py"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import weights
from lrp_amd.explainers.gridtd import GridTDEngine
from oracle import lrp_oracle as O
torch.set_num_threads(8)
n_img = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sd = weights.make_gridtd_state(seed=0, vocab_size=64)
eng = GridTDEngine(sd)
img = torch.from_numpy(weights.make_images(0, n_img))
eng.vgg.forward(img.cuda())
torch.cuda.synchronize()
acts, zs = eng.vgg.trace_views()
sdt = O.state_to_torch(sd)
f32, _, s32 = O.vgg_forward(sdt, img)
f64, _, s64 = O.vgg_forward({k: v.double() for k, v in sdt.items()}, img.double())
s32.append(f32); s64.append(f64)
tot = {"relu32": 0, "relu64": 0, "pool32": 0, "pool64": 0, "ref_relu": 0, "ref_pool": 0}
print("layer            | GPU vs oneDNN fp32 | GPU vs fp64 | oneDNN fp32 vs fp64 | max-rel err GPU-fp64 / oneDNN-fp64")
for l, (kind, idx, cin, cout) in enumerate(O.vgg_layers()):
    hw, c = eng.vgg.ACT_DIMS[l + 1] if l + 1 < len(eng.vgg.ACT_DIMS) else (14, 512)
    out_gpu = acts[l + 1][:n_img].cpu().reshape(n_img, hw, hw, -1)[..., :c].permute(0, 3, 1, 2)
    o32, o64 = s32[l + 1], s64[l + 1]
    if kind == "conv":
        a, b, r = ((out_gpu > 0) != (o32 > 0)).sum().item(), ((out_gpu > 0) != (o64 > 0)).sum().item(), ((o32 > 0) != (o64 > 0)).sum().item()
        tot["relu32"] += a; tot["relu64"] += b; tot["ref_relu"] += r
        e1 = ((out_gpu.double() - o64).abs().max() / o64.abs().max()).item()
        e2 = ((o32.double() - o64).abs().max() / o64.abs().max()).item()
        print(f"conv {l:2d} ({hw:3d}^2) | relu flips {a:6d} | {b:6d} | {r:6d} | {e1:.2e} / {e2:.2e}")
    else:
        x_gpu = acts[l][:n_img].cpu().reshape(n_img, 2 * hw, 2 * hw, -1)[..., :c].permute(0, 3, 1, 2)
        _, ig = F.max_pool2d(x_gpu, 2, 2, return_indices=True)
        _, i32 = F.max_pool2d(s32[l], 2, 2, return_indices=True)
        _, i64 = F.max_pool2d(s64[l], 2, 2, return_indices=True)
        live = F.max_pool2d(s64[l], 2, 2) > 0
        a, b, r = ((ig != i32) & live).sum().item(), ((ig != i64) & live).sum().item(), ((i32 != i64) & live).sum().item()
        tot["pool32"] += a; tot["pool64"] += b; tot["ref_pool"] += r
        print(f"pool {l:2d}         | winner flips {a:4d} | {b:6d} | {r:6d} | of {int(live.sum())} live windows")
print("total ReLU flips: GPU-oneDNN %d, GPU-fp64 %d, oneDNN-fp64 %d; pool-winner flips: %d, %d, %d  (%d images; switches: %s)" % (
    tot["relu32"], tot["relu64"], tot["ref_relu"], tot["pool32"], tot["pool64"], tot["ref_pool"], n_img,
    " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("LRPX_")) or "defaults"))
