#!/usr/bin/env python3
"""distance of the GPU forward trace's features from an fp64 evaluation of the same VGG16 (1 image, CPU double), per forward variant:
python tools/dbg/fwd_f64_probe.py   (LRPX_FWD_KSPLIT28=4 etc. in the environment select the switches)"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrp_amd  # noqa
from lrp_amd import _lib, ops, weights
from oracle import lrp_oracle as O
lib = _lib.load()
sd = weights.make_gridtd_state(seed=0, vocab_size=64)
names = [k for k in sd if k.startswith("img_encoder.encoder.") and k.endswith(".weight")]
vgg = ops.Vgg16([torch.from_numpy(sd[k]).cuda() for k in names], [torch.from_numpy(sd[k.replace(".weight", ".bias")]).cuda() for k in names])
img = torch.from_numpy(weights.make_images(0, 1))
x = img.double()
torch.set_num_threads(16)
for kind, idx, cin, cout in O.vgg_layers():
    if kind == "conv":
        x = F.relu(F.conv2d(x, torch.from_numpy(sd[f"img_encoder.encoder.{idx}.weight"]).double(), torch.from_numpy(sd[f"img_encoder.encoder.{idx}.bias"]).double(), padding=1))
    else:
        x = F.max_pool2d(x, 2, 2)
f64 = x[0].reshape(512, 196).t()           # (P, C)
x32 = img
for kind, idx, cin, cout in O.vgg_layers():
    if kind == "conv":
        x32 = F.relu(F.conv2d(x32, torch.from_numpy(sd[f"img_encoder.encoder.{idx}.weight"]), torch.from_numpy(sd[f"img_encoder.encoder.{idx}.bias"]), padding=1))
    else:
        x32 = F.max_pool2d(x32, 2, 2)
print(f"CPU fp32 (oneDNN) forward vs fp64: {((x32[0].reshape(512, 196).t().double() - f64).abs().max() / f64.abs().max()).item():.2e} of the feature maximum")
for name, mode, f16 in (("mode 0 fp32 MFMA", 0, 0), ("mode 1 exact bf16 splits", 1, 0), ("mode 2 + fp16 forward", 2, 1)):
    vgg.conv_mode, vgg.forward_f16 = mode, f16
    f = vgg.forward(img.cuda())[0].double().cpu()
    e = (f - f64).abs()
    print(f"{name}: max {(e.max() / f64.abs().max()).item():.2e}, rms {(e.pow(2).mean().sqrt() / f64.abs().max()).item():.2e} of the feature maximum (switches: "
          f"KSPLIT14={os.environ.get('LRPX_FWD_KSPLIT', '8')} KSPLIT28={os.environ.get('LRPX_FWD_KSPLIT28', '1')})", flush=True)
