#!/usr/bin/env python3
"""Calibration of FETCH_SIZE / WRITE_SIZE on kernels of KNOWN traffic (run under rocprofv3 --pmc FETCH_SIZE  /  --pmc WRITE_SIZE):
a 1 GiB device copy (torch), lrpx_cumsum_maps over 320 maps (193 MB in, 193 MB out, float4 accesses) and lrpx_amax_maps (read only)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import ops
x = torch.rand(256 * 1024 * 1024, device="cuda")          # 1 GiB
for _ in range(3):
    y = x.clone()
m = torch.rand(320, 3, 224, 224, device="cuda")           # 192.7 MB
for _ in range(3):
    c = ops.cumsum_maps(m, 16, 20)
    a = ops.amax_maps(m, 320)
torch.cuda.synchronize()
print("done")
