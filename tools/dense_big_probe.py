#!/usr/bin/env python3
"""Latency of the many-row dense f16x3 GEMM (REL epilogue) for the shapes of configs 2 and 5; LRPX_DENSE_WIDE=1 -> 256-row tiles."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import ops, _lib
for n_maps, P, K, N, n_img in ((320, 196, 512, 512, 16), (640, 36, 512, 2048, 32), (640, 36, 512, 512, 32), (1280, 196, 512, 512, 64)):
    a = torch.randn(n_maps, P, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.05
    x = torch.randn(n_img, P, N, device="cuda"); m2i = (torch.arange(n_maps, device="cuda") * n_img // n_maps).to(torch.int32)
    out = torch.empty(n_maps, P, N, device="cuda")
    wh = ops.pack_weights_f16x2(w, K, N, _lib.PACK_BWD_PLAIN, taps=1)
    am = ops.amax_maps(a, n_maps)
    for _ in range(5):
        ops.conv_mfma(a, wh, n_maps, 0, K, N, 1, _lib.EPI_REL, pix_per_map=P, oc_split=N, x=x, map2img=m2i, out0=out, f16x3=1, in_amax=am)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.conv_mfma(a, wh, n_maps, 0, K, N, 1, _lib.EPI_REL, pix_per_map=P, oc_split=N, x=x, map2img=m2i, out0=out, f16x3=1, in_amax=am)
    e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1000 / 50
    fl = 2.0 * n_maps * P * K * N
    print(f"{n_maps * P} x {K} -> {N}: {us:.1f} us = {fl / us / 1e6:.0f} algorithmic TFLOP/s, {(a.numel() + out.numel()) * 4 / us / 1e3:.0f} GB/s of A + out")
