#!/usr/bin/env python3
"""Latency of the few-row dense f16x3 rule (the lock-step gate rules: 640 rows x 512 -> 1536 at config 5, 320 at config 2)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import ops, _lib
for rows, K, N, n_src in ((640, 512, 1536, 32), (320, 512, 1536, 16), (1280, 512, 2048, 64)):
    a = torch.randn(rows, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.05
    x = torch.randn(n_src, N, device="cuda"); u = torch.randn(rows, N, device="cuda")
    src = (torch.arange(rows, device="cuda") * n_src // rows).to(torch.int32)
    out = torch.empty(rows, N, device="cuda")
    wh = ops.pack_weights_f16x2(w, K, N, _lib.PACK_BWD_PLAIN, taps=1)
    def run():
        ops.conv_mfma(a, wh, rows, 0, K, N, 1, _lib.EPI_REL, pix_per_map=1, oc_split=N, x=x, u=u, map2img=src, out0=out, f16x3=1)
    for _ in range(10): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); e1.synchronize()
    print(f"{rows} x {K} -> {N}: {e0.elapsed_time(e1) * 1000 / 200:.2f} us per launch (back to back on one stream)")
