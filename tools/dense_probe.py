#!/usr/bin/env python3
"""Latency of the lock-step dense rule GEMM (rows x 512 -> N, REL epilogue with a row gather) through lrpx_conv_mfma: the f16x3 few-row
kernel against the fp32 dense_small kernel, HIP events over 200 back-to-back launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import ops, _lib
for rows, N in ((320, 1536), (320, 2048), (640, 1536), (1280, 1536)):
    K = 512
    a = torch.randn(rows, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.05
    x = torch.randn(rows, N, device="cuda"); src = torch.randperm(rows, device="cuda").to(torch.int32)
    out = torch.empty(rows, N, device="cuda")
    wh = ops.pack_weights_f16x2(w, K, N, _lib.PACK_BWD_PLAIN, taps=1)
    wf = ops.pack_weights(w, K, N, 1, _lib.PACK_DENSE_T, 32)
    res = []
    for name, wp, f16 in (("f16x3", wh, 1), ("fp32", wf, 0)):
        for _ in range(20):
            ops.conv_mfma(a, wp, rows, 0, K, N, 1, _lib.EPI_REL, pix_per_map=1, oc_split=N, x=x, map2img=src, out0=out, f16x3=f16)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            ops.conv_mfma(a, wp, rows, 0, K, N, 1, _lib.EPI_REL, pix_per_map=1, oc_split=N, x=x, map2img=src, out0=out, f16x3=f16)
        e1.record(); e1.synchronize()
        res.append("%s %.1f us" % (name, e0.elapsed_time(e1) * 1000 / 200))
    print(f"rows {rows} x {K} -> {N}: " + "   ".join(res))
