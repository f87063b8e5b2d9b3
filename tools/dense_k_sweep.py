#!/usr/bin/env python3
"""Many-row dense f16x3 GEMM (REL, out0 only) over K at fixed rows / columns: time = overhead (prologue + epilogue + launch) + slope * K.
The slope against the matrix time per K (3 fp16 MFMA products) says how well the K loop runs, the intercept what the rest costs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import ops, _lib
for n_maps, P, N, n_img in ((640, 36, 2048, 32), (320, 196, 512, 16)):
    res = []
    for K in (128, 256, 512, 1024, 2048):
        a = torch.randn(n_maps, P, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.05
        x = torch.randn(n_img, P, N, device="cuda"); m2i = (torch.arange(n_maps, device="cuda") * n_img // n_maps).to(torch.int32)
        out = torch.empty(n_maps, P, N, device="cuda")
        wh = ops.pack_weights_f16x2(w, K, N, _lib.PACK_BWD_PLAIN, taps=1)
        am = ops.amax_maps(a, n_maps)
        def run():
            ops.conv_mfma(a, wh, n_maps, 0, K, N, 1, _lib.EPI_REL, pix_per_map=P, oc_split=N, x=x, map2img=m2i, out0=out, f16x3=1, in_amax=am)
        for _ in range(5): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(30): run()
        e1.record(); e1.synchronize()
        res.append((K, e0.elapsed_time(e1) * 1000 / 30))
    slope = (res[-1][1] - res[2][1]) / (res[-1][0] - res[2][0])
    fl = 2.0 * n_maps * P * N * 3          # MFMA flops per unit of K
    print(f"{n_maps * P} rows x {N} cols: " + "  ".join(f"K={k}: {t:.1f} us" for k, t in res) +
          f"   slope {slope * 512:.1f} us per 512 of K = {fl / slope / 1e6:.0f} TFLOP/s of fp16 MFMA, intercept {res[2][1] - slope * 512:.1f} us")
