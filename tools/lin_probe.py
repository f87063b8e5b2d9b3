#!/usr/bin/env python3
"""Latency of the decoder's skinny GEMMs at the headline shapes (B=16): lrpx_linear_small and the lock-step eps-rule GEMM."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import _lib
lib = _lib.load()
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (B, K, N) in [(16, 1536, 2560), (16, 1536, 2048), (16, 512, 9600), (64, 1536, 2560)]:
    x = torch.randn(B, K, device="cuda"); w = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
    out = torch.empty(B, N, device="cuda")
    # two weight sets alternated (the decoder alternates LSTM1 / LSTM2 / attention between calls)
    w2 = torch.randn(N, K, device="cuda") * 0.02
    ws = [w, w2]
    i = [0]
    def f():
        i[0] ^= 1
        _lib.check(lib.lrpx_linear_small(_lib.ptr(x), K, _lib.ptr(ws[i[0]]), _lib.ptr(b), _lib.ptr(out), N, B, K, N, 0, _lib.stream_ptr()))
    us = timeit(f)
    ref = x @ w2.t() + b if i[0] == 1 else x @ w.t() + b
    print(f"linear_small B={B} K={K} N={N}: {us:.1f} us  ({N*K*4/us/1e6:.2f} TB/s of weights)  err {float((out-ref).abs().max()):.2e}")
