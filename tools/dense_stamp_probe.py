#!/usr/bin/env python3
"""Timeline of dense_f16x3_n256_kernel from per-wave s_memtime stamps (a -DLRPX_EXPERIMENTS -DLRPX_STAMP build of dense_f16x3.hip:
csrc/variant.sh dnstamp dense_f16x3.hip -DLRPX_EXPERIMENTS -DLRPX_STAMP; LRPX_LIB_PATH=.../variants/liblrpx_dnstamp.so):
phase lengths per wave, and per CU how much of the time its two workgroups are in which phase."""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import ops, _lib
raw = C.CDLL(os.environ.get("LRPX_LIB_PATH", _lib.LIB_PATH))
shapes = ((640, 36, 512, 2048, 32), (320, 196, 512, 512, 16))
for n_maps, P, K, N, n_img in shapes:
    a = torch.randn(n_maps, P, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.05
    x = torch.randn(n_img, P, N, device="cuda"); m2i = (torch.arange(n_maps, device="cuda") * n_img // n_maps).to(torch.int32)
    out = torch.empty(n_maps, P, N, device="cuda")
    wh = ops.pack_weights_f16x2(w, K, N, _lib.PACK_BWD_PLAIN, taps=1)
    am = ops.amax_maps(a, n_maps)
    def run():
        ops.conv_mfma(a, wh, n_maps, 0, K, N, 1, _lib.EPI_REL, pix_per_map=P, oc_split=N, x=x, map2img=m2i, out0=out, f16x3=1, in_amax=am)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); run(); e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1000
    m_tiles, n_blocks = -(-n_maps * P // 128), -(-N // 256)
    grid = -(-m_tiles * n_blocks // 8) * 8
    nw = grid * 4
    buf = np.zeros(nw * 10, dtype=np.uint64)
    assert raw.lrpx_debug_stamps_dn(buf.ctypes.data_as(C.c_void_p), nw, 0) == 0
    r = buf.reshape(nw, 10).astype(np.int64)
    r = r[r[:, 0] > 0]
    hw, xcc = r[:, 8], r[:, 9] & 15
    cu = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
    simd = (hw >> 4) & 3
    pro, loop, epi, drain = r[:, 1] - r[:, 0], r[:, 2] - r[:, 1], r[:, 3] - r[:, 2], r[:, 4] - r[:, 3]
    spans = [r[cu == c, 4].max() - r[cu == c, 0].min() for c in np.unique(cu)]          # (one counter per XCD: compare inside a CU only)
    span = max(spans)
    tpu = span / us
    print(f"== {n_maps * P} x {K} -> {N}: {us:.1f} us, {len(r)} waves, {len(np.unique(cu))} CUs, longest CU span {span} ticks = {tpu:.0f} ticks/us")
    f = lambda v: f"{v.mean() / tpu:6.2f} us (min {v.min() / tpu:.2f} max {v.max() / tpu:.2f})"
    print(f"   prologue {f(pro)}\n   K loop   {f(loop)}   of it: issue+reads+MFMA {r[:, 5].mean() / tpu:.2f}, commit {r[:, 6].mean() / tpu:.2f}, barrier {r[:, 7].mean() / tpu:.2f}"
          f"\n   epilogue {f(epi)}\n   drain    {f(drain)}\n   wave     {f(r[:, 4] - r[:, 0])}")
    # per CU: fraction of the kernel's span with 0 / 1 / 2 waves of SIMD 0 inside their K loop, and workgroups per CU
    occ = np.zeros(3); nwg = []
    for c in np.unique(cu):
        sel = r[(cu == c) & (simd == 0)]
        nwg.append(len(sel))
        ev = sorted([(t, 1) for t in sel[:, 1]] + [(t, -1) for t in sel[:, 2]])
        lo, hi = r[cu == c, 0].min(), r[cu == c, 4].max()
        t_prev, k = lo, 0
        for t, d in ev:
            occ[min(k, 2)] += t - t_prev
            t_prev, k = t, k + d
        occ[min(k, 2)] += hi - t_prev
    occ /= occ.sum()
    print(f"   SIMD 0 of a CU: {100 * occ[0]:.0f} % of its time no wave in the K loop, {100 * occ[1]:.0f} % one, {100 * occ[2]:.0f} % two;  waves of SIMD 0 per CU: min {min(nwg)} max {max(nwg)}")
    c = np.unique(cu)[5]
    sel = r[(cu == c) & (simd == 0)]
    sel = sel[np.argsort(sel[:, 0])]
    t00 = sel[:, 0].min()
    print("   block ids on that CU:", " ".join(str(int(v >> 8)) for v in sel[:, 9]), " and on the next:", " ".join(str(int(v >> 8)) for v in r[(cu == np.unique(cu)[6]) & (simd == 0)][:, 9]))
    print("   one CU, SIMD 0, (start, loop start, loop end, end) in us: " + "  ".join("(" + " ".join(f"{(v - t00) / tpu:.0f}" for v in (w_[0], w_[1], w_[2], w_[4])) + ")" for w_ in sel))
