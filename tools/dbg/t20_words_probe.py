"""T = 20 gridTD r_words rows of the two golden images inside a B = 16 batch: distance to the reference's fp32 rows and to the fp64
evaluation, for the decoder GEMM kinds (fp32 / f16x3) - the data behind the bound of tests/test_gpu_t20.py.
usage: python tools/dbg/t20_words_probe.py   (GPU box; env switches of the library apply)"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import lrp_amd  # noqa: E402,F401
from lrp_amd import weights  # noqa: E402
from lrp_amd.explainers.gridtd import GridTDEngine  # noqa: E402
import test_gpu_t20 as t20  # noqa: E402

g = np.load(os.path.join(t20.GOLDEN, "t20.npz"))
g64 = np.load(os.path.join(t20.GOLDEN, "t20_f64.npz"))
V, B = int(g["grid_V"]), 16
T, caps = t20._batch(g, B, "grid_caption", V, 61)
eng = GridTDEngine(weights.make_gridtd_state(seed=int(g["seed"]), vocab_size=V))
imgs = t20._images(g, B)
for mode in (1, 0):
    eng.vgg.conv_mode = mode
    for f16 in (False, True):
        for b6 in ((True, False) if not f16 else (True,)):
            eng.force_f16, eng.dense_bf16x6 = f16, b6
            maps, r_words, r_feat, tr, enc = eng.explain_batch(imgs, caps, accumulate=True, return_features=True)
            torch.cuda.synchronize()
            rw = r_words.cpu().numpy()
            rows = []
            for k, p in enumerate(t20.POS):
                for t in range(T):
                    ref32, ref64 = g[f"grid{k}_r_words_{t}"], g64[f"grid{k}_r_words64_{t}"]
                    got = rw[p, t, :t + 1]
                    rows.append((float(np.abs(got - ref32).max()), float(np.abs(got.astype(np.float64) - ref64).max()),
                                 float(np.abs(ref32.astype(np.float64) - ref64).max()), k, t))
            rows.sort(reverse=True)
            print(f"conv mode {mode}, decoder {'f16x3' if f16 else 'fp32'}{'' if f16 else (' + bf16x6 rules' if b6 else ' (fp32 MFMA rules)')}: "
                  f"worst |GPU - ref32| {rows[0][0]:.2e}; top rows (vs ref32, vs fp64, ref32 vs fp64, image, word): "
                  + ", ".join(f"({a:.1e}, {b:.1e}, {c:.1e}, {k}, {t})" for a, b, c, k, t in rows[:4]))
