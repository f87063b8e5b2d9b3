#!/usr/bin/env python3
"""Where does the AoA r_words error at T=20 come from?  Row (golden image 0, word 17) of tests/golden/t20.npz with the
forward trace on the f16x3 kernels (default) and on the exact-split bf16x6 kernels, and the feature error of both."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lrp_amd  # noqa
from lrp_amd import weights, _lib
from lrp_amd.explainers.aoa import AOAEngine
from oracle import lrp_oracle as O
g = np.load(os.path.join(ROOT, "tests/golden/t20.npz"))
T, V = 20, int(g["aoa_V"])
lib = _lib.load()
sd = weights.make_aoa_state(seed=0, vocab_size=V)
eng = AOAEngine(sd)
imgs = torch.from_numpy(weights.make_images(50, 2))
caps = torch.from_numpy(g["aoa_caption"])
sdt = O.state_to_torch(sd)
f64 = {k: v.double() for k, v in sdt.items()}
feats64, _, _ = O.vgg_forward(f64, imgs.double())
feats32, _, _ = O.vgg_forward(sdt, imgs)
print("cpu fp32 forward vs fp64: %.2e" % ((feats32.double() - feats64).abs().max() / feats64.abs().max()).item())
for fwd16 in (1, 0):
    lib.lrpx_set_forward_f16(fwd16)
    enc = eng.encode(imgs.cuda())
    fe = enc["feats"].cpu().double().reshape(2, 196, 512).permute(0, 2, 1).reshape(2, 512, 14, 14)
    print("forward_f16=%d: features vs fp64 %.2e" % (fwd16, ((fe - feats64).abs().max() / feats64.abs().max()).item()))
    tr = eng.trace(enc, caps.cuda(), predictions=False)
    r_feat, r_words, _ = eng.relevance(enc, tr, 0)
    rw = r_words.view(2, T, T).cpu()
    for k in range(2):
        errs = [np.abs(rw[k, t, :t + 1].numpy() - g[f"aoa{k}_h0_r_words_{t}"]).max() for t in range(T)]
        print("   image %d r_words worst %.2e at t=%d; all: %s" % (k, max(errs), int(np.argmax(errs)), " ".join("%.0e" % e for e in errs)))
lib.lrpx_set_forward_f16(1)
