#!/usr/bin/env python3
"""one image, 20 maps: the VGG16 relevance chain on 1 / 2 / 4 HIP streams (ops.Vgg16.relevance(streams=k): the maps are independent, every
stream runs the whole chain on its share) - time by HIP events and bit-identity against the one-stream chain, per conv mode"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrp_amd  # noqa
from lrp_amd import weights
from lrp_amd.explainers.gridtd import GridTDEngine
V, T = 9586, 20
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=V))
img = torch.from_numpy(weights.make_images(100, B)).cuda()
cap = torch.from_numpy(weights.make_captions(200, B, T, V)).cuda()
for mode in (1, 3):
    eng.vgg.conv_mode = mode
    enc = eng.encode(img)
    tr = eng.trace(enc, cap, predictions=False)
    r_feat, r_words, row2img = eng.relevance(enc, tr)
    ref = None
    for k in (1, 2, 3, 4, 5):
        out = torch.empty(B * T, 3, 224, 224, device="cuda")
        for _ in range(3):
            eng.vgg.relevance(r_feat, row2img, out=out, streams=k)
        torch.cuda.synchronize()
        ts = []
        for _ in range(15):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.vgg.relevance(r_feat, row2img, out=out, streams=k); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ms = sorted(ts)[len(ts) // 2]
        if ref is None:
            ref = out.clone()
        print(f"conv mode {mode}, {B * T} maps, {k} stream(s): chain {ms:.2f} ms, bit-identical to one stream: {torch.equal(out, ref)}", flush=True)
