#!/usr/bin/env python3
"""Throughput of the other explainers of gridTD at BASELINE config-2 size (16 images x 20 words): guided backprop,
plain gradient, Grad-CAM - maps/s next to the LRP number of bench.py (same engine, same synthetic inputs)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrp_amd  # noqa
from lrp_amd import weights
from lrp_amd.explainers.gridtd import GridTDEngine
B, T, V = 16, 20, 9586
eng = GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=V))
images = torch.from_numpy(weights.make_images(100, B)).cuda()
caps = torch.from_numpy(weights.make_captions(200, B, T, V)).cuda()
sizes = {}
runs = {"LRP (explain_batch)": lambda: eng.explain_batch(images, caps),
        "guided backprop (explain_batch_guided)": lambda: eng.explain_batch_guided(images, caps),
        "plain gradient (explain_batch_gradient)": lambda: eng.explain_batch_gradient(images, caps),
        "Grad-CAM (explain_batch_gradient cam=True)": lambda: eng.explain_batch_gradient(images, caps, cam=True)}
from lrp_amd.explainers.aoa import AOAEngine
VA = 11027
aoa = AOAEngine(weights.make_aoa_state(seed=0, vocab_size=VA))
caps_a = torch.from_numpy(weights.make_captions(201, B, T, VA)).cuda()
runs["AoA LRP, head 0 (config 3 model, B=16)"] = lambda: aoa.explain_batch(caps_a, 0, images=images)
bu = AOAEngine(weights.make_aoa_state(seed=0, vocab_size=VA, feat_dim=2048, with_encoder=False))
feats_bu = torch.from_numpy(weights.make_bu_features(5, 32)).cuda()
caps_b = torch.from_numpy(weights.make_captions(202, 32, T, VA)).cuda()
runs["AoA bottom-up LRP, head 0 (config 5, B=32, no CNN)"] = lambda: bu.explain_batch(caps_b, 0, features=feats_bu)
sizes = {"AoA bottom-up LRP, head 0 (config 5, B=32, no CNN)": 32 * T}
for name, fn in runs.items():
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:52s} {dt*1e3:8.2f} ms/step  {sizes.get(name, B*T)/dt:9.1f} maps/s")

# device-side consumers of the maps (evaluation.py reductions): 320 maps that never leave HBM
from lrp_amd import evaluation as ev
maps, _ = eng.explain_batch(images, caps)
m = maps.view(B * T, 3, 224, 224)
boxes = torch.tensor([[30, 40, 150, 200]] * (B * T))
def consumers():
    sp = ev.spatial_relevance(m, "mean")
    ev.block_image(sp, 8, 20)
    ev.overlapped_pixels(ev.project_maxabs(ev.spatial_relevance(m, "pos")), boxes)
    ev.map_statistics(sp)
consumers(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): consumers()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
gb = B * T * 224 * 224 * 4 * (3 + 1 + 1 + 1 + 3 + 1 + 1 + 1 + 10 + 1) / 1e9
print(f"{'map consumers (mask, bbox ratios x10, statistics)':52s} {dt*1e3:8.2f} ms/step  {B*T/dt:9.1f} maps/s  ~{gb/dt:.0f} GB/s of algorithmic traffic")

sp_ = ev.spatial_relevance(m, "mean")
ev.map_quantiles(sp_); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): ev.map_quantiles(sp_)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 10
print(f"{'100-point quantiles (per-map sort)':52s} {dt*1e3:8.2f} ms/step  {B*T/dt:9.1f} maps/s")

# LRP-inference decoding (GridTDEngine.sample_lrp) next to plain greedy decoding: 16 images x 20 words
enc = eng.encode(images)
wm = weights.make_word_map(V)
skip = [wm[k] for k in ('<start>', '<end>', '<pad>', '<unk>')]
for name, fn in (("greedy decoding (greedy)", lambda: eng.greedy(enc, T + 1, wm['<start>'], wm['<end>'])),
                 ("LRP-inference decoding (sample_lrp)", lambda: eng.sample_lrp(enc, T, wm['<start>'], wm['<end>'], skip))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"{name:52s} {dt*1e3:8.2f} ms/step  {B*T/dt:9.1f} words/s")

# AoA LRP with three batches in flight (AOAEngine.explain_stream), config-3 model
def aoa_stream(n):
    return sum(1 for _ in aoa.explain_stream(((images, caps_a) for _ in range(n)), 0, depth=3))
aoa_stream(4); torch.cuda.synchronize()
t0 = time.perf_counter()
aoa_stream(12)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 12
print(f"{'AoA LRP, head 0, 3 batches in flight':52s} {dt*1e3:8.2f} ms/step  {B*T/dt:9.1f} maps/s")
