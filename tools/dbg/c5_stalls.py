"""where does the config-5 step stall?  completion gaps of the pipelined steps + allocator activity inside the timed region"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import lrp_amd
from lrp_amd import weights
from lrp_amd.explainers.aoa import AOAEngine
B, T, V, NP = 32, 20, 11027, int(sys.argv[1]) if len(sys.argv) > 1 else 3
eng = AOAEngine(weights.make_aoa_state(seed=0, vocab_size=V, feat_dim=2048, with_encoder=False))
if len(sys.argv) > 2:
    eng.fused_rel = sys.argv[2] != "nofuse"
feats = torch.from_numpy(weights.make_bu_features(100, B)).cuda()
caps = torch.from_numpy(weights.make_captions(200, B, T, V)).cuda()
engines = [eng] + [eng.replica() for _ in range(NP - 1)]
for e in engines:
    e.fused_rel = eng.fused_rel
streams = [torch.cuda.Stream() for _ in range(NP)]
def step(k):
    with torch.cuda.stream(streams[k]):
        e = engines[k]
        enc = e.encode(features=feats)
        tr = e.trace(enc, caps, predictions=True)
        out = e.relevance(enc, tr, 0)
        ev = torch.cuda.Event(enable_timing=True); ev.record()
    return ev, out
for i in range(12):
    step(i % NP)
torch.cuda.synchronize()
st0 = torch.cuda.memory_stats()
ev0 = torch.cuda.Event(enable_timing=True); ev0.record()
t0 = time.perf_counter()
evs, host = [], []
N = 200
for i in range(N):
    h0 = time.perf_counter()
    ev, _ = step(i % NP)
    host.append(time.perf_counter() - h0)
    evs.append(ev)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st1 = torch.cuda.memory_stats()
ends = [ev0.elapsed_time(e) for e in evs]
gaps = [ends[i] - ends[i - 1] for i in range(1, N)]
host_ms = sorted(h * 1e3 for h in host)
print(f"pipeline {NP} fused_rel {eng.fused_rel}: wall {dt / N * 1e3:.3f} ms/step; host issue time per step: median {host_ms[N // 2]:.3f} ms, "
      f"p90 {host_ms[int(N * 0.9)]:.3f}, max {host_ms[-1]:.3f}; sum {sum(host) * 1e3:.1f} ms of {dt * 1e3:.1f}")
print("device allocs during the region:", st1["num_device_alloc"] - st0["num_device_alloc"], "retries", st1["num_alloc_retries"] - st0["num_alloc_retries"],
      "allocation calls", st1["allocation.all.allocated"] - st0["allocation.all.allocated"])
big = [(i, round(g, 2)) for i, g in enumerate(gaps) if g > 2.0]
print("completion gaps > 2 ms:", big[:20], "count", len(big))
