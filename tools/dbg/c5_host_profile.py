"""host-side cost of a config-5 step: cProfile over 300 steps (one stream; the GPU runs behind)"""
import sys, os, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import lrp_amd
from lrp_amd import weights
from lrp_amd.explainers.aoa import AOAEngine
B, T, V = 32, 20, 11027
eng = AOAEngine(weights.make_aoa_state(seed=0, vocab_size=V, feat_dim=2048, with_encoder=False))
feats = torch.from_numpy(weights.make_bu_features(100, B)).cuda()
caps = torch.from_numpy(weights.make_captions(200, B, T, V)).cuda()
def step():
    enc = eng.encode(features=feats)
    tr = eng.trace(enc, caps, predictions=True)
    return eng.relevance(enc, tr, 0)
for _ in range(10):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
