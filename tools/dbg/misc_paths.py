"""the rarely profiled entry points once each (for rocprofv3 --stats): LRP-inference decoding, forwardlrp_context, beam search, evaluation
consumers, heat maps, Grad-CAM / Guided-Grad-CAM, the generic add_lrp driver"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import lrp_amd
from lrp_amd import weights, evaluation as ev
from lrp_amd.explainers.gridtd import GridTDEngine
from lrp_amd.explainers.aoa import AOAEngine
V, B, T = 9586, 16, 20
wm = weights.make_word_map(V)
eng = GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=V))
img = torch.from_numpy(weights.make_images(1, B)).cuda()
cap = torch.from_numpy(weights.make_captions(2, B, T, V)).cuda()
for rep in range(3):
    enc = eng.encode(img)
    eng.sample_lrp(enc, T, wm['<start>'], wm['<end>'], [1, 2, 3])
    eng.forwardlrp_context(enc, cap, [T + 1] * B, [1, 2, 3])
    eng.greedy(enc, T + 1, wm['<start>'], wm['<end>'])
    enc1 = eng.encode(img[:1])
    eng.beam_search(enc1, 2, 50, wm['<start>'], wm['<end>'])
    maps, rw = eng.explain_batch(img, cap, accumulate=True)
    flat = maps.view(-1, 3, 224, 224)
    heat = ev.spatial_relevance(flat, "mean")
    ev.project_maxabs(heat); ev.block_image(heat); ev.map_statistics(heat)
    eng.explain_batch_gradient(img, cap, cam=True)
    eng.explain_batch_guided(img[:4], cap[:4], gradcam=True)
torch.cuda.synchronize()
aoa = AOAEngine(weights.make_aoa_state(seed=0, vocab_size=11027))
cap2 = torch.from_numpy(weights.make_captions(2, B, T, 11027)).cuda()
for rep in range(3):
    enc = aoa.encode(img)
    aoa.sample_lrp(enc, T, 11025, 11026, [1, 2, 3])
    aoa.beam_search(aoa.encode(img[:1]), 3, 20, 11025, 11026)
    aoa.explain_batch_gradient(cap2[:4], 3, img[:4], kind="guided_gradcam")
torch.cuda.synchronize()
