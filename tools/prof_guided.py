import os, sys, torch
sys.path.insert(0, "/root/repo")
import lrp_amd
from lrp_amd import weights
from lrp_amd.explainers.gridtd import GridTDEngine
B,T,V=16,20,9586
eng=GridTDEngine(weights.make_gridtd_state(seed=0,vocab_size=V))
images=torch.from_numpy(weights.make_images(100,B)).cuda(); caps=torch.from_numpy(weights.make_captions(200,B,T,V)).cuda()
for _ in range(3): eng.explain_batch_guided(images,caps)
torch.cuda.synchronize()
