import os, sys, torch, subprocess, numpy as np
sys.path.insert(0, "/root/repo")
import lrp_amd
from lrp_amd import weights, ops
sd = weights.make_gridtd_state(seed=0, vocab_size=64)
names = [k for k in sd if k.startswith("img_encoder.encoder.") and k.endswith(".weight")]
vgg = ops.Vgg16([torch.from_numpy(sd[k]).cuda() for k in names], [torch.from_numpy(sd[k.replace(".weight", ".bias")]).cuda() for k in names])
img = torch.from_numpy(weights.make_images(0, int(sys.argv[1]))).cuda()
f = vgg.forward(img)
torch.cuda.synchronize()
np.save(sys.argv[2], f.cpu().numpy())
print("ok", float(f.abs().max()), float(f.abs().mean()))
