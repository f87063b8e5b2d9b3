import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import lrp_amd
from lrp_amd import weights, ops
from lrp_amd.explainers.gridtd import GridTDEngine
g=np.load(os.path.join(os.environ.get("GRAFT_REPO_ROOT","/root/repo"),'tests/golden/gridtd_T3.npz'))
sd=weights.make_gridtd_state(seed=0,vocab_size=9586)
eng=GridTDEngine(sd)
img=torch.from_numpy(weights.make_images(0,1))
cap=torch.from_numpy(g['caption']).view(1,-1)
maps,_=eng.explain_batch(img,cap,accumulate=True)
m=maps[0,2].cpu().double(); w=torch.from_numpy(g['map_full_2'][0]).double()
d=(m-w).abs()/w.abs().max()
print('relL2',((m-w).norm()/w.norm()).item(),'max',d.max().item())
for q in (0.9,0.99,0.999,0.9999): print(q, torch.quantile(d.flatten(),q).item())
print('frac>1e-4',(d>1e-4).double().mean().item(),'frac>1e-3',(d>1e-3).double().mean().item())
# count pool flips between GPU forward and oracle forward
from oracle import lrp_oracle as O
import torch.nn.functional as F
sdt=O.state_to_torch(sd)
feats,_,saved=O.vgg_forward(sdt,img)
acts,zs=eng.vgg.trace_views()
for l,(kind,idx,cin,cout) in enumerate(O.vgg_layers()):
    if kind=='pool':
        hw,c=eng.vgg.ACT_DIMS[l]
        ga=acts[l][0].cpu().reshape(hw,hw,c).permute(2,0,1)[None]
        _,ia=F.max_pool2d(ga,2,2,return_indices=True); _,ib=F.max_pool2d(saved[l],2,2,return_indices=True)
        act=F.max_pool2d(saved[l],2,2)>0
        print('pool',l,'flips',int(((ia!=ib)&act).sum()),'of',int(act.sum()))
