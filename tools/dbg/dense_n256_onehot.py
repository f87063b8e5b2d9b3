import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrp_amd
from lrp_amd import ops, _lib
n_maps, P, K, N, n_img = 120, 36, 512, 2048, 1
rows = n_maps * P
a = torch.zeros(rows, K, device="cuda")
a[torch.arange(rows), torch.arange(rows) % K] = 1.0
w = (torch.arange(K, device="cuda").view(-1, 1) * 4096 + torch.arange(N, device="cuda").view(1, -1)).float()
x = torch.ones(n_img, P, N, device="cuda")
m2i = torch.zeros(n_maps, dtype=torch.int32, device="cuda")
wp = ops.pack_weights_f16x2(w, K, N, _lib.PACK_BWD_PLAIN, taps=1)
am = ops.amax_maps(a.view(n_maps, P, K), n_maps)
out = torch.zeros(rows, N, device="cuda")
ops.conv_mfma(a, wp, n_maps, 0, K, N, 1, _lib.EPI_REL, pix_per_map=P, oc_split=N, x=x, map2img=m2i, out0=out, f16x3=1, in_amax=am)
torch.cuda.synchronize()
want = w[torch.arange(rows, device="cuda") % K]
bad = (out != want)
print("mismatches", bad.sum().item(), "of", bad.numel())
if bad.any():
    r, c = bad.nonzero(as_tuple=True)
    print("rows", r.unique()[:40].tolist(), "n", r.unique().numel())
    print("cols", c.unique()[:40].tolist(), "n", c.unique().numel())
    for i in range(0, min(10, r.numel())):
        rr, cc = r[i].item(), c[i].item()
        g = out[rr, cc].item(); print(rr, cc, "got", g, "-> k", int(g) // 4096, "col", int(g) % 4096, "want k", rr % K)
