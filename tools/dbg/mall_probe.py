"""Does the first-layer kernel read S1 faster when conv1_2 just wrote it and it fits the 256 MB Infinity Cache?  Per-layer HIP-event times of the
chain for few maps (S1 = 12.8 MB per map) against many."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrp_amd  # noqa
from lrp_amd import ops, weights
sd = weights.make_gridtd_state(seed=0, vocab_size=64)
names = [k for k in sd if k.startswith("img_encoder.encoder.") and k.endswith(".weight")]
vgg = ops.Vgg16([torch.from_numpy(sd[k]).cuda() for k in names], [torch.from_numpy(sd[k.replace(".weight", ".bias")]).cuda() for k in names])
img = torch.from_numpy(weights.make_images(0, 16)).cuda()
vgg.forward(img)
for maps in (4, 8, 16, 32, 64, 128, 320):
    r_feat = torch.randn(maps, 196, 512, device="cuda")
    m2i = (torch.arange(maps, device="cuda") * 16 // maps).to(torch.int32)
    out = vgg.relevance(r_feat, m2i)
    lm = (C.c_float * 17)()
    best = None
    for _ in range(4):
        vgg.relevance(r_feat, m2i, out=out, layer_ms=lm)
        v = list(lm)
        best = v if best is None else [min(a, b) for a, b in zip(best, v)]
    print(f"{maps:4d} maps: first layer {best[0] / maps * 1e3:6.2f} us/map  conv1_2 {best[1] / maps * 1e3:6.2f} us/map  conv2_1 {best[3] / maps * 1e3:6.2f}  conv2_2 {best[4] / maps * 1e3:6.2f}  chain {sum(best) / maps * 1e3:7.2f} us/map")
