"""GPU diagnosis (one box): the failing map of test_chain_hostile_weights_all_modes layer by layer.  The exact chain runs on the CPU in
float64; at every conv layer the GPU's single-layer rule (tests/test_gpu_vgg.py:gpu_conv_rule, conv modes f16x3 = 1 / 2 and fp32) gets the
EXACT relevance of the layer's output and is compared with the exact result of that layer: separates per-layer operand errors from
chain effects (recorded amax words, pooled staging, derived multiplicands)."""
import sys, os
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lrp_amd
from lrp_amd import weights, ops, _lib
from oracle import lrp_oracle as O
import test_gpu_vgg as T

torch.set_num_threads(16)
sd = T._trained_like_vgg_state(41, 1.5, 0.10, True, 0.05)
sdt = O.state_to_torch(sd)
img2 = torch.from_numpy(weights.make_images(43, 2))
g = np.load(os.path.join(ROOT, "tests", "golden", "gridtd_T3.npz"))
vgg = T._vgg(ops, sd)
feats = vgg.forward(img2.cuda())
m2i = (torch.arange(40, device="cuda") * 2 // 40).to(torch.int32)
r_feat, names = T._hostile_targets(feats, m2i, 40, g)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
b = int(m2i[k])
img = img2[b:b + 1]
_, _, saved = O.vgg_forward(sdt, img)
R = T.from_nhwc(r_feat[k:k + 1].cpu(), 512, 14, 14).double()
print("map", k, names[k], "image", b, "max|r_feat|", float(R.abs().max()))
layers = O.vgg_layers()
for l in range(len(layers) - 1, 0, -1):
    kind, idx, cin, cout = layers[l]
    x = saved[l].double()
    if kind == "pool":
        z = F.max_pool2d(x, 2, 2)
        s = R / (z + 1e-7 * (z == 0))
        _, ind = F.max_pool2d(x, 2, 2, return_indices=True)
        R = x * F.max_unpool2d(s, ind, 2, 2, output_size=x.shape[-2:])
        continue
    w = sdt[f"img_encoder.encoder.{idx}.weight"].double()
    z = F.conv2d(x, w.clamp(min=0), padding=1)
    s = R / (z + 1e-7 * (z == 0))
    exact = x * F.conv_transpose2d(s, w.clamp(min=0), padding=1)
    line = f"layer {l:2d} conv{cin}->{cout} {x.shape[-1]:3d}^2: max|S| {float(s.abs().max()):.2e} max|R_in| {float(exact.abs().max()):.2e}"
    if x.shape[-1] <= 56:
        for name, kw in (("fp32", {}), ("f16x3", dict(f16x3=1)), ("f16+f6", dict(f16x3=2))):
            got, _, zg = T.gpu_conv_rule(ops, x.float(), w.float(), R.float(), **kw)
            e = ((got.double() - exact).abs().max() / exact.abs().max()).item()
            line += f"  {name} {e:.2e}"
    print(line, flush=True)
    R = exact
