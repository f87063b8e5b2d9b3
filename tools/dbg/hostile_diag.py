"""CPU diagnosis of the split-product modes on hostile VGG16 weights: per layer of the relevance chain, the dynamic range of S = R / Z+
inside one map and what an fp16 hi/lo split behind one per-map power-of-two scale does to the layer's result (float64 reference)."""
import sys, os
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import lrp_amd
from lrp_amd import weights
from oracle import lrp_oracle as O
import test_gpu_vgg as T

FLUSH = "flush" in sys.argv
ZERO = "zero" in sys.argv
def split16(x):
    """per-map scale so that max lies in [2^14, 2^15), hi + lo in fp16 (subnormals kept, or flushed), back to float64"""
    m = x.abs().max()
    k = 14 - torch.floor(torch.log2(m))
    xs = (x * 2.0 ** k).float()
    hi = xs.half()
    lo = (xs - hi.float()).half()
    if FLUSH:
        hi = torch.where(hi.abs().float() < 2.0 ** -14, torch.zeros_like(hi), hi)
        lo = torch.where(lo.abs().float() < 2.0 ** -14, torch.zeros_like(lo), lo)
    return (hi.double() + lo.double()) * 2.0 ** (-k)

torch.set_num_threads(8)
fam = dict(sigma=float(sys.argv[1]) if len(sys.argv) > 1 else 1.5, dead_frac=0.10, heavy=True, bias_std=0.05)
balance = "balance" in sys.argv
sd = T._trained_like_vgg_state(41, fam["sigma"], fam["dead_frac"], fam["heavy"], fam["bias_std"])
sdt = O.state_to_torch(sd)
img = torch.from_numpy(weights.make_images(43, 1))
feats, _, saved = O.vgg_forward(sdt, img)
g = torch.Generator().manual_seed(3)
r = torch.randn(1, 512, 14, 14, generator=g) * (torch.rand(1, 512, 14, 14, generator=g) < 0.002)
r.view(-1)[12345] = 1e4
layers = O.vgg_layers()
R = r.double()
for l in range(len(layers) - 1, -1, -1):
    kind, idx, cin, cout = layers[l]
    x = saved[l].double()
    if kind == "pool":
        z = F.max_pool2d(x, 2, 2)
        s = R / (z + 1e-7 * (z == 0))
        _, ind = F.max_pool2d(x, 2, 2, return_indices=True)
        R = x * F.max_unpool2d(s, ind, 2, 2, output_size=x.shape[-2:])
        continue
    w = sdt[f"img_encoder.encoder.{idx}.weight"].double()
    wp, wn = w.clamp(min=0), w.clamp(max=0)
    xp, xn = x.clamp(min=0), x.clamp(max=0)
    z = F.conv2d(xp, wp, padding=1) + F.conv2d(xn, wn, padding=1)
    rs = torch.ones(cout, dtype=torch.float64)
    if balance:
        rowmax = (w.abs() if l == 0 else wp).amax(dim=(1, 2, 3))
        d = torch.where(rowmax > 0, 2.0 ** torch.floor(torch.log2(rowmax.clamp_min(1e-300))), torch.zeros_like(rowmax))
        rs = torch.where(d > 0, d.max() / d, torch.ones_like(d))
    zs = z * rs.view(1, -1, 1, 1)
    s = R / (zs + 1e-7 * (zs == 0))
    if ZERO:
        s = torch.where(zs == 0, torch.zeros_like(s), s)
    wps, wns = wp * rs.view(-1, 1, 1, 1), wn * rs.view(-1, 1, 1, 1)
    exact = xp * F.conv_transpose2d(s, wps, padding=1) + xn * F.conv_transpose2d(s, wns, padding=1)
    sq = split16(s)
    wq = split16(wps) if l else wps
    approx = xp * F.conv_transpose2d(sq, wq, padding=1) + xn * F.conv_transpose2d(sq, wns, padding=1)
    err = ((approx - exact).abs().max() / exact.abs().max()).item()
    a = s.abs().flatten(); a = a[a > 0]
    lg = torch.log2(a / a.max())
    big = a.argmax()
    cbig = (s.abs().flatten().argmax() // (s.shape[2] * s.shape[3])).item()
    print(f"layer {l:2d} conv{cin}->{cout} {x.shape[-1]:3d}^2: max|S| {a.max():.2e} median {a.median():.2e}  log2 quantiles rel. max: "
          f"1% {lg.quantile(0.01):.0f} 50% {lg.median():.0f} 99% {lg.quantile(0.99):.0f} 99.99% {lg.quantile(0.9999):.0f};  split error of this layer {err:.2e}; "
          f"max in channel {cbig}: rowmax W+ {wp[cbig].max():.2e} (layer max {wp.max():.2e}) bias {sdt[f'img_encoder.encoder.{idx}.bias'][cbig]:.2e} Z+ there {z.flatten()[s.abs().flatten().argmax()]:.2e}")
    R = exact
