"""debug: where do the f16+fp6 kernels differ from the fp32 kernel?  (single conv layer, structured by pixel / channel)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import lrp_amd  # noqa
from lrp_amd import ops
from test_gpu_vgg import gpu_conv_rule

def run(hw, cin, cout, n_img, n_maps, mode):
    g = torch.Generator().manual_seed(hw * 91 + cin)
    x = torch.relu(torch.randn(n_img, cin, hw, hw, generator=g)) + 0.1
    w = torch.rand(cout, cin, 3, 3, generator=g) * 0.03
    if mode == "tap":      # one tap only
        w2 = torch.zeros_like(w); w2[:, :, TAP // 3, TAP % 3] = w[:, :, TAP // 3, TAP % 3]; w = w2
    r = torch.randn(n_maps, cout, hw, hw, generator=g)
    m2i = [i % n_img for i in range(n_maps)]
    got, _, _ = gpu_conv_rule(ops, x, w, r, m2i, f16x3=2)
    got32, _, _ = gpu_conv_rule(ops, x, w, r, m2i)
    d = (got - got32).abs() / got32.abs().amax(dim=(1, 2, 3), keepdim=True)
    return d.max().item()

TAP = 0
for shape in [(224, 64, 64, 1, 1), (14, 256, 32, 1, 2), (56, 256, 32, 1, 1), (14, 64, 96, 1, 2), (112, 64, 128, 1, 1)]:
    full = run(*shape, "full")
    taps = []
    for TAP in range(9):
        taps.append(run(*shape, "tap"))
    print("%-22s full %.1e | taps %s" % (shape, full, " ".join("%.0e" % t for t in taps)), flush=True)
