"""debug: pooled-input f16+fp6 kernel with identity centre-tap weights: out must be the unpooled S (winner position keeps the value)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
import lrp_amd  # noqa
from lrp_amd import ops, _lib

def run(hw, c, n_maps, f8):
    dev = "cuda"
    ho = hw // 2
    g = torch.Generator().manual_seed(hw)
    s_lo = (torch.randint(1, 9, (n_maps, ho * ho, c), generator=g).float()).to(dev)
    am = torch.randint(0, 4, (1, ho * ho, c), generator=g, dtype=torch.uint8).to(dev)
    w = torch.zeros(c, c, 3, 3); w[torch.arange(c), torch.arange(c), 1, 1] = 1.0
    wb = (ops.pack_weights_f16f8 if f8 else ops.pack_weights_f16x2)(w.to(dev), c, c, _lib.PACK_BWD_POS)
    xg = torch.ones(1, hw * hw, c, device=dev)
    out = torch.empty(n_maps, hw * hw, c, device=dev)
    m2i = torch.zeros(n_maps, dtype=torch.int32, device=dev)
    ops.conv_mfma(s_lo, wb, n_maps, hw, c, c, 9, _lib.EPI_REL_MUL, oc_split=c, x=xg, map2img=m2i, out0=out, f16x3=2 if f8 else 1,
                  in_amax=ops.amax_maps(s_lo, n_maps), pool_am=am)
    torch.cuda.synchronize()
    want = torch.zeros(n_maps, hw, hw, c, device=dev)
    sl = s_lo.view(n_maps, ho, ho, c); a = am.view(1, ho, ho, c)
    for pos in range(4):
        want[:, pos // 2::2, pos % 2::2, :] = sl * (a == pos)
    got = out.view(n_maps, hw, hw, c)
    d = (got - want).abs()
    print(f"hw {hw} c {c} f8 {f8}: max diff {d.max().item():.3g}; wrong entries {(d > 1e-3).sum().item()} of {d.numel()}")
    if d.max() > 1e-3:
        idx = (d > 1e-3).nonzero()[:12]
        for i in idx.tolist():
            n, y, x, ch = i
            print("   map %d y %d x %d ch %d: got %g want %g (s_lo %g, winner %d, pos %d)" % (n, y, x, ch, got[n, y, x, ch].item(), want[n, y, x, ch].item(),
                  sl[n, y // 2, x // 2, ch].item(), a[0, y // 2, x // 2, ch].item(), (y % 2) * 2 + x % 2))
        bych = (d > 1e-3).sum(dim=(0, 1, 2))
        print("   wrong by channel:", bych.tolist())
        bypos = [(d[:, p // 2::2, p % 2::2, :] > 1e-3).sum().item() for p in range(4)]
        print("   wrong by position:", bypos)

for hw in (56, 28):
    run(hw, 32, 1, False)
    run(hw, 32, 1, True)
