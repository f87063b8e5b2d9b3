import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lrp_amd  # noqa
from lrp_amd import ops, _lib
rows, N, n_src = 640, 1536, 32
for K in (64, 128, 256, 384, 512):
    a = torch.randn(rows, K, device="cuda"); w = torch.randn(K, N, device="cuda") * 0.05
    x = torch.randn(n_src, N, device="cuda"); u = torch.randn(rows, N, device="cuda")
    src = (torch.arange(rows, device="cuda") * n_src // rows).to(torch.int32)
    out = torch.empty(rows, N, device="cuda")
    wh = ops.pack_weights_f16x2(w, K, N, _lib.PACK_BWD_PLAIN, taps=1)
    def run():
        ops.conv_mfma(a, wh, rows, 0, K, N, 1, _lib.EPI_REL, pix_per_map=1, oc_split=N, x=x, u=u, map2img=src, out0=out, f16x3=1)
    for _ in range(10): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): run()
    e1.record(); e1.synchronize()
    print(f"K={K}: {e0.elapsed_time(e1) * 1000 / 200:.2f} us")
# an empty kernel's launch-to-launch time on this stream for comparison
z = torch.zeros(1, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200): z.add_(1.0)
e1.record(); e1.synchronize()
print(f"tiny torch kernel back to back: {e0.elapsed_time(e1) * 1000 / 200:.2f} us")
