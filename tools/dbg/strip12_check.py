#!/usr/bin/env python3
"""(needs tools/experiments/r5_conv12_strip_kernel.patch applied and `make`: the strip kernel is not in the shipped library)
conv1_2's strip-persistent relevance kernel (csrc/conv_inst_strip12.hip, LRPX_STRIP12=1) against the generic 2-row-tile kernel
(LRPX_STRIP12=0): the switch is read once per process, so this script runs itself twice and compares the maps bit for bit; per-layer
times by the chain's own HIP events.   python tools/dbg/strip12_check.py [images] [maps]"""
import ctypes as C
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def child(images, maps, out):
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import lrp_amd  # noqa: F401
    from lrp_amd import ops, weights
    sd = weights.make_gridtd_state(seed=0, vocab_size=64)
    names = [k for k in sd if k.startswith("img_encoder.encoder.") and k.endswith(".weight")]
    vgg = ops.Vgg16([torch.from_numpy(sd[k]).cuda() for k in names], [torch.from_numpy(sd[k.replace(".weight", ".bias")]).cuda() for k in names])
    img = torch.from_numpy(weights.make_images(0, images)).cuda()
    vgg.forward(img)
    torch.manual_seed(0)
    r_feat = torch.randn(maps, 196, 512, device="cuda")
    m2i = (torch.arange(maps, device="cuda") * images // maps).to(torch.int32)
    res = vgg.relevance(r_feat, m2i)
    torch.cuda.synchronize()
    ms = (C.c_float * 17)()
    best = None
    for _ in range(5):
        vgg.relevance(r_feat, m2i, out=res, layer_ms=ms)
        torch.cuda.synchronize()
        cur = [ms[i] for i in range(17)]
        best = cur if best is None else [min(a, b) for a, b in zip(best, cur)]
    a = res.cpu().numpy()
    np.save(out, a)
    print(f"STRIP12={os.environ.get('LRPX_STRIP12')}: sha {hashlib.sha1(a.tobytes()).hexdigest()[:12]}  finite {np.isfinite(a).all()}  max|R| {np.abs(a).max():.4e}  "
          f"conv1_2 {best[1]:.3f} ms  first {best[0]:.3f}  conv2_1 {best[3]:.3f}  chain {sum(best):.3f} ms", flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
        sys.exit(0)
    images = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    maps = int(sys.argv[2]) if len(sys.argv) > 2 else 320
    import numpy as np
    outs = []
    for v in ("0", "1"):
        out = f"/tmp/strip12_{v}.npy"
        env = dict(os.environ, LRPX_STRIP12=v)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(images), str(maps), out], env=env)
        if r.returncode != 0:
            sys.exit(f"child LRPX_STRIP12={v} failed ({r.returncode})")
        outs.append(np.load(out))
    a, b = outs
    d = np.abs(a - b)
    print(f"images {images} maps {maps}: bit-identical {np.array_equal(a, b)}  max|diff| {d.max():.3e} of max|R| {np.abs(a).max():.3e}  "
          f"differing pixels {np.count_nonzero(d)} of {d.size}")
