#!/usr/bin/env python3
"""BUILD CONTAINER ONLY (imports /root/reference read-only through the harness of tests/golden/make_golden.py).

BASELINE.md §3 / SURVEY §8(d): the CPU baseline that travels to the GPU box is oracle/ref_equiv.py, a replay of the reference's
op sequence.  This script times BOTH on the same image, the same weights, the same 20-word caption and the same thread count -
the reference's own `ExplainGridTDAttention.explain_caption` and `ref_equiv.explain_words` - checks that they return the same
maps, and writes the two wall times to profiles/<tag>_ref_equiv_vs_reference.json (claim: within +-10 %).

    python tools/ref_timing.py [--threads 8] [--words 20] [--tag r04]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--words", type=int, default=20)
    ap.add_argument("--tag", default="r04")
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    import make_golden as G
    G.install_stubs()
    weights = G.load_pkg()
    G.ref_vgg_patch()
    import models.gridTDmodel as gtd
    from oracle import lrp_oracle as O, ref_equiv as RE
    T, V = a.words, 9586
    sd = weights.make_gridtd_state(seed=0, vocab_size=V)
    img = weights.make_images(0, 1)
    cap = weights.make_captions(1, 1, T, V)[0]
    # ---- the reference itself (trace + every word, the whole explain_caption)
    model = gtd.GridTDModel(512, 512, V, 'vgg16')
    model.load_state_dict(G.to_torch_sd(sd))
    with tempfile.TemporaryDirectory() as tmp:
        ex = gtd.ExplainGridTDAttention(G.make_args(tmp), weights.make_word_map(V), model=model)
        G._patch_explainer(ex, img, cap)
        t0 = time.time()
        maps_ref, rws_ref = ex.explain_caption("synthetic.jpg")
        t_ref = time.time() - t0
    print(f"reference explain_caption: {t_ref:.1f} s for {T} words on {a.threads} threads = {T / t_ref:.3f} maps/s", flush=True)
    # ---- oracle/ref_equiv.py on the same inputs
    sdt = O.state_to_torch(sd)
    t0 = time.time()
    maps, rws, t_trace, t_words = RE.explain_words(sdt, torch.from_numpy(img), cap, list(range(T)))
    t_eq = time.time() - t0
    print(f"ref_equiv.explain_words : {t_eq:.1f} s (trace {t_trace:.1f} + words {t_words:.1f}) = {T / t_eq:.3f} maps/s", flush=True)
    # same results: the reference returns running sums over the words (sample.grad is never cleared), ref_equiv too
    err = max(float((maps[t] - maps_ref[t]).abs().max() / maps_ref[t].abs().max()) for t in range(T))
    err_w = max(float((rws[t] - rws_ref[t].detach()).abs().max()) for t in range(T))
    out = {"threads": a.threads, "words": T, "vocab": V, "reference_s": round(t_ref, 2), "ref_equiv_s": round(t_eq, 2),
           "reference_maps_per_s": round(T / t_ref, 4), "ref_equiv_maps_per_s": round(T / t_eq, 4),
           "ratio_ref_equiv_over_reference_time": round(t_eq / t_ref, 4),
           "max_rel_map_difference": err, "max_r_words_difference": err_w,
           "host": os.uname().nodename, "cpu_count": os.cpu_count(),
           "what": "models/gridTDmodel.py ExplainGridTDAttention.explain_caption (imported from /root/reference) against "
                   "oracle/ref_equiv.py explain_words, same image / weights / caption / thread count, build container"}
    path = os.path.join(ROOT, "profiles", f"{a.tag}_ref_equiv_vs_reference.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out))
    assert err < 2e-3 and err_w < 1e-4, (err, err_w)       # (pool-winner ties between the two conv back-ends aside: DESIGN §3)


if __name__ == "__main__":
    main()
