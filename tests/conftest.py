import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


_TIE_STATS = []


def pytest_sessionfinish(session, exitstatus):
    """observed end-to-end deviations (assert_close_modulo_pool_ties) of this session -> gpurun_out/pool_tie_stats.json,
    only when LRPX_TIE_STATS=1 asks for it (a test session does not write into the tree otherwise: ADVICE r2)"""
    if _TIE_STATS and os.environ.get("LRPX_TIE_STATS") == "1":
        import json
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "pool_tie_stats.json"), "w") as f:
            json.dump(_TIE_STATS, f, indent=0)


def rel_err(a, b):
    """max|a-b| / max|b|  (the parity metric of SURVEY.md §8(d))"""
    import torch
    a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-300)).item()


def cosine(a, b):
    import torch
    a, b = torch.as_tensor(a).double().flatten(), torch.as_tensor(b).double().flatten()
    return (a @ b / (a.norm() * b.norm()).clamp_min(1e-300)).item()


# Worst observation per test of the LRP end-to-end comparisons (fraction of the pixels off by more than 1e-4 of max|R|, largest
# deviation, relative L2), round 4, default kernels (gpurun_out/pool_tie_stats.json of a LRPX_TIE_STATS=1 session).  A test fails at
# 10x ITS OWN observation - but never below what ONE more flipped pool winner adds (a near-tied window of the
# forward, DESIGN.md §3: about 0.1 % of the pixels, 7e-4, 8e-5 on the golden image; whether a box's clock order flips one more is chance).
OBSERVED = {
    "test_add_lrp_compute_lrp_accumulates_like_reference": (1.2e-03, 6.8e-04, 7.9e-05),
    "test_aoa_batch_vs_oracle": (0.0e+00, 1.9e-05, 1.6e-05),
    "test_aoa_vs_reference": (1.4e-03, 8.0e-04, 9.7e-05),
    "test_batch_of_images_vs_oracle": (2.8e-03, 4.9e-04, 1.8e-04),
    "test_explainer_class_drop_in": (1.2e-03, 6.8e-04, 7.9e-05),
    "test_gridtd_t20_rows_inside_b16_batch": (4.3e-04, 1.1e-04, 3.5e-05),
    "test_maps_end_to_end_vs_reference": (1.2e-03, 6.8e-04, 7.9e-05),
    "test_vgg_chain_multi_image_vs_oracle": (2.6e-04, 4.2e-04, 5.5e-05),
    "test_vgg_relevance_vs_reference_maps_end_to_end": (1.2e-03, 6.8e-04, 7.9e-05),
}
ONE_FLIP = (2e-3, 2e-3, 2.5e-4)           # floor of a bound: one more flipped winner than in the observed run, with margin
GLOBAL_DEFAULT = (1e-2, 8e-3, 1.5e-3)     # tests without an entry (and the ceiling of every derived bound)


def assert_close_modulo_pool_ties(got, want, frac=None, hard=None, l2=None, what="", cos=0.99999):
    """End-to-end comparison of pixel relevance maps whose FORWARD passes were computed by different conv
    implementations.  Relevance through MaxPool2d goes to the arg-max of each 2x2 window
    (LRPtools/lrp_modules.py:182-195); rounding-level differences of the forward flip the winner of a few
    near-tied windows out of 1.5 M, each moving one channel's relevance by one pixel — the reference itself moves
    by 1.6e-4 of max|R| between PyTorch's oneDNN and native CPU convs (2 flips); GPU vs oneDNN on the golden image:
    4 flips, relative L2 error 2.1e-4, 99th percentile 2.4e-5, 0.17 % of the pixels above 1e-4, max 1.8e-3
    (tests/e2e_stats.py).  So: cosine >= 0.99999, relative L2 error < `l2`, at most `frac` of the
    pixels off by more than 1e-4 of max|R|, none by more than `hard`.  The strict 1e-4 bound is asserted
    separately on identical activations.
    Bounds not given by the caller come from the calling test's own entry in OBSERVED: 10x its worst observation, floored at one
    flipped winner's worth and capped by the old suite-wide default; the assertion message states observation and bound.
    LRPX_TIE_STATS=1 writes the observations of a session to gpurun_out/pool_tie_stats.json."""
    import torch
    name = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0].split("::")[-1].split("[")[0]
    obs = OBSERVED.get(name)
    derived = tuple(min(max(10.0 * o, f), c) for o, f, c in zip(obs, ONE_FLIP, GLOBAL_DEFAULT)) if obs else GLOBAL_DEFAULT
    frac = derived[0] if frac is None else frac
    hard = derived[1] if hard is None else hard
    l2 = derived[2] if l2 is None else l2
    got, want = torch.as_tensor(got).double(), torch.as_tensor(want).double()
    scale = want.abs().max().clamp_min(1e-300)
    d = (got - want).abs() / scale
    o_frac, o_max = (d > 1e-4).double().mean().item(), d.max().item()
    o_l2, o_cos = ((got - want).norm() / want.norm().clamp_min(1e-300)).item(), cosine(got, want)
    _TIE_STATS.append({"test": os.environ.get("PYTEST_CURRENT_TEST", ""), "bounds": [frac, hard, l2], "what": str(what), "frac_gt_1e-4": o_frac, "max": o_max,
                       "rel_l2": o_l2, "cos": o_cos})
    msg = (f"{name} {what}: fraction of pixels off by > 1e-4 of max|R| {o_frac:.2e} (bound {frac:.1e}), largest deviation {o_max:.2e} "
           f"(bound {hard:.1e}), relative L2 {o_l2:.2e} (bound {l2:.1e}), cosine {o_cos:.7f} (bound {cos}); "
           f"this test's recorded worst observation: {obs if obs else 'none (suite default bounds)'}")
    assert o_cos > cos, msg
    assert o_l2 < l2, msg
    assert o_frac < frac, msg
    assert o_max < hard, msg


def forward_flips(vgg, sd, img):
    """Discrete decisions of `vgg`'s CURRENT forward trace (lrp_amd.ops.Vgg16 after forward(img)) that differ from the CPU oracle's
    oneDNN forward - what the reference itself runs: [(layer index, 'conv' | 'pool', map size of the layer's OUTPUT, flips)] with ReLU
    sign flips per conv layer and arg-max flips of live windows per pool.  The end-to-end deviation of a gradient-family map is made
    of exactly these: the tests choose their end-to-end bound from the count of the run at hand."""
    import torch
    import torch.nn.functional as F
    from oracle import lrp_oracle as O
    acts, _ = vgg.trace_views()
    sdt = O.state_to_torch(sd)
    feats, _, saved = O.vgg_forward(sdt, img)
    saved = saved + [feats]
    out = []
    for l, (kind, idx, cin, cout) in enumerate(O.vgg_layers()):
        want = saved[l + 1]
        n, c, hw = want.shape[0], want.shape[1], want.shape[2]
        got = acts[l + 1][:n].cpu().reshape(n, hw, hw, -1)[..., :c].permute(0, 3, 1, 2)
        if kind == "conv":
            out.append((l, "conv", hw, int(((got > 0) != (want > 0)).sum())))
        else:
            x_gpu = acts[l][:n].cpu().reshape(n, 2 * hw, 2 * hw, -1)[..., :c].permute(0, 3, 1, 2)
            _, ig = F.max_pool2d(x_gpu, 2, 2, return_indices=True)
            _, iw = F.max_pool2d(saved[l], 2, 2, return_indices=True)
            out.append((l, "pool", hw, int(((ig != iw) & (want > 0)).sum())))
    return out


def gradient_e2e_bounds(flips, what=""):
    """End-to-end bounds of a plain-gradient / guided map against the reference's golden, chosen from the flips of THIS run
    (VERDICT r3 item 4): without a flipped decision at the deep layers (outputs of 28 x 28 pixels and below: conv4_x, conv5_x, pools 3
    and 4 - one such ReLU path spans a 100-pixel receptive field and moves a fifth of the pixels) the strict set applies, otherwise the
    set a couple of deep flips produce (gpurun_out/r3d: 12 % / 16 % of the pixels, relative L2 7e-3 with two)."""
    deep = sum(f for _, _, hw, f in flips if hw <= 28)
    total = sum(f for _, _, _, f in flips)
    print(f"{what}: {total} flipped decisions of the forward against the oracle's, {deep} of them at the deep layers (<= 28 x 28): "
          + ", ".join(f"layer {l} ({k} {hw}): {f}" for l, k, hw, f in flips if f))
    if deep == 0:
        return dict(frac=0.02, l2=6e-3, cos=0.99997, hard=0.1)
    return dict(frac=0.3, l2=3e-2, cos=0.9995, hard=0.15)


import pytest as _pytest


@_pytest.fixture(autouse=True)
def _lrpx_test_diag(request):
    """LRPX_TEST_DIAG=1: one line per test (open file descriptors, threads, reserved / allocated device memory) appended to
    gpurun_out/test_diag.txt BEFORE the test runs - what a process that dies without a message leaves behind."""
    if os.environ.get("LRPX_TEST_DIAG") == "1":
        import resource
        import threading
        line = f"{request.node.nodeid} fds={len(os.listdir('/proc/self/fd'))}/{resource.getrlimit(resource.RLIMIT_NOFILE)[0]} threads={threading.active_count()}"
        try:
            import torch
            if torch.cuda.is_available():
                free, total = torch.cuda.mem_get_info()
                line += f" reserved={torch.cuda.memory_reserved() >> 20}MiB allocated={torch.cuda.memory_allocated() >> 20}MiB free={free >> 20}MiB of {total >> 20}MiB"
            with open("/proc/self/status") as f:
                st = f.read()
            line += " " + " ".join(l.replace("\t", "") for l in st.splitlines() if l.startswith(("VmRSS", "Threads")))
        except Exception as e:  # noqa: BLE001
            line += f" ({e})"
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "test_diag.txt"), "a") as f:
            f.write(line + "\n")
    yield
