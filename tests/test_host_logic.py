"""Host-side logic that needs no GPU: seeded generator, hook-API validation, no-CPU-fallback guarantees."""
import numpy as np
import pytest
import torch
import torch.nn as nn

import lrp_amd  # noqa: F401
from lrp_amd import weights
from lrp_amd.LRPtools import lrp_modules, lrp_wrapper, utils


def test_generator_is_deterministic_and_matches_reference_names():
    a = weights.make_gridtd_state(seed=5, vocab_size=40)
    b = weights.make_gridtd_state(seed=5, vocab_size=40)
    assert list(a) == list(b) and all(np.array_equal(a[k], b[k]) for k in a)
    assert a["img_encoder.encoder.28.weight"].shape == (512, 512, 3, 3)      # last conv of features[0:-1]
    assert a["AdaLSTM.lstm_cell.weight_ih"].shape == (2048, 1536) and a["fc.weight"].shape == (40, 512)
    assert a["embedding.weight"].dtype == np.float32
    cap = weights.make_captions(1, 3, 5, 40)
    assert cap.shape == (3, 6) and (cap[:, 0] == 38).all() and cap[:, 1:].min() >= 1 and cap[:, 1:].max() <= 36
    wm = weights.make_word_map(40)
    assert wm["<pad>"] == 0 and wm["<unk>"] == 37 and wm["<start>"] == 38 and wm["<end>"] == 39


def test_constants_match_reference():
    assert (utils.EPSILON, utils.Z_EPSILON, utils.RELEVANCE_RECT) == (0.01, 1e-7, -1e-6)
    assert lrp_wrapper.SequentialPresetA().lrp_params == {"alpha": 1., "beta": 0., "ignore_bias": True}


def test_unknown_leaf_raises_value_error_like_reference():
    with pytest.raises(ValueError, match="not known"):
        lrp_modules.get_lrp_module(nn.Sigmoid())
    with pytest.raises(ValueError):
        lrp_wrapper.add_lrp(nn.Sequential(nn.Conv2d(3, 8, 3, padding=1), nn.Sigmoid()))
    with pytest.raises(NotImplementedError):           # lrp_modules.py:152
        lrp_modules.Conv2d().propagate_relevance(nn.Conv2d(3, 8, 3, padding=1), None, (torch.zeros(1),), "epsilon", {})


def test_dispatch_table_matches_reference():
    # LRPtools/lrp_modules.py:321-341: every leaf type of the reference's table, incl. the reference's own resnet.Add /
    # resnet.Flatten classes by name
    table = {nn.Linear(4, 4): "Linear", nn.ReLU(): "ReLU", nn.Conv2d(3, 8, 3): "Conv2d", nn.MaxPool2d(2): "Pool2d",
             nn.AvgPool2d(2): "Pool2d",
             nn.BatchNorm2d(4): "BatchNorm2d", nn.BatchNorm1d(4): "BatchNorm1d", nn.Dropout(): "Dropout",
             nn.Dropout2d(): "Dropout", lrp_modules.resFlatten(): "Flatten", lrp_modules.resAdd(): "Add"}
    for mod, name in table.items():
        assert type(lrp_modules.get_lrp_module(mod)).__name__ == name

    class Add(nn.Module):        # a module class named like the reference's, living in a module called `resnet`
        pass
    Add.__module__ = "models.resnet"
    assert type(lrp_modules.get_lrp_module(Add())).__name__ == "Add"
    with pytest.raises(ValueError, match="not known"):
        lrp_modules.get_lrp_module(nn.AdaptiveAvgPool2d(1))


def test_vgg16_structure_matching():
    def vgg(cfg):
        mods, cin = [], 3
        for v in cfg:
            if v == 'M':
                mods.append(nn.MaxPool2d(2, 2))
            else:
                mods += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
                cin = v
        return nn.Sequential(*mods)
    leaves = lambda m: [x for x in m.modules() if len(list(x.children())) == 0]
    assert lrp_wrapper._match_vgg16(leaves(vgg(lrp_wrapper.VGG16_FEATURES)))
    assert not lrp_wrapper._match_vgg16(leaves(vgg(lrp_wrapper.VGG16_FEATURES + ['M'])))
    assert not lrp_wrapper._match_vgg16(leaves(vgg([64, 'M', 128])))


def test_no_cpu_fallback():
    from lrp_amd import _lib, ops
    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    with pytest.raises(_lib.LrpxError):
        from lrp_amd.explainers.gridtd import GridTDEngine
        GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=16))
    with pytest.raises(ValueError):
        ops.conv_mfma(torch.zeros(4), torch.zeros(4), 1, 14, 32, 32, 9, 1)
    cpu_vgg = nn.Sequential(*[m for v in [64] for m in (nn.Conv2d(3, 64, 3, padding=1), nn.ReLU())])
    with pytest.raises(_lib.LrpxError, match="no CPU path"):      # generic driver: refuses a model on the CPU
        lrp_wrapper.add_lrp(cpu_vgg)
    pools = nn.Sequential(nn.MaxPool2d(2, 2), nn.ReLU())          # nothing to tell the device from: refused at the call
    lrp_wrapper.add_lrp(pools)
    with pytest.raises(_lib.LrpxError, match="no CPU path"):
        pools.compute_lrp(torch.zeros(1, 2, 4, 4), target=torch.ones(1, 2, 2, 2))


def test_device_pointer_helper_refuses_host_tensors():
    """`_lib.ptr` / `ptr_at` hand addresses to HIP kernels: a host tensor's address there is a GPU page fault that comes and goes with
    what the runtime has mapped (round 5).  They refuse it - on any machine, before the library is even loaded."""
    from lrp_amd import _lib
    assert _lib.ptr(None) is None
    for f in (_lib.ptr, _lib.ptr_at):
        with pytest.raises(TypeError, match="no CPU path"):
            f(torch.zeros(4))


def test_generic_add_lrp_hooks_every_leaf_once():
    """lrp_wrapper.py:37-59 hooks every leaf of ANY model; a second add_lrp must not stack hooks (the reference does)"""
    net = nn.Sequential(nn.MaxPool2d(2, 2), nn.ReLU(), nn.Dropout())
    lrp_wrapper.add_lrp(net)
    lrp_wrapper.add_lrp(net)
    assert all(len(m._forward_hooks) == 1 for m in net) and callable(net.compute_lrp)
    x = torch.rand(1, 2, 4, 4)
    net.eval()(x)
    assert net[0].input[0] is x and len(net._lrpx_tape) == 3      # save_input_hook (:24-25) + the call order
    assert [lrp_wrapper._rule_name(m) for m in (nn.Linear(2, 2), nn.BatchNorm2d(2), nn.ReLU(), nn.Conv2d(2, 2, 3), nn.MaxPool2d(2))] == \
        ["epsilon", "epsilon", "identity", "alpha_beta", "alpha_beta"]


def test_bench_host_cores_respects_cgroup():
    import bench
    n = bench.host_cores()
    assert 1 <= n <= 16


def test_bench_arguments_of_every_baseline_config():
    """bench.py: the defaults of every BASELINE config line and of the `configs` block the default run appends"""
    import bench
    a = bench.parse([])
    assert (a.config, a.batch, a.words, a.vocab, a.pipeline, a.explainer, a.gpus) == (2, 16, 20, 9586, 3, "lrp", 1)
    assert (a.steps, a.warmup) == (20, 4) and not a.no_configs
    for cfg, want in ((3, (64, 11027, 2, "lrp")), (4, (32, 9586, 2, "lrp+guided")), (5, (32, 11027, 3, "lrp"))):
        b = bench.parse(["--config", str(cfg)])
        assert (b.batch, b.vocab, b.pipeline, b.explainer) == want
    b64 = bench.parse(["--config", "2", "--batch", "64", "--pipeline", "2"])
    assert (b64.batch, b64.pipeline, b64.vocab) == (64, 2, 9586)
    assert bench.parse(["--config", "3", "--all-heads"]).all_heads


def test_zeros_arena_views_are_zero_disjoint_and_aligned():
    """ops.zeros_arena (the decoder traces: one allocation, one fill): every view has its shape, is contiguous, zero, starts on a
    256-byte boundary and shares no element with another view"""
    from lrp_amd import ops
    shapes = {"a": (3, 5, 7), "b": (2, 64), "c": (1,), "d": (4, 3, 2, 5)}
    v = ops.zeros_arena("cpu", shapes)
    base = min(t.data_ptr() for t in v.values())
    spans = []
    for k, shp in shapes.items():
        t = v[k]
        assert tuple(t.shape) == shp and t.is_contiguous() and t.dtype == torch.float32 and not t.any()
        assert (t.data_ptr() - base) % 256 == 0
        spans.append((t.data_ptr(), t.data_ptr() + 4 * t.numel()))
    spans.sort()
    assert all(spans[i][1] <= spans[i + 1][0] for i in range(len(spans) - 1))
    v["a"].fill_(1.0)
    assert not v["b"].any() and not v["c"].any() and not v["d"].any()


def test_ragged_rows_of_unequal_caption_lengths():
    """explainers/ragged.py: the row tables of a batch whose captions differ in length (models/gridTDmodel.py:1147-1153 explains
    `caption_length` words per image): valid rows image-major, their images, the first compact row of every image; cached per
    pattern; bad patterns refused"""
    from lrp_amd.explainers.ragged import RaggedRows, ragged
    r = RaggedRows([3, 0, 1, 4], 4, 4, "cpu")
    assert r.n == 8 and not r.full
    assert r.rows.tolist() == [0, 1, 2, 8, 12, 13, 14, 15] and r.row2img.tolist() == [0, 0, 0, 2, 3, 3, 3, 3]
    assert r.offs.tolist() == [0, 3, 3, 4] and r.lens.tolist() == [3, 0, 1, 4]
    assert r.rows.dtype == torch.int32 and r.offs.dtype == torch.int32
    assert RaggedRows([2, 2], 2, 2, "cpu").full and RaggedRows([0, 0], 2, 5, "cpu").n == 0
    assert ragged(None, 2, 2, "cpu") is None
    a = ragged([1, 2], 2, 2, "cpu")
    assert ragged(torch.tensor([1, 2]), 2, 2, "cpu") is a and ragged(a, 2, 2, "cpu") is a
    with pytest.raises(ValueError):
        RaggedRows([1, 2, 3], 2, 4, "cpu")
    with pytest.raises(ValueError):
        RaggedRows([1, 5], 2, 4, "cpu")
    with pytest.raises(ValueError):
        RaggedRows([-1, 2], 2, 4, "cpu")


def test_engine_cache_keys(tmp_path):
    """explainers/engine_cache.py: one engine per weight set - a checkpoint path keys by file identity, torch tensors by (pointer,
    version): an in-place update or a rewritten file gives a new key; numpy arrays are never cached"""
    from lrp_amd.explainers import engine_cache as ec
    ec.clear()
    sd = {"a": torch.zeros(3), "b": torch.ones(2, 2)}
    k1 = ec.fingerprint("gridtd", sd)
    assert k1 is not None and k1 == ec.fingerprint("gridtd", sd) and k1 != ec.fingerprint("aoa", sd, (8,))
    sd["a"].add_(1.0)
    assert ec.fingerprint("gridtd", sd) != k1
    lin = torch.nn.Linear(2, 2)
    assert ec.fingerprint("gridtd", lin) == ec.fingerprint("gridtd", lin)
    assert ec.fingerprint("gridtd", {"a": np.zeros(3)}) is None and ec.fingerprint("gridtd", {}) is None
    f = tmp_path / "ckpt.pth"
    f.write_bytes(b"x" * 10)
    kp = ec.fingerprint("gridtd", str(f))
    assert kp is not None and ec.fingerprint("gridtd", str(f)) == kp
    f.write_bytes(b"y" * 11)
    assert ec.fingerprint("gridtd", str(f)) != kp and ec.fingerprint("gridtd", str(tmp_path / "missing.pth")) is None
    built = []
    mk = lambda: built.append(1) or object()
    a = ec.get(k1, mk)
    assert ec.get(k1, mk) is a and len(built) == 1 and ec.get(None, mk) is not a and len(built) == 2
    for i in range(ec.MAX_ENGINES + 1):
        ec.get(("k", i), mk)
    assert ec.get(k1, mk) is not a          # evicted: the cache holds MAX_ENGINES engines
    ec.clear()


def test_engine_cache_two_checkpoints_loaded_one_after_the_other(tmp_path):
    """two DIFFERENT checkpoints `torch.load`ed in sequence, the first state dict freed before the second is read - the
    allocator hands the second one the same addresses and `_version` is 0 after every load.  The entry holds its source tensors
    (addresses cannot be reused while it lives) and the key carries a content digest: two different engines come back."""
    import gc
    from lrp_amd.explainers import engine_cache as ec
    ec.clear()
    g = torch.Generator().manual_seed(0)
    for i in range(2):
        torch.save({"state_dict": {"w": torch.randn(256, 256, generator=g), "b": torch.randn(256, generator=g)}}, tmp_path / f"c{i}.pth")
    engines, keys = [], []
    for hold in (False, True):          # the digest alone, then as the explainers call it (entry holds the tensors)
        ec.clear()
        engines.clear(); keys.clear()
        for i in range(2):
            sd = torch.load(tmp_path / f"c{i}.pth")["state_dict"]
            k = ec.fingerprint("gridtd", sd)
            keys.append(k)
            engines.append(ec.get(k, lambda i=i: ("engine", i), hold=ec.source_tensors(sd) if hold else None))
            del sd
            gc.collect()
        assert keys[0] != keys[1] and engines[0] != engines[1], hold
    # the same state dict again: a hit
    sd = torch.load(tmp_path / "c1.pth")["state_dict"]
    built = []
    e1 = ec.get(ec.fingerprint("gridtd", sd), lambda: built.append(1) or "x", hold=ec.source_tensors(sd))
    assert ec.get(ec.fingerprint("gridtd", sd), lambda: built.append(1) or "y", hold=ec.source_tensors(sd)) is e1 and len(built) == 1
    # an edit through .data bumps no version counter: the digest sees it (element 0 is always sampled)
    sd["w"].data[0, 0] += 1.0
    assert ec.get(ec.fingerprint("gridtd", sd), lambda: built.append(1) or "z") == "z"
    ec.clear()


def test_bench_pmc_traffic_parser_and_precedence(tmp_path):
    """bench.py's roofline.traffic: the counters of its own `rocprofv3 --pmc` child passes (one *_counter_collection.csv per pass, a row
    per dispatch and counter, FETCH_SIZE / WRITE_SIZE in kilobytes) -> bytes per launch = (2 x FETCH + WRITE) x 1024 / launches, scaled to
    the maps of the line; a live measurement takes precedence over the summary committed under profiles/; kernels it did not see fall back."""
    import bench
    name = "void lrpx::conv_f16x3_kernel<224, 2, 2, false, 5, true, true>(lrpx::ConvArgs, int, int)"
    for sub, rows in (("fetch/runA/1", [(name, "FETCH_SIZE", 1000.0), (name, "GRBM_GUI_ACTIVE", 5.0), (name, "FETCH_SIZE", 3000.0),
                                        ("lrpx::amax_maps_kernel<7>(float const*)", "FETCH_SIZE", 7.0)]),
                      ("write/runB/2", [(name, "WRITE_SIZE", 4000.0), (name, "TCC_HIT_sum", 30.0), (name, "TCC_MISS_sum", 10.0),
                                        (name, "WRITE_SIZE", 4000.0)])):
        d = tmp_path / sub
        d.mkdir(parents=True)
        with open(d / "77_counter_collection.csv", "w") as f:
            f.write("Correlation_Id,Dispatch_Id,Kernel_Name,Counter_Name,Counter_Value\n")
            for i, (k, c, v) in enumerate(rows):
                f.write(f'{i},{i},"{k}",{c},{v}\n')
    doc = bench.pmc_traffic_doc(str(tmp_path), 320)
    short = "conv_f16x3_kernel<224, 2, 2, false, 5, true, true>"
    assert list(doc["kernels"]) == [short]                       # only conv / first-layer kernels with both byte counters
    k = doc["kernels"][short]
    assert k == {"launches": 2, "fetch_bytes": 4000.0 * 1024, "write_bytes": 8000.0 * 1024, "l2_hit": 0.75}
    saved = dict(bench.LIVE_TRAFFIC)
    try:
        bench.LIVE_TRAFFIC.update(doc=doc, note="live")
        t, src = bench.read_traffic(short, 320)
        assert t == round((2 * 4000.0 + 8000.0) * 1024 / 2) and src == "live"
        t2, _ = bench.read_traffic(short, 640)                   # per map, scaled to the line's maps
        assert t2 == 2 * t
        t3, src3 = bench.read_traffic("conv_f16x3_kernel<14, 1, 8, true, 5, false, true>", 320)
        assert src3 != "live" and (t3 is None or t3 > 0)         # not measured live: the committed summary (if it holds the kernel)
    finally:
        bench.LIVE_TRAFFIC.clear()
        bench.LIVE_TRAFFIC.update(saved)


def test_decoder_arithmetic_rule_and_its_overrides(monkeypatch):
    """lrp_amd.ops.decoder_f16: the decoders' GEMMs take the fp16 split products (22 operand bits) only with the opt-in conv modes 2 / 3 -
    an engine's own mode, else the process default - so that nothing on the default path is narrower than the reference's fp32
    (models/gridTDmodel.py:744-765 `lrp_linear_eps` is fp32 matmul); LRPX_DECODER_F16 = 0 / 1 overrides either way.  Host logic only."""
    from lrp_amd import _lib, ops
    lib = _lib.load()
    monkeypatch.delenv("LRPX_DECODER_F16", raising=False)
    keep = lib.lrpx_set_conv_mode(-1)
    try:
        for mode in (0, 1, 2, 3):
            assert ops.decoder_f16(mode) is (mode >= 2)
            lib.lrpx_set_conv_mode(mode)
            assert ops.decoder_f16() is (mode >= 2)          # (None: the process default)
        monkeypatch.setenv("LRPX_DECODER_F16", "1")
        assert ops.decoder_f16(1) is True and ops.decoder_f16(0) is True
        monkeypatch.setenv("LRPX_DECODER_F16", "0")
        assert ops.decoder_f16(3) is False
        monkeypatch.setenv("LRPX_DECODER_F16", "")
        assert ops.decoder_f16(3) is True and ops.decoder_f16(1) is False
    finally:
        lib.lrpx_set_conv_mode(keep)
