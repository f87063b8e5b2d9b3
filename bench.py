#!/usr/bin/env python3
"""Headline benchmark: LRP relevance maps/sec (VGG16 + gridTD, 224x224, 20-token caption).

    python bench.py --gpus N --steps K --warmup W        (N > 1 without RANK/WORLD_SIZE: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch per GPU.  Default = BASELINE.json configs[1]: 16 images x 20 words
= 320 relevance maps, V=9586: VGG16 forward trace (+Z+), decoder trace incl. the (T,V) predictions the reference's
explainer keeps (evaluation.py:109 reads them), decoder relevance, VGG16 relevance, running sums of the maps (what
`explain_caption` returns: lrp_wrapper.py:64-82).  Inputs (images, token ids) are resident in HBM before the timed
region; weights are random-init from the seeded generator (no network for checkpoints), data synthetic.  Images are
independent, so ranks shard the batch with no data-path collective (weak scaling: every rank runs the same per-GPU
batch); --gather adds the RCCL gather of the maps to rank 0 that north_star mentions.

Other BASELINE configs as optional lines (the headline stays config 2):
    --config 3                 AoA 8-head decoder, B=64, V=11027, head 0 (1280 maps / step)
    --config 4                 gridTD, B=32 per GPU, LRP + Guided-Backprop side by side on the same encoder trace
                               (= --explainer lrp+guided --batch 32; 2 x 640 maps / step)
    --config 5                 AoA bottom-up, 36x2048 region features, B=32 per GPU, relevance back to the features

Arithmetic of the line (round 6): `value`, `dtype`, `ms_per_step` and `roofline` are measured in CONV MODE 1 - every fp32 operand of the VGG16
contractions split exactly into three bf16 parts (24 significand bits, fp32's exponent range), six matrix-core products, fp32 accumulate: no
narrower than the reference's fp32 convolutions (LRPtools/lrp_modules.py:124-150, utils.py:21-31); it is also the library's process default.
The opt-in speed modes are reported beside it as top-level scalars: `value_f16x3`, `value_f16f6` (and `value_fp32_mfma`), each with
`dev_chain_<mode>` / `dev_step_<mode>` = its worst-map deviation from the fp32-MFMA chain measured in this run on the step's own 320 maps.

Rank 0 prints ONE JSON line.  At N=1 it also carries
  roofline     : MFMA roofline of the dominant kernel = the relevance conv kernel NAME with the largest total time over the
                 12 conv launches of one pass (30.69 GFLOP per map), picked from this run's own per-launch HIP-event times
                 (on the launch stream); `chain_frac`: the same fraction over all 13 layers; `per_layer`: every layer's;
                 `traffic` = HBM bytes per launch from the PMC counters: measured in this run by two `rocprofv3 --pmc` child passes
                 over the chain alone before the timed region (`traffic_source` says so; --no-live-traffic or any failure: the PMC
                 summary committed under profiles/, separate --pmc passes of tools/pmc_passes.sh);
                 `modes`: the same step and chain in every matrix-core mode (0 fp32 MFMA ... 3 fp16+fp6), same process, with `dev_vs_fp32`
  sustained    : the same step repeated for >= --sustain seconds (power-limited clocks show here, not in 20 steps)
  median_ms    : median interval between step completions (HIP events) inside the timed region
  cpu_baseline : the reference-equivalent CPU mode (oracle/ref_equiv.py, kind "port") on a bounded sample.
  configs      : the other BASELINE configs measured in the SAME invocation after the headline (driver-witnessed):
                 "3" AoA B=64 head 0 (+ `all_heads`: the 8 heads of every word, 10 240 maps per step), "4" LRP + Guided-
                 Backprop B=32, "5" bottom-up B=32, and "b64": the headline model at the 64-image batch north_star's
                 target names; each with value, ms_per_step and the dominant kernel's roofline fraction (--no-configs skips)
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_MAP = 30.69          # one transposed conv per VGG16 layer = 15.35 GMAC (algorithmic, fp32)
PEAK_FP32_MFMA_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_16BIT_MFMA_TF = 2500.0    # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (no sparsity)
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E
# VGG16 cfg 'D' without the last pool: conv layer index of the fused chain -> (name, map size, cin, cout).  The relevance step of
# layer l is one transposed 3x3 conv: 2*9*cin*cout*hw*hw algorithmic flop per map (the 13 layers sum to 30.69 GFLOP).
# The matrix cores execute PRODUCTS[mode] 16-bit MFMA products per fp32 product (operand splits, csrc/conv_f16x3.h /
# conv_bf16x6.h), so `achieved` counts executed MFMA flop against the 16-bit dense peak; `algorithmic_tflops` is the
# fp32-equivalent rate.
VGG_CONV = {0: ("conv1_1", 224, 3, 64), 1: ("conv1_2", 224, 64, 64), 3: ("conv2_1", 112, 64, 128), 4: ("conv2_2", 112, 128, 128),
            6: ("conv3_1", 56, 128, 256), 7: ("conv3_2", 56, 256, 256), 8: ("conv3_3", 56, 256, 256),
            10: ("conv4_1", 28, 256, 512), 11: ("conv4_2", 28, 512, 512), 12: ("conv4_3", 28, 512, 512),
            14: ("conv5_1", 14, 512, 512), 15: ("conv5_2", 14, 512, 512), 16: ("conv5_3", 14, 512, 512)}
POOL_ABOVE = {1, 4, 8, 12}        # the relevance of these layers arrives at the resolution of the pool above them
EPI_ID = {"EPI_FWD_DUAL": 0, "EPI_REL": 1, "EPI_FIRST": 2, "EPI_PLAIN": 3, "EPI_GUIDED": 4, "EPI_REL_MUL": 5}


def layer_flop_per_map(l):
    _, hw, cin, cout = VGG_CONV[l]
    return 2.0 * 9 * cin * cout * hw * hw


def rel_launcher(l, mode):
    """the launch function csrc/lrpx_vgg.hip:conv_dispatch picks for the relevance step of conv layer l (default switches)"""
    _, hw, cin, _ = VGG_CONV[l]
    n_oc, pooled = cin, l in POOL_ABOVE
    if l == 0:
        return "first_layer_mfma_kernel" if mode >= 2 else "first_layer_rel_kernel"
    if mode == 3:
        if pooled:
            return {224: "launch_h8_224_pool", 112: "launch_h8_112_pool", 56: "launch_h8_56w_pool", 28: "launch_h8_28w_pool"}[hw]
        if n_oc >= 256 and hw in (56, 28, 14):
            return f"launch_h8_{hw}w_rel"
        return "launch_h8_112n_rel" if (hw == 112 and n_oc <= 64) else f"launch_h8_{hw}_rel"
    if mode == 2:
        if pooled:
            return f"launch_h3_{hw}_pool"
        return "launch_h3_112n_rel" if (hw == 112 and n_oc <= 64) else f"launch_h3_{hw}_rel"
    if mode == 1:      # conv_f16x3.h with B6 (exact bf16 splits): fused multiplicand, pooled-input staging
        if pooled:
            return f"launch_b6_{hw}_pool"
        if hw == 14:      # K split in two as a property of the layer (csrc/lrpx_vgg.hip, b6_rel_ksplit14): PLAIN partials + rel_mul_finish
            return "launch_b6_14_plain"
        return "launch_b6_112n_rel" if (hw == 112 and n_oc <= 64) else f"launch_b6_{hw}_rel"
    return {224: "launch_conv_224_8_2_2_9_rel", 112: "launch_conv_112_8_2_2_9_rel" if n_oc <= 64 else "launch_conv_112_8_1_4_9_rel",
            56: "launch_conv_56_16_1_4_9_rel", 28: "launch_conv_28_16_1_4_9_rel", 14: "launch_conv_14_16_1_4_9_rel"}[hw]


_INST = {}


def kernel_name(launcher):
    """kernel name as rocprofv3 prints it, read from the explicit instantiations in csrc/conv_inst_*.hip (so that the name in
    the bench line follows the source, whatever tile shape a layer is built with)"""
    import glob
    import re
    if not launcher.startswith("launch_"):
        return launcher
    if not _INST:
        for f in glob.glob(os.path.join(ROOT, "lrp-imagecaptioning-pytorch_amd", "csrc", "conv_inst_*.hip")):
            for m in re.finditer(r"int (launch_\w+)\([^)]*\)\s*\{\s*return (launch_conv_cfg|launch_conv_f16x3|launch_conv_bf16x6)<([^>]*)>", open(f).read()):
                args = [EPI_ID.get(x.strip(), x.strip()) for x in m.group(3).split(",")]
                kern = {"launch_conv_cfg": "conv_mfma_kernel", "launch_conv_f16x3": "conv_f16x3_kernel", "launch_conv_bf16x6": "conv_bf16x6_kernel"}[m.group(2)]
                if kern == "conv_f16x3_kernel":
                    args += ["false"] * (8 - len(args))           # POOL, F8, B6 default to false
                _INST[m.group(1)] = f"{kern}<{', '.join(str(x) for x in args)}>"
    return _INST.get(launcher, launcher)


MODE_NAME = {0: "fp32 MFMA (v_mfma_f32_32x32x2_f32)",
             1: "bf16x6: both operands split EXACTLY into three bf16 parts (24 significand bits, fp32's exponent range, no operand scales), the six "
                "products with i + j <= 2 on v_mfma_f32_32x32x16_bf16, fp32 accumulate (dropped terms <= 3 x 2^-24 of a product)",
             2: "f16x3: per-map power-of-two scaling, 2-way fp16 split, 3 products, fp32 accumulate",
             3: "f16+f6x2: as f16x3, the two cross products (2^-11 of the result) as block-scaled fp6 e2m3 MFMAs "
                "(v_mfma_scale_f32_32x32x64_f8f6f4, one E8M0 exponent per 16-channel slice; rounds 1-2: fp8 e4m3)"}
# mode 3: one fp16 product + two fp6 products; the fp6 dense peak is four times the fp16 one (MI355X_MICROARCH.md: 32 cycles per
# 32x32x64 MFMA, measured tools/micro/mfma_f6.hip), so an fp6 flop counts a quarter: `achieved` / `peak` is then (time the matrix
# cores need at their peaks) / (measured time), as in the other modes.  (Rounds 1-2 ran the cross products in fp8: 2.0.)
PRODUCTS = {0: 1, 1: 6, 2: 3, 3: 1.5}
# arithmetic the contractions run in (tensors in HBM are fp32 in every mode; everything outside the convolutions is fp32 VALU)
# (the decoders follow the conv mode - lrp_amd.ops.decoder_f16: in modes 0 / 1 their (word, pixel) rules run on the same exact bf16 splits
# (csrc/dense_f16x3.hip, B6) and every other decoder GEMM on the fp32 MFMA / fp32 VALU; the fp16 split products only in modes 2 / 3)
DEC_EXACT = "decoder GEMMs: exact 3 x bf16 splits ((word, pixel) rules) and f32 MFMA"
DEC_F16 = "decoder GEMMs: f16x3 split products (22 operand bits, per-row scale)"
MODE_DTYPE = {0: "f32 (v_mfma_f32_32x32x2_f32); " + DEC_EXACT,
              1: "f32 operands as exact 3 x bf16 splits (24 significand bits, f32 exponent range), 6 bf16 MFMA products, f32 accumulate; " + DEC_EXACT,
              2: "f16x3 split products, f32 accumulate; " + DEC_F16, 3: "f16 + 2 x f6(e2m3, block-scaled) split products, f32 accumulate; " + DEC_F16}
MODE_KEY = {0: "fp32_mfma", 1: "bf16x6", 2: "f16x3", 3: "f16f6"}


def dtype_of(mode, has_vgg=True):
    """the line's `dtype`: the conv mode's arithmetic + what the decoders' GEMMs run on (LRPX_DECODER_F16 can override the rule: say so)"""
    from lrp_amd import ops
    d = MODE_DTYPE[mode] if has_vgg else "f32; " + (DEC_F16 if mode >= 2 else DEC_EXACT)
    if ops.decoder_f16(mode) != (mode >= 2):
        d = d.replace(DEC_F16, DEC_EXACT + " (LRPX_DECODER_F16=0)") if mode >= 2 else d.replace(DEC_EXACT, DEC_F16 + " (LRPX_DECODER_F16=1)")
    return d
# The headline mode: the fastest mode whose operands keep 24 significand bits and fp32's exponent range (a number at
# arithmetic narrower than the reference's fp32 convolutions - LRPtools/lrp_modules.py:124-150, utils.py:21-31 - is not creditable).
# Modes 2 / 3 are reported beside it (value_f16x3 / value_f16f6) with their same-run deviation from the fp32-MFMA chain.
HEADLINE_MODE = 1
# HBM traffic per launch: tools/prof_summary.py traffic-json writes this file from the two --pmc passes (FETCH_SIZE /
# WRITE_SIZE, separate from any tracing); bytes = 2 x FETCH_SIZE raw [gfx950 reports half of wide streaming reads,
# MI355X_MICROARCH.md §HBM; re-calibrated by tools/pmc_calibrate.py] + WRITE_SIZE, divided by the launches and scaled to the maps of this run
TRAFFIC_FILES = ["profiles/r06_pmc_traffic.json", "profiles/r05_pmc_traffic.json", "profiles/r04_pmc_traffic.json", "profiles/r03_pmc_traffic.json", "profiles/r02_pmc_traffic.json", "profiles/r01_pmc_traffic.json"]


def host_cores():
    """CPU cores this process may actually use: affinity mask capped by the cgroup CPU quota (the GPU box
    gives one GPU's share of a large host; oversubscribing OpenMP threads there stalls the run)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("LRPX_CPU_THREADS", "16"))))


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(V, T, words):
    """reference-equivalent CPU path on a bounded sample: 1 image, T words traced, `words` explained."""
    from oracle import lrp_oracle as O, ref_equiv as RE
    from lrp_amd import weights
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu_baseline on {cores} threads ...")
    sd = O.state_to_torch(weights.make_gridtd_state(seed=0, vocab_size=V))
    img = torch.from_numpy(weights.make_images(0, 1))
    cap = weights.make_captions(1, 1, T, V)[0]
    _, _, t_trace, t_words = RE.explain_words(sd, img, cap, words)
    per_image = t_trace + t_words * T / len(words)       # trace once per image (word cost pro-rated if a subset was asked for)
    note = "all of them, nothing extrapolated" if len(words) == T else f"scaled to {T} words"
    return {"value": round(T / per_image, 4), "unit": "maps/s", "cores": cores, "kind": "port",
            "sample": f"1 image, T={T}, V={V}: trace {t_trace:.1f} s + {len(words)} words explained by oracle/ref_equiv.py in {t_words:.1f} s ({note}); "
                      "ref_equiv.py against the imported reference on the same image in the build container: profiles/r04_ref_equiv_vs_reference.json"}


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child `torch.distributed.run` and relay its
    output (rank 0's JSON line goes to our stdout).  This process never touches the GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("starting ranks: " + " ".join(cmd))
    return subprocess.run(cmd, env=env).returncode


LIVE_TRAFFIC = {"doc": None, "note": None}


def pmc_traffic_doc(out_dir, n_maps):
    """rocprofv3 --pmc output directories (any depth below out_dir, one *_counter_collection.csv per pass: columns Kernel_Name, Counter_Name,
    Counter_Value, one row per dispatch and counter) -> {"maps_per_launch", "kernels": {short name: launches, fetch_bytes, write_bytes, l2_hit}}
    for the conv / first-layer kernels; FETCH_SIZE / WRITE_SIZE count kilobytes."""
    import csv
    import glob
    import re
    data = {}
    for f in glob.glob(os.path.join(out_dir, "**", "*_counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = re.sub(r"\(.*", "", re.sub(r"void lrpx::|lrpx::", "", row["Kernel_Name"]))
            data.setdefault(k, {}).setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
    doc = {"maps_per_launch": n_maps, "kernels": {}}
    for k, c in data.items():
        if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c or ("conv_" not in k and "first_layer" not in k):
            continue
        hit, miss = sum(c.get("TCC_HIT_sum", [0])), sum(c.get("TCC_MISS_sum", [0]))
        doc["kernels"][k] = {"launches": len(c["FETCH_SIZE"]), "fetch_bytes": sum(c["FETCH_SIZE"]) * 1024,
                             "write_bytes": sum(c["WRITE_SIZE"]) * 1024, "l2_hit": round(hit / max(hit + miss, 1), 4)}
    return doc


def live_traffic(n_img, n_maps, mode):
    """HBM traffic of the chain's kernels measured IN THIS RUN (the committed summary is not driver-witnessed):
    two `rocprofv3 --pmc` child passes over tools/bench_vgg.py (the relevance chain on the same library: 2 forward + 2 relevance passes of
    `n_maps` maps over `n_img` images) - counters only, one pass per counter group as MI355X_MICROARCH.md's HBM section prescribes, the
    program itself after `--`, cwd and TMPDIR in /tmp.  Runs BEFORE this process touches the GPU (the children need the card to
    themselves, and nothing is exec'd from a process that holds a HIP context).  Any failure leaves the committed summary in charge."""
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None:
        LIVE_TRAFFIC["note"] = "rocprofv3 not on PATH"
        return
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        LIVE_TRAFFIC["note"] = "this process runs under a profiler itself"
        return
    out = tempfile.mkdtemp(prefix="lrpx_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    passes = (("fetch", ["FETCH_SIZE", "GRBM_GUI_ACTIVE"]), ("write", ["WRITE_SIZE", "TCC_HIT_sum", "TCC_MISS_sum"]))
    t0 = time.time()
    try:
        for name, ctr in passes:
            cmd = [exe, "--pmc", *ctr, "--output-format", "csv", "-d", os.path.join(out, name), "--", "python3",
                   os.path.join(ROOT, "tools", "bench_vgg.py"), "--images", str(n_img), "--maps", str(n_maps), "--iters", "1", "--mode", str(mode)]
            r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=240)
            if r.returncode != 0:
                LIVE_TRAFFIC["note"] = f"rocprofv3 --pmc pass '{name}' exited {r.returncode}: {r.stderr.decode(errors='replace')[-200:]}"
                return
        doc = pmc_traffic_doc(out, n_maps)
        if doc["kernels"]:
            LIVE_TRAFFIC["doc"] = doc
            LIVE_TRAFFIC["note"] = (f"measured in this run: two rocprofv3 --pmc child passes (FETCH_SIZE GRBM_GUI_ACTIVE / WRITE_SIZE TCC_HIT_sum "
                                    f"TCC_MISS_sum) over tools/bench_vgg.py --images {n_img} --maps {n_maps} --mode {mode} before the timed region, "
                                    f"{time.time() - t0:.0f} s; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / launches "
                                    "(gfx950 reports half of wide streaming reads: MI355X_MICROARCH.md, HBM section; tools/pmc_calibrate.py)")
        else:
            LIVE_TRAFFIC["note"] = "the --pmc passes returned no conv kernels"
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        LIVE_TRAFFIC["note"] = f"live --pmc passes failed: {type(e).__name__}: {e}"
    finally:
        shutil.rmtree(out, ignore_errors=True)


def read_traffic(kernel, n_maps):
    """(bytes per launch of `kernel` for `n_maps` maps, source) - from this run's own --pmc passes (live_traffic) when they ran, else
    from the PMC summary under profiles/."""
    doc = LIVE_TRAFFIC["doc"]
    if doc is not None and kernel in doc["kernels"]:
        k = doc["kernels"][kernel]
        return round((2.0 * k["fetch_bytes"] + k["write_bytes"]) / k["launches"] / doc["maps_per_launch"] * n_maps), LIVE_TRAFFIC["note"]
    for rel in TRAFFIC_FILES:
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        try:
            doc = json.load(open(path))
            k = doc["kernels"][kernel]
            per_map = (2.0 * k["fetch_bytes"] + k["write_bytes"]) / k["launches"] / doc["maps_per_launch"]
            return round(per_map * n_maps), rel
        except (KeyError, ValueError, ZeroDivisionError):
            continue
    return None, None


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5], help="BASELINE.json configs[] (1-based): see the docstring")
    ap.add_argument("--explainer", default=None, choices=["lrp", "lrp+guided"], help="config 2 / 4: LRP only or LRP + Guided-Backprop")
    ap.add_argument("--batch", type=int, default=None, help="images per GPU per step (default: the config's)")
    ap.add_argument("--words", type=int, default=20)
    ap.add_argument("--vocab", type=int, default=None)
    ap.add_argument("--head", type=int, default=0, help="AoA head explained (configs 3 / 5)")
    ap.add_argument("--gather", nargs="?", const="maps", default=None, choices=["maps", "heatmap", "stats", "none"],
                    help="terminal collective of a step (north_star's 'trivial gather', RCCL over xGMI): maps = the fp32 maps to rank 0 from a "
                         "double-buffered copy on a side stream (overlaps the next step); heatmap = the channel mean, reduced on the device "
                         "first (a third of the bytes), all_gather_into_tensor on preallocated buffers; stats = the tpfp statistics of that "
                         "heat map (4 floats per map); none / absent = no collective")
    ap.add_argument("--lens", default=None, choices=[None, "uniform"],
                    help="config 2: captions of unequal length, words per image ~U[8, --words] (seeded); the step is "
                         "explain_batch(lens=...) and maps/s counts the VALID (image, word) maps only")
    ap.add_argument("--only-dropin", action="store_true", help="print only the configs.dropin_b1 block (the drop-in's one-image calling pattern)")
    ap.add_argument("--allow-experiment", action="store_true",
                    help="print a line even when liblrpx.so is a timing-experiment / profiling build (lrpx_build_flags() non-empty); "
                         "the line is marked value_valid=false")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-modes", action="store_true", help="skip the roofline.modes sweep (every conv mode, same process)")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not run the two rocprofv3 --pmc child passes that measure roofline.traffic in this run (the committed PMC summary under profiles/ is used)")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (BASELINE configs 3 / 4 / 5 and the B=64 line, same process)")
    ap.add_argument("--all-heads", action="store_true", help="config 3: explain all 8 heads of every word in the step (8 x B x T maps)")
    ap.add_argument("--sustain", type=float, default=5.0, help="seconds of the sustained sub-measurement (0 = off)")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph per batch in flight (config 2 LRP, config 5); measured slower than eager launches with batches in flight in both (DESIGN.md)")
    ap.add_argument("--replay", type=int, default=None, choices=[0, 1],
                    help="config 5: issue the step from its recorded call list (AOAEngine.explain_batch_replay: inputs copied into static buffers, then the "
                         "same launches without the interpreter's per-call cost); default 1 for config 5 (the step is host-bound eagerly), 0 elsewhere")
    ap.add_argument("--pipeline", type=int, default=None,
                    help="independent batches in flight on separate HIP streams (1 = serial steps; default 3, 2 for the "
                         "large configs 3 / 4); every step is still one full pass over one batch, the decoder's "
                         "latency-bound kernels of one batch overlap the MFMA-bound CNN chain of another")
    ap.add_argument("--host-threads", action="store_true", help="one host thread per batch in flight issues that batch's launches (A/B: config 5 is "
                    "bound by the GPU's dispatch rate of dependent small kernels, not by the host: 430 000 against 480 000 maps/s)")
    ap.add_argument("--chain-streams", type=int, default=1, help="HIP streams the maps of one VGG16 relevance pass are split over (maps are independent)")
    ap.add_argument("--fp32-mfma", action="store_true", help="same as --conv-mode 0")
    ap.add_argument("--conv-mode", type=int, default=HEADLINE_MODE, choices=[0, 1, 2, 3],
                    help="matrix-core mode of the VGG16 chains: 0 fp32 MFMA, 1 bf16x6 exact splits (default: arithmetic no narrower than the "
                         "reference's fp32), 2 f16x3, 3 fp16 + fp6 cross products (the opt-in speed modes)")
    a = ap.parse_args(argv)
    if a.gather == "none":
        a.gather = None
    if a.explainer is None:
        a.explainer = "lrp+guided" if a.config == 4 else "lrp"
    if a.batch is None:
        a.batch = {2: 16, 3: 64, 4: 32, 5: 32}[a.config]
    if a.vocab is None:
        a.vocab = 9586 if a.config in (2, 4) else 11027
    if a.pipeline is None:
        a.pipeline = 3 if a.config in (2, 5) else 2
    if a.replay is None:
        a.replay = 1 if (a.config == 5 and not a.graph) else 0
    return a


def main():
    a = parse()
    if a.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))          # before anything touches the GPU

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # LRPX_BENCH_BACKEND=gloo + LRPX_BENCH_ONE_GPU=1: rehearsal of the multi-rank path on a one-GPU box (all ranks on cuda:0)
        backend = os.environ.get("LRPX_BENCH_BACKEND", "nccl")
        if os.environ.get("LRPX_BENCH_ONE_GPU") == "1":
            local = 0
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        if (a.config == 2 and a.explainer == "lrp" and not a.no_configs and not a.graph and a.batch == 16 and not a.no_live_traffic
                and not a.only_dropin and not a.fp32_mfma):
            log("HBM traffic of the chain's kernels: two rocprofv3 --pmc child passes (before this process touches the GPU) ...")
            live_traffic(a.batch, a.batch * a.words, a.conv_mode)
            log(f"  -> {LIVE_TRAFFIC['note']}")
        torch.cuda.set_device(0)

    if a.only_dropin:
        print(json.dumps({"dropin_b1": dropin_b1(a.conv_mode)}), flush=True)
        return
    out = run_config(a, dist, rank, world)
    if rank == 0:
        if world == 1 and a.config == 2 and a.explainer == "lrp" and not a.no_configs and not a.graph and a.batch == 16:
            out["configs"] = other_configs(a)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def other_configs(a):
    """BASELINE configs 3 / 4 / 5 and the 64-image gridTD batch, measured in this process after the headline: 4 warm-up (one
    more than the batches in flight: every replica - its buffers come from an allocator emptied just before - has run once) +
    `n` timed steps each with that config's own batches in flight, no sustained leg, no mode sweep, no CPU baseline."""
    import gc
    res = {}
    todo = [("3", ["--config", "3"], 6), ("3_all_heads", ["--config", "3", "--all-heads"], 2),
            ("4", ["--config", "4"], 6), ("5", ["--config", "5"], 80), ("b64", ["--config", "2", "--batch", "64", "--pipeline", "2"], 6),
            ("varlen", ["--config", "2", "--lens", "uniform", "--batch", "23"], 12)]
    # the same lines in the opt-in speed mode 3 (fp16 + fp6 cross products): reported beside the headline-mode values, never instead
    fast = [("3", ["--config", "3"], 4), ("4", ["--config", "4"], 4), ("5", ["--config", "5"], 60), ("b64", ["--config", "2", "--batch", "64", "--pipeline", "2"], 4)]
    todo += [(k + "#f16f6", argv, n) for k, argv, n in fast] if a.conv_mode != 3 else []
    for key, argv, n in todo:
        gc.collect()
        torch.cuda.empty_cache()
        fast_mode = key.endswith("#f16f6")
        b = parse(argv + ["--steps", str(n), "--warmup", "4", "--sustain", "0", "--no-modes", "--no-cpu-baseline", "--no-configs",
                          "--conv-mode", "3" if fast_mode else str(a.conv_mode)])
        o = run_config(b, None, 0, 1)
        if fast_mode:
            k0 = key.split("#")[0]
            if k0 == "5":       # (no CNN stage: what mode 3 changes here is the decoder's GEMMs - fp16 split products instead of exact splits / fp32)
                res[k0]["value_f16x3"] = o["value"]
                res[k0]["value_f16x3_note"] = f"{DEC_F16}: narrower than fp32, opt-in (conv modes 2 / 3 or LRPX_DECODER_F16=1), {n} timed steps, same process"
            else:
                res[k0]["value_f16f6"] = o["value"]
                res[k0]["value_f16f6_note"] = (f"conv mode 3 (fp16 + block-scaled fp6 cross products; {DEC_F16}: narrower than fp32, opt-in), "
                                               f"{n} timed steps, same process")
            log(f"configs[{k0}] mode 3 (f16+f6, f16x3 decoder): {o['value']:.0f} maps/s")
            continue
        r = o.get("roofline") or {}
        line = {"value": o["value"], "unit": o["unit"], "dtype": o["dtype"], "ms_per_step": o["ms_per_step"], "steps": n, "maps_per_step": o["config"]["maps_per_step"],
                "batches_in_flight": o["config"]["batches_in_flight"], "workload": o["config"]["workload"],
                "roofline": {k: r.get(k) for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "frac_algorithmic", "ms_per_launch", "chain_frac") if k in r}}
        if "chain" in r:
            line["roofline"]["chain_ms"] = r["chain"]["ms_per_step"]
        log(f"configs[{key}]: {o['value']:.0f} maps/s, {o['ms_per_step']:.2f} ms/step")
        if key == "3_all_heads":
            res["3"]["all_heads"] = {k: line[k] for k in ("value", "unit", "ms_per_step", "steps", "maps_per_step")}
            res["3"]["all_heads"]["note"] = "encoder + decoder trace once per batch, decoder relevance + CNN relevance + running sums for each of the 8 heads"
        else:
            res[key] = line
    gc.collect()
    torch.cuda.empty_cache()
    res["dropin_b1"] = dropin_b1(a.conv_mode)
    gc.collect()
    torch.cuda.empty_cache()
    return res


def dropin_b1(conv_mode):
    """The drop-in's OWN calling pattern: one image per `explain_caption` call, as evaluation.py:806-838 and
    models/gridTDmodel.py:1141-1156 run it - a 20-word caption handed over (`caption_encode=`) and the explainer's own beam search
    (beam 2, up to 50 words, :935); the explainer constructed once, and constructed anew for every image as the reference's
    evaluation loop does (the device engine is then shared per weight set: explainers/engine_cache.py)."""
    import types
    import lrp_amd  # noqa: F401
    from lrp_amd import _lib, weights
    from lrp_amd.explainers import engine_cache
    from lrp_amd.explainers.gridtd import ExplainGridTDAttention
    _lib.load().lrpx_set_conv_mode(conv_mode)
    V, T = 9586, 20
    sd = {k: torch.from_numpy(v) for k, v in weights.make_gridtd_state(seed=0, vocab_size=V).items()}
    wm = weights.make_word_map(V)
    args = types.SimpleNamespace(embed_dim=512, hidden_dim=512, encoder="vgg16", weight="", save_path="/tmp", dataset="synthetic",
                                 height=224, width=224)
    img = torch.from_numpy(weights.make_images(100, 1)).cuda()
    cap = [int(c) for c in weights.make_captions(200, 1, T, V)[0]]

    def timed(fn, n):
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return sorted(ts)[len(ts) // 2]
    engine_cache.clear()
    t_cold = timed(lambda: ExplainGridTDAttention(args, wm, model=sd), 1)          # weight upload + every weight pack
    ex = ExplainGridTDAttention(args, wm, model=sd)
    for _ in range(3):
        ex.explain_caption(img, caption_encode=cap)
    ms_warm = timed(lambda: ex.explain_caption(img, caption_encode=cap), 10)
    ms_new = timed(lambda: ExplainGridTDAttention(args, wm, model=sd).explain_caption(img, caption_encode=cap), 5)
    ms_fast = None
    if conv_mode != 3:          # the same call in the opt-in speed mode (per-context mode of this explainer's engine: nothing global changes)
        ex.engine.vgg.conv_mode = 3
        for _ in range(3):
            ex.explain_caption(img, caption_encode=cap)
        ms_fast = timed(lambda: ex.explain_caption(img, caption_encode=cap), 10)
        ex.engine.vgg.conv_mode = None
    ex.explain_caption(img)
    n_beam = ex.caption_length
    ms_beam = timed(lambda: ex.explain_caption(img), 5)
    res = {"workload": "ExplainGridTDAttention.explain_caption on ONE resident 224x224 image (B = 1), V=9586, conv mode %d" % conv_mode,
           "given_caption": {"words": T, "ms_per_call": round(ms_warm, 3), "maps_per_s": round(T / ms_warm * 1e3, 1),
                             "note": "explainer constructed once; median of 10 calls, each synchronised (the caller reads the maps)",
                             "ms_per_call_f16f6": None if ms_fast is None else round(ms_fast, 3)},
           "new_explainer_per_image": {"words": T, "ms_per_call": round(ms_new, 3), "maps_per_s": round(T / ms_new * 1e3, 1),
                                       "note": "ExplainGridTDAttention(args, word_map, model) + explain_caption per image (evaluation.py:811-838); "
                                               "the device engine of the weight set is reused (explainers/engine_cache.py)"},
           "first_construction_ms": round(t_cold, 1),
           "beam_search_caption": {"words": n_beam, "ms_per_call": round(ms_beam, 3), "maps_per_s": round(n_beam / ms_beam * 1e3, 1) if n_beam else 0.0,
                                   "note": "the explainer captions the image itself: beam 2, up to 50 steps (models/gridTDmodel.py:935), host "
                                           "bookkeeping with one small device->host read per step as the reference's"}}
    log(f"configs[dropin_b1]: {ms_warm:.2f} ms per 20-word image warm" + (f" ({ms_fast:.2f} in conv mode 3)" if ms_fast else "") + f", {ms_new:.2f} ms with a new explainer per image, first construction "
        f"{t_cold:.0f} ms, beam-search caption of {n_beam} words {ms_beam:.2f} ms")
    engine_cache.clear()
    del ex
    # the AoA explainer the same way: evaluation.py:637 `explain_caption(img_filepath, head_idx)`, models/aoamodel.py:1171-1176
    from lrp_amd.explainers.aoa import ExplainAOAAttention
    Va = 11027
    sda = {k: torch.from_numpy(v) for k, v in weights.make_aoa_state(seed=0, vocab_size=Va).items()}
    wma = weights.make_word_map(Va)
    args.num_head = 8
    capa = [int(c) for c in weights.make_captions(201, 1, T, Va)[0]]
    exa = ExplainAOAAttention(args, wma, model=sda)
    for _ in range(3):
        exa.explain_caption(img, 0, caption_encode=capa)
    ms_a = timed(lambda: exa.explain_caption(img, 0, caption_encode=capa), 10)
    ms_a_new = timed(lambda: ExplainAOAAttention(args, wma, model=sda).explain_caption(img, 0, caption_encode=capa), 5)
    res["aoa_given_caption"] = {"words": T, "head": 0, "vocab": Va, "ms_per_call": round(ms_a, 3), "maps_per_s": round(T / ms_a * 1e3, 1),
                                "new_explainer_per_image_ms": round(ms_a_new, 3),
                                "note": "ExplainAOAAttention.explain_caption(img, head_idx, caption_encode=...) on the same image, V=11027"}
    log(f"configs[dropin_b1]: AoA {ms_a:.2f} ms per 20-word image warm, {ms_a_new:.2f} ms with a new explainer per image")
    engine_cache.clear()
    return res


def run_config(a, dist, rank, world):
    """one measurement (the contract's timed region + the roofline of its dominant kernel) -> the dict of the JSON line"""
    import lrp_amd  # noqa: F401
    from lrp_amd import _lib, ops, weights

    B, T, V = a.batch, a.words, a.vocab
    lib = _lib.load()
    mode = 0 if a.fp32_mfma else a.conv_mode
    lib.lrpx_set_conv_mode(mode)
    torch.set_num_threads(min(8, host_cores()))
    if rank == 0:
        log(f"config {a.config} ({a.explainer}): building weights + engine (B={B}, T={T}, V={V}, world={world})")
    n_pipe = max(1, a.pipeline)
    streams = [torch.cuda.Stream() for _ in range(n_pipe)] if n_pipe > 1 else [None]
    caps = torch.from_numpy(weights.make_captions(200 + rank, B, T, V)).cuda()
    state = {"caps": caps}
    guided = a.explainer == "lrp+guided"
    heads = list(range(8)) if (a.all_heads and a.config == 3) else [a.head]
    maps_per_gpu = B * T * (2 if guided else 1) * len(heads)
    has_vgg = a.config != 5
    build_flags = lib.lrpx_build_flags().decode()
    if build_flags and not a.allow_experiment:
        # a timing-experiment / STAMP library (loaded through LRPX_LIB_PATH) computes wrong results on purpose; its maps/s
        # must never be recorded as a product number
        print(f"bench.py: liblrpx.so is an experiment build ({build_flags!r}); pass --allow-experiment for a line marked invalid",
              file=sys.stderr)
        sys.exit(3)
    lens = None
    if a.lens == "uniform":
        assert a.config == 2 and not guided, "--lens applies to config 2 (LRP)"
        import numpy as np
        lens = np.random.RandomState(300 + rank).randint(8, T + 1, size=B).tolist()
        maps_per_gpu = int(sum(lens))

    def buf(name, k, *shape):
        key = f"{name}{k}"
        if key not in state:
            state[key] = torch.empty(*shape, device="cuda")
        return state[key]

    if a.config in (2, 4):
        from lrp_amd.explainers.gridtd import GridTDEngine
        eng = GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=V))
        images = torch.from_numpy(weights.make_images(100 + rank, B)).cuda()     # every rank: its own shard (seed offset)
        state["images"] = images

        def one_step(e, k):
            if lens is not None:          # captions of unequal length: the public entry, padded (B,T,...) results with running sums
                maps, r_words, _ = e.explain_batch(images, caps, lens=lens, accumulate=True, predictions=True)
                return maps, r_words
            enc = e.encode(images)
            tr = e.trace(enc, caps, predictions=True)
            r_feat, r_words, row2img = e.relevance(enc, tr)
            maps = e.vgg.relevance(r_feat, row2img, out=buf("maps", k, B * T, 3, 224, 224), streams=a.chain_streams)
            cum = ops.cumsum_maps(maps, B, T, out=buf("cum", k, B * T, 3, 224, 224))   # explain_caption's running sums
            if k == 0:
                state["chain_in"] = (r_feat, row2img)
            if guided:     # config 4: guided backprop of the same words on the same encoder trace (own decoder trace, :1323-1422)
                trg = e.trace(enc, caps, predictions=False, grad=True)
                d_feat, _, _ = e.guided_gradient(enc, trg)
                state["gmaps"] = e.vgg.guided_backprop(d_feat, row2img, out=buf("gmaps", k, B * T, 3, 224, 224))
            return cum, r_words
        workload = (f"BASELINE configs[{a.config - 1}]: batch-{B} 224x224 images x {T}-word captions per GPU, VGG16+gridTD, "
                    f"LRP alpha1beta0 (conv) + epsilon (decoder){' + Guided-Backprop side by side' if guided else ''}, V={V}, random-init")
        if lens is not None:
            workload += (f"; captions of UNEQUAL length, words per image ~U[8, {T}] = {lens} ({sum(lens)} valid maps per step, "
                         "padded words cost no (word, pixel) rule and no VGG16 chain time; maps/s counts valid maps only)")
    elif a.config == 3:
        from lrp_amd.explainers.aoa import AOAEngine
        eng = AOAEngine(weights.make_aoa_state(seed=0, vocab_size=V))
        images = torch.from_numpy(weights.make_images(100 + rank, B)).cuda()

        def one_step(e, k):
            enc = e.encode(images)
            tr = e.trace(enc, caps, predictions=True)
            for hd in heads:          # (--all-heads: the heads share the traces, everything behind `lrp_mha` is per head)
                r_feat, r_words, row2img = e.relevance(enc, tr, hd)
                maps = e.vgg.relevance(r_feat, row2img, out=buf("maps", k, B * T, 3, 224, 224), streams=a.chain_streams)
                cum = ops.cumsum_maps(maps, B, T, out=buf(f"cum{hd}_", k, B * T, 3, 224, 224))
            if k == 0:
                state["chain_in"] = (r_feat, row2img)
            return cum, r_words
        workload = (f"BASELINE configs[2]: batch-{B} 224x224 images x {T}-word captions per GPU, VGG16 + AoA 8-head decoder, "
                    f"LRP through multi-head attention ({'all 8 heads' if len(heads) > 1 else f'head {a.head}'}) + FC predictor, V={V}, random-init")
    else:
        from lrp_amd.explainers.aoa import AOAEngine
        eng = AOAEngine(weights.make_aoa_state(seed=0, vocab_size=V, feat_dim=2048, with_encoder=False))
        feats = torch.from_numpy(weights.make_bu_features(100 + rank, B)).cuda()

        def one_step(e, k):
            if a.graph:          # the step replayed from a HIP graph per batch in flight (the path is host-launch-bound eagerly)
                return e.explain_batch_graph(caps, a.head, features=feats, predictions=True)
            if a.replay:         # the step as a recorded call list (explain_batch_replay): same kernels as ordinary launches, ~1.5 us of host time each
                return e.explain_batch_replay(caps, a.head, features=feats, predictions=True)
            enc = e.encode(features=feats)
            tr = e.trace(enc, caps, predictions=True)
            r_feat, r_words, _ = e.relevance(enc, tr, a.head)
            return r_feat, r_words
        workload = (f"BASELINE configs[4]: batch-{B} x 36x2048 bottom-up region features x {T}-word captions per GPU, AoA-BU "
                    f"decoder, LRP relevance back to the region features (head {a.head}), V={V}, random-init")
    engines = [eng] + [eng.replica() for _ in range(n_pipe - 1)]     # shared weights, own trace / workspace buffers
    gloo = dist is not None and dist.get_backend() == "gloo"
    og = None
    if a.gather and world > 1 and has_vgg:
        # lrp_amd.shard.OverlappedGather: copy + collective on a side stream, one slot more than batches in flight.  The rehearsal
        # backend (gloo) moves host tensors only: its slots are pinned host memory and the copy is waited for before the collective.
        from lrp_amd import shard
        shape = {"maps": (B * T, 3, 224, 224), "heatmap": (B * T, 224, 224), "stats": (B * T, 4)}[a.gather]
        og = shard.OverlappedGather(shape, device="cpu" if gloo else "cuda", depth=n_pipe + 1,
                                    mode="gather" if a.gather == "maps" else "all_gather")

    def gather_maps(m):
        """the terminal collective of north_star: every rank's (reduced) maps travel from a side stream (RCCL over xGMI; device tensors
        end to end) while the next step computes"""
        from lrp_amd import shard
        red = shard.reduce_for_gather(m.view(B * T, 3, 224, 224), a.gather)
        og.submit(red)

    step_no = [0]

    def step(end_events=None):
        k = step_no[0] % n_pipe
        step_no[0] += 1
        use_graph = a.graph and a.config == 2 and not guided
        if n_pipe > 1:
            with torch.cuda.stream(streams[k]):
                # --graph: the step replayed from a HIP graph captured per replica (same kernels, same buffers every step)
                out = engines[k].explain_batch_graph(images, caps, accumulate=True, predictions=True) if use_graph \
                    else one_step(engines[k], k)
                if og is not None:
                    gather_maps(out[0])
                if end_events is not None:
                    ev = torch.cuda.Event(enable_timing=True)
                    ev.record()
                    end_events.append(ev)
            return out
        elif use_graph:
            out = eng.explain_batch_graph(images, caps, accumulate=True, predictions=True)
        else:
            out = one_step(eng, 0)
        if og is not None:
            gather_maps(out[0])
        if end_events is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record()
            end_events.append(ev)
        return out

    def barrier():
        if og is not None:
            og.finish()                 # every collective of the region has delivered before the clock stops
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    host_threads = a.host_threads and n_pipe > 1 and not a.gather

    def run_steps(n, events):
        """n steps in issue order: one host thread round-robin over the streams, or (--host-threads) one host thread per
        batch in flight - thread k issues steps k, k + n_pipe, ... on its own stream and replica (the native step loops
        release the interpreter lock while they issue their launches).  Measured on config 5 (~330 small launches per
        step): no gain - three dependent chains of 4 - 25 us kernels overlap 1.7x whoever issues them, the ceiling is
        the GPU's dispatch rate (~250 000 launches/s), so the default stays one thread."""
        if not host_threads:
            out = None
            for _ in range(n):
                out = step(events)
            return out
        import threading
        base = step_no[0]
        step_no[0] += n
        outs = [None] * n_pipe

        def worker(k):
            torch.cuda.set_device(torch.cuda.current_device())
            with torch.cuda.stream(streams[k]):
                for i in range(base, base + n):
                    if i % n_pipe == k:
                        outs[k] = one_step(engines[k], k)
        ths = [threading.Thread(target=worker, args=(k,)) for k in range(n_pipe)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        last = (base + n - 1) % n_pipe
        if events is not None:          # (per-step completion events are not taken in this mode: one end event)
            ev = torch.cuda.Event(enable_timing=True)
            for st_ in streams:
                torch.cuda.current_stream().wait_stream(st_)
            ev.record()
            events.append(ev)
        return outs[last]

    def timed(n, events=None):
        """n steps bracketed by barrier + synchronize on both sides; MAX over ranks"""
        barrier()
        t0 = time.perf_counter()
        out = run_steps(n, events)
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = tmax.item()
        return dt, out

    run_steps(a.warmup, None)
    barrier()
    # Host hygiene of a serving loop: everything built so far (weights as numpy arrays, 10 000-entry word maps, engines, the interpreter's
    # modules) is long-lived - moved to the permanent generation, the cyclic collector no longer walks it.  Without this a full collection
    # lands inside the timed region now and then (~20 ms each); the bottom-up step is host-bound at 0.67 ms and showed it as 0.86 - 0.90 ms
    # per step against 0.68 with the collector off (same box, 4 runs each).  The collector stays ON.
    import gc
    gc.collect()
    gc.freeze()
    if rank == 0:
        log("warm-up done, timing")
    ev0 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    ev0.record()
    events = [ev0]
    dt, outs = timed(a.steps, events)
    maps = outs[0]
    if not build_flags:                                       # (a timing-experiment build computes garbage on purpose: tools/ab_chain.sh)
        ops.check_relevance(maps, finite=True, nonzero=True)  # the reference's asserts, outside the timed region
    # step completions by HIP events; with n_pipe batches in flight completions come in bursts, so the per-step time is
    # taken over windows of n_pipe consecutive completions: (end[i] - end[i - n_pipe]) / n_pipe, median over the region
    ends = [0.0] + [ev0.elapsed_time(e) for e in events[1:]]
    w = min(n_pipe, a.steps)
    if host_threads:                   # one end event for the whole region in this mode
        ends = [0.0] * a.steps + [ends[-1]]
        median_ms = ends[-1] / a.steps
    else:
        gaps = sorted((ends[i] - ends[i - w]) / w for i in range(w, len(ends)))
        median_ms = gaps[len(gaps) // 2] if len(gaps) % 2 else 0.5 * (gaps[len(gaps) // 2 - 1] + gaps[len(gaps) // 2])
    if rank == 0:
        log(f"timed region done: {dt / a.steps * 1e3:.2f} ms/step (median step interval by HIP events {median_ms:.2f} ms)")
    sustained = None
    if a.sustain > 0:
        n_sus = max(a.steps, int(math.ceil(a.sustain / (dt / a.steps))))
        if dist is not None:           # the same count on every rank
            t = torch.tensor([n_sus], device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            n_sus = int(t.item())
        dts, _ = timed(n_sus)
        sustained = {"seconds": round(dts, 2), "steps": n_sus, "ms_per_step": round(dts / n_sus * 1e3, 3),
                     "value": round(world * maps_per_gpu * n_sus / dts, 2), "unit": "maps/s"}
        if rank == 0:
            log(f"sustained: {n_sus} steps in {dts:.1f} s = {dts / n_sus * 1e3:.2f} ms/step")

    if rank == 0:
        n_maps = world * maps_per_gpu * a.steps
        metric = {2: "LRP relevance maps/sec (VGG16+gridTD, 224x224, 20-token caption)",
                  4: "LRP relevance maps/sec (VGG16+gridTD, 224x224, 20-token caption)" + ("; LRP + Guided-Backprop maps" if guided else ""),
                  3: "LRP relevance maps/sec (VGG16+AoA, 224x224, 20-token caption)",
                  5: "LRP relevance maps/sec (AoA bottom-up 36x2048 features, 20-token caption)"}[a.config]
        out = {"metric": metric, "value": round(n_maps / dt, 2), "unit": "maps/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": dtype_of(mode, has_vgg), "data": "synthetic",
               "config": {"workload": workload, "images_per_gpu": B, "words": T, "vocab": V,
                          "maps_per_step": world * maps_per_gpu,
                          "unit_of_work": "trace (VGG16 forward + decoder, predictions kept) + decoder relevance + CNN relevance + "
                                          "running sums of the maps" if has_vgg else "decoder trace (predictions kept) + decoder relevance",
                          "sharding": f"images x{world}, no data-path collective" + (f" + terminal gather of the {a.gather} (side stream, double-buffered)" if a.gather else ""),
                          "batches_in_flight": n_pipe, "launch": ("HIP graph replay per batch in flight" if a.graph else ("recorded call list replayed as ordinary launches (explain_batch_replay)" if (a.replay and a.config == 5) else "eager")) + (", one host thread per batch in flight" if host_threads else "")},
               "median_ms": round(median_ms, 3), "hip_event_ms_per_step": round(ends[-1] / a.steps, 3), "sustained": sustained,
               "build_flags": build_flags, "value_valid": not build_flags}
        if build_flags:
            out["metric"] = "INVALID (experiment build of liblrpx.so: " + build_flags + ") " + out["metric"]
        out["median_ms_note"] = (f"median over the timed region of (completion[i] - completion[i-{w}]) / {w} by HIP events "
                                 f"({w} batches in flight complete in bursts); hip_event_ms_per_step = last completion / steps")
        if world == 1 and has_vgg and lens is None:
            out["roofline"] = roofline(a, lib, eng, state, maps, B, T, mode)
            if a.config == 2 and not guided and not a.no_modes and not a.graph:      # (the sweep switches modes between eager steps)
                modes = mode_sweep(lib, engines, streams, one_step, state, B, T, mode)
                out["roofline"]["modes"] = modes
                # every mode's maps/s and its worst-map deviation from the fp32-MFMA chain as TOP-LEVEL scalars: `value` is
                # the headline mode's 20-step region above; value_<mode> are the sweep's 6-step figures of the same process
                for m_, key_ in MODE_KEY.items():
                    out["value_" + key_] = modes[str(m_)]["maps_per_s"]
                    if m_ != 0:
                        out["dev_chain_" + key_] = modes[str(m_)]["dev_vs_fp32"]["chain_worst_map"]
                        out["dev_step_" + key_] = modes[str(m_)]["dev_vs_fp32"]["step_worst_map"]
                out["value_modes_note"] = ("`value` / `dtype`: conv mode %d (%s).  value_<mode>: the same step in every matrix-core mode, 2 warm-up + 6 timed steps "
                                           "each in this process.  dev_chain_<mode>: worst map of max|R - R_fp32| / max|R_fp32| over this step's %d maps, the VGG16 "
                                           "relevance chain of that mode against the fp32-MFMA chain (mode 0) on ONE trace and the step's own decoder relevance; "
                                           "dev_step_<mode>: the same for the whole step (each mode runs its own forward trace: includes max-pool tie flips, "
                                           "DESIGN.md 3)" % (mode, MODE_DTYPE[mode], B * T))
        elif world == 1 and not has_vgg:
            # config 5: no CNN stage; HBM-bound projector / v_proj rules.  Algorithmic bytes per map: read F,
            # write R_feat, read the projected features: 3 x 36 x 2048 x 4 B = 0.9 MB (weights amortised over the batch)
            byts = 3.0 * 36 * 2048 * 4 * B * T
            gbs = byts / (dt / a.steps) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "whole step (decoder relevance back to 36x2048 region features)",
                               "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                               "traffic": None}
        if world == 1 and a.config == 2 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(V, T, list(range(T)))       # every word of one image: nothing extrapolated
        return out
    return None


def e_images(state):
    return state["images"]


def chain_ms(eng, r_feat, row2img, out, reps):
    ms = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        eng.vgg.relevance(r_feat, row2img, out=out)
        e1.record()
        e1.synchronize()
        ms.append(e0.elapsed_time(e1))
    return sorted(ms)[len(ms) // 2]          # median: one repetition now and then takes twice as long (clock / power event)


def roofline(a, lib, eng, state, maps, B, T, mode):
    import ctypes as C
    torch.cuda.synchronize()
    if "chain_in" not in state:          # (--graph: the step ran inside a captured graph) one eager decoder pass for the chain's input
        e = eng
        enc = e.encode(e_images(state))
        state["chain_in"] = e.relevance(enc, e.trace(enc, state["caps"], predictions=False))[::2]
        state["maps0"] = torch.empty(B * T, 3, 224, 224, device="cuda")
    r_feat, row2img = state["chain_in"]
    out = state["maps0"]
    reps = max(3, min(a.steps, 10))
    # (1) the whole VGG16 relevance chain (12 conv launches + first-layer kernel), HIP events
    c_ms = chain_ms(eng, r_feat, row2img, out, reps)
    # (2) per-launch times: HIP events recorded by the library on the launch stream around every conv launch of the same
    # chain on the same inputs
    per_layer = [0.0] * 17
    lib.lrpx_vgg16_layer_timing(1, None)
    buf = (C.c_float * 17)()
    for _ in range(reps):
        eng.vgg.relevance(r_feat, row2img, out=out)
        lib.lrpx_vgg16_layer_timing(-1, buf)
        per_layer = [p + float(v) for p, v in zip(per_layer, buf)]
    lib.lrpx_vgg16_layer_timing(0, None)
    per_layer = [p / reps for p in per_layer]
    # The dominant kernel = the kernel NAME with the largest total time over the launches of one pass (what a rocprofv3 --stats
    # summary of the serial step ranks first among the chain's kernels), chosen from THIS run's per-layer times, not hard-coded.
    peak = PEAK_FP32_MFMA_TF if mode == 0 else PEAK_16BIT_MFMA_TF
    n = B * T
    groups, table = {}, []
    for l in sorted(VGG_CONV):
        if per_layer[l] <= 0:
            continue
        name = kernel_name(rel_launcher(l, mode))
        fl = layer_flop_per_map(l) * n
        alg_l = fl / per_layer[l] / 1e9
        g = groups.setdefault(name, {"ms": 0.0, "flop": 0.0, "layers": []})
        g["ms"] += per_layer[l]; g["flop"] += fl; g["layers"].append(VGG_CONV[l][0])
        row = {"layer": VGG_CONV[l][0], "kernel": name, "ms": round(per_layer[l], 4), "algorithmic_tflops": round(alg_l, 1),
               "frac": round(PRODUCTS[mode] * alg_l / peak, 4)}
        if l == 0:          # the 3-channel first layer: 0.3 % of the flop, bound by reading S (64 channels x 224 x 224 fp32 per map)
            gbs = 224 * 224 * 64 * 4.0 * n / per_layer[l] / 1e6
            row.update({"bound": "hbm", "algorithmic_gbs": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 4)})
        table.append(row)
    conv_groups = {k: v for k, v in groups.items() if not k.startswith("first_layer")}
    dom_name = max(conv_groups, key=lambda k: conv_groups[k]["ms"])
    dom = conv_groups[dom_name]
    dom_ms = dom["ms"] / len(dom["layers"])                                   # average launch of that kernel
    flop = dom["flop"] / len(dom["layers"])                                   # average algorithmic flop / launch
    alg = flop / dom_ms / 1e9                                                 # TFLOP/s, fp32-equivalent
    exe = PRODUCTS[mode] * alg
    traffic, src = read_traffic(dom_name, n)
    sum_ms = sum(per_layer)
    chain_alg = GFLOP_PER_MAP * n / sum_ms                                    # all 13 layers' flop over all 13 launches' time
    return {
        "bound": "mfma", "kernel": dom_name,
        "kernel_note": f"relevance step of {' / '.join(dom['layers'])}: {n} maps per launch, {len(dom['layers'])} launch(es) per step; the "
                       f"kernel name with the largest total time in this run's chain ({dom['ms']:.3f} of {sum_ms:.3f} ms)",
        "achieved": round(exe, 1), "peak": peak, "unit": "TFLOP/s", "frac": round(exe / peak, 4),
        "frac_algorithmic": round(alg / peak, 4), "mfma_dtype": MODE_NAME[mode],
        "mfma_products_per_fp32_product": PRODUCTS[mode], "algorithmic_tflops": round(alg, 1),
        "ms_per_launch": round(dom_ms, 4), "flop_per_launch": flop,
        "traffic": traffic, "traffic_source": src,
        #  `kernel` is named by bench.rel_launcher, a restatement of csrc/lrpx_vgg.hip:conv_dispatch for the default switches:
        # true when a kernel of exactly that name was profiled in the PMC passes of `traffic_source` (same library, same dispatch)
        "kernel_name_verified": traffic is not None,
        "chain_frac": round(PRODUCTS[mode] * chain_alg / peak, 4),
        "chain_frac_note": "sum of the 13 layers' algorithmic flop / sum of their launch times (HIP events per launch), same accounting as frac",
        "per_layer": table,
        "peak_note": ("`peak` is the guide's nominal dense 16-bit rate.  Measured on this part the matrix cores' rate depends on the operands "
                      "(power-limited clock): v_mfma_f32_32x32x16_bf16 from registers, six products per accumulator as in this kernel, runs at "
                      "2489 TFLOP/s on zeros and at 1830 - 1880 on the three planes of split random data (tools/micro/mfma_bf16_peak.hip, "
                      "profiles/r06_mfma_bf16_peak.txt): the power-limited roof of mode 1 is ~310 fp32-equivalent TFLOP/s = 0.74 of `peak`; "
                      "v_mfma_f32_32x32x16_f16: 2484 / 1656 (profiles/r04_mfma_f16_peak.txt).  rocm-smi under the mode-1 chain: 1280 - 1320 W of package "
                      "power, shader clock 1.97 - 2.02 GHz of 2.4 (profiles/r06_clocks_under_load.txt): the chain runs at the part's power cap"),
        "accounting": ("executed matrix flop = algorithmic x products; mode 3 executes 1 fp16 + 2 fp6 products per fp32 "
                       "product and an fp6 flop counts 1/4 (fp6 dense peak = 4 x fp16 peak; the path's roof is 2500 / 1.5 = 1667 "
                       "algorithmic TFLOP/s, 1250 with the fp8 cross products of rounds 1-2), so achieved/peak = matrix "
                       "time at peak / measured time; frac_algorithmic = fp32-equivalent flop against the same peak") if mode == 3 else
                      "executed matrix flop = algorithmic x products; frac_algorithmic = fp32-equivalent flop against the same peak",
        "chain": {"ms_per_step": round(c_ms, 3), "sum_of_launches_ms": round(sum_ms, 3), "flop_per_step": GFLOP_PER_MAP * 1e9 * n,
                  "algorithmic_tflops": round(GFLOP_PER_MAP * n / c_ms, 1),
                  "conv_ms_by_layer": {str(l): round(v, 3) for l, v in enumerate(per_layer) if v > 0}}}


def mode_sweep(lib, engines, streams, one_step, state, B, T, mode_now):
    """The same step (and the chain alone) in every matrix-core mode, measured in this process: 2 warm-up + 6 timed steps
    with the batches in flight of the headline; chain = HIP events around lrpx_vgg16_relevance, median of 5 repetitions."""
    res = {}
    n_pipe = len(engines)
    r_feat, row2img = state["chain_in"]
    # (a) the relevance chain of every mode on ONE trace (forward pass of mode 0) and the step's own decoder relevance: isolates the
    # arithmetic of the relevance convolutions; worst / mean over the maps of max|R_m - R_0| / max|R_0|
    torch.cuda.synchronize()
    lib.lrpx_set_conv_mode(0)
    engines[0].encode(e_images(state))
    dev_chain = {}
    ref = None
    for m in (0, 1, 2, 3):
        lib.lrpx_set_conv_mode(m)
        got = engines[0].vgg.relevance(r_feat, row2img, out=state["maps0"])
        torch.cuda.synchronize()
        if m == 0:
            ref = got.clone()
            ref_max = ref.flatten(1).abs().amax(1).clamp_min(1e-30)
        else:
            e = (got - ref).flatten(1).abs().amax(1) / ref_max
            dev_chain[m] = (e.max().item(), e.mean().item())
    step_ref = None
    for m in (0, 1, 2, 3):
        torch.cuda.synchronize()
        lib.lrpx_set_conv_mode(m)

        def run(n):
            for i in range(n):
                k = i % n_pipe
                if streams[k] is None:
                    one_step(engines[k], k)
                else:
                    with torch.cuda.stream(streams[k]):
                        one_step(engines[k], k)
        run(2)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(6)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 6
        # (b) the whole step in this mode (its own forward trace, decoder trace and relevance) against the whole step in mode 0
        one_step(engines[0], 0)
        torch.cuda.synchronize()
        maps_m = state["maps0"]
        dev = None
        if m == 0:
            step_ref = maps_m.clone()
            step_max = step_ref.flatten(1).abs().amax(1).clamp_min(1e-30)
        else:
            d = (maps_m - step_ref).flatten(1).abs() / step_max[:, None]
            e = d.amax(1)
            dev = {"chain_worst_map": float("%.3e" % dev_chain[m][0]), "chain_mean_map": float("%.3e" % dev_chain[m][1]),
                   "step_worst_map": float("%.3e" % e.max().item()), "step_mean_map": float("%.3e" % e.mean().item()),
                   "step_frac_pixels_above_1e-4": float("%.3e" % (d > 1e-4).float().mean().item())}
            del d
        c_ms = chain_ms(engines[0], r_feat, row2img, state["maps0"], 5)
        res[str(m)] = {"maps_per_s": round(B * T / dt, 1), "ms_per_step": round(dt * 1e3, 3), "chain_ms": round(c_ms, 3),
                       "dtype": MODE_DTYPE[m]}
        if dev is not None:
            res[str(m)]["dev_vs_fp32"] = dev
        log(f"mode {m}: {B * T / dt:.0f} maps/s, chain {c_ms:.2f} ms" + (f", worst map vs fp32: chain {dev['chain_worst_map']:.1e}, step {dev['step_worst_map']:.1e}" if dev else ""))
    torch.cuda.synchronize()
    lib.lrpx_set_conv_mode(mode_now)
    return res


if __name__ == "__main__":
    main()
