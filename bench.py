#!/usr/bin/env python3
"""Headline benchmark: LRP relevance maps/sec (VGG16 + gridTD, 224x224, 20-token caption).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path over one batch per GPU (BASELINE.json configs[1]: 16 images x 20 words
= 320 relevance maps, V=9586): VGG16 forward trace (+Z+), decoder trace, decoder relevance, VGG16 relevance.
Inputs (images, token ids) are resident in HBM before the timed region; weights are random-init from the
seeded generator (no network for checkpoints), data synthetic.  Images are independent, so ranks shard the
batch with no data-path collective (weak scaling: every rank runs the same per-GPU batch); --gather adds the
RCCL gather of the maps to rank 0 that north_star mentions.

Rank 0 prints ONE JSON line.  At N=1 it also carries
  roofline     : fp32-MFMA roofline of the dominant kernel family (conv_mfma_kernel relevance pass, 13 launches
                 per step = 30.69 GFLOP per map), timed live with HIP events on the launch stream
  cpu_baseline : the reference-equivalent CPU mode (oracle/ref_equiv.py, kind "port") on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GFLOP_PER_MAP = 30.69          # one transposed conv per VGG16 layer = 15.35 GMAC (algorithmic, fp32)
PEAK_FP32_MFMA_TF = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_BF16_MFMA_TF = 2500.0     # MI355X_MICROARCH.md: dense bf16 MFMA peak (no sparsity)
# The 12 relevance convs with >= 64 input channels run as "bf16x6": each fp32 product is evaluated as 6 bf16 MFMA
# products of exact operand thirds (fp32-accurate, csrc/conv_bf16x6.h); the 3-channel first layer is a VALU kernel.
# Executed matrix work = 6 x the algorithmic flops of those 12 layers (15.26 of the 15.35 GMAC).
BF16X6_SHARE = (15.35 - 0.087) / 15.35
# HBM traffic of one relevance pass over 320 maps, from rocprofv3 --pmc (separate passes, tools/pmc_passes.sh;
# profiles/r01_pmc_traffic_bf16x6.txt): FETCH_SIZE 37.4 GB raw (x2 for wide coalesced streams on gfx950, per
# MI355X_MICROARCH.md §HBM) + WRITE_SIZE 21.5 GB.  Scaled linearly with the map count below.
TRAFFIC_BYTES_PER_MAP = (2 * 37.4e9 + 21.5e9) / 320


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
    per_image = t_trace + t_words * T / len(words)       # trace once per image, word cost pro-rated
    return {"value": round(T / per_image, 4), "unit": "maps/s", "cores": cores, "kind": "port",
            "sample": f"1 image, T={T}, V={V}, words {words} explained by oracle/ref_equiv.py "
                      f"({t_words:.1f} s) + trace {t_trace:.1f} s, scaled to {T} words"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step (config 2)")
    ap.add_argument("--words", type=int, default=20)
    ap.add_argument("--vocab", type=int, default=9586)
    ap.add_argument("--gather", action="store_true", help="gather the maps to rank 0 over RCCL inside the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph")
    ap.add_argument("--fp32-mfma", action="store_true", help="keep every conv on the fp32 MFMA (disable bf16x6)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist = None
        torch.cuda.set_device(0)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    import lrp_amd  # noqa: F401
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    from lrp_amd import ops

    B, T, V = a.batch, a.words, a.vocab
    if a.fp32_mfma:
        from lrp_amd import _lib
        _lib.load().lrpx_set_bf16x6(0)
    torch.set_num_threads(min(8, host_cores()))
    if rank == 0:
        log(f"building weights + engine (B={B}, T={T}, V={V}, world={world})")
    eng = GridTDEngine(weights.make_gridtd_state(seed=0, vocab_size=V))
    # every rank gets its own shard of a global synthetic batch (seed offset by rank)
    images = torch.from_numpy(weights.make_images(100 + rank, B)).cuda()
    caps = torch.from_numpy(weights.make_captions(200 + rank, B, T, V)).cuda()
    gathered = None
    if a.gather and world > 1 and rank == 0:
        gathered = [torch.empty(B * T, 3, 224, 224, device="cuda") for _ in range(world)]

    state = {}

    def step(timed):
        if a.graph:
            maps, r_words = eng.explain_batch_graph(images, caps)
            if a.gather and world > 1:
                dist.gather(maps.view(B * T, 3, 224, 224), gathered, dst=0)
            return maps, r_words
        enc = eng.encode(images)
        tr = eng.trace(enc, caps, predictions=False)
        r_feat, r_words, row2img = eng.relevance(enc, tr)
        maps = eng.vgg.relevance(r_feat, row2img)
        state["chain_in"] = (r_feat, row2img)
        if a.gather and world > 1:
            dist.gather(maps, gathered, dst=0)
        return maps, r_words

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    barrier()
    if rank == 0:
        log("warm-up done, timing")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        maps, _ = step(True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = tmax.item()
    ops.check_relevance(maps, finite=True, nonzero=True)      # the reference's asserts, outside the timed region
    if rank == 0:
        log(f"timed region done: {dt / a.steps * 1e3:.1f} ms/step")

    if rank == 0:
        n_maps = world * B * T * a.steps
        out = {"metric": "LRP relevance maps/sec (VGG16+gridTD, 224x224, 20-token caption)",
               "value": round(n_maps / dt, 2), "unit": "maps/s", "n_gpus": world, "steps": a.steps,
               "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "BASELINE configs[1]: batch-16 224x224 images x 20-word captions per GPU, "
                                      "VGG16+gridTD, LRP alpha1beta0 (conv) + epsilon (decoder), V=9586, random-init",
                          "images_per_gpu": B, "words": T, "vocab": V, "maps_per_step": world * B * T,
                          "sharding": f"images x{world}, no data-path collective" + (" + gather" if a.gather else "")}}
        if world == 1:
            # roofline of the dominant kernel family: the 13 conv_mfma launches (+ pool / first-layer kernels between
            # them) of one relevance pass, timed live with HIP events on the launch stream, outside the headline timing
            if "chain_in" not in state:
                enc = eng.encode(images)
                state["chain_in"] = eng.relevance(enc, eng.trace(enc, caps, predictions=False))[::2]
            r_feat, row2img = state["chain_in"]
            ms = []
            for _ in range(max(3, a.steps)):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                eng.vgg.relevance(r_feat, row2img, out=maps.view(B * T, 3, 224, 224))
                e1.record()
                e1.synchronize()
                ms.append(e0.elapsed_time(e1))
            avg = sum(ms) / len(ms)
            tf = GFLOP_PER_MAP * B * T / avg          # GFLOP / ms = TFLOP/s (algorithmic, fp32-equivalent)
            from lrp_amd import _lib
            x6 = bool(_lib.load().lrpx_set_bf16x6(-1))
            if x6:
                exe = 6.0 * BF16X6_SHARE * tf         # bf16 flops actually issued to the matrix cores
                rf = {"bound": "mfma", "kernel": "conv_bf16x6_kernel relevance pass (12 launches/step; + first-layer "
                                                 "VALU kernel and 4 pool kernels inside the timed chain)",
                      "achieved": round(exe, 1), "peak": PEAK_BF16_MFMA_TF, "unit": "TFLOP/s",
                      "frac": round(exe / PEAK_BF16_MFMA_TF, 4), "mfma_dtype": "bf16 (exact 3-way split, 6 products, "
                      "fp32 accumulate)", "algorithmic_tflops_fp32": round(tf, 2),
                      "vs_fp32_mfma_peak": round(tf / PEAK_FP32_MFMA_TF, 3)}
            else:
                rf = {"bound": "mfma", "kernel": "conv_mfma_kernel relevance pass (13 launches/step)",
                      "achieved": round(tf, 2), "peak": PEAK_FP32_MFMA_TF, "unit": "TFLOP/s",
                      "frac": round(tf / PEAK_FP32_MFMA_TF, 4), "mfma_dtype": "f32"}
            out["roofline"] = {**rf,
                               "traffic": round(TRAFFIC_BYTES_PER_MAP * B * T),
                               "ms_per_step": round(avg, 3), "flop_per_step": GFLOP_PER_MAP * 1e9 * B * T}
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(V, T, [0, 5, 10, 15, 19])
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
