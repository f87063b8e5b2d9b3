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
  roofline     : MFMA roofline of the dominant kernel (the relevance conv kernel with the largest total time of the
                 12 conv launches per step = 30.69 GFLOP per map), timed live with HIP events on the launch stream
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
PEAK_16BIT_MFMA_TF = 2500.0    # MI355X_MICROARCH.md: dense bf16 / fp16 MFMA peak (no sparsity)
# The dominant kernel: the relevance step of a 256->256 conv on 56x56 maps (conv3_x) or of a 512->512 conv on 28x28 maps
# (conv4_x) - the same work; the three conv3 and three conv4 launches are 55 % of the chain, and in the default mode the
# 28x28 kernel has the largest total time (3 launches).  Algorithmic work per launch and map: one transposed 3x3 conv =
# 2*9*256*256*56*56 = 2*9*512*512*28*28 flop.  The matrix cores execute PRODUCTS[mode] 16-bit MFMA products per fp32 product (operand
# splits, csrc/conv_f16x3.h / conv_bf16x6.h), so `achieved` counts executed MFMA flop against the 16-bit dense peak;
# `algorithmic_tflops` is the fp32-equivalent rate.
DOM_FLOP_PER_MAP = 2.0 * 9 * 256 * 256 * 56 * 56
MODE_NAME = {0: "fp32 MFMA (v_mfma_f32_32x32x2_f32)", 1: "bf16x6: exact 3-way bf16 split, 6 products, fp32 accumulate",
             2: "f16x3: per-map power-of-two scaling, 2-way fp16 split, 3 products, fp32 accumulate",
             3: "f16+f8x2: as f16x3, the two cross products (2^-11 of the result) as fp8 e4m3 MFMAs (v_mfma_f32_32x32x64_f8f6f4)"}
MODE_KERNEL = {0: "conv_mfma_kernel<56,16,1,4,9,REL>", 1: "conv_bf16x6_kernel<56,1,4,true,REL>",
               2: "conv_f16x3_kernel<56,1,4,true,REL_MUL,false>", 3: "conv_f16x3_kernel<28,1,4,true,REL_MUL,false,true>"}
# mode 3: one fp16 product + two fp8 products; the fp8 dense peak is twice the fp16 one, so an fp8 flop counts half:
# `achieved` / `peak` is then (time the matrix cores need at their peaks) / (measured time), as in the other modes
PRODUCTS = {0: 1, 1: 6, 2: 3, 3: 2}
# arithmetic the contractions run in: all tensors are fp32; modes 1 / 2 evaluate each fp32 product as split 16-bit MFMA
# products with fp32 accumulation (fp32-grade results), everything else is fp32 VALU
MODE_DTYPE = {0: "f32", 1: "f32 (bf16x6 split-product MFMA, f32 accumulate)", 2: "f32 (f16x3 split-product MFMA, f32 accumulate)",
              3: "f32 (fp16 + 2 fp8 split-product MFMA, f32 accumulate)"}
# HBM traffic of ONE launch of that kernel over 320 maps, from rocprofv3 --pmc (separate passes, tools/pmc_passes.sh;
# profiles/r01_pmc_traffic_f16x3.txt, ..._bf16x6.txt, r01_pmc_traffic.txt): (2 x FETCH_SIZE raw [gfx950 reports half of wide streaming reads,
# MI355X_MICROARCH.md §HBM] + WRITE_SIZE) per launch.  Scaled linearly with the map count.
DOM_TRAFFIC_BYTES_PER_MAP = {0: (2 * 1.9e9 + 0.86e9) / 320, 1: (2 * 1.55e9 + 0.86e9) / 320, 2: (2 * 1.22e9 + 0.77e9) / 320,
                             3: (2 * 5.72e9 + 2.80e9) / 6 / 320}   # 28x28 kernel, 6 launches: profiles/r01_pmc_traffic_f16f8.txt


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
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=4)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU per step (config 2)")
    ap.add_argument("--words", type=int, default=20)
    ap.add_argument("--vocab", type=int, default=9586)
    ap.add_argument("--gather", action="store_true", help="gather the maps to rank 0 over RCCL inside the step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph")
    ap.add_argument("--pipeline", type=int, default=3,
                    help="independent batches in flight on separate HIP streams (1 = serial steps); every step is still "
                         "one full pass over one batch, the decoder's latency-bound kernels of one batch overlap the "
                         "MFMA-bound CNN chain of another")
    ap.add_argument("--fp32-mfma", action="store_true", help="same as --conv-mode 0")
    ap.add_argument("--conv-mode", type=int, default=3, choices=[0, 1, 2, 3],
                    help="matrix-core mode of the VGG16 chains: 0 fp32 MFMA, 1 bf16x6, 2 f16x3 relevance (default)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
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
        torch.cuda.set_device(0)
    assert a.gpus == world, f"--gpus {a.gpus} but WORLD_SIZE={world}"

    import lrp_amd  # noqa: F401
    from lrp_amd import weights
    from lrp_amd.explainers.gridtd import GridTDEngine
    from lrp_amd import ops

    B, T, V = a.batch, a.words, a.vocab
    from lrp_amd import _lib
    mode = 0 if a.fp32_mfma else a.conv_mode
    _lib.load().lrpx_set_conv_mode(mode)
    torch.set_num_threads(min(8, host_cores()))
    if rank == 0:
        log(f"building weights + engine (B={B}, T={T}, V={V}, world={world})")
    sd0 = weights.make_gridtd_state(seed=0, vocab_size=V)
    eng = GridTDEngine(sd0)
    n_pipe = 1 if a.graph else max(1, a.pipeline)
    engines = [eng] + [eng.replica() for _ in range(n_pipe - 1)]     # shared weights, own trace / workspace buffers
    streams = [torch.cuda.Stream() for _ in range(n_pipe)] if n_pipe > 1 else [None]
    # every rank gets its own shard of a global synthetic batch (seed offset by rank)
    images = torch.from_numpy(weights.make_images(100 + rank, B)).cuda()
    caps = torch.from_numpy(weights.make_captions(200 + rank, B, T, V)).cuda()
    gathered = None
    if a.gather and world > 1 and rank == 0:
        gathered = [torch.empty(B * T, 3, 224, 224, device="cuda") for _ in range(world)]

    state = {}

    step_no = [0]

    def step(timed):
        if n_pipe > 1:
            k = step_no[0] % n_pipe
            step_no[0] += 1
            with torch.cuda.stream(streams[k]):
                e = engines[k]
                enc = e.encode(images)
                tr = e.trace(enc, caps, predictions=False)
                r_feat, r_words, row2img = e.relevance(enc, tr)
                if "maps%d" % k not in state:
                    state["maps%d" % k] = torch.empty(B * T, 3, 224, 224, device="cuda")
                maps = e.vgg.relevance(r_feat, row2img, out=state["maps%d" % k])
                if k == 0:
                    state["chain_in"] = (r_feat, row2img)
                if a.gather and world > 1:
                    dist.gather(maps, gathered, dst=0)
            return maps, r_words
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
               "scaling": "weak", "vs_baseline": None, "dtype": MODE_DTYPE[mode], "data": "synthetic",
               "config": {"workload": "BASELINE configs[1]: batch-16 224x224 images x 20-word captions per GPU, "
                                      "VGG16+gridTD, LRP alpha1beta0 (conv) + epsilon (decoder), V=9586, random-init",
                          "images_per_gpu": B, "words": T, "vocab": V, "maps_per_step": world * B * T,
                          "sharding": f"images x{world}, no data-path collective" + (" + gather" if a.gather else ""),
                          "batches_in_flight": n_pipe}}
        if world == 1:
            # (1) the whole VGG16 relevance chain (12 conv launches + first-layer kernel + 4 pool kernels), HIP events
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
            chain_ms = sum(ms) / len(ms)
            # (2) roofline of the dominant kernel (default mode: the 28x28 relevance conv kernel, 3 launches per pass: conv4_3,
            # conv4_2 with 512 and conv4_1 with 256 output channels), HIP events recorded by the library on the launch
            # stream around every conv launch of the same chain on the same inputs
            import ctypes as C
            lib = _lib.load()
            per_layer = [0.0] * 17
            reps = max(3, a.steps)
            lib.lrpx_vgg16_layer_timing(1, None)
            buf = (C.c_float * 17)()
            for _ in range(reps):
                eng.vgg.relevance(r_feat, row2img, out=maps.view(B * T, 3, 224, 224))
                lib.lrpx_vgg16_layer_timing(-1, buf)
                per_layer = [p + float(v) for p, v in zip(per_layer, buf)]
            lib.lrpx_vgg16_layer_timing(0, None)
            per_layer = [p / reps for p in per_layer]
            # launches of that kernel NAME per pass: conv3_1 (128 output channels) and conv3_2 (256); in mode 2 conv3_3
            # is the pooled-input variant of the kernel (own name in rocprof), in modes 0/1 it is the same kernel
            # mode 3: the kernel with the largest total time is the 28x28 one (conv4_1 with 256 output channels, conv4_2,
            # conv4_3 behind the unpool scatter): same flop per full launch as the 56x56 layers (2*9*512*512*28*28)
            dom_layers, dom_w = ([10, 11, 12], [0.5, 1.0, 1.0]) if mode == 3 else (([6, 7], [0.5, 1.0]) if mode == 2 else ([6, 7, 8], [0.5, 1.0, 1.0]))
            dom_desc = ("conv4_1/conv4_2/conv4_3 on 28x28 maps" if mode == 3 else
                        ("conv3_1/conv3_2 on 56x56 maps" if mode == 2 else "conv3_1/conv3_2/conv3_3 on 56x56 maps"))
            dom_ms = sum(per_layer[l] for l in dom_layers) / len(dom_layers)          # average launch of that kernel
            flop = DOM_FLOP_PER_MAP * B * T * sum(dom_w) / len(dom_w)                 # average algorithmic flop / launch
            alg = flop / dom_ms / 1e9                                        # TFLOP/s, fp32-equivalent
            exe = PRODUCTS[mode] * alg
            peak = PEAK_FP32_MFMA_TF if mode == 0 else PEAK_16BIT_MFMA_TF
            out["roofline"] = {
                "bound": "mfma", "kernel": MODE_KERNEL[mode] + f" (relevance step of {dom_desc}, {B * T} maps per launch, {len(dom_layers)} launches per step)", "achieved": round(exe, 1), "peak": peak,
                "unit": "TFLOP/s", "frac": round(exe / peak, 4), "mfma_dtype": MODE_NAME[mode],
                "mfma_products_per_fp32_product": PRODUCTS[mode], "algorithmic_tflops": round(alg, 1),
                "ms_per_launch": round(dom_ms, 4), "flop_per_launch": flop,
                "traffic": round(DOM_TRAFFIC_BYTES_PER_MAP[mode] * B * T),
                "accounting": ("executed matrix flop = algorithmic x products; mode 3 executes 1 fp16 + 2 fp8 products per fp32 "
                               "product and an fp8 flop counts 1/2 (fp8 dense peak = 2 x fp16 peak), so achieved/peak = matrix "
                               "time at peak / measured time; the same layer with 3 fp16 products (--conv-mode 2) reaches "
                               "frac 0.55-0.57 on 1.5x the matrix work and is 1.19x slower") if mode == 3 else
                              "executed matrix flop = algorithmic x products",
                "chain": {"ms_per_step": round(chain_ms, 3), "flop_per_step": GFLOP_PER_MAP * 1e9 * B * T,
                          "algorithmic_tflops": round(GFLOP_PER_MAP * B * T / chain_ms, 1),
                          "conv_ms_by_layer": {str(l): round(v, 3) for l, v in enumerate(per_layer) if v > 0}}}
            if not a.no_cpu_baseline:
                out["cpu_baseline"] = cpu_baseline(V, T, sorted({round(i * (T - 1) / 9) for i in range(10)}))   # 10 words, mean index (T-1)/2
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
