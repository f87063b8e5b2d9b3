#!/bin/bash
# copy the evidence of one tools/r6_final.sh run (gpurun_out/<tag>/) into profiles/r06_*
R=gpurun_out/${1:-r06}
cp $R/bench.json profiles/r06_bench.json
hdr() { echo "# $1"; }
(hdr "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 (conv mode 1 = the default, 3 batches in flight); tools/r6_final.sh"; cat $R/kernel_stats_bench.txt) > profiles/r06_bench_kernel_stats.txt
(hdr "rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 2 --pipeline 1 (conv mode 1, SERIAL steps: 12 steps of 320 maps + one-time packing)"; cat $R/kernel_stats_serial.txt) > profiles/r06_bench_kernel_stats_serial.txt
(hdr "config 3 (AoA, B = 64), serial steps, conv mode 1"; cat $R/kernel_stats_config3_serial.txt) > profiles/r06_config3_kernel_stats_serial.txt
(hdr "config 5 (AoA bottom-up, B = 32), serial eager steps"; cat $R/kernel_stats_config5_serial.txt) > profiles/r06_config5_kernel_stats_serial.txt
(hdr "the drop-in's one-image pattern (tools/dbg/b1_phases.py 1: gridTD, B = 1, 20 words, conv mode 1; 24 calls): rocprofv3 --kernel-trace --stats"; cat $R/kernel_stats_dropin_b1.txt) > profiles/r06_dropin_b1_kernel_stats.txt
(hdr "phase table of the same pattern by HIP events, median of 20 calls (tools/dbg/b1_phases.py 1 3)"; cat $R/b1_phases.txt) > profiles/r06_dropin_b1_phases.txt
(hdr "rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum (separate passes, tools/pmc_passes.sh BC) over tools/bench_vgg.py --images 16 --maps 320 (conv mode 1): 2 forward + 2 relevance passes"; cat $R/pmc_traffic.txt) > profiles/r06_pmc_traffic.txt
cp $R/pmc_traffic.json profiles/r06_pmc_traffic.json
(hdr "matrix-pipe utilisation per kernel, conv mode 1 (tools/prof_summary.py pipe over the SQ passes A / E / F / G of tools/r6_final.sh): SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs against GRBM_GUI_ACTIVE / 8 XCDs, MFMA operations by type (1e9), LDS bank conflicts / LDS active cycles"; grep -E "^kernel|conv_|first_layer" $R/pmc_pipe.txt) > profiles/r06_pmc_sq_pipe.txt
(hdr "all SQ counters per kernel, per-launch means (tools/prof_summary.py sq), same passes as r06_pmc_sq_pipe.txt"; cat $R/pmc_sq.txt) > profiles/r06_pmc_sq_counters.txt
grep "^{" $R/bench_2rank_gloo_gather.json > profiles/r06_bench_2rank_gloo_gather_one_gpu.json
grep "^{" $R/bench_2rank_gloo_heatmap.json > profiles/r06_bench_2rank_gloo_heatmap_one_gpu.json
(hdr "python -m pytest tests -q -m gpu -s (LRPX_TIE_STATS=1), one MI355X box, round-6 final tree"; grep -E "^conv mode|^mode [0-9]|in-slice|T=20|passed|failed|opt-in|hostile" $R/tests.log | cut -c1-400) > profiles/r06_gpu_tests.txt
ls -la profiles/r06_*
