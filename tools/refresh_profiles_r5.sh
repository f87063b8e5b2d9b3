#!/bin/bash
# copy the evidence of one tools/r5_final.sh run (gpurun_out/<tag>/) into profiles/r05_*
R=gpurun_out/${1:-r05}
cp $R/bench.json profiles/r05_bench.json
cp $R/kernel_stats_bench.txt profiles/r05_bench_kernel_stats.txt
cp $R/kernel_stats_serial.txt profiles/r05_bench_kernel_stats_serial.txt
cp $R/kernel_stats_config3_serial.txt profiles/r05_config3_kernel_stats_serial.txt
cp $R/kernel_stats_config5_serial.txt profiles/r05_config5_kernel_stats_serial.txt
cp $R/pmc_traffic.txt profiles/r05_pmc_traffic.txt
cp $R/pmc_traffic.json profiles/r05_pmc_traffic.json
(head -3 profiles/r04_pmc_sq_pipe.txt | sed "s/r4_final/r5_final/"; grep -E "^kernel|conv_|first_layer" $R/pmc_pipe.txt) > /tmp/pipe.txt && mv /tmp/pipe.txt profiles/r05_pmc_sq_pipe.txt
(echo "# all SQ counters per kernel, per-launch means (tools/prof_summary.py sq), same passes as r05_pmc_sq_pipe.txt"; cat $R/pmc_sq.txt) > profiles/r05_pmc_sq_counters.txt
grep -E "passed|failed" $R/tests.log | tail -1 > profiles/r05_gpu_tests.txt
