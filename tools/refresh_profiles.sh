#!/bin/bash
# copy the evidence of one tools/r4_final.sh run (gpurun_out/<tag>/) into profiles/r04_*
R=gpurun_out/${1:-r04}
cp $R/bench.json profiles/r04_bench.json
cp $R/kernel_stats_bench.txt profiles/r04_bench_kernel_stats.txt
cp $R/kernel_stats_serial.txt profiles/r04_bench_kernel_stats_serial.txt
cp $R/kernel_stats_config3_serial.txt profiles/r04_config3_kernel_stats_serial.txt
cp $R/kernel_stats_config5_serial.txt profiles/r04_config5_kernel_stats_serial.txt
cp $R/pmc_traffic.txt profiles/r04_pmc_traffic.txt
cp $R/pmc_traffic.json profiles/r04_pmc_traffic.json
(head -3 profiles/r03_pmc_sq_pipe.txt | sed "s/r3_final/r4_final/"; grep -E "^kernel|conv_|first_layer" $R/pmc_pipe.txt) > /tmp/pipe.txt && mv /tmp/pipe.txt profiles/r04_pmc_sq_pipe.txt
(echo "# all SQ counters per kernel, per-launch means (tools/prof_summary.py sq), same passes as r04_pmc_sq_pipe.txt"; cat $R/pmc_sq.txt) > profiles/r04_pmc_sq_counters.txt
grep "^{" $R/bench_2rank_gloo_gather.json > profiles/r04_bench_2rank_gloo_gather_one_gpu.json
