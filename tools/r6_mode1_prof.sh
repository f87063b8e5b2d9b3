#!/bin/bash
# round 6: rocprofv3 kernel stats of the serial headline step in conv mode 1 (bf16x6) + the bf16 MFMA power-limited peak
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r6_mode1}
MODE=${2:-1}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for m in 0 1 2; do tools/micro/mfma_bf16_peak $m 2; done > $OUT/mfma_bf16_peak.txt 2>&1
cat $OUT/mfma_bf16_peak.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -- python3 $ROOT/bench.py --conv-mode $MODE --steps 6 --warmup 2 --pipeline 1 --no-cpu-baseline --no-modes --no-configs --sustain 0 --no-live-traffic > $OUT/prof_serial.log 2>&1; echo "prof serial rc=$?")
python tools/prof_summary.py stats $OUT/prof_serial > $OUT/kernel_stats_serial.txt 2>&1
rm -rf $OUT/prof_serial/*/*.db
head -30 $OUT/kernel_stats_serial.txt
tail -2 $OUT/prof_serial.log | head -c 1500
