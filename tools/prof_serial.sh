#!/bin/bash
# rocprofv3 kernel statistics of the serial bench step (one batch in flight): tools/prof_serial.sh <tag> [env assignments]
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
for v in "$@"; do export $v; done
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-modes --sustain 0 --pipeline 1 > $OUT/prof_serial.log 2>&1; echo "prof serial rc=$?")
python $ROOT/tools/prof_summary.py stats $OUT/prof_serial > $OUT/kernel_stats_serial.txt 2>&1
rm -rf $OUT/prof_serial/*/*.db
head -40 $OUT/kernel_stats_serial.txt
