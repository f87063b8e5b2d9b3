#!/bin/bash
# One GPU-box session: full GPU test suite, headline bench, rocprofv3 kernel stats of the bench, PMC traffic passes.
# usage: tools/gpu_round.sh <tag> [tests|bench|prof|pmc ...]   (default: all four); outputs under gpurun_out/<tag>/
TAG=$1; shift
WHAT=${*:-tests bench prof pmc}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
if [[ $WHAT == *tests* ]]; then
  timeout -k 10 900 python -m pytest tests -q -m gpu -x > $OUT/tests.log 2>&1; echo "tests rc=$?" | tee -a $OUT/tests.log
  tail -5 $OUT/tests.log
fi
if [[ $WHAT == *bench* ]]; then
  timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
  cat $OUT/bench.json
fi
if [[ $WHAT == *prof* ]]; then
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-modes --sustain 0 > $OUT/prof.log 2>&1; echo "prof rc=$?")
  python tools/prof_summary.py stats $OUT/prof > $OUT/kernel_stats.txt 2>&1; head -30 $OUT/kernel_stats.txt
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_serial -- python3 $ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-modes --sustain 0 --pipeline 1 > $OUT/prof_serial.log 2>&1; echo "prof serial rc=$?")
  python tools/prof_summary.py stats $OUT/prof_serial > $OUT/kernel_stats_serial.txt 2>&1
  rm -rf $OUT/prof/*/*.db $OUT/prof_serial/*/*.db
fi
if [[ $WHAT == *pmc* ]]; then
  tools/pmc_passes.sh $OUT/pmc 16 320 BC
  python tools/prof_summary.py traffic $OUT/pmc/B $OUT/pmc/C > $OUT/pmc_traffic.txt 2>&1
  python tools/prof_summary.py traffic-json 320 $OUT/pmc/B $OUT/pmc/C > $OUT/pmc_traffic.json 2>&1
  head -50 $OUT/pmc_traffic.txt
fi
exit 0
