#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3e
mkdir -p $OUT
cd $ROOT
i=0
for V in "A=1" "LRPX_FWD_KSPLIT28=1" "LRPX_FWD_KSPLIT=4 LRPX_FWD_KSPLIT28=1" "LRPX_FWD_KSPLIT=1 LRPX_FWD_KSPLIT28=1 LRPX_CONV11_F16=0" "LRPX_CONV11_F16=0"; do
  echo "== $V"
  env $V timeout -k 10 300 python tools/flip_probe.py 2 > $OUT/flips_$i.txt 2>&1; tail -22 $OUT/flips_$i.txt
  env $V timeout -k 10 300 python -m pytest tests/test_gpu_t20.py -q -m gpu -s -k "aoa_t20_rows_inside_b16 or forward_features or bottom_up" > $OUT/t20_$i.log 2>&1
  grep -E "T=20|forward features|passed|failed" $OUT/t20_$i.log
  i=$((i+1))
done
