#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3d
mkdir -p $OUT
cd $ROOT
for V in "A=1" "LRPX_CONV11_F16=0" "LRPX_FWD_KSPLIT28=1" "LRPX_FWD_KSPLIT=4 LRPX_FWD_KSPLIT28=1 LRPX_CONV11_F16=0" "LRPX_FWD_KSPLIT=1 LRPX_FWD_KSPLIT28=1 LRPX_CONV11_F16=0"; do
  echo "== $V"
  env $V timeout -k 10 200 python -m pytest tests/test_gpu_gradient.py tests/test_gpu_aoa_gradient.py -q -m gpu -k "decoder_and_maps" > $OUT/log.txt 2>&1
  tail -1 $OUT/log.txt
  python - <<PY
import json
for r in json.load(open("gpurun_out/pool_tie_stats.json")):
    print("  ", r["test"].split("::")[1][:40], r["what"], "frac %.4f max %.2e l2 %.2e cos %.7f" % (r["frac_gt_1e-4"], r["max"], r["rel_l2"], r["cos"]))
PY
done
