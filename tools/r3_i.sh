#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3i}
mkdir -p $OUT
cd $ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_aoa.py tests/test_gpu_aoa_gradient.py tests/test_gpu_t20.py -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -3 $OUT/tests.log
for ARGS in "--config 5 --pipeline 3" "--config 5 --pipeline 2" "--config 5 --pipeline 4" "--config 5 --pipeline 1"; do
  timeout -k 10 200 python bench.py $ARGS --steps 200 --warmup 20 --sustain 2 > $OUT/c5.json 2> $OUT/c5.err
  python - <<PY
import json
d = json.load(open("$OUT/c5.json"))
print("$ARGS ->", d["value"], "maps/s", d["ms_per_step"], "ms/step; sustained", d["sustained"]["value"])
PY
done
