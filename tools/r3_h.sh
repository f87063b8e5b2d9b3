#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3h}
mkdir -p $OUT
cd $ROOT
for ARGS in "--config 5 --pipeline 3" "--config 5 --pipeline 4" "--config 5 --pipeline 6" "--config 5 --pipeline 8"; do
  timeout -k 10 200 python bench.py $ARGS --steps 200 --warmup 20 --sustain 2 > $OUT/c5.json 2> $OUT/c5.err
  python - <<PY
import json
d = json.load(open("$OUT/c5.json"))
print("$ARGS ->", d["value"], "maps/s", d["ms_per_step"], "ms/step; sustained", d["sustained"]["value"])
PY
done
timeout -k 10 100 python tools/fwd_only.py 16 30 | tail -1
timeout -k 10 200 python tools/phase_times.py | tail -1
