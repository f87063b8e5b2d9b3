#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3i}
mkdir -p $OUT
cd $ROOT
echo skip tests
for ARGS in "--config 5 --pipeline 3" "--config 5 --pipeline 3 --one-host-thread" "--config 5 --pipeline 4" "--config 5 --pipeline 6" "--config 5 --pipeline 8" "--config 5 --graph --pipeline 3"; do
  timeout -k 10 200 python bench.py $ARGS --steps 200 --warmup 20 --sustain 2 > $OUT/c5.json 2> $OUT/c5.err
  python - <<PY
import json
d = json.load(open("$OUT/c5.json"))
print("$ARGS ->", d["value"], "maps/s", d["ms_per_step"], "ms/step; sustained", d["sustained"]["value"])
PY
done
