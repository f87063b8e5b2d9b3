#!/bin/bash
# round 3, GPU session b: whole GPU suite, bench line with the configs block, config 5 / phase times
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3b}
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py tests/test_gpu_hooks.py -q -m gpu -s -x > $OUT/new.log 2>&1; echo "new rc=$?"
grep -E "dense f16x3|generic add_lrp|passed|failed|Error|error" $OUT/new.log | head -30
LRPX_TIE_STATS=1 timeout -k 10 1100 python -m pytest tests -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?"
tail -15 $OUT/tests.log
timeout -k 10 500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
tail -3 $OUT/bench.err
python - <<PY
import json
d = json.load(open("$OUT/bench.json"))
print("headline", d["value"], d["ms_per_step"], "frac", d["roofline"]["frac"], "chain", d["roofline"]["chain"]["ms_per_step"])
for k, v in d.get("configs", {}).items():
    print("config", k, v["value"], v["ms_per_step"], v["roofline"].get("frac"), v.get("all_heads", {}).get("value"))
print(d.get("cpu_baseline"))
PY
timeout -k 10 200 python tools/phase_times.py > $OUT/phase.txt 2>&1; cat $OUT/phase.txt
exit 0
