#!/bin/bash
# A/B of the relevance chain on ONE box: tools/ab_chain.sh "<env assignments A>" "<env assignments B>" ...
# prints chain ms + per-layer ms (HIP events of the library) for every variant, twice (boxes drift by ~1 %)
for rep in 1 2; do
for v in "$@"; do
  out=$(env $v timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-modes --no-configs --sustain 0 2>/dev/null)
  python - "$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
c = d["roofline"]["chain"]
print("%-28s step %.2f ms  chain %.3f ms  layers %s" % (sys.argv[1] or "(default)", d["ms_per_step"], c["ms_per_step"],
      " ".join("%s:%.2f" % kv for kv in c["conv_ms_by_layer"].items())))
PY
done
done
