#!/bin/bash
# A/B of the whole pipelined step on ONE box: tools/ab_bench.sh "<env A>" "<env B>" ... ; 20 steps + 5 s sustained, twice
for rep in 1 2; do
for v in "$@"; do
  out=$(env $v timeout -k 10 300 python bench.py --no-cpu-baseline --no-modes --no-configs 2>/dev/null)
  python - "$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print("%-28s %8.1f maps/s (%.2f ms/step)  sustained %8.1f maps/s (%.2f ms)  chain %.2f ms" % (sys.argv[1], d["value"], d["ms_per_step"],
      d["sustained"]["value"], d["sustained"]["ms_per_step"], d["roofline"]["chain"]["ms_per_step"]))
PY
done
done
