#!/bin/bash
# A/B of bench.py argument sets on ONE box: tools/ab_args.sh "<args A>" "<args B>" ... (env from the caller), twice
for rep in 1 2; do
for v in "$@"; do
  out=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-modes $v 2>/dev/null)
  python - "$v" "$out" <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print("%-28s %8.1f maps/s (%.2f ms/step)  sustained %8.1f maps/s (%.2f ms)" % (sys.argv[1], d["value"], d["ms_per_step"],
      d["sustained"]["value"], d["sustained"]["ms_per_step"]))
PY
done
done
