#!/bin/bash
# On the GPU box: rebuild the f16+f8 instantiation files (conv_inst_h8*) with different -D flags and print the chain's
# per-layer conv times (bench.py --conv-mode 3, HIP events around every launch).
# usage: tools/variant_sweep3.sh "<flags1>" "<flags2>" ...
cd $(dirname $0)/..
for v in "$@"; do
  rm -f lrp-imagecaptioning-pytorch_amd/csrc/build/conv_inst_h8*.o lrp-imagecaptioning-pytorch_amd/csrc/build/conv_inst_h3*.o lrp-imagecaptioning-pytorch_amd/csrc/build/lrpx_vgg.o
  make -C lrp-imagecaptioning-pytorch_amd/csrc -j16 EXTRA="$v" > /tmp/make.log 2>&1 || { tail -5 /tmp/make.log; exit 1; }
  timeout -k 10 200 python3 bench.py --conv-mode 3 --no-cpu-baseline --steps 6 --warmup 2 2>/dev/null | grep metric > /tmp/b.json || exit 1
  python3 - "$v" <<'PY'
import json,sys
d=json.loads(open('/tmp/b.json').read()); r=d["roofline"]
print("== %-40s %8.1f maps/s chain %.2f ms  " % (sys.argv[1], d["value"], r["chain"]["ms_per_step"]) + " ".join("%s:%.2f" % kv for kv in r["chain"]["conv_ms_by_layer"].items()), flush=True)
PY
done
