#!/bin/bash
# VGPRs / spills / scratch of every kernel in the given instantiation files (csrc/conv_inst_<name>.hip ...)
# usage: tools/kernel_regs.sh h8 h8w ...   [EXTRA="-D..."]
cd "$(dirname "$0")/../lrp-imagecaptioning-pytorch_amd/csrc"
for f in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function $EXTRA -c conv_inst_$f.hip -o /tmp/kr_$$.o \
      -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys
cur = {}
for line in sys.stdin:
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1); cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            n = re.sub(r"_ZN4lrpx17conv_f16x3_kernelI|EEvNS_8ConvArgsEii", "", cur["name"]).replace("Li", "").replace("Lb", "").replace("E", ",")
            print("%-6s %-40s vgpr %-4s agpr %-4s spill %-4s scratch %-5s occ %s" % (sys.argv[1], n[:40], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("VGPRs Spill"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]")))
' "$f"
done
rm -f /tmp/kr_$$.o
