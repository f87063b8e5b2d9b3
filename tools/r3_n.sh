#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
V=$ROOT/lrp-imagecaptioning-pytorch_amd/csrc/variants
for L in "" lib_b.so lib_c.so lib_d.so "" lib_b.so lib_c.so; do
  if [ -z "$L" ]; then echo -n "default: "; timeout -k 10 100 python tools/fwd_only.py 16 40 | tail -1
  else echo -n "$L: "; LRPX_LIB_PATH=$V/$L timeout -k 10 100 python tools/fwd_only.py 16 40 | tail -1; fi
done
