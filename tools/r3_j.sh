#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
for V in 0 1 2 4 8 15 0; do
  echo -n "LRPX_FWD_WIDE=$V: "; LRPX_FWD_WIDE=$V timeout -k 10 100 python tools/fwd_only.py 16 40 | tail -1
done
LRPX_FWD_WIDE=15 timeout -k 10 300 python -m pytest tests/test_gpu_switches.py -q -m gpu -k "FWD_WIDE" 2>&1 | tail -2
