#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
echo "== default (SCHED 1)"; timeout -k 10 100 python tools/dense_big_probe.py | head -3
for E in 0 2; do echo "== SCHED $E"; LRPX_LIB_PATH=$ROOT/lrp-imagecaptioning-pytorch_amd/csrc/variants/lib_s$E.so timeout -k 10 100 python tools/dense_big_probe.py | head -3; done
LRPX_LIB_PATH=$ROOT/lrp-imagecaptioning-pytorch_amd/csrc/variants/lib_s2.so timeout -k 10 200 python -m pytest tests/test_gpu_dense.py -q -m gpu 2>&1 | tail -2
