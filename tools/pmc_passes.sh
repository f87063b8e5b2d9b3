#!/bin/bash
# PMC passes for the VGG relevance chain (separate passes: kernel-trace/stats are never combined with --pmc).
# usage (on the GPU box): tools/pmc_passes.sh <outdir> [images] [maps]
set -e
OUT=$1; IMG=${2:-4}; MAPS=${3:-80}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout -k 10 280 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/bench_vgg.py --images $IMG --maps $MAPS --iters 1 > $OUT/$name.log 2>&1
  echo "pass $name done"
}
run A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run B FETCH_SIZE GRBM_GUI_ACTIVE
run C WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run D SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU
