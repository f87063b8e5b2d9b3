#!/bin/bash
# PMC passes for the VGG relevance chain (separate passes: kernel-trace/stats are never combined with --pmc).
# usage (on the GPU box): tools/pmc_passes.sh <outdir> [images] [maps] [passes]
set -e
OUT=$1; IMG=${2:-4}; MAPS=${3:-80}; PASSES=${4:-ABCD}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # name counters...
  local name=$1; shift
  timeout -k 10 280 rocprofv3 --pmc "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/bench_vgg.py --images $IMG --maps $MAPS --iters 1 > $OUT/$name.log 2>&1
  echo "pass $name done"
}
[[ $PASSES == *A* ]] && run A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
[[ $PASSES == *B* ]] && run B FETCH_SIZE GRBM_GUI_ACTIVE
[[ $PASSES == *C* ]] && run C WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
[[ $PASSES == *D* ]] && run D SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_UNALIGNED_STALL SQ_INSTS_VALU
[[ $PASSES == *E* ]] && run E SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU
exit 0
