#!/bin/bash
# SQ counter passes (as tools/r4_final.sh's "sq" leg) over tools/dense_big_probe.py: matrix-pipe busy and LDS conflicts of the dense kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-pmc_dense}; mkdir -p $OUT/pmc
cd /tmp && export TMPDIR=/tmp
run() { local name=$1; shift
  timeout -k 10 280 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc/$name -- python3 $ROOT/tools/dense_big_probe.py > $OUT/pmc_$name.log 2>&1; echo "pass $name rc=$?"; }
run A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run E SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU
run F SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F8 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F6F4 GRBM_GUI_ACTIVE
run G GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
cd $ROOT
python tools/prof_summary.py pipe $OUT/pmc/A $OUT/pmc/E $OUT/pmc/F $OUT/pmc/G > $OUT/pmc_pipe.txt 2>&1; cat $OUT/pmc_pipe.txt
python tools/prof_summary.py sq $OUT/pmc/A $OUT/pmc/E $OUT/pmc/F $OUT/pmc/G > $OUT/pmc_sq.txt 2>&1
