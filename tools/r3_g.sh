#!/bin/bash
# config 5 (graph replay / pipeline depth), t20 + switches + flips tests, matrix-pipe PMC passes of the relevance chain
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3g}
mkdir -p $OUT
cd $ROOT
for ARGS in "--config 5 --no-graph" "--config 5" "--config 5 --pipeline 2" "--config 5 --pipeline 4" "--config 5 --pipeline 6"; do
  timeout -k 10 200 python bench.py $ARGS --steps 60 --warmup 6 --sustain 2 > $OUT/c5.json 2> $OUT/c5.err
  python - <<PY
import json
d = json.load(open("$OUT/c5.json"))
print("$ARGS ->", d["value"], "maps/s", d["ms_per_step"], "ms/step; sustained", d["sustained"]["value"])
PY
done
LRPX_TIE_STATS=1 timeout -k 10 900 python -m pytest tests/test_gpu_t20.py tests/test_gpu_switches.py tests/test_gpu_vgg.py tests/test_gpu_aoa.py -q -m gpu -s > $OUT/tests.log 2>&1; echo "tests rc=$?"
grep -E "T=20|forward|flips|passed|failed|Error" $OUT/tests.log | head -40
# ---- PMC
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters_avail.txt 2>&1
grep -o "SQ_INSTS_VALU_MFMA_MOPS_[A-Z0-9]*\|SQ_VALU_MFMA_BUSY_CYCLES\|SQ_INSTS_MFMA" $OUT/counters_avail.txt | sort -u > $OUT/mfma_counter_names.txt
cat $OUT/mfma_counter_names.txt | tr '\n' ' '; echo
MOPS=$(grep MOPS $OUT/mfma_counter_names.txt | head -6 | tr '\n' ' ')
run() { local name=$1; shift
  timeout -k 10 280 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc/$name -- python3 $ROOT/tools/bench_vgg.py --images 16 --maps 320 --iters 1 > $OUT/pmc_$name.log 2>&1; echo "pass $name rc=$?"; }
run A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
run E SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU
[ -n "$MOPS" ] && run F $MOPS GRBM_GUI_ACTIVE
run G GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
cd $ROOT
python tools/prof_summary.py sq $OUT/pmc/A $OUT/pmc/E $OUT/pmc/F $OUT/pmc/G > $OUT/pmc_sq.txt 2>&1
head -80 $OUT/pmc_sq.txt
rm -rf $OUT/pmc/*/*/*.db
exit 0
