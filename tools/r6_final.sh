#!/bin/bash
# round-6 evidence run (conv mode 1 = the process default and the bench headline): tests, bench (+configs), rocprof kernel stats
# (pipelined + serial, config 3 serial, config 5 serial, the one-image drop-in), PMC traffic + SQ passes of the mode-1 chain
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r06}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
WHAT=${2:-tests bench prof pmc sq b1 smoke gloo}
if [[ $WHAT == *tests* ]]; then
  LRPX_TIE_STATS=1 timeout -k 10 1100 python -m pytest tests -q -m gpu -s > $OUT/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed" $OUT/tests.log | tail -2
fi
if [[ $WHAT == *bench* ]]; then
  timeout -k 10 700 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; cat $OUT/bench.json | head -c 700; echo
fi
prof() { # name, bench args...
  local name=$1; shift
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -- python3 $ROOT/bench.py "$@" --no-cpu-baseline --no-modes --no-configs --sustain 0 --no-live-traffic > $OUT/prof_$name.log 2>&1; echo "prof $name rc=$?")
  python tools/prof_summary.py stats $OUT/prof_$name > $OUT/kernel_stats_$name.txt 2>&1
  rm -rf $OUT/prof_$name/*/*.db
}
if [[ $WHAT == *prof* ]]; then
  prof bench --steps 10 --warmup 2
  prof serial --steps 10 --warmup 2 --pipeline 1
  prof config3_serial --config 3 --steps 4 --warmup 1 --pipeline 1
  prof config5_serial --config 5 --steps 20 --warmup 4 --pipeline 1 --replay 0
  head -24 $OUT/kernel_stats_serial.txt
fi
if [[ $WHAT == *b1* ]]; then
  timeout -k 10 200 python tools/dbg/b1_phases.py 1 3 2>&1 | grep -v amdgpu > $OUT/b1_phases.txt; cat $OUT/b1_phases.txt
  (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_b1 -- python3 $ROOT/tools/dbg/b1_phases.py 1 > $OUT/prof_b1.log 2>&1; echo "prof b1 rc=$?")
  python tools/prof_summary.py stats $OUT/prof_b1 > $OUT/kernel_stats_dropin_b1.txt 2>&1
  rm -rf $OUT/prof_b1/*/*.db
fi
if [[ $WHAT == *pmc* ]]; then
  tools/pmc_passes.sh $OUT/pmc 16 320 BC
  python tools/prof_summary.py traffic $OUT/pmc/B $OUT/pmc/C > $OUT/pmc_traffic.txt 2>&1
  python tools/prof_summary.py traffic-json 320 $OUT/pmc/B $OUT/pmc/C > $OUT/pmc_traffic.json 2>&1
  head -30 $OUT/pmc_traffic.txt
fi
if [[ $WHAT == *sq* ]]; then
  cd /tmp && export TMPDIR=/tmp
  run() { local name=$1; shift
    timeout -k 10 280 rocprofv3 --pmc "$@" --output-format csv -d $OUT/pmc/$name -- python3 $ROOT/tools/bench_vgg.py --images 16 --maps 320 --iters 1 > $OUT/pmc_$name.log 2>&1; echo "pass $name rc=$?"; }
  run A SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
  run E SQ_INSTS_MFMA SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VALU
  run F SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_F6F4 GRBM_GUI_ACTIVE
  run G GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL
  cd $ROOT
  python tools/prof_summary.py sq $OUT/pmc/A $OUT/pmc/E $OUT/pmc/F $OUT/pmc/G > $OUT/pmc_sq.txt 2>&1
  python tools/prof_summary.py pipe $OUT/pmc/A $OUT/pmc/E $OUT/pmc/F $OUT/pmc/G > $OUT/pmc_pipe.txt 2>&1; cat $OUT/pmc_pipe.txt
fi
if [[ $WHAT == *smoke* ]]; then
  timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
fi
if [[ $WHAT == *gloo* ]]; then
  LRPX_BENCH_BACKEND=gloo LRPX_BENCH_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --gather --steps 6 --warmup 2 --no-cpu-baseline --no-modes --no-configs --sustain 0 --pipeline 2 > $OUT/bench_2rank_gloo_gather.json 2> $OUT/bench_2rank_gloo_gather.err; echo "gloo gather rc=$?"; cat $OUT/bench_2rank_gloo_gather.json | head -c 400; echo
  LRPX_BENCH_BACKEND=gloo LRPX_BENCH_ONE_GPU=1 timeout -k 10 400 python bench.py --gpus 2 --gather heatmap --steps 6 --warmup 2 --no-cpu-baseline --no-modes --no-configs --sustain 0 --pipeline 2 > $OUT/bench_2rank_gloo_heatmap.json 2> $OUT/bench_2rank_gloo_heatmap.err; echo "gloo heatmap rc=$?"; cat $OUT/bench_2rank_gloo_heatmap.json | head -c 300; echo
fi
rm -rf $OUT/pmc/*/*/*.db
exit 0
