#!/bin/bash
# round 3, first GPU session: MFMA rounding probe, new parity tests, forward K-split A/B
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r3a
mkdir -p $OUT
cd $ROOT
hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_round.hip -o $OUT/mfma_round > /dev/null 2>&1 && timeout -k 5 60 $OUT/mfma_round > $OUT/mfma_round.txt 2>&1
cat $OUT/mfma_round.txt
echo "--- t20 tests"
timeout -k 10 600 python -m pytest tests/test_gpu_t20.py -q -m gpu -s > $OUT/t20.log 2>&1; echo "t20 rc=$?"
grep -E "T=20|forward features|config-3|passed|failed|Error|assert" $OUT/t20.log | head -40
echo "--- hostile"
timeout -k 10 400 python -m pytest tests/test_gpu_vgg.py -q -m gpu -s -k "hostile" > $OUT/hostile.log 2>&1; echo "hostile rc=$?"
grep -E "hostile|passed|failed|Error" $OUT/hostile.log | head -30
echo "--- forward probe (K split default / off)"
timeout -k 10 200 python tools/fwd_probe.py > $OUT/fwd_default.txt 2>&1; cat $OUT/fwd_default.txt
LRPX_FWD_KSPLIT=4 LRPX_FWD_KSPLIT28=1 timeout -k 10 200 python tools/fwd_probe.py > $OUT/fwd_r2.txt 2>&1; cat $OUT/fwd_r2.txt
LRPX_FWD_KSPLIT=1 LRPX_FWD_KSPLIT28=1 timeout -k 10 200 python tools/fwd_probe.py > $OUT/fwd_nosplit.txt 2>&1; cat $OUT/fwd_nosplit.txt
timeout -k 10 200 python tools/phase_times.py > $OUT/phase.txt 2>&1; cat $OUT/phase.txt
exit 0
