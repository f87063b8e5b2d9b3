#!/bin/bash
# shader clock / power of the GPU while the VGG16 relevance chain runs in a given conv mode (rocm-smi samples every 0.4 s):
# evidence for the power-limited matrix-core rate behind DESIGN 5.1i.  usage: tools/dbg/clock_probe.sh <mode> [iters]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
MODE=${1:-1}; ITERS=${2:-160}
echo "== idle"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | head -4
python3 $ROOT/tools/bench_vgg.py --images 16 --maps 320 --iters $ITERS --mode $MODE > /tmp/clock_probe_bench.log 2>&1 &
PID=$!
sleep 6     # import + weight packing + the forward passes
for i in 1 2 3 4 5 6 7 8; do
  kill -0 $PID 2>/dev/null || break
  echo "== sample $i (conv mode $MODE chain running)"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" | head -3
  sleep 0.4
done
wait $PID; grep -E "relevance|forward" /tmp/clock_probe_bench.log
