#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3l}
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py -q -m gpu > $OUT/dense.log 2>&1; echo "dense rc=$?"; tail -1 $OUT/dense.log
timeout -k 10 200 python tools/phase_times.py | tail -1
timeout -k 10 200 python tools/phase_times.py --lockstep-fp32 | tail -1
timeout -k 10 200 python bench.py --config 5 --steps 200 --warmup 20 --sustain 2 --no-configs > $OUT/c5.json 2> $OUT/c5.err
python -c "
import json; d=json.load(open('$OUT/c5.json')); print('config 5', d['value'], d['ms_per_step'], 'sustained', d['sustained']['value'])"
for S in 1 2 3; do echo -n "chain streams=$S: "; timeout -k 10 200 python tools/bench_vgg.py --images 16 --maps 320 --iters 5 --streams $S 2>&1 | tail -1; done
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $ROOT/tools/phase_times.py > $OUT/prof.log 2>&1)
python tools/prof_summary.py stats $OUT/prof 2>&1 | grep -E "dense|linear" 
rm -rf $OUT/prof/*/*.db
