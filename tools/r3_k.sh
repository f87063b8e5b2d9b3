#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3k}
mkdir -p $OUT
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py -q -m gpu -s > $OUT/dense.log 2>&1; echo "dense rc=$?"; grep -E "few rows|passed|failed|Error" $OUT/dense.log
timeout -k 10 900 python -m pytest tests/test_gpu_t20.py tests/test_gpu_gridtd.py tests/test_gpu_aoa.py tests/test_gpu_guided.py -q -m gpu -s > $OUT/tests.log 2>&1; echo "tests rc=$?"
grep -E "T=20|passed|failed|Error" $OUT/tests.log | head -20
timeout -k 10 200 python tools/phase_times.py | tail -1
timeout -k 10 200 python tools/phase_times.py --lockstep-fp32 | tail -1
timeout -k 10 200 python bench.py --config 5 --steps 200 --warmup 20 --sustain 2 --no-configs > $OUT/c5.json 2> $OUT/c5.err
python -c "
import json; d=json.load(open('$OUT/c5.json')); print('config 5', d['value'], d['ms_per_step'], 'sustained', d['sustained']['value'])"
timeout -k 10 300 python bench.py --no-configs --no-modes --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err
python -c "
import json; d=json.load(open('$OUT/bench.json')); print('headline', d['value'], d['ms_per_step'], 'sustained', d['sustained']['value'], 'frac', d['roofline']['frac'], 'chain', d['roofline']['chain']['ms_per_step'])"
