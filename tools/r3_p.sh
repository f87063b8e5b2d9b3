#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
timeout -k 10 300 python -m pytest tests/test_gpu_dense.py -q -m gpu -s 2>&1 | grep -E "PLAIN|passed|failed|Error"
timeout -k 10 600 python -m pytest tests/test_gpu_gridtd.py tests/test_gpu_aoa.py tests/test_gpu_hooks.py -q -m gpu 2>&1 | tail -2
timeout -k 10 200 python tools/phase_times.py | tail -1
timeout -k 10 200 python bench.py --config 5 --steps 200 --warmup 20 --sustain 2 --no-configs > /tmp/c5.json 2> /tmp/c5.err
python -c "
import json; d=json.load(open('/tmp/c5.json')); print('config 5', d['value'], d['ms_per_step'], 'sustained', d['sustained']['value'])"
