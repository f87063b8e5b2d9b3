#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3m}
mkdir -p $OUT
cd $ROOT
for ARGS in "--chain-streams 1" "--chain-streams 2" "--chain-streams 1 --pipeline 2" "--chain-streams 2 --pipeline 2" "--chain-streams 2 --pipeline 1" "--chain-streams 1"; do
  timeout -k 10 200 python bench.py $ARGS --steps 30 --warmup 6 --sustain 3 --no-configs --no-modes --no-cpu-baseline > $OUT/b.json 2> $OUT/b.err
  python -c "
import json; d=json.load(open('$OUT/b.json')); print('$ARGS ->', d['value'], d['ms_per_step'], 'sustained', d['sustained']['value'])"
done
