#!/bin/bash
# round-4 working run: [tests] GPU test suite, [ab] chain A/B of the in-tree library against csrc/variants/liblrpx_base.so on the same box
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-r4x}; WHAT=${2:-tests ab}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p $OUT; cd $ROOT
if [[ $WHAT == *tests* ]]; then
  timeout -k 10 900 python -m pytest tests -q -m gpu -x > $OUT/tests.log 2>&1; echo "tests rc=$?"; tail -15 $OUT/tests.log
fi
if [[ $WHAT == *vgg* ]]; then
  timeout -k 10 600 python -m pytest tests/test_gpu_vgg.py -q -m gpu -x > $OUT/tests_vgg.log 2>&1; echo "vgg tests rc=$?"; tail -25 $OUT/tests_vgg.log
fi
if [[ $WHAT == *ab* ]]; then
  tools/ab_chain.sh "LRPX_LIB_PATH=$ROOT/lrp-imagecaptioning-pytorch_amd/csrc/variants/liblrpx_base.so" "" > $OUT/ab_chain.txt 2>&1; cat $OUT/ab_chain.txt
fi
if [[ $WHAT == *bench* ]]; then
  tools/ab_bench.sh "LRPX_LIB_PATH=$ROOT/lrp-imagecaptioning-pytorch_amd/csrc/variants/liblrpx_base.so" "" > $OUT/ab_bench.txt 2>&1; cat $OUT/ab_bench.txt
fi
exit 0
