#!/bin/bash
# forward A/B (conv1_1 f16x3 on/off) + forward kernel stats
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/${1:-r3c}
mkdir -p $OUT
cd $ROOT
timeout -k 10 100 python tools/fwd_only.py 16 30 > $OUT/fwd16.txt 2>&1; cat $OUT/fwd16.txt | tail -1
LRPX_CONV11_F16=0 timeout -k 10 100 python tools/fwd_only.py 16 30 > $OUT/fwd16_c11fp32.txt 2>&1; tail -1 $OUT/fwd16_c11fp32.txt
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_fwd -- python3 $ROOT/tools/fwd_only.py 16 30 > $OUT/prof_fwd.log 2>&1; echo "prof rc=$?")
python tools/prof_summary.py stats $OUT/prof_fwd > $OUT/fwd_kernel_stats.txt 2>&1; head -40 $OUT/fwd_kernel_stats.txt
rm -rf $OUT/prof_fwd/*/*.db
