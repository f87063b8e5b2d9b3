#!/bin/bash
# On the GPU box: rebuild the f16x3 instantiation files with different -D flags and time the chain's kernels.
# usage: tools/variant_sweep2.sh "<flags1>" "<flags2>" ...
cd $(dirname $0)/..
for v in "$@"; do
  rm -f lrp-imagecaptioning-pytorch_amd/csrc/build/conv_inst_h3*.o
  make -C lrp-imagecaptioning-pytorch_amd/csrc -j16 EXTRA="$v" > /tmp/make.log 2>&1 || { tail -5 /tmp/make.log; exit 1; }
  echo "== variant: $v"
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/vs && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vs -- python3 $OLDPWD/tools/bench_vgg.py --images 16 --maps 320 --iters 2 > /tmp/vs.log 2>&1; grep relevance /tmp/vs.log )
  python3 - <<'PY'
import csv,glob,re
f=glob.glob('/tmp/vs/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'f16x3' in r['Name']:
        n=re.sub(r"void lrpx::|lrpx::","",r['Name']); n=re.sub(r"\(.*","",n)
        print(f"   {n[:50]:50s} {float(r['AverageNs'])/1e3:10.1f} us")
PY
done
