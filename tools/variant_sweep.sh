#!/bin/bash
# On the GPU box: rebuild conv_inst_x6.o (56/28/14 relevance kernels) with different -D flags and time the chain.
# usage: tools/variant_sweep.sh "<flags1>" "<flags2>" ...
cd $(dirname $0)/..
for v in "$@"; do
  rm -f lrp-imagecaptioning-pytorch_amd/csrc/build/conv_inst_x6.o
  make -C lrp-imagecaptioning-pytorch_amd/csrc -j16 EXTRA="$v" > /tmp/make.log 2>&1 || { tail -5 /tmp/make.log; exit 1; }
  echo "== variant: $v"
  ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/vs && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/vs -- python3 $OLDPWD/tools/bench_vgg.py --images 16 --maps 320 --iters 2 > /dev/null 2>&1 )
  python3 - <<'PY'
import csv,glob,re
f=glob.glob('/tmp/vs/*/*_kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'bf16x6' in r['Name'] and ', 1>' in r['Name'] and ('<56' in r['Name'] or '<28' in r['Name'] or '<14' in r['Name']):
        n=re.sub(r"void lrpx::|lrpx::","",r['Name']); n=re.sub(r"\(.*","",n)
        print(f"   {n[:50]:50s} {float(r['AverageNs'])/1e3:10.1f} us")
PY
done
