#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch)."""
import csv, glob, re, sys, collections
root = sys.argv[1]
data = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/*/runc/*_counter_collection.csv") + glob.glob(f"{root}/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"void lrpx::|lrpx::", "", r["Kernel_Name"]); name = re.sub(r"\(.*", "", name)
        if "conv_mfma" not in name and "first_layer_rel" not in name and "maxpool_rel" not in name:
            continue
        data[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
cols = sorted({c for k in data.values() for c in k})
for name, cs in sorted(data.items()):
    n = max(len(v) for v in cs.values())
    print(f"\n{name}  (dispatches {n})")
    for c in cols:
        if c in cs:
            v = cs[c]
            print(f"   {c:28s} mean {sum(v)/len(v):16.1f}")
