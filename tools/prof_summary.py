#!/usr/bin/env python3
"""Summaries of rocprofv3 output directories for profiles/:
   prof_summary.py stats <dir>            kernel table (calls, total ms, avg us, %) from *_kernel_stats.csv
   prof_summary.py traffic <dirB> <dirC>  per kernel FETCH_SIZE / WRITE_SIZE sums (GB) and L2 hit rate from two --pmc passes
   prof_summary.py traffic-json <maps_per_launch> <dirB> <dirC>   the same as JSON (read by bench.py for roofline.traffic)
   prof_summary.py sq <dir> [<dir>...]    per kernel means of every SQ counter found (per launch)
   prof_summary.py pipe <dir> [<dir>...]  matrix-pipe utilisation per kernel (MFMA busy cycles / GPU cycles), MFMA op mix"""
import collections, csv, glob, os, re, sys


def newest(pattern):
    fs = glob.glob(pattern, recursive=True)
    if not fs:
        sys.exit(f"no file matches {pattern}")
    return max(fs, key=os.path.getmtime)


def short(name):
    n = re.sub(r"void lrpx::|lrpx::", "", name)
    return re.sub(r"\(.*", "", n)


def stats(d):
    rows = list(csv.DictReader(open(newest(f"{d}/**/*_kernel_stats.csv"))))
    print(f"{'kernel':70s} {'calls':>6s} {'total ms':>10s} {'avg us':>10s} {'%':>6s}")
    for r in rows:
        print(f"{short(r['Name'])[:70]:70s} {int(r['Calls']):6d} {float(r['TotalDurationNs'])/1e6:10.2f} "
              f"{float(r['AverageNs'])/1e3:10.1f} {float(r['Percentage']):6.2f}")


def counters(dirs):
    data = collections.defaultdict(lambda: collections.defaultdict(list))
    for d in dirs:
        for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                data[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return data


def traffic(dirs):
    data = counters(dirs)
    print(f"{'kernel':56s} {'launches':>8s} {'FETCH GB':>10s} {'WRITE GB':>10s} {'L2 hit %':>9s}")
    for k, c in sorted(data.items()):
        if "FETCH_SIZE" not in c:
            continue
        hit, miss = sum(c.get("TCC_HIT_sum", [0])), sum(c.get("TCC_MISS_sum", [0]))
        print(f"{k[:56]:56s} {len(c['FETCH_SIZE']):8d} {sum(c['FETCH_SIZE'])*1024/1e9:10.2f} "
              f"{sum(c.get('WRITE_SIZE', [0]))*1024/1e9:10.2f} {100*hit/max(hit+miss,1):9.1f}")


def traffic_json(args):
    import json
    maps, dirs = int(args[0]), args[1:]
    data = counters(dirs)
    doc = {"maps_per_launch": maps, "source": "rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE / --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum "
           "(separate passes of tools/pmc_passes.sh BC); bytes = counter x 1024 summed over the launches; FETCH_SIZE raw "
           "(HBM bytes = 2 x FETCH + WRITE on gfx950)", "kernels": {}}
    for k, c in sorted(data.items()):
        if "FETCH_SIZE" not in c or ("conv_" not in k and "first_layer" not in k):
            continue
        hit, miss = sum(c.get("TCC_HIT_sum", [0])), sum(c.get("TCC_MISS_sum", [0]))
        doc["kernels"][k] = {"launches": len(c["FETCH_SIZE"]), "fetch_bytes": sum(c["FETCH_SIZE"]) * 1024,
                             "write_bytes": sum(c.get("WRITE_SIZE", [0])) * 1024, "l2_hit": round(hit / max(hit + miss, 1), 4)}
    print(json.dumps(doc, indent=1))


def sq(dirs):
    data = counters(dirs)
    for k, c in sorted(data.items()):
        if "conv_" not in k and "first_layer" not in k:
            continue
        print(k)
        for name, v in sorted(c.items()):
            print(f"   {name:36s} {sum(v)/len(v):16.0f}")


def pipe(dirs):
    """matrix-pipe utilisation per kernel from the counters: SQ_VALU_MFMA_BUSY_CYCLES (summed over the 1024 SIMDs) / 1024 against
    GRBM_GUI_ACTIVE / 8 (rocprofv3 sums the 8 XCDs), per-launch means; MFMA operations by type; LDS bank-conflict share"""
    data = counters(dirs)
    print(f"{'kernel':64s} {'launches':>8s} {'Mcycles':>9s} {'pipe busy':>10s} {'MOPS f16':>10s} {'MOPS bf16':>10s} {'MOPS f8':>10s} {'MOPS f6f4':>10s} {'MOPS f32':>10s} {'LDS confl':>10s}")
    for k, c in sorted(data.items()):
        if "GRBM_GUI_ACTIVE" not in c or "SQ_VALU_MFMA_BUSY_CYCLES" not in c:
            continue
        m = lambda n: sum(c[n]) / len(c[n]) if n in c else 0.0
        gui = m("GRBM_GUI_ACTIVE") / 8
        print(f"{k[:64]:64s} {len(c['SQ_VALU_MFMA_BUSY_CYCLES']):8d} {gui/1e6:9.3f} {m('SQ_VALU_MFMA_BUSY_CYCLES')/1024/max(gui,1):10.3f} "
              f"{m('SQ_INSTS_VALU_MFMA_MOPS_F16')/1e9:10.2f} {m('SQ_INSTS_VALU_MFMA_MOPS_BF16')/1e9:10.2f} {m('SQ_INSTS_VALU_MFMA_MOPS_F8')/1e9:10.2f} {m('SQ_INSTS_VALU_MFMA_MOPS_F6F4')/1e9:10.2f} {m('SQ_INSTS_VALU_MFMA_MOPS_F32')/1e9:10.2f} "
              f"{m('SQ_LDS_BANK_CONFLICT')/max(m('SQ_LDS_IDX_ACTIVE'),1):10.3f}")


if __name__ == "__main__":
    {"stats": lambda a: stats(a[0]), "traffic": traffic, "traffic-json": traffic_json, "sq": sq, "pipe": pipe}[sys.argv[1]](sys.argv[2:])
