import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*_kernel_stats.csv")[0]
steps = float(sys.argv[2])
for r in list(csv.DictReader(open(f)))[:45]:
    n = re.sub(r"void lrpx::|lrpx::", "", r["Name"]); n = re.sub(r"\(.*", "", n)
    if "f16x3" in n and ", 5, " in n:
        continue
    print("%-58s %5d %8.2f ms/step %9.1f us" % (n[:58], int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3))
