"""Per-kernel averages of rocprofv3 --pmc counters (csv): python tools/pmc_summary.py <dir> [kernel-substring]"""
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ssvio::", "")
    if flt in k and r["Grid_Size"] not in ("", None):
        agg[(k, r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, g), cs in agg.items():
    print("%-24s grid %9s " % (k[:24], g) + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))
