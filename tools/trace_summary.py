"""Summarise a rocprofv3 kernel trace CSV: per-kernel mean duration by grid shape, and inter-kernel gaps."""
import csv, sys, collections, glob
f = sorted(glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows:
    k = (r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ssvio::", ""), r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])
    agg.setdefault(k, []).append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    print("%-28s grid %6s %5s %4s  n=%3d  mean %8.2f us  min %8.2f" % (k[0][:28], k[1], k[2], k[3], len(v), sum(v) / len(v) / 1e3, min(v) / 1e3))
gaps = []
prev = None
for r in rows[len(rows)//2:]:
    st, en = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if prev is not None: gaps.append(st - prev)
    prev = en
gaps.sort()
print("gaps: median %.2f us, p10 %.2f, p90 %.2f" % (gaps[len(gaps)//2] / 1e3, gaps[len(gaps)//10] / 1e3, gaps[len(gaps)*9//10] / 1e3))
