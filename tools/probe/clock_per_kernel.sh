# GPU box: the shader clock each kernel of a step runs at, exclusive launches: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / launch duration
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/clkk -- python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --no-batched-matching --no-next-rows --sync-steps > $R/gpurun_out/clkk.log 2>&1 || { tail -5 $R/gpurun_out/clkk.log; exit 1; }
python3 - $R/gpurun_out/clkk <<'PY'
import csv, sys, glob, collections
f = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
print(list(rows[0].keys()))
agg = collections.defaultdict(list)
for r in rows:
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("ssvio::", "")
    dt = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    agg[(k[:28], r["Grid_Size"])].append((float(r["Counter_Value"]) / 8, dt))
for k, v in agg.items():
    cyc = sum(a for a, _ in v) / len(v); ns = sum(b for _, b in v) / len(v)
    print("%-28s grid %9s  n=%3d  %9.0f cycles  %8.1f us  %.3f GHz" % (k[0], k[1], len(v), cyc, ns / 1e3, cyc / ns))
PY
