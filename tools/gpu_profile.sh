#!/bin/bash
# usage (GPU box): tools/gpu_profile.sh <tag> [extra bench.py arguments, e.g. --width 1920 --height 1080 --features 4000 --match-mode ratio --batch 32]
#   1. rocprofv3 --kernel-trace --stats over the default bench command (per-kernel average durations)
#   2. two --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, MI355X_MICROARCH.md "HBM / rocprofv3") -> HBM traffic per launch
# Results land in gpurun_out/<tag>_*; copy what should be judged into profiles/.
tag=${1:-prof}; shift; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp SSVIO_BENCH_SOFT_EXIT=1
BENCH="$R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-configs --no-batched-matching --no-next-rows $*"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $BENCH > $R/gpurun_out/${tag}_trace.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_trace.log; exit 1; }
grep '"metric"' $R/gpurun_out/${tag}_trace.log > $R/gpurun_out/${tag}_bench_under_trace.json
cp $(ls $R/gpurun_out/${tag}_trace/*/*kernel_stats.csv | tail -1) $R/gpurun_out/${tag}_kernel_stats.csv && head -14 $R/gpurun_out/${tag}_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/${tag}_pmc_$c -- python3 $BENCH --streams 1 > $R/gpurun_out/${tag}_pmc_$c.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_pmc_$c.log; exit 1; }
done
python3 - <<PY
import csv, glob, json, collections
out = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = sorted(glob.glob("$R/gpurun_out/${tag}_pmc_%s/**/*counter_collection.csv" % c, recursive=True))[-1]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == c:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        # rocprofv3 reports KB; FETCH_SIZE counts 128-B requests as 64 B on gfx950 -> doubled (guide, HBM section)
        kb = sum(v) / len(v)
        out[k][c] = {"launches": len(v), "avg_KB_raw": kb, "avg_bytes_corrected": kb * 1024 * (2 if c == "FETCH_SIZE" else 1)}
json.dump(out, open("$R/gpurun_out/${tag}_pmc_hbm.json", "w"), indent=1)
import sys
sys.path.insert(0, "$R")
import bench
open("$R/gpurun_out/${tag}_source_hash.txt", "w").write(bench.kernel_source_hash() + "\n")
for k, v in out.items():
    print("%-40s fetch %10.0f KB (x2 corrected)  write %10.0f KB" % (k[:40], v.get("FETCH_SIZE", {}).get("avg_bytes_corrected", 0) / 1024, v.get("WRITE_SIZE", {}).get("avg_bytes_corrected", 0) / 1024))
PY
