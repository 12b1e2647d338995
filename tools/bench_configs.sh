#!/bin/bash
# usage (GPU box): tools/bench_configs.sh <tag>   -- one bench line per BASELINE config that fits one GPU
# config 1: VGA pair-style latency (one frame per step, 500 features, host waits every step); config 2: the default line;
# config 3: 1080p, 4000 features, kNN2 ratio.  Configs 4/5 are the batched_matching section of the default line.
tag=${1:-cfg}; mkdir -p gpurun_out
run() { name=$1; shift; timeout -k 10 280 python -u bench.py "$@" > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err || { tail -5 gpurun_out/${tag}_${name}.err; exit 1; }
        python - <<PY
import json
d=json.loads(open("gpurun_out/${tag}_${name}.json").read().strip().splitlines()[-1])
print("${name}", d["value"], d["unit"], "ms/step", d["ms_per_step"], "cpu", (d.get("cpu_baseline") or {}).get("value"), "all-cores", (d.get("cpu_baseline_all_cores") or {}).get("value"))
PY
}
run config1_latency --batch 1 --features 500 --sync-steps --streams 1 --steps 300 --no-batched-matching --no-next-rows && \
run config2 && \
run config3 --width 1920 --height 1080 --features 4000 --match-mode ratio --batch 16 --steps 50 --no-batched-matching --no-next-rows
