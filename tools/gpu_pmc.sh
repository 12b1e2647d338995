#!/bin/bash
# usage (GPU box): tools/gpu_pmc.sh <tag> "<counters>" [kernel-substring]  -- one rocprofv3 --pmc pass over a short bench run
tag=$1; ctr=$2; flt=$3; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp SSVIO_BENCH_SOFT_EXIT=1
timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/$tag -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-batched-matching --no-next-rows --sync-steps > $R/gpurun_out/$tag.log 2>&1 || { tail -5 $R/gpurun_out/$tag.log; exit 1; }
python3 $R/tools/pmc_summary.py $R/gpurun_out/$tag $flt
