#!/bin/bash
# usage (GPU box): tools/gpu_trace.sh <tag>  -- rocprofv3 kernel trace of tools/timing_probe.py + summary
tag=${1:-tr}; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp SSVIO_BENCH_SOFT_EXIT=1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/timing_probe.py > $R/gpurun_out/$tag.log 2>&1
grep batch $R/gpurun_out/$tag.log
python3 $R/tools/trace_summary.py $R/gpurun_out/$tag
