#!/bin/bash
# usage (GPU box): tools/gpu_trace.sh <tag>  -- rocprofv3 kernel trace of tools/timing_probe.py + summary
tag=${1:-tr}; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp SSVIO_BENCH_SOFT_EXIT=1
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/$tag -- python3 $R/tools/timing_probe.py > $R/gpurun_out/$tag.log 2>&1
grep batch $R/gpurun_out/$tag.log
python3 $R/tools/trace_summary.py $R/gpurun_out/$tag
# the bench line's 256-frame launches, one step in flight, k_blur started after k_fast (SSVIO_BLUR_AFTER_FAST: the default starts it beside
# k_fast for batches of this size, which stretches both): the exclusive duration of every kernel as rocprofv3 sees it
echo "== 256-frame launches, one step in flight, k_blur after k_fast"
SSVIO_BLUR_AFTER_FAST=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_256 -- python3 $R/tools/timing_probe.py 256 > $R/gpurun_out/${tag}_256.log 2>&1
grep batch $R/gpurun_out/${tag}_256.log
python3 $R/tools/trace_summary.py $R/gpurun_out/${tag}_256
echo "== 768-frame launches (the bench line's default step), one step in flight, k_blur after k_fast"
SSVIO_BLUR_AFTER_FAST=1 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${tag}_768 -- python3 $R/tools/timing_probe.py 768 > $R/gpurun_out/${tag}_768.log 2>&1
grep batch $R/gpurun_out/${tag}_768.log
python3 $R/tools/trace_summary.py $R/gpurun_out/${tag}_768
