#!/bin/bash
# usage (here, after tools/gpu_final.sh on the GPU box): copies the merged gpurun_out/r2_* results into profiles/r2/
cd "$(dirname "$0")/.." && cd gpurun_out && cp r2_final_kernel_stats.csv r2_final_pmc_hbm.json r2_final_pmc_sq.txt r2_final_pmc_lds.txt r2_final_source_hash.txt \
   r2_bench_default.json r2_bench_200steps.json r2_trace_by_batch_size.txt ../profiles/r2/ && cp r2_final_bench_under_trace.json ../profiles/r2/r2_final_bench_under_rocprof.json && echo installed
