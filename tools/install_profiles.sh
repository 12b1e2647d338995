#!/bin/bash
# usage (here, after tools/gpu_final.sh on the GPU box): copies the merged gpurun_out/r3_* results into profiles/r3/
cd "$(dirname "$0")/.." && cd gpurun_out && cp r3_final_kernel_stats.csv r3_final_pmc_hbm.json r3_final_pmc_sq.txt r3_final_pmc_lds.txt r3_final_source_hash.txt \
   r3_bench_default.json r3_bench_200steps.json r3_trace_by_batch_size.txt ../profiles/r3/ && cp r3_final_bench_under_trace.json ../profiles/r3/r3_final_bench_under_rocprof.json && cp r3_mfma_probe.txt r3_mfma_pmc.txt r3_valu_pmc.txt r3_1080p_kernel_stats.csv r3_1080p_pmc_hbm.json ../profiles/r3/ && echo installed
