#!/bin/bash
# usage (GPU box): tools/gpu_final.sh  -- everything profiles/r2 holds for the bench line, in one go:
#   kernel trace + HBM passes (gpu_profile.sh), SQ and LDS passes (gpu_pmc.sh), per-batch-size trace (gpu_trace.sh), then the default
#   bench line (with the CPU baseline) and the 200-step one.  Copy gpurun_out/r2_* into profiles/r2/ afterwards (tools/install_profiles.sh).
R=$GRAFT_REPO_ROOT; cd $R
bash tools/gpu_profile.sh r2_final > gpurun_out/r2_final_profile.log 2>&1 || exit 1
bash tools/gpu_pmc.sh r2_final_sq "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" > gpurun_out/r2_final_pmc_sq.txt 2>&1 || exit 1
bash tools/gpu_pmc.sh r2_final_lds "SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" > gpurun_out/r2_final_pmc_lds.txt 2>&1 || exit 1
bash tools/gpu_trace.sh r2f_tr > gpurun_out/r2_trace_by_batch_size.txt 2>&1 || exit 1
cd $R
# the profile-sourced fields of the line come from profiles/r2: install the fresh files first
cp gpurun_out/r2_final_kernel_stats.csv gpurun_out/r2_final_pmc_hbm.json gpurun_out/r2_final_pmc_sq.txt gpurun_out/r2_final_pmc_lds.txt gpurun_out/r2_final_source_hash.txt profiles/r2/
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r2_bench_default.json 2> gpurun_out/r2_bench_default.err || exit 1
timeout -k 10 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline > gpurun_out/r2_bench_200steps.json 2> /dev/null || exit 1
echo done
