#!/bin/bash
# usage (GPU box): tools/gpu_final.sh  -- everything profiles/r3 holds for the bench line, in one go:
#   kernel trace + HBM passes (gpu_profile.sh), SQ and LDS passes (gpu_pmc.sh), per-batch-size trace (gpu_trace.sh), then the default
#   bench line (with the CPU baseline) and the 200-step one.  Copy gpurun_out/r3_* into profiles/r3/ afterwards (tools/install_profiles.sh).
R=$GRAFT_REPO_ROOT; cd $R
echo profile; bash tools/gpu_profile.sh r3_final > gpurun_out/r3_final_profile.log 2>&1 || exit 1
bash tools/gpu_pmc.sh r3_final_sq "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES SQ_WAVE_CYCLES" > gpurun_out/r3_final_pmc_sq.txt 2>&1 || exit 1
bash tools/gpu_pmc.sh r3_final_lds "SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" > gpurun_out/r3_final_pmc_lds.txt 2>&1 || exit 1
echo trace; bash tools/gpu_trace.sh r3f_tr > gpurun_out/r3_trace_by_batch_size.txt 2>&1 || exit 1
cd $R
# 1080p (configs[2]'s shape): kernel stats + HBM passes
bash tools/gpu_profile.sh r3_1080p --width 1920 --height 1080 --features 4000 --match-mode ratio --batch 32 > gpurun_out/r3_1080p_profile.log 2>&1 || exit 1
cp gpurun_out/r3_1080p_kernel_stats.csv gpurun_out/r3_1080p_pmc_hbm.json profiles/r3/
# the profile-sourced fields of the line come from profiles/r3: install the fresh files first
cp gpurun_out/r3_final_kernel_stats.csv gpurun_out/r3_final_pmc_hbm.json gpurun_out/r3_final_pmc_sq.txt gpurun_out/r3_final_pmc_lds.txt gpurun_out/r3_final_source_hash.txt profiles/r3/
echo bench; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench_default.json 2> gpurun_out/r3_bench_default.err || exit 1
echo bench; timeout -k 10 300 python bench.py --steps 200 --warmup 5 --no-cpu-baseline > gpurun_out/r3_bench_200steps.json 2> /dev/null || exit 1
# the matrix-core matcher: probe table + counters on 8000 x 8000 blocks, both kernels
echo mfma; timeout -k 10 300 python -u tools/probe/mfma_probe.py > gpurun_out/r3_mfma_probe.txt 2>&1 || exit 1
bash tools/probe/mfma_pmc.sh r3_mfma mfma 8000 32 > gpurun_out/r3_mfma_pmc.txt 2>&1 || exit 1
bash tools/probe/mfma_pmc.sh r3_valu valu 8000 32 > gpurun_out/r3_valu_pmc.txt 2>&1 || exit 1
echo done
