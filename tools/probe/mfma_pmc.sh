#!/bin/bash
# usage (GPU box): tools/probe/mfma_pmc.sh <tag> <valu|mfma> <rows> <npairs>  -- kernel trace + SQ / GRBM counter passes of one matcher kernel
tag=$1; k=$2; rows=$3; np=$4; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/tools/probe/mfma_one.py $k $rows $np 20 > $R/gpurun_out/${tag}_trace.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_trace.log; exit 1; }
cp $(ls $R/gpurun_out/${tag}_trace/*/*kernel_stats.csv | tail -1) $R/gpurun_out/${tag}_kernel_stats.csv && head -6 $R/gpurun_out/${tag}_kernel_stats.csv
i=0
for ctr in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAVES" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 90 rocprofv3 --pmc $ctr --output-format csv -d $R/gpurun_out/${tag}_pmc$i -- python3 $R/tools/probe/mfma_one.py $k $rows $np 4 > $R/gpurun_out/${tag}_pmc$i.log 2>&1 || { tail -5 $R/gpurun_out/${tag}_pmc$i.log; continue; }
  python3 $R/tools/pmc_summary.py $R/gpurun_out/${tag}_pmc$i knn2
done
