#!/bin/bash
# PMC passes for the f4 screen kernel (guarded: one counter group per run, `timeout` around rocprofv3).
R=${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-60000}
cd /tmp && export TMPDIR=/tmp DGMI_SKIP_BUILD=1
i=0
for grp in "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
           "SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_knn/g$i -- python3 $R/tools/knn_profile.py $N 4 > $R/gpurun_out/pmc_knn_g$i.log 2>&1 || { echo "group $i failed"; exit 1; }
  echo "group $i done"
done
