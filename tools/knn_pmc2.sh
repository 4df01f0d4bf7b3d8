#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp DGMI_SKIP_BUILD=1
i=0
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_2ph/g$i -- python3 $R/tools/knn_profile.py 100000 4 > $R/gpurun_out/pmc_2ph_g$i.log 2>&1 || { echo "group $i failed"; tail -3 $R/gpurun_out/pmc_2ph_g$i.log; exit 1; }
  echo "group $i done"
done
