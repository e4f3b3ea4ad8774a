#!/bin/bash
# (prepared in round 4, run in round 5) one rocprofv3 --pmc pass per counter set over
# tools/gpu_window_pmc.py; counters the build does not know are skipped.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out/window_pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > $R/gpurun_out/window_pmc/list_avail.txt 2>&1
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_WRREQ_STALL_sum" "TCC_TAG_STALL_sum" \
           "TCC_EA0_WRREQ_DRAM_sum" "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum" \
           "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum" "TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_LEVEL_sum" \
           "GRBM_GUI_ACTIVE"; do
  tag=$(echo $set | tr ' ' '+')
  timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/window_pmc/$tag -o w -- \
      python $R/tools/gpu_window_pmc.py > $R/gpurun_out/window_pmc/$tag.log 2>&1 || echo "skipped: $set"
done
cd $R
python tools/gpu_window_pmc.py --analyse gpurun_out/window_pmc | tee gpurun_out/r05_window_pmc.txt
