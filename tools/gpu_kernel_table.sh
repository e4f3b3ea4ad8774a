#!/bin/bash
# One row per bench configuration: rocprofv3 kernel time (--kernel-trace --stats) and, in
# a SEPARATE pass, the SQ instruction counters of the dominant kernel.
# usage: gpu_kernel_table.sh OUTFILE  < lines "tag | bench args"
R=${GRAFT_REPO_ROOT:-$PWD}
OUTFILE=${1:-$R/gpurun_out/kernel_table.txt}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out
: > $OUTFILE
while IFS='|' read -r TAG ARGS; do
  TAG=$(echo $TAG); [ -z "$TAG" ] && continue
  OUT=$R/gpurun_out/kt_$TAG
  rm -rf $OUT; mkdir -p $OUT
  cd /tmp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- \
    python $R/bench.py --steps 10 --warmup 2 --settle 0 --traffic committed --no-cpu-baseline $ARGS > $OUT/stats.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
    --output-format csv -d $OUT/sq -o q -- \
    python $R/bench.py --steps 3 --warmup 1 --settle 0 --traffic committed --placement plain --no-cpu-baseline $ARGS > $OUT/sq.log 2>&1
  cd $R
  python tools/kernel_table_row.py $OUT "$TAG" "$ARGS" | tee -a $OUTFILE
done
