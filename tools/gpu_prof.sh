#!/bin/bash
# rocprofv3 on the bench: kernel-trace stats, then HBM counters in separate passes.
# usage: gpu_prof.sh <tag> [bench args...]
set -x
TAG=${1:-dg_f32}; shift
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o $TAG -- \
  python $R/bench.py --steps 20 --warmup 5 --settle 0 --traffic committed --no-cpu-baseline "$@" > $OUT/stats.log 2>&1
grep '^{' $OUT/stats.log | tail -1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o $TAG -- \
  python $R/bench.py --steps 5 --warmup 1 --settle 0 --traffic committed --no-cpu-baseline "$@" > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o $TAG -- \
  python $R/bench.py --steps 5 --warmup 1 --settle 0 --traffic committed --no-cpu-baseline "$@" > $OUT/pmc_write.log 2>&1
find $OUT -type f | head -40
python $R/tools/summarize_prof.py $OUT $TAG | tee $OUT/summary.txt
