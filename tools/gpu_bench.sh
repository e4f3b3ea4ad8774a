#!/bin/bash
# GPU-box script: smoke, bench, rocprofv3 kernel-trace stats (+ optional PMC passes).
set -x
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out
mkdir -p $OUT
cd $R
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
python bench.py --steps 20 --warmup 3 2>&1 | tee $OUT/bench_default.log | tail -3
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o dg_f32 -- python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/prof_stats.log 2>&1
tail -3 $OUT/prof_stats.log
ls -R $OUT/prof_stats | head -30
