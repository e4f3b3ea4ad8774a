#!/bin/bash
# The round-2 library (commit ece4e7e) as optiland_amd/lib/variant_r02.so: the "before" arm
# of the round-3 interleaved A/Bs (tools/gpu_ab_r03.sh).  CPU only (hipcc cross-compiles).
set -e
R=$(cd $(dirname $0)/.. && pwd)
T=${TMPDIR:-/tmp}/r02tree
rm -rf $T; git -C $R worktree prune; git -C $R worktree add -f $T ece4e7e > /dev/null
(cd $T && python -c "
import sys; sys.path.insert(0, '$T')
from optiland_amd import build
print(build.build_library(force=True))")
cp $T/optiland_amd/lib/liboptiland_hip.so $R/optiland_amd/lib/variant_r02.so
git -C $R worktree remove --force $T
echo $R/optiland_amd/lib/variant_r02.so
