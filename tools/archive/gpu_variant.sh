#!/bin/bash
# A/B of library variants built with different compile-time knobs (same box, alternating)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value=%.4g kernel_ms=%.4f moved=%.0f GB/s'%(d['value'],d['roofline']['kernel_ms'],d['roofline']['moved_GBps']))"; }
for rep in 1 2 3; do
 echo "## default"; python bench.py --steps 30 --warmup 3 --no-cpu-baseline $EXTRA 2>/dev/null | show
 for v in "$@"; do
  echo "## $v"; OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/$v.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline $EXTRA 2>/dev/null | show
 done
done
