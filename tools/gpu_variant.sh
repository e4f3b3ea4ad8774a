#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value=%.4g kernel_ms=%.4f moved=%.0f GB/s'%(d['value'],d['roofline']['kernel_ms'],d['roofline']['moved_GBps']))"; }
for rep in 1 2 3; do
 echo "## default (nt scalar)"; python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show
 echo "## plain scalar"; OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_plain_scalar.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show
done
echo "## f64 default"; python bench.py --steps 30 --warmup 3 --no-cpu-baseline --dtype f64 2>/dev/null | show
echo "## f64 plain"; OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_plain_scalar.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline --dtype f64 2>/dev/null | show
