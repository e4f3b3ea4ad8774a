#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for rep in 1 2; do
for al in 256 16384 1048576 2097152; do
 echo "## align=$al"; OPTILAND_RECORD_ALIGN=$al python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value=%.4g kernel_ms=%.4f moved=%.0f GB/s'%(d['value'],d['roofline']['kernel_ms'],d['roofline']['moved_GBps']))"
done
done
echo "## default"; python bench.py --steps 30 --warmup 3 --no-cpu-baseline --dtype f64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   f64 value=%.4g kernel_ms=%.4f moved=%.0f GB/s'%(d['value'],d['roofline']['kernel_ms'],d['roofline']['moved_GBps']))"
OPTILAND_RECORD_ALIGN=256 python bench.py --steps 30 --warmup 3 --no-cpu-baseline --dtype f64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   f64 align256 value=%.4g kernel_ms=%.4f moved=%.0f GB/s'%(d['value'],d['roofline']['kernel_ms'],d['roofline']['moved_GBps']))"
