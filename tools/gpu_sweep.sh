#!/bin/bash
# quick A/B sweeps of bench.py variants (no cpu baseline)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
run() { echo "## $*"; env "$@" python bench.py --steps 20 --warmup 3 --no-cpu-baseline $EXTRA 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value=%.4g rs/s  kernel_ms=%.4f  GB/s=%.0f  frac=%.3f'%(d['value'],d['roofline']['kernel_ms'],d['roofline']['achieved'],d['roofline']['frac']))"; }
for rpt in 4 2 1; do run OL_TRACE_RPT=$rpt; done
EXTRA="--mode last" run OL_TRACE_RPT=4
EXTRA="--mode last" run OL_TRACE_RPT=2
EXTRA="--mode last" run OL_TRACE_RPT=1
EXTRA="--dtype f64" run OL_TRACE_RPT=2
EXTRA="--dtype f64" run OL_TRACE_RPT=1
EXTRA="--dtype f64 --mode last" run OL_TRACE_RPT=2
