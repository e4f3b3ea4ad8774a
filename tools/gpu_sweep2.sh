#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
run() { echo "## $*"; env "$@" python bench.py --steps 10 --warmup 2 --no-cpu-baseline $EXTRA 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   value=%.4g rs/s  kernel_ms=%.4f  GB/s=%.0f  frac=%.3f'%(d['value'],d['roofline']['kernel_ms'],d['roofline']['achieved'],d['roofline']['frac']))"; }
for wl in rc_asphere zernike_fresnel cooke; do
 for dt in f32 f64; do
  for rpt in 0 1; do
   EXTRA="--workload $wl --dtype $dt" run OL_TRACE_RPT=$rpt
  done
 done
done
EXTRA="--workload rc_asphere --mode last" run OL_TRACE_RPT=0
EXTRA="--workload zernike_fresnel --mode last" run OL_TRACE_RPT=0
EXTRA="--workload zernike_fresnel --mode last" run OL_TRACE_RPT=1
