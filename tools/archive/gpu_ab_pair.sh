#!/bin/bash
# A/B: fp32 record-all / record-last with one ray per lane (default), the 16-byte vector
# and the 8-byte packed pair (OL_TRACE_RPT=3), alternating on one box
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('   kernel_ms=%.4f moved=%.0f GB/s'%(d['roofline']['kernel_ms'],d['roofline']['moved_GBps']))"; }
for mode in record last; do
 for rep in 1 2 3; do
  for rpt in 0 3 2; do
   echo -n "$mode rpt=$rpt"; OL_TRACE_RPT=$rpt python bench.py --mode $mode --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show
  done
 done
done
python -m pytest tests/test_gpu_parity.py -q -k "bit_identical and double_gauss" 2>&1 | tail -1
