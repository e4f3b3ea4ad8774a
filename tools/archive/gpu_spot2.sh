#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_spot.py -q 2>&1 | tail -15 | tee gpurun_out/pytest_spot.log
for tiles in 0 1 2 4 8; do
  for dt in f32 f64; do
    echo "== spot tiles=$tiles $dt"
    OL_SPOT_TILES=$tiles timeout 300 python bench.py --mode spot --dtype $dt --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
echo "== f32 rpt1 tiles 1"
OL_TRACE_RPT=1 OL_SPOT_TILES=1 timeout 300 python bench.py --mode spot --no-cpu-baseline --steps 30 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
