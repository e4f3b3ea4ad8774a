#!/bin/bash
# Fused spot kernel: parity tests, then timing of fused vs un-fused pipelines.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_spot.py -q -x 2>&1 | tail -15 | tee gpurun_out/pytest_spot.log
for rpt in 0 1 2; do
  for dt in f32 f64; do
    echo "== spot rpt=$rpt $dt"
    OL_TRACE_RPT=$rpt timeout 300 python bench.py --mode spot --dtype $dt --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
echo "== last f32 (un-fused trace only)"
timeout 300 python bench.py --mode last --no-cpu-baseline --steps 20 2>&1 | tail -1 | cut -c1-200
for w in rc_asphere cooke; do
  echo "== spot $w"
  timeout 300 python bench.py --mode spot --workload $w --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'])"
done
