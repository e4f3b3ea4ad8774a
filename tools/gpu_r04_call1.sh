#!/bin/bash
# Round 4, GPU call 1: parity suite on the ABI-8 library, default bench line, one-rank
# exchange overhead (one launch per step), clock / power next to per-launch durations (C5 vs DG)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -8 | tee gpurun_out/r04_pytest_gpu_1.log
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/r04_bench_default_1.json
timeout 600 bash tools/gpu_dist1.sh 2>&1 | tee gpurun_out/r04_dist1.txt | tail -12
for cfg in "zernike_fresnel f32" "double_gauss f32" "zernike_fresnel f64"; do
  set -- $cfg
  timeout 120 python tools/gpu_clock_spread.py --workload $1 --dtype $2 --launches 300 \
     > gpurun_out/r04_clock_spread_$1_$2.txt 2>&1
  grep '^{' gpurun_out/r04_clock_spread_$1_$2.txt | cut -c1-900
done
timeout 120 python tools/gpu_clock_spread.py --workload zernike_fresnel --launches 100 --idle-ms 5 \
     > gpurun_out/r04_clock_spread_zf_idle5.txt 2>&1
grep '^{' gpurun_out/r04_clock_spread_zf_idle5.txt | cut -c1-900
python -c "
import json; d=json.load(open('gpurun_out/r04_bench_default_1.json')); print(d['value'], d['ms_per_step'], json.dumps(d['roofline'])[:600])"
