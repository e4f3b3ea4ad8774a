#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_live_reference.py -m gpu -q -x 2>&1 | tail -15
python tools/gpu_r03_dropin.py > gpurun_out/r03_dropin.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r03_dropin.json'))
print(json.dumps(d['trace_generic_1e7_float32'])); print(json.dumps(d['trace_generic_1e7_float64'])); print(json.dumps(d['set_radius_then_trace_100_rays']))"
grep -A34 "EncircledEnergy(1e6" gpurun_out/r03_analyses_profile.txt | cut -c1-170
bash tools/gpu_dist1.sh 2>&1 | tee gpurun_out/r03_dist1.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03_bench_default.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_default.json')); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], json.dumps(d['dropin']))"
