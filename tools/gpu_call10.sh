#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r03e_pytest_gpu.log
python tools/gpu_r03_dropin.py > gpurun_out/r03_dropin.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r03_dropin.json'))
print(json.dumps(d['trace_generic_1e7_float32'])); print(json.dumps(d['set_radius_then_trace_100_rays'])); print(json.dumps(d['reference_analyses_cooke_fp64']['with_seams']))"
OPTILAND_HIP_DEFER_VIEWS=0 python tools/gpu_r03_dropin.py > gpurun_out/r03_dropin_bound.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r03_dropin.json'))
print('views bound inside the call:'); print(json.dumps(d['trace_generic_1e7_float32'])); print(json.dumps(d['set_radius_then_trace_100_rays']))"
timeout 900 python tools/gpu_ref_consumers.py > gpurun_out/r03_reference_consumers_on_device.txt 2>&1
tail -8 gpurun_out/r03_reference_consumers_on_device.txt | cut -c1-330
python tools/gpu_r03_dropin.py > gpurun_out/r03_dropin.log 2>&1
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03_bench_default.json
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_default.json')); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], json.dumps(d['dropin']))"
