#!/bin/bash
# Round 3, final GPU pass on one box: parity suite, smoke, the default bench line, its
# rocprofv3 stats + PMC passes (profiles/traffic.json), drop-in end to end, entry-point matrix.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r03_pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee gpurun_out/r03_smoke.log
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03_bench_default.json
bash tools/gpu_prof.sh r03_dg_f32_gen > /dev/null 2>&1
bash tools/gpu_prof.sh r03_zf_f32_gen --workload zernike_fresnel > /dev/null 2>&1
python tools/gpu_r03_dropin.py > gpurun_out/r03_dropin.log 2>&1
python tools/gpu_dropin_matrix.py > gpurun_out/r03_dropin_matrix.log 2>&1
python tools/gpu_pol_e2e.py > /dev/null 2>&1
bash tools/gpu_dist1.sh 2>&1 | tee gpurun_out/r03_dist1.txt | tail -9
python -c "
import json; d=json.load(open('gpurun_out/r03_bench_default.json')); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['frac'], d['roofline']['traffic'], json.dumps(d['dropin']))
d=json.load(open('gpurun_out/r03_dropin.json')); print(json.dumps(d['trace_generic_1e7_float32'])); print(json.dumps(d['set_radius_then_trace_100_rays'])); print(json.dumps(d['reference_analyses_cooke_fp64']['with_seams']))"
grep -E "DoubleGauss|ZernikeFresnelPolarized" gpurun_out/r03_dropin_matrix.log | cut -c1-420
cut -c1-200 gpurun_out/prof_r03_dg_f32_gen/summary.txt | grep -v "at::native" | head -12
