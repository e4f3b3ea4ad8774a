#!/bin/bash
# Full GPU round: parity tests, smoke, default bench (with CPU baseline), profiles.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee gpurun_out/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
python bench.py 2>/dev/null | tail -1 | tee gpurun_out/bench_default.json
bash tools/gpu_prof.sh dg_f32 > /dev/null 2>&1
bash tools/gpu_prof.sh dg_f64 --dtype f64 > /dev/null 2>&1
bash tools/gpu_prof.sh rc_f32 --workload rc_asphere > /dev/null 2>&1
bash tools/gpu_prof.sh zf_f32 --workload zernike_fresnel > /dev/null 2>&1
bash tools/gpu_prof.sh dg_f32_last --mode last > /dev/null 2>&1
bash tools/gpu_prof.sh dg_f32_copy --object-row copy > /dev/null 2>&1
bash tools/gpu_prof.sh dg_f32_spot --mode spot > /dev/null 2>&1
bash tools/gpu_prof.sh dg_f64_spot --mode spot --dtype f64 > /dev/null 2>&1
for t in dg_f32 dg_f64 rc_f32 zf_f32 dg_f32_last dg_f32_copy dg_f32_spot dg_f64_spot; do echo "=== $t"; grep '^{' gpurun_out/prof_$t/stats.log | tail -1 | cut -c1-400; grep -E "trace_kernel|spot" gpurun_out/prof_$t/summary.txt | cut -c1-260; done
