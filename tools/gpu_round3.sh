#!/bin/bash
# Round 3 GPU pass: parity suite, smoke, default bench line, rocprofv3 stats + PMC of the
# default bench and of the configurations whose kernels changed, SQ counter table, drop-in
# end-to-end numbers.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_live_reference.py -m gpu -q -x -k "fused_seam_on_device" 2>&1 | tail -60 > gpurun_out/r03_seam_fail.log
python -m pytest tests -m gpu -q 2>&1 | tail -12 | tee gpurun_out/r03_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee gpurun_out/r03_smoke.log
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03_bench_default.json
python bench.py --mode record --no-ref-baselines --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_record.json
bash tools/gpu_prof.sh r03_dg_f32_gen > /dev/null 2>&1
bash tools/gpu_prof.sh r03_zf_f64 --workload zernike_fresnel --dtype f64 --mode record > /dev/null 2>&1
bash tools/gpu_prof.sh r03_rc_f64 --workload rc_asphere --dtype f64 --mode record > /dev/null 2>&1
bash tools/gpu_prof.sh r03_z_opd --workload zernike --mode opd > /dev/null 2>&1
bash tools/gpu_prof.sh r03_zf_f32_gen --workload zernike_fresnel > /dev/null 2>&1
bash tools/gpu_prof.sh r03_rc_f32_gen --workload rc_asphere > /dev/null 2>&1
for t in r03_dg_f32_gen r03_zf_f64 r03_rc_f64 r03_z_opd r03_zf_f32_gen r03_rc_f32_gen; do echo "=== $t"; grep '^{' gpurun_out/prof_$t/stats.log | tail -1 | cut -c1-300; cat gpurun_out/prof_$t/summary.txt | cut -c1-260 | head -14; done > gpurun_out/r03_prof_summaries.txt
python tools/gpu_r03_dropin.py > gpurun_out/r03_dropin.log 2>&1
bash tools/gpu_kernel_table.sh $R/gpurun_out/r03_kernel_table_after.txt > /dev/null 2>&1 <<'CFG'
dg_f32_gen  |
dg_f32_rec  | --mode record
rc_f32_rec  | --workload rc_asphere --mode record
zf_f32_rec  | --workload zernike_fresnel --mode record
dg_f64_rec  | --dtype f64 --mode record
rc_f64_rec  | --workload rc_asphere --dtype f64 --mode record
zf_f64_rec  | --workload zernike_fresnel --dtype f64 --mode record
z_f32_rec   | --workload zernike --mode record
z_f64_rec   | --workload zernike --dtype f64 --mode record
dg_f32_spot | --mode spot
dg_f64_spot | --mode spot --dtype f64
rc_f32_spot | --workload rc_asphere --mode spot
rc_f64_spot | --workload rc_asphere --mode spot --dtype f64
z_f32_spot  | --workload zernike --mode spot
z_f64_spot  | --workload zernike --mode spot --dtype f64
dg_opd      | --mode opd
rc_opd      | --workload rc_asphere --mode opd
z_opd       | --workload zernike --mode opd
CFG
cat gpurun_out/r03_seam_fail.log | tail -40
cat gpurun_out/r03_bench_default.json | cut -c1-1500
