#!/bin/bash
# Round 3, call 1: the kernels VERDICT r2 lists as unmeasured, as they stand.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
OUT=gpurun_out/${1:-r03_kernel_table_before.txt}
bash tools/gpu_kernel_table.sh $R/$OUT <<'CFG'
dg_f32_rec  |
rc_f32_rec  | --workload rc_asphere
zf_f32_rec  | --workload zernike_fresnel
dg_f64_rec  | --dtype f64
rc_f64_rec  | --workload rc_asphere --dtype f64
zf_f64_rec  | --workload zernike_fresnel --dtype f64
z_f32_rec   | --workload zernike
z_f64_rec   | --workload zernike --dtype f64
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
echo; cat $OUT
