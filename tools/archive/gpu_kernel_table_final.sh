#!/bin/bash
# The per-kernel table (rocprofv3 duration, SQ instruction counts per ray, vector-issue time,
# moved bytes, fraction) on the FINAL round-3 library.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
bash tools/gpu_kernel_table.sh $R/gpurun_out/r03_kernel_table_final.txt > /dev/null 2>&1 <<'CFG'
dg_f32_gen  |
dg_f32_rec  | --mode record
rc_f32_gen  | --workload rc_asphere
zf_f32_gen  | --workload zernike_fresnel
zf_f32_rec  | --workload zernike_fresnel --mode record
dg_f64_gen  | --dtype f64
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
python - <<'PY'
import sys, json
print(f"{'tag':<12} {'kernel':<58} {'us':>8} {'VALU/ray':>9} {'SALU/ray':>9} {'SMEM/ray':>8} {'issue_ms':>8} {'movedGB':>8} {'TB/s':>6} {'frac':>6}")
for ln in open("gpurun_out/r03_kernel_table_final.txt"):
    if not ln.startswith('{'): continue
    r=json.loads(ln)
    print(f"{r['tag']:<12} {r.get('kernel','?')[:58]:<58} {r.get('avg_us',0):8.1f} {r.get('VALU_per_ray',0):9.0f} {r.get('SALU_per_ray',0):9.0f} {r.get('SMEM_per_ray',0):8.0f} {r.get('valu_issue_ms',0):8.3f} {r.get('moved_GB',0):8.3f} {r.get('TBps_moved',0):6.2f} {r.get('frac',0):6.3f}")
PY
