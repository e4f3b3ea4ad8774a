#!/bin/bash
# Round 3, closing A/B: the product library as it ships against the round-2 library
# (variant_r02.so, built from its commit by tools/build_r02_variant.sh), interleaved on ONE
# box, every kernel class whose table / argument access changed this round.  kernel_ms = mean
# HIP-event time of 20 launches at 1e7 rays (tools/ab_kernel.py); summary by
# tools/ab_summary.py.  Output: gpurun_out/r03_ab_final.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/r03_ab_final.txt; mkdir -p $R/gpurun_out; : > $OUT
ROUNDS=${ROUNDS:-2}
run() { local label=$1 lib=$2; shift 2
  echo -n "$label   " >> $OUT
  if [ -n "$lib" ]; then
    OPTILAND_HIP_ALLOW_ABI5=1 OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$lib.so \
      timeout 120 python tools/ab_kernel.py "$@" 2>/dev/null | tail -1 >> $OUT
  else
    timeout 120 python tools/ab_kernel.py "$@" 2>/dev/null | tail -1 >> $OUT
  fi
  echo >> $OUT
}
order() { if [ $(($1 % 2)) -eq 1 ]; then echo "${@:2}"; else echo "${@:2}" | tr ' ' '\n' | tac | tr '\n' ' '; fi; }
echo "# $(date -u) interleaved A/B, 1e7 rays, arms alternate order every round" >> $OUT
ab() { local tag=$1 arms=$2; shift 2
  for rep in $(seq 1 $ROUNDS); do
    for v in $(order $rep $arms); do run "$tag $v" "${v/product/}" "$@"; done
  done
}
ab zf_f32_rec   "product r02" --workload zernike_fresnel
ab zf_f64_rec   "product r02" --workload zernike_fresnel --dtype f64
ab z_f32_rec    "product r02" --workload zernike
ab z_f64_rec    "product r02" --workload zernike --dtype f64
ab rc_f32_rec   "product r02" --workload rc_asphere
ab rc_f64_rec   "product r02" --workload rc_asphere --dtype f64
ab z_f32_spot   "product r02" --workload zernike --mode spot
ab z_f64_spot   "product r02" --workload zernike --mode spot --dtype f64
ab rc_f32_spot  "product r02" --workload rc_asphere --mode spot
ab rc_f64_spot  "product r02" --workload rc_asphere --mode spot --dtype f64
ab dg_f32_spot  "product r02" --mode spot
ab dg_f64_spot  "product r02" --mode spot --dtype f64
ab z_opd        "product r02" --workload zernike --mode opd
ab rc_opd       "product r02" --workload rc_asphere --mode opd
ab dg_opd       "product r02" --mode opd
ab dg_f32_rec   "product r02"
ab dg_f64_rec   "product r02" --dtype f64
# store flavour / workgroup size for the fp64 record-all kernels (C3 runs these)
ab dg_f64_rec_stores "product plain_stores block128 block512" --dtype f64
ab dg_f64_gen_stores "product plain_stores block128 block512" --dtype f64 --mode gen
python tools/ab_summary.py $OUT | tee gpurun_out/r03_ab_final_summary.txt
