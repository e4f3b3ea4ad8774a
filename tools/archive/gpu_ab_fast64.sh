#!/bin/bash
# Round 3: fp64 division / square root from the v_rcp_f64 / v_rsq_f64 seeds + two refinement
# steps (surface_math.h: OL_FAST_F64, the product default) against the compiler's IEEE
# sequences (variant_fast64_0.so).  Interleaved A/B on one box, kernel_ms = mean HIP-event
# time of 20 launches at 1e7 rays (tools/ab_kernel.py).  Output: gpurun_out/r03_ab_fast64.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/r03_ab_fast64.txt; mkdir -p $R/gpurun_out; : > $OUT
ROUNDS=${ROUNDS:-3}
run() { # label, lib ('' = product), args...
  local label=$1 lib=$2; shift 2
  echo -n "$label   " >> $OUT
  if [ -n "$lib" ]; then
    OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$lib.so \
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
ab dg_f64_rec   "product fast64_0" --dtype f64
ab dg_f64_spot  "product fast64_0" --mode spot --dtype f64
ab rc_f64_spot  "product fast64_0" --workload rc_asphere --mode spot --dtype f64
ab z_f64_spot   "product fast64_0" --workload zernike --mode spot --dtype f64
ab dg_opd       "product fast64_0" --mode opd
ab z_opd        "product fast64_0" --workload zernike --mode opd
ab zf_f64_rec   "product fast64_0" --workload zernike_fresnel --dtype f64
