#!/bin/bash
# Workgroup size for the generating record-all kernel (the default bench step): 256 (product)
# against 128 / 512.  Interleaved, one box.  Output: gpurun_out/r03_ab_block_gen.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/r03_ab_block_gen.txt; mkdir -p $R/gpurun_out; : > $OUT
run() { local label=$1 lib=$2; shift 2
  echo -n "$label   " >> $OUT
  if [ -n "$lib" ]; then
    OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$lib.so timeout 120 python tools/ab_kernel.py "$@" 2>/dev/null | tail -1 >> $OUT
  else
    timeout 120 python tools/ab_kernel.py "$@" 2>/dev/null | tail -1 >> $OUT
  fi
  echo >> $OUT
}
order() { if [ $(($1 % 2)) -eq 1 ]; then echo "${@:2}"; else echo "${@:2}" | tr ' ' '\n' | tac | tr '\n' ' '; fi; }
echo "# $(date -u) interleaved A/B, 1e7 rays" >> $OUT
ab() { local tag=$1 arms=$2; shift 2
  for rep in 1 2 3 4; do for v in $(order $rep $arms); do run "$tag $v" "${v/product/}" "$@"; done; done; }
ab dg_f32_gen "product block128 block512" --mode gen
ab rc_f32_gen "product block128 block512" --workload rc_asphere --mode gen
ab zf_f32_gen "product block128 block512" --workload zernike_fresnel --mode gen
python tools/ab_summary.py $OUT
