#!/bin/bash
# fp32 unrolled Zernike: two blocks (product) vs the 8-dword window vs round 2; the random
# pupil drawn on the device (EncircledEnergy at 1e6 rays); edge-case tests.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_parity.py tests/test_gpu_hostmath.py -m gpu -q -x 2>&1 | tail -3
OUT=$R/gpurun_out/r03_ab_zmono32.txt; : > $OUT
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
echo "# $(date -u) interleaved A/B, 1e7 rays" >> $OUT
ab() { local tag=$1 arms=$2; shift 2
  for rep in 1 2 3 4; do for v in $(order $rep $arms); do run "$tag $v" "${v/product/}" "$@"; done; done; }
ab zf_f32_rec  "product zmono_window32 r02" --workload zernike_fresnel
ab z_f32_rec   "product zmono_window32 r02" --workload zernike
ab z_f32_spot  "product zmono_window32 r02" --workload zernike --mode spot
ab zf_f32_gen  "product zmono_window32" --workload zernike_fresnel --mode gen
python tools/ab_summary.py $OUT
python tools/gpu_r03_dropin.py > gpurun_out/r03_dropin.log 2>&1
python -c "
import json; d=json.load(open('gpurun_out/r03_dropin.json'))
print(json.dumps(d['reference_analyses_cooke_fp64']['with_seams']))"
