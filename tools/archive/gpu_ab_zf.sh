#!/bin/bash
# GPU box: interleaved A/B of library variants on the Zernike + Fresnel workload (C5).
# The order of the arms ALTERNATES from round to round (A B, B A, ...): back-to-back bench
# processes on one box showed a 1-2 % advantage for whichever arm ran second (r02: two arms
# with byte-identical kernels differed by 1.3 % in a fixed order).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/ab_zf.txt; mkdir -p $R/gpurun_out; : > $OUT
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   kernel_ms=%.4f moved=%.0f GB/s frac=%.3f'%(r['kernel_ms'],r['achieved'],r['frac']))"; }
arm() {
  local v=$1
  echo -n "zf_f32_record ${v:-product}" >> $OUT
  if [ -n "$v" ] && [ "$v" != "product" ]; then
    OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$v.so python bench.py --workload zernike_fresnel --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $OUT
  else
    python bench.py --workload zernike_fresnel --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $OUT
  fi
}
echo "# $(date -u) C5 (zernike_fresnel, 1e7 rays fp32 record-all), 30 launches per arm, arm order alternating" >> $OUT
ARMS=(product "$@")
for rep in 1 2 3 4 5 6; do
  if [ $((rep % 2)) -eq 1 ]; then for v in "${ARMS[@]}"; do arm "$v"; done
  else for ((i=${#ARMS[@]}-1; i>=0; i--)); do arm "${ARMS[$i]}"; done; fi
done
cat $OUT
