#!/bin/bash
# GPU box: interleaved A/B of library variants on the Zernike + Fresnel workload (C5)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/ab_zf.txt; mkdir -p $R/gpurun_out; : > $OUT
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   kernel_ms=%.4f moved=%.0f GB/s frac=%.3f'%(r['kernel_ms'],r['achieved'],r['frac']))"; }
echo "# $(date -u) C5 (zernike_fresnel, 1e7 rays fp32 record-all), interleaved, 30 launches each" >> $OUT
for rep in 1 2 3 4; do
  for v in "" "$@"; do
    echo -n "zf_f32_record ${v:-product}" >> $OUT
    if [ -n "$v" ]; then
      OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$v.so python bench.py --workload zernike_fresnel --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $OUT
    else
      python bench.py --workload zernike_fresnel --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $OUT
    fi
  done
done
cat $OUT
