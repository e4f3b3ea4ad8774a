#!/bin/bash
# GPU box: interleaved A/B of the addressing form (product: SGPR base + shared lane offset;
# variant_vaddr: 64-bit per-lane addresses) on every bench workload.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/ab_addr.txt; mkdir -p $R/gpurun_out; : > $OUT
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   kernel_ms=%.4f moved=%.0f GB/s frac=%.3f'%(r['kernel_ms'],r['achieved'],r['frac']))"; }
echo "# $(date -u) interleaved, 30 launches per arm, 1e7 rays" >> $OUT
# (arm order alternates per round, see gpu_ab_zf.sh)
for rep in 1 2 3 4; do
 for args in "--workload double_gauss" "--workload double_gauss --dtype f64" "--workload zernike_fresnel" "--workload rc_asphere" "--workload double_gauss --mode last" "--workload double_gauss --mode spot"; do
  if [ $((rep % 2)) -eq 1 ]; then arms="product vaddr"; else arms="vaddr product"; fi
  for v in $arms; do v=${v/product/}
    echo -n "[$args] ${v:-product}" >> $OUT
    if [ -n "$v" ]; then
      OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$v.so python bench.py $args --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $OUT
    else
      python bench.py $args --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $OUT
    fi
  done
 done
done
cat $OUT
