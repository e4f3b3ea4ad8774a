#!/bin/bash
# GPU box (round 3 opener): interleaved A/B of the hot-block prefetch policy on the kernels
# it has not been measured on -- fp64 record-all (C3's per-GPU shard), fp64 fused spot, the
# fused spot on a Newton system.  Build the arms first (build container):
#     python tools/build_variants.py noprefetch_f64 noprefetch_fused_nr nr_prefetch
# Arm order alternates (ABBA); 30 launches per arm.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/ab_prefetch.txt; mkdir -p $R/gpurun_out; : > $OUT
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('   ms_per_step=%.4f kernel_ms=%s value=%.4g'%(d['ms_per_step'], r.get('kernel_ms'), d['value']))"; }
arm() {  # arm <label> <variant|product> <bench args...>
  local label=$1 v=$2; shift 2
  echo -n "$label $v" >> $OUT
  if [ "$v" = product ]; then python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | show >> $OUT
  else OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$v.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | show >> $OUT; fi
}
echo "# $(date -u) hot-block prefetch policy, 1e7 rays, 30 launches per arm, arm order alternating" >> $OUT
for rep in 1 2; do
  for pair in "dg_f64_record noprefetch_f64 --dtype f64" "dg_f64_spot noprefetch_f64 --dtype f64 --mode spot" \
              "rc_f32_spot noprefetch_fused_nr --workload rc_asphere --mode spot" \
              "rc_f32_record nr_prefetch --workload rc_asphere" "zf_f32_record nr_prefetch --workload zernike_fresnel"; do
    set -- $pair; label=$1; v=$2; shift 2
    arm $label product "$@"; arm $label $v "$@"; arm $label $v "$@"; arm $label product "$@"
  done
done
cat $OUT
