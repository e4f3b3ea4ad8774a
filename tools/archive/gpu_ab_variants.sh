#!/bin/bash
# GPU box: interleaved A/B of the product library against compile-time variants
# (tools/build_variants.py) and of the run-time knobs (rays per lane, __shfl compaction),
# ONE box, alternating runs, so that box-to-box and DVFS drift hit every arm equally.
# Output: gpurun_out/ab_variants.txt  (copied to profiles/r02_ab_*.txt by hand).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/ab_variants.txt; mkdir -p $R/gpurun_out; : > $OUT
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   kernel_ms=%.4f moved=%.0f GB/s frac=%.3f'%(r['kernel_ms'],r['achieved'],r['frac']))"; }
run() { # label, lib ('' = product), bench args...
  local label=$1 lib=$2; shift 2
  echo -n "$label" >> $OUT
  if [ -n "$lib" ]; then
    OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$lib.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | show >> $OUT
  else
    python bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | show >> $OUT
  fi
}
echo "# $(date -u) interleaved A/B, 1e7 rays, kernel_ms = mean HIP-event time of 30 launches" >> $OUT
# arm order alternates from round to round: a fixed order favours the arm that runs second
# by 1-2 % on these boxes (methodological note)
order() { if [ $(($1 % 2)) -eq 1 ]; then echo "${@:2}"; else echo "${@:2}" | tr ' ' '\n' | tac | tr '\n' ' '; fi; }
for rep in 1 2 3 4; do
  echo "## round $rep: double Gauss fp32 record-all" >> $OUT
  for v in $(order $rep product lds_table plain_stores block128 block512); do run "dg_f32_record $v" "${v/product/}"; done
  echo "## round $rep: double Gauss fp32 record-last" >> $OUT
  for v in $(order $rep product lds_table nt_vector); do run "dg_f32_last $v" "${v/product/}" --mode last; done
  echo "## round $rep: Zernike + Fresnel fp32 record-all" >> $OUT
  for v in $(order $rep product lds_table); do run "zf_f32_record $v" "${v/product/}" --workload zernike_fresnel; done
  echo "## round $rep: RC + asphere fp32 record-all" >> $OUT
  for v in $(order $rep product lds_table); do run "rc_f32_record $v" "${v/product/}" --workload rc_asphere; done
done
echo "# run-time knobs, in-process interleaved (tools/ab_bench.py): vector+compaction / vector / one ray per lane" >> $OUT
for w in rc_asphere zernike_fresnel; do
  python tools/ab_bench.py --workload $w --dtype f32 --mode record --rounds 5 >> $OUT 2>&1
done
python tools/ab_bench.py --system-json tests/golden/aspheric_singlet.json --hy 0.0 --tol 1e-12 --dtype f32 --mode last --rounds 5 >> $OUT 2>&1
python tools/ab_bench.py --system-json tests/golden/nr_family.json --hy 0.0 --dtype f32 --mode last --rounds 5 >> $OUT 2>&1
python tools/ab_bench.py --workload double_gauss --dtype f32 --mode record --rounds 5 >> $OUT 2>&1
python tools/ab_bench.py --workload double_gauss --dtype f64 --mode record --rounds 5 >> $OUT 2>&1
python tools/ab_bench.py --workload double_gauss --dtype f32 --mode last --rounds 5 >> $OUT 2>&1
cat $OUT
