#!/bin/bash
# Round 3: which FETCH LEVEL per kernel class (trace_kernel.hip: OL_FETCH_*).  Interleaved
# A/B on one box; arms = the product library, the round-2 library (variant_r02.so) and the
# compile-time variants of tools/build_variants.py.  kernel_ms = mean HIP-event time of 20
# launches at 1e7 rays (tools/ab_kernel.py).  Output: gpurun_out/r03_ab_fetch_levels.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/${1:-r03_ab_fetch_levels.txt}; mkdir -p $R/gpurun_out; : > $OUT
ROUNDS=${ROUNDS:-3}
run() { # label, lib ('' = product), args...
  local label=$1 lib=$2; shift 2
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
ab() { # tag, arms, args
  local tag=$1 arms=$2; shift 2
  for rep in $(seq 1 $ROUNDS); do
    for v in $(order $rep $arms); do run "$tag $v" "${v/product/}" "$@"; done
  done
}
ab zf_f32_rec   "product r02 nr32_0 nr32_1 nr32_2_waves0 nr32_0_waves0" --workload zernike_fresnel
ab z_f32_rec    "product r02 nr32_0 nr32_1" --workload zernike
ab rc_f32_rec   "product r02 nr32_0 nr32_1" --workload rc_asphere
ab z_f32_spot   "product r02 nr32_0 nr32_1" --workload zernike --mode spot
ab rc_f32_spot  "product r02 nr32_0 nr32_1" --workload rc_asphere --mode spot
ab zf_f64_rec   "product r02 nr64_1" --workload zernike_fresnel --dtype f64
ab z_f64_rec    "product r02 nr64_1" --workload zernike --dtype f64
ab rc_f64_rec   "product r02 nr64_1" --workload rc_asphere --dtype f64
ab z_f64_spot   "product r02 nr64_1" --workload zernike --mode spot --dtype f64
ab rc_f64_spot  "product r02 nr64_1" --workload rc_asphere --mode spot --dtype f64
ab z_opd        "product r02 nr64_1" --workload zernike --mode opd
ab rc_opd       "product r02 nr64_1" --workload rc_asphere --mode opd
ab dg_f32_spot  "product r02 lean32_1 lean32_2" --mode spot
ab dg_f64_spot  "product r02 lean64_0 lean64_2" --mode spot --dtype f64
ab dg_opd       "product r02 lean64_0 lean64_2" --mode opd
ab dg_f32_rec   "product r02"
ab dg_f32_gen   "product" --mode gen
ab dg_f64_rec   "product r02" --dtype f64
ab zf_f32_gen   "product" --workload zernike_fresnel --mode gen
python - "$OUT" <<'PY' | tee -a $OUT
import re, sys, collections
d = collections.defaultdict(list)
for ln in open(sys.argv[1]):
    m = re.match(r"(\S+) (\S+)\s+kernel_ms=([\d.]+)", ln)
    if m: d[(m.group(1), m.group(2))].append(float(m.group(3)))
print("# summary: config arm  median_ms  (all)   ratio to r02")
tags = []
for (t, a) in d:
    if t not in tags: tags.append(t)
for t in tags:
    base = sorted(d.get((t, "r02"), [float("nan")]))
    base = base[len(base)//2]
    for (tt, a), v in d.items():
        if tt != t: continue
        v2 = sorted(v); med = v2[len(v2)//2]
        print(f"{t:<12} {a:<16} {med:8.4f}  {' '.join('%.4f'%x for x in v)}   {med/base:6.3f}")
PY
