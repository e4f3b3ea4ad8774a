#!/bin/bash
# Round 3: interleaved A/B on ONE box, product library (phase-local table / kernarg reads,
# windowed coefficient stream, host-formed generator constants) against the round-2 library
# built from its commit (variant_r02.so) and the compile-time variants that isolate each
# change.  kernel_ms = mean HIP-event time of the bench's 20 launches at 1e7 rays.
# Output: gpurun_out/r03_ab_spills.txt
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/${1:-r03_ab_spills.txt}; mkdir -p $R/gpurun_out; : > $OUT
ROUNDS=${ROUNDS:-3}
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   kernel_ms=%.4f moved=%.0f GB/s frac=%.3f'%(r['kernel_ms'],r['achieved'],r['frac']))" 2>/dev/null || echo "   FAILED"; }
run() { # label, lib ('' = product), bench args...
  local label=$1 lib=$2; shift 2
  echo -n "$label" >> $OUT
  if [ -n "$lib" ]; then
    OPTILAND_HIP_ALLOW_ABI5=1 OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$lib.so python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | show >> $OUT
  else
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | show >> $OUT
  fi
}
order() { if [ $(($1 % 2)) -eq 1 ]; then echo "${@:2}"; else echo "${@:2}" | tr ' ' '\n' | tac | tr '\n' ' '; fi; }
echo "# $(date -u) interleaved A/B, 1e7 rays, arms alternate order every round" >> $OUT
ab() { # tag, arms, bench args
  local tag=$1 arms=$2; shift 2
  for rep in $(seq 1 $ROUNDS); do
    for v in $(order $rep $arms); do run "$tag $v" "${v/product/}" "$@"; done
  done
}
ab zf_f32_rec   "product r02 polnr_waves0 zmono_nosplit nr_byvalue" --workload zernike_fresnel
ab zf_f64_rec   "product r02 zmono_nosplit nr_byvalue" --workload zernike_fresnel --dtype f64
ab z_f32_rec    "product r02" --workload zernike
ab z_f64_rec    "product r02 zmono_nosplit" --workload zernike --dtype f64
ab rc_f32_rec   "product r02" --workload rc_asphere
ab rc_f64_rec   "product r02" --workload rc_asphere --dtype f64
ab z_f32_spot   "product r02" --workload zernike --mode spot
ab z_f64_spot   "product r02 zmono_nosplit" --workload zernike --mode spot --dtype f64
ab rc_f32_spot  "product r02" --workload rc_asphere --mode spot
ab rc_f64_spot  "product r02" --workload rc_asphere --mode spot --dtype f64
ab dg_f32_spot  "product r02 leanspot_byvalue" --mode spot
ab dg_f64_spot  "product r02 f64_byvalue" --mode spot --dtype f64
ab z_opd        "product r02 zmono_nosplit" --workload zernike --mode opd
ab rc_opd       "product r02" --workload rc_asphere --mode opd
ab dg_opd       "product r02 f64_byvalue" --mode opd
ab dg_f32_rec   "product r02" 
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
        print(f"{t:<12} {a:<18} {med:8.4f}  {' '.join('%.4f'%x for x in v)}   {med/base:6.3f}")
PY
