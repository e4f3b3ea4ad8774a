R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out/ab_prefetch_f64.txt; : > $OUT
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d.get('roofline') or {}; print('   ms_per_step=%.4f kernel_ms=%s'%(d['ms_per_step'], r.get('kernel_ms')))"; }
arm() { local label=$1 v=$2; shift 2; echo -n "$label $v" >> $OUT
  if [ "$v" = product ]; then python bench.py --steps 20 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | show >> $OUT
  else OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$v.so python bench.py --steps 20 --warmup 2 --no-cpu-baseline "$@" 2>/dev/null | show >> $OUT; fi; }
for v in product noprefetch_f64 noprefetch_f64 product; do arm dg_f64_spot $v --dtype f64 --mode spot; done
for v in product noprefetch_f64 noprefetch_f64 product; do arm dg_f64_record $v --dtype f64; done
cat $OUT
