#!/bin/bash
# Final tree of round 2: the whole GPU suite, smoke, the default bench line, C5 / C4 bench
# lines, rocprofv3 stats + HBM counters of the C5 and C4 kernels, fused-spot A/B of the
# single-family kernels.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out; mkdir -p $OUT
timeout 170 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4 | tee $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -i smoke
python bench.py 2>/dev/null | tail -1 > $OUT/bench_default.json; cut -c1-600 $OUT/bench_default.json
for w in zernike_fresnel rc_asphere; do python bench.py --workload $w --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_$w.json; cut -c1-330 $OUT/bench_$w.json; done
bash tools/gpu_prof.sh zf_f32 --workload zernike_fresnel > /dev/null 2>&1
bash tools/gpu_prof.sh rc_f32 --workload rc_asphere > /dev/null 2>&1
for t in zf_f32 rc_f32; do grep -E "trace_kernel" $OUT/prof_$t/summary.txt | cut -c1-260; done
for m in family generic generic family; do
  if [ $m = generic ]; then export OPTILAND_HIP_NR_FAMILY=0; else unset OPTILAND_HIP_NR_FAMILY; fi
  for w in zernike_fresnel rc_asphere; do
    echo -n "spot $w $m " >> $OUT/ab_nrfam_spot.txt
    python bench.py --workload $w --mode spot --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ms_per_step=%.4f value=%.4g'%(d['ms_per_step'], d['value']))" >> $OUT/ab_nrfam_spot.txt
  done
done
unset OPTILAND_HIP_NR_FAMILY
cat $OUT/ab_nrfam_spot.txt
