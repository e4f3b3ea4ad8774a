#!/bin/bash
# Round 3, second GPU pass: parity suite on the rebuilt library (fp64 division / square root
# from the hardware seeds), which earlier test the fused-seam spot test depends on, the
# drop-in end to end again (re-pack loop), A/B of OL_FAST_F64, the write-schedule
# microbenchmark, the default bench line.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -70 > gpurun_out/r03b_pytest_gpu.log
tail -8 gpurun_out/r03b_pytest_gpu.log
T="tests/test_gpu_live_reference.py::test_spot_diagram_and_encircled_energy_through_the_fused_seam_on_device[float64-CookeTriplet]"
if grep -q "fused_seam_on_device" gpurun_out/r03b_pytest_gpu.log; then
  : > gpurun_out/r03b_bisect.txt
  for f in $(ls tests/test_*.py | sort); do
    [ "$f" = "tests/test_gpu_live_reference.py" ] && break
    grep -q "mark.gpu\|pytestmark" $f || continue
    r=$(timeout 300 python -m pytest $f "$T" -m gpu -q -p no:cacheprovider 2>&1 | tail -1)
    echo "$f :: $r" >> gpurun_out/r03b_bisect.txt
  done
  r=$(timeout 300 python -m pytest tests/test_gpu_live_reference.py -m gpu -q 2>&1 | tail -3 | tr '\n' ' ')
  echo "tests/test_gpu_live_reference.py alone :: $r" >> gpurun_out/r03b_bisect.txt
  cat gpurun_out/r03b_bisect.txt
fi
python tools/gpu_r03_dropin.py > gpurun_out/r03_dropin.log 2>&1
grep -A12 set_radius gpurun_out/r03_dropin.json | head -30
ROUNDS=3 bash tools/gpu_ab_fast64.sh
cat gpurun_out/r03_ab_fast64.txt
(cd tools/microbench && timeout 300 ./rw_schedule > $R/gpurun_out/r03_rw_schedule.txt 2>&1)
cat gpurun_out/r03_rw_schedule.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03_bench_default.json
python bench.py --dtype f64 --no-ref-baselines --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03_bench_f64_gen.json
cut -c1-400 gpurun_out/r03_bench_default.json
