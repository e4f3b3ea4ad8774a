#!/bin/bash
# Round 4, GPU call 9: the placed record block in the bench line (all workloads), its test, the
# one-step fp64 quotient A/B.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_math_probe.py -m gpu -q -x 2>&1 | tail -4
timeout 300 python bench.py 2>gpurun_out/r04_bench_default_4.err | tail -1 > gpurun_out/r04_bench_default_4.json
tail -3 gpurun_out/r04_bench_default_4.err
timeout 300 python bench.py --placement plain --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_default_plain.json
timeout 200 python bench.py --dtype f64 --warmup 50 --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_dg_f64.json
timeout 200 python bench.py --workload zernike_fresnel --warmup 150 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_c5_steady.json
timeout 200 python bench.py --workload zernike_fresnel --dtype f64 --warmup 150 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_c5_f64_steady.json
timeout 200 python bench.py --workload rc_asphere --warmup 150 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_c4_steady.json
timeout 200 python bench.py --mode record --warmup 50 --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_record.json
python - <<'PY'
import json
for f in ("r04_bench_default_4","r04_bench_default_plain","r04_bench_dg_f64","r04_bench_c5_steady","r04_bench_c5_f64_steady","r04_bench_c4_steady","r04_bench_record"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); r=d["roofline"]; p=r.get("record_placement") or {}
        print(f, "value=%.4g ms/step=%.4f kernel_ms=%.4f frac=%.3f steady=%.4f fill=%s ceil=%s | placed=%s off=%.2fGiB probe best/med=%.0f/%.0f GB/s plain_ms=%s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], (r.get("steady_state") or {}).get("kernel_ms", float('nan')), r.get("stream_fill_GBps") and round(r["stream_fill_GBps"]), r.get("frac_of_write_ceiling") and round(r["frac_of_write_ceiling"],3), p.get("placed"), (p.get("window_offset_bytes") or 0)/2**30, p.get("probe_best_GBps") or 0, p.get("probe_median_GBps") or 0, p.get("kernel_ms_plain_block")))
    except Exception as e: print(f, "failed", e)
PY
OUT=$R/gpurun_out/r04_ab_div64.txt; : > $OUT
run() { local label=$1 lib=$2; shift 2
  echo -n "$label   " >> $OUT
  if [ -n "$lib" ]; then
    OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$lib.so timeout 120 python tools/ab_kernel.py --sustained --warmup 100 --steps 60 "$@" 2>/dev/null | tail -1 >> $OUT
  else
    timeout 120 python tools/ab_kernel.py --sustained --warmup 100 --steps 60 "$@" 2>/dev/null | tail -1 >> $OUT
  fi
  echo >> $OUT
}
for rep in 1 2; do
  for v in product div64_2steps; do
    run "dg_f64_spot $v" "${v/product/}" --mode spot --dtype f64
    run "dg_opd $v" "${v/product/}" --mode opd
    run "z_opd $v" "${v/product/}" --workload zernike --mode opd
  done
done
python tools/ab_summary.py $OUT
