#!/bin/bash
# Round 4, GPU call 2: parity suite (hardware-seed ray generator / reference sphere, math probe),
# STEADY-STATE A/Bs (launches queued back to back, 150 warm-up + 60 timed: past the power
# transient of profiles/r04_clock_transient.txt), SQ counters per kernel, bench lines.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
(hostname; rocm-smi --showserial --showproductname 2>/dev/null | grep -iE "serial|card series|GPU\[0\]" | head -4) > gpurun_out/r04_box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 | tee gpurun_out/r04_pytest_gpu_2.log
OUT=$R/gpurun_out/r04_ab_steady.txt; : > $OUT
run() { local label=$1 lib=$2; shift 2
  echo -n "$label   " >> $OUT
  if [ -n "$lib" ]; then
    OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$lib.so \
      timeout 120 python tools/ab_kernel.py --sustained --warmup 150 --steps 60 "$@" 2>/dev/null | tail -1 >> $OUT
  else
    timeout 120 python tools/ab_kernel.py --sustained --warmup 150 --steps 60 "$@" 2>/dev/null | tail -1 >> $OUT
  fi
  echo >> $OUT
}
order() { if [ $(($1 % 2)) -eq 1 ]; then echo "${@:2}"; else echo "${@:2}" | tr ' ' '\n' | tac | tr '\n' ' '; fi; }
ab() { local tag=$1 arms=$2; shift 2
  for rep in 1 2; do
    for v in $(order $rep $arms); do run "$tag $v" "${v/product/}" "$@"; done
  done
}
echo "# $(date -u) interleaved A/B in steady state (150 queued warm-up launches, 60 timed), 1e7 rays" >> $OUT
ab zf_f32_gen  "product raygen_ieee nr32_1 nr32_2 polnr_waves0" --workload zernike_fresnel --mode gen
ab zf_f64_gen  "product raygen_ieee" --workload zernike_fresnel --mode gen --dtype f64
ab dg_f32_gen  "product raygen_ieee" --mode gen
ab dg_f64_gen  "product raygen_ieee" --mode gen --dtype f64
ab dg_f64_spot "product raygen_ieee" --mode spot --dtype f64
ab dg_f32_spot "product raygen_ieee" --mode spot
ab dg_opd      "product raygen_ieee" --mode opd
ab z_opd       "product raygen_ieee" --workload zernike --mode opd
ab rc_f32_gen  "product raygen_ieee" --workload rc_asphere --mode gen
python tools/ab_summary.py $OUT | tee gpurun_out/r04_ab_steady_summary.txt
# SQ counters + rocprof durations per kernel
bash tools/gpu_kernel_table.sh $R/gpurun_out/r04_kernel_table.txt > /dev/null 2>&1 <<'CFG'
dg_f32_gen  |
zf_f32_gen  | --workload zernike_fresnel
zf_f64_gen  | --workload zernike_fresnel --dtype f64
rc_f32_gen  | --workload rc_asphere
dg_f64_gen  | --dtype f64
dg_f32_spot | --mode spot
dg_f64_spot | --mode spot --dtype f64
dg_opd      | --mode opd
z_opd       | --workload zernike --mode opd
CFG
python - <<'PY'
import json
print(f"{'tag':<12} {'kernel':<58} {'us':>8} {'VALU/ray':>9} {'SALU/ray':>9} {'SMEM/ray':>8} {'issue_ms':>8} {'movedGB':>8} {'TB/s':>6} {'frac':>6}")
for ln in open("gpurun_out/r04_kernel_table.txt"):
    if not ln.startswith('{'): continue
    r=json.loads(ln)
    print(f"{r['tag']:<12} {r.get('kernel','?')[:58]:<58} {r.get('avg_us',0):8.1f} {r.get('VALU_per_ray',0):9.0f} {r.get('SALU_per_ray',0):9.0f} {r.get('SMEM_per_ray',0):8.0f} {r.get('valu_issue_ms',0):8.3f} {r.get('moved_GB',0):8.3f} {r.get('TBps_moved',0):6.2f} {r.get('frac',0):6.3f}")
PY
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/r04_bench_default_2.json
timeout 200 python bench.py --workload zernike_fresnel --warmup 150 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_c5_steady.json
timeout 200 python bench.py --workload zernike_fresnel --dtype f64 --warmup 150 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_c5_f64_steady.json
timeout 200 python bench.py --workload rc_asphere --warmup 150 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_c4_steady.json
python - <<'PY'
import json
for f in ("r04_bench_default_2","r04_bench_c5_steady","r04_bench_c5_f64_steady","r04_bench_c4_steady"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); r=d["roofline"]
        print(f, "value=%.4g ms/step=%.4f kernel_ms=%.4f frac=%.3f minmax=%s steady=%s fill=%s ceil=%s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("kernel_us_minmax"), json.dumps(r.get("steady_state")), r.get("stream_fill_GBps"), r.get("frac_of_write_ceiling")))
    except Exception as e: print(f, "failed", e)
PY
