#!/bin/bash
# Round 4, closing GPU pass on one box: parity suite, smoke, the default bench line, its rocprofv3
# stats + PMC passes (profiles/traffic.json), the other workloads in steady state, the drop-in end
# to end, the one-rank exchange overhead.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
(hostname; rocm-smi --showserial 2>/dev/null | grep -i "serial" | head -2) > gpurun_out/r04_box.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r04_pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee gpurun_out/r04_smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04_bench_default.json
bash tools/gpu_prof.sh r04_dg_f32_gen > /dev/null 2>&1
bash tools/gpu_prof.sh r04_dg_f64_gen --dtype f64 > /dev/null 2>&1
bash tools/gpu_prof.sh r04_rc_f32_gen --workload rc_asphere > /dev/null 2>&1
bash tools/gpu_prof.sh r04_zf_f32_gen --workload zernike_fresnel > /dev/null 2>&1
timeout 200 python bench.py --dtype f64 --warmup 50 --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_dg_f64.json
timeout 200 python bench.py --workload zernike_fresnel --warmup 150 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_c5_steady.json
timeout 200 python bench.py --workload zernike_fresnel --dtype f64 --warmup 150 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_c5_f64_steady.json
timeout 200 python bench.py --workload rc_asphere --warmup 150 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_c4_steady.json
timeout 300 python bench.py --config c3 --no-cpu-baseline --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/r04_bench_c3_n1.json
timeout 900 python tools/gpu_r04_dropin.py > gpurun_out/r04_dropin.log 2>&1; tail -2 gpurun_out/r04_dropin.log
timeout 600 bash tools/gpu_dist1.sh 2>&1 | tee gpurun_out/r04_dist1.txt | tail -9
python - <<'PY'
import json
for f in ("r04_bench_default","r04_bench_dg_f64","r04_bench_c5_steady","r04_bench_c5_f64_steady","r04_bench_c4_steady","r04_bench_c3_n1"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); r=d["roofline"]; p=r.get("record_placement") or {}
        print(f, "value=%.4g ms/step=%.4f kernel_ms=%.4f frac=%.3f steady=%s ceil=%s | placed=%s probe best/med=%.0f/%.0f plain_ms=%s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], (r.get("steady_state") or {}).get("kernel_ms"), r.get("frac_of_write_ceiling"), p.get("placed"), p.get("probe_best_GBps") or 0, p.get("probe_median_GBps") or 0, p.get("kernel_ms_plain_block")))
    except Exception as e: print(f, "failed", e)
d=json.load(open("gpurun_out/r04_dropin.json"))
print(json.dumps(d.get("reference_analyses_cooke_fp64_with_seams")))
d=json.load(open("gpurun_out/r04_bench_default.json")); print(json.dumps(d.get("dropin")), json.dumps(d.get("cpu_baseline"))[:300])
PY
cut -c1-200 gpurun_out/prof_r04_dg_f32_gen/summary.txt | grep -v "at::native" | head -12
