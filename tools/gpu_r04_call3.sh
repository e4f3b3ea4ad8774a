#!/bin/bash
# Round 4, GPU call 3: parity suite (device pupil samplers, math probe), the drop-in end to end
# (per-ray fields / apodisation in one launch, first-call latency, object protocol on the device,
# analyses), one-rank exchange overhead with its breakdown, bench line with the store-pattern
# yardstick.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee gpurun_out/r04_pytest_gpu_3.log
timeout 900 python tools/gpu_r04_dropin.py > gpurun_out/r04_dropin.log 2>&1; tail -3 gpurun_out/r04_dropin.log
timeout 600 bash tools/gpu_dist1.sh 2>&1 | tee gpurun_out/r04_dist1.txt | tail -12
timeout 300 python bench.py 2>/dev/null | tail -1 > gpurun_out/r04_bench_default_3.json
timeout 200 python bench.py --workload zernike_fresnel --warmup 150 --steps 100 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_c5_steady.json
timeout 200 python bench.py --dtype f64 --warmup 50 --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r04_bench_dg_f64.json
python - <<'PY'
import json
for f in ("r04_bench_default_3","r04_bench_c5_steady","r04_bench_dg_f64"):
    try:
        d=json.load(open(f"gpurun_out/{f}.json")); r=d["roofline"]
        print(f, "value=%.4g ms/step=%.4f kernel_ms=%.4f frac=%.3f minmax=%s steady=%s fill=%s planes_ceiling=%s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], r.get("kernel_us_minmax"), json.dumps(r.get("steady_state")), r.get("stream_fill_GBps"), r.get("frac_of_write_ceiling")))
    except Exception as e: print(f, "failed", e)
d=json.load(open("gpurun_out/r04_dropin.json"))
for k in ("trace_generic_1e7_float32","trace_generic_1e7_float64","first_call_latency","object_protocol_on_device","reference_analyses_cooke_fp64_with_seams"):
    print(k, json.dumps(d.get(k)))
PY
cat gpurun_out/math_probe.json 2>/dev/null | tr -d '\n' | cut -c1-900
