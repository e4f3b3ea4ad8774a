#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x 2>&1 | tail -2
timeout 600 bash tools/gpu_dist1.sh 2>&1 | tee gpurun_out/r04_dist1.txt | tail -9
timeout 300 python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r04_bench_default.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_bench_default.json")); r=d["roofline"]; p=r.get("record_placement") or {}
print("default value=%.4g ms/step=%.4f kernel_ms=%.4f frac=%.3f steady=%s placed=%s arenas=%s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], (r.get("steady_state") or {}).get("kernel_ms"), p.get("placed"), p.get("arenas_tried")))
print(json.dumps(d.get("cpu_baseline"))[:400])
PY
