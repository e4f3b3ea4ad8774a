#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
for k in 1 2 3; do
timeout 300 python bench.py --steps 20 --warmup 5 $( [ $k -gt 1 ] && echo --no-cpu-baseline ) 2>/dev/null | tail -1 > gpurun_out/r04_bench_default_$k.json
python - $k <<'PY'
import json, sys
d=json.load(open(f"gpurun_out/r04_bench_default_{sys.argv[1]}.json")); r=d["roofline"]; p=r.get("record_placement") or {}
print("default value=%.4g ms/step=%.4f kernel_ms=%.4f frac=%.3f steady=%s placed=%s arenas=%s first5=%s" % (d["value"], d["ms_per_step"], r["kernel_ms"], r["frac"], (r.get("steady_state") or {}).get("kernel_ms"), p.get("placed"), p.get("arenas_tried"), [round(v) for v in r["kernel_us_each"][:5]]))
PY
done
bash tools/gpu_prof.sh r04_dg_f32_gen > /dev/null 2>&1
cut -c1-200 gpurun_out/prof_r04_dg_f32_gen/summary.txt | grep -v "at::native" | head -12
timeout 300 bash tools/gpu_dist1.sh 2>&1 | tee gpurun_out/r04_dist1.txt | tail -9
