#!/bin/bash
# How often does a fresh process find a fast window for the record block, and in which arena?
# N default bench runs in fresh processes (the driver's command line, without the CPU baseline).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
N=${1:-8}
out=gpurun_out/r04_placement_attempts.txt
echo "# $N x python bench.py --steps 20 --warmup 5 --no-cpu-baseline (fresh processes, one box)" > $out
for k in $(seq 1 $N); do
  extra=""
  # every other run after some allocator churn in ANOTHER process (a GPU test module)
  if [ $((k % 2)) -eq 0 ]; then timeout 300 python -m pytest tests/test_gpu_spot.py -m gpu -q -x > /dev/null 2>&1; extra="(after a pytest module)"; fi
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - "$k" "$extra" <<'PY' >> $out
import json, sys
d = json.load(open("/tmp/b.json")); r = d["roofline"]; p = r.get("record_placement") or {}
print(f"run {sys.argv[1]}: value={d['value']:.4g} ms/step={d['ms_per_step']:.4f} kernel_ms={r['kernel_ms']:.4f} "
      f"frac={r['frac']:.3f} placed={p.get('placed')} arenas_tried={p.get('arenas_tried')} probes={p.get('probes')} "
      f"best/median={p.get('probe_best_GBps', 0):.0f}/{p.get('probe_median_GBps', 0):.0f} GB/s "
      f"offset={p.get('window_offset_bytes', 0) / 2**30:.2f} GiB {sys.argv[2]}")
PY
done
cat $out
