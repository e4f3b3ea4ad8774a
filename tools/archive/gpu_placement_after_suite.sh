#!/bin/bash
# The sequence of the closing pass that once ended with placed=False: the whole -m gpu suite,
# smoke, then default bench runs in fresh processes.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
out=gpurun_out/r04_placement_after_suite.txt
echo "# pytest -m gpu (whole suite), smoke(), then 4 x python bench.py --steps 20 --warmup 5 --no-cpu-baseline" > $out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -1 >> $out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke >> $out
for k in 1 2 3 4; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - "$k" <<'PY' >> $out
import json, sys
d = json.load(open("/tmp/b.json")); r = d["roofline"]; p = r.get("record_placement") or {}
print(f"run {sys.argv[1]}: value={d['value']:.4g} ms/step={d['ms_per_step']:.4f} kernel_ms={r['kernel_ms']:.4f} "
      f"frac={r['frac']:.3f} placed={p.get('placed')} arenas_tried={p.get('arenas_tried')} probes={p.get('probes')} "
      f"best/median={p.get('probe_best_GBps', 0):.0f}/{p.get('probe_median_GBps', 0):.0f} GB/s "
      f"offset={p.get('window_offset_bytes', 0) / 2**30:.2f} GiB")
PY
done
cat $out
