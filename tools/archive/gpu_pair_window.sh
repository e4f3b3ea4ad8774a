#!/bin/bash
# record-mode line (rays read from planes) with one ray per lane vs one packed PAIR per lane
# (OL_TRACE_RPT=3, v_pk_*_f32), in the driver's 5 + 20 window and in steady state, placed block.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
out=gpurun_out/r04_pair_window.txt
echo "# bench.py --mode record --steps 20 --warmup 5 --no-cpu-baseline, OL_TRACE_RPT = 1 | 3 (alternating, fresh processes)" > $out
for k in 1 2 3; do
 for rpt in 1 3; do
  OL_TRACE_RPT=$rpt timeout 300 python bench.py --mode record --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/b.json
  python - "$rpt" <<'PY' >> $out
import json, sys
d = json.load(open("/tmp/b.json")); r = d["roofline"]; p = r.get("record_placement") or {}; s = r.get("steady_state") or {}
print(f"rpt={sys.argv[1]}: ms/step={d['ms_per_step']:.4f} kernel_ms(window)={r['kernel_ms']:.4f} first5={[round(v) for v in r['kernel_us_each'][:5]]} last5={[round(v) for v in r['kernel_us_each'][-5:]]} steady={s.get('kernel_ms', 0):.4f} placed={p.get('placed')} kernel={r.get('kernel', '')[:60]}")
PY
 done
done
cat $out
