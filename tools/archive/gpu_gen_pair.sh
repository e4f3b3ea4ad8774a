#!/bin/bash
# The generating launch on packed pairs (trace_kernel<float, 2, true, 0, 0, false, kGenUniform>):
# parity, then the default bench line with one ray per lane (OL_TRACE_RPT=1) and with the pair
# (default), fresh processes alternating; placed and plain block; the drop-in's trace_generic.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_generate_fused.py tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py tests/test_gpu_live_reference.py -m gpu -q -x 2>&1 | tail -4 | tee gpurun_out/r04_gen_pair_pytest.log
out=gpurun_out/r04_gen_pair.txt
echo "# bench.py --steps 20 --warmup 5 --no-cpu-baseline [--placement plain], OL_TRACE_RPT=1 (one ray per lane) | default (pair)" > $out
for k in 1 2 3; do
 for rpt in 1 0; do
  for pl in probe plain; do
   OL_TRACE_RPT=$rpt timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --placement $pl 2>/dev/null | tail -1 > /tmp/b.json
   python - "$rpt" "$pl" <<'PY' >> $out
import json, sys
d = json.load(open("/tmp/b.json")); r = d["roofline"]; p = r.get("record_placement") or {}; s = r.get("steady_state") or {}
print(f"rpt={sys.argv[1]} {sys.argv[2]:5s}: value={d['value']:.4g} ms/step={d['ms_per_step']:.4f} kernel_ms(window)={r['kernel_ms']:.4f} first5={[round(v) for v in r['kernel_us_each'][:5]]} last5={[round(v) for v in r['kernel_us_each'][-5:]]} steady={s.get('kernel_ms', 0):.4f} placed={p.get('placed')} dropin_ms={(d.get('dropin') or {}).get('ms_per_call')}")
PY
  done
 done
done
cat $out
