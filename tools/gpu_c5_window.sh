cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for arm in 1 3; do
  OL_TRACE_RPT=$arm python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic committed --workload zernike_fresnel > gpurun_out/c5w_${arm}_${rep}.json 2>/dev/null
  python - gpurun_out/c5w_${arm}_${rep}.json $arm $rep <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]; st = r.get("steady_state") or {}
print(f"rpt={sys.argv[2]} rep={sys.argv[3]} ms_per_step {d['ms_per_step']:.4f} kernel_ms {r['kernel_ms']:.4f} {r['kernel_us_minmax']} frac {r['frac']:.3f} steady {st.get('kernel_ms')} {st.get('frac')}")
PY
done; done
