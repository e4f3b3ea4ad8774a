#!/bin/bash
# closing pass c: two lines of tools/gpu_dist1.sh ran ~2.7x slow (DoubleGauss gen f32 under
# torchrun 1.65 ms, C5 with the exchange 0.85 ms).  The same lines again, with the placement
# record and the per-launch times.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
out=gpurun_out/r04_anomaly.txt
: > $out
show() { grep "^{" | tail -1 > /tmp/b.json; python - "$1" <<'PY' >> gpurun_out/r04_anomaly.txt
import json, sys
d = json.load(open("/tmp/b.json")); r = d["roofline"]; p = r.get("record_placement") or {}
e = r["kernel_us_each"]
print(f"{sys.argv[1]}: ms/step={d['ms_per_step']:.4f} kernel_ms={r['kernel_ms']:.4f} first5={[round(v) for v in e[:5]]} mid5={[round(v) for v in e[28:33]]} last5={[round(v) for v in e[-5:]]} placed={p.get('placed')} arenas={p.get('arenas_tried')} probes={p.get('probes')} best/med={p.get('probe_best_GBps', 0):.0f}/{p.get('probe_median_GBps', 0):.0f}")
PY
}
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 --steps 60 --warmup 20 --settle 0 --no-cpu-baseline "${@:2}" 2>&1 | show "torchrun ${*:2}"; }
for k in 1 2 3; do
  python bench.py --steps 60 --warmup 20 --settle 0 --no-cpu-baseline 2>/dev/null | show "plain"
  run 2953$k --exchange none
  run 2954$k --force-exchange --workload zernike_fresnel
done
cat $out
