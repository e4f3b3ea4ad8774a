#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
out=gpurun_out/r04_anomaly3.txt
echo "# after: + the cyclic GC collected and disabled around the timed region" > $out
show() { grep "^{" | tail -1 > /tmp/b.json; python - "$1" <<'PY' >> gpurun_out/r04_anomaly3.txt
import json, sys
d = json.load(open("/tmp/b.json")); r = d["roofline"]; p = r.get("record_placement") or {}
e = r["kernel_us_each"]
print(f"{sys.argv[1]}: ms/step={d['ms_per_step']:.4f} kernel_ms={r['kernel_ms']:.4f} first5={[round(v) for v in e[:5]]} last5={[round(v) for v in e[-5:]]} placed={p.get('placed')} arenas={p.get('arenas_tried')}")
PY
}
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 --steps 60 --warmup 20 --settle 0 --no-cpu-baseline "${@:2}" 2>&1 | show "torchrun ${*:2}"; }
for k in 1 2 3; do
  run 2953$k --exchange none
  run 2954$k --force-exchange --workload zernike_fresnel
  run 2955$k --force-exchange --exchange reduce
done
cat $out
