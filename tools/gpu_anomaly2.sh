#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
out=gpurun_out/r04_anomaly2.txt
: > $out
show() { grep "^{" | tail -1 > /tmp/b.json; python - "$1" <<'PY' >> gpurun_out/r04_anomaly2.txt
import json, sys
d = json.load(open("/tmp/b.json")); r = d["roofline"]; p = r.get("record_placement") or {}
e = r["kernel_us_each"]
print(f"{sys.argv[1]}: ms/step={d['ms_per_step']:.4f} kernel_ms={r['kernel_ms']:.4f} first5={[round(v) for v in e[:5]]} last5={[round(v) for v in e[-5:]]} placed={p.get('placed')} arenas={p.get('arenas_tried')}")
PY
}
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 $2 --gpus 1 $FLAGS --no-cpu-baseline "${@:3}" 2>&1 | show "torchrun $2 ${*:3}"; }
for FLAGS in "--steps 60 --warmup 20 --settle 0" "--steps 60 --warmup 20" "--steps 20 --warmup 5 --settle 0" "--steps 20 --warmup 20"; do
  echo "## $FLAGS" >> $out
  run 29531 bench.py --exchange none
  run 29541 tools/_bench_noempty.py --exchange none
  run 29551 bench.py --exchange none --placement plain
  OL_TRACE_RPT=1 run 29561 bench.py --exchange none
done
cat $out
