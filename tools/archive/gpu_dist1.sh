#!/bin/bash
# torchrun path on ONE rank with the image-plane exchange forced (RCCL path check + overhead)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
show() { grep "^{" | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); e=d.get('exchange') or {}; print('  ', d['config']['workload'].split(' (')[0], d['config']['mode'], d['dtype'], 'exchange='+d['config']['exchange'], 'ms/step=%.4f kernel_ms=%.4f value=%.4g' % (d['ms_per_step'], d['roofline']['kernel_ms'], d['value']), ('| with=%.4f without=%.4f exchange_ms_per_step=%.4f other(%s)=%.4f' % (e['ms_per_step_with'], e['ms_per_step_without'], e['exchange_ms_per_step'], (e.get('other') or {}).get('kind','-')[:14], (e.get('other') or {}).get('exchange_ms_per_step', float('nan')))) if e else '')"; }
run() { python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 1 --steps 60 --warmup 20 --settle 0 --no-cpu-baseline "${@:2}" 2>&1 | show; }
echo "plain (no process group)"; python bench.py --steps 60 --warmup 20 --settle 0 --no-cpu-baseline 2>/dev/null | show
run 29531 --exchange none
run 29532 --force-exchange --exchange reduce
run 29533 --force-exchange --exchange gather
run 29534 --force-exchange --dtype f64 --rays 1.25e7
run 29535 --force-exchange --mode spot
run 29536 --force-exchange --mode last
run 29537 --force-exchange --workload zernike_fresnel
