#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
for ex in reduce gather; do
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --force-exchange --exchange $ex 2>&1 | tail -2 | cut -c1-600
done
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --dtype f64 --rays 1.25e7 --force-exchange 2>&1 | tail -1 | cut -c1-400
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29535 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline --mode spot --force-exchange 2>&1 | tail -1 | cut -c1-300
