#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -x -k "pool or placed" 2>&1 | tail -12
timeout 600 python tools/gpu_pool_dropin.py 2>&1 | tee gpurun_out/r04_pool_dropin.txt | tail -8
