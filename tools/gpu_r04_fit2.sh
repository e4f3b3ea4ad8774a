#!/bin/bash
# after the finishing-sum / grid change of ol_wavefront_fit: its tests and its cost again
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_wavefront_fit.py tests/test_wavefront.py tests/test_gpu_live_reference.py -m gpu -q 2>&1 | tail -5 | tee gpurun_out/r04_fit_pytest.log
timeout 300 python tools/gpu_fit_timing.py 2>&1 | tee gpurun_out/r04_fit_timing.txt
