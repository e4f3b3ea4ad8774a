#!/bin/bash
# Round 4, the fitted-reference pipeline on the GPU: its tests, the whole -m gpu suite, the cost
# of the passes, the reference's OPD classes through the seam.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_wavefront_fit.py tests/test_wavefront.py -m gpu -q 2>&1 | tail -5 | tee gpurun_out/r04_fit_pytest.log
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r04_pytest_gpu_final.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke | tee gpurun_out/r04_smoke.log
timeout 300 python tools/gpu_fit_timing.py 2>&1 | tee gpurun_out/r04_fit_timing.txt
timeout 900 python tools/gpu_r04_dropin.py > gpurun_out/r04_dropin.log 2>&1; tail -3 gpurun_out/r04_dropin.log
