#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee gpurun_out/r03f_pytest_gpu.log
python tools/gpu_pol_e2e.py 2>&1 | tail -16
OPTILAND_HIP_FUSE_INTENSITY=0 python tools/gpu_pol_e2e.py 2>&1 | tail -16
python tools/gpu_dropin_matrix.py 2>&1 | grep -E "ZernikeFresnelPolarized|DoubleGauss" | cut -c1-400
