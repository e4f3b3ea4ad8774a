#!/bin/bash
# Round 3, third GPU pass: parity suite, the reference's own consumer tests stock vs drop-in
# (+ seams), drop-in end to end (re-pack loop), profile of the seamed analyses.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r03c_pytest_gpu.log
tail -5 gpurun_out/r03c_pytest_gpu.log
timeout 900 python tools/gpu_ref_consumers.py > gpurun_out/r03_reference_consumers_on_device.txt 2>&1
cat gpurun_out/r03_reference_consumers_on_device.txt | cut -c1-400 | tail -12
python tools/gpu_r03_dropin.py > gpurun_out/r03_dropin.log 2>&1
grep -A8 set_radius gpurun_out/r03_dropin.json
cut -c1-160 gpurun_out/r03_analyses_profile.txt | head -120
