#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 300 python tools/gpu_block_placement.py f32 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_block_placement_f32.txt
timeout 300 python tools/gpu_block_placement.py f64 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_block_placement_f64.txt
timeout 300 python -m pytest tests/test_gpu_math_probe.py -m gpu -q 2>&1 | tail -3
cat gpurun_out/math_probe.json | tr -d '\n' | cut -c1-900
