#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 400 python tools/gpu_window_scan.py f32 128 2 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_window_scan_f32.txt
timeout 400 python tools/gpu_window_scan.py f64 128 4 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04_window_scan_f64.txt
