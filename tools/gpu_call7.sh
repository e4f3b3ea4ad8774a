#!/bin/bash
# Round 3, closing GPU pass: parity suite on the final library, closing A/B against the
# round-2 library, drop-in end to end, table create / update latency, default bench line.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/r03d_pytest_gpu.log
tail -3 gpurun_out/r03d_pytest_gpu.log
python tools/gpu_create_lat.py 2>&1 | tee gpurun_out/r03_create_update_latency.txt
python tools/gpu_r03_dropin.py > gpurun_out/r03_dropin.log 2>&1
grep -A8 set_radius gpurun_out/r03_dropin.json
grep -B2 -A12 with_seams gpurun_out/r03_dropin.json | head -30
ROUNDS=2 bash tools/gpu_ab_final_r03.sh > /dev/null 2>&1
cat gpurun_out/r03_ab_final_summary.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/r03_bench_default.json
cut -c1-300 gpurun_out/r03_bench_default.json
