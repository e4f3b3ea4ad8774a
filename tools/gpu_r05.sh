#!/bin/bash
# Round 5: ONE parameterised runner for the GPU box (`gpurun -- 'bash tools/gpu_r05.sh <leg> ...'`).
# Every leg writes under gpurun_out/; what is judged is copied to profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for leg in "$@"; do
  echo "=== leg $leg ($(date +%T))"
  case $leg in
    regions)   # the map behind fast record blocks (tools/microbench/vmm_regions.hip)
      timeout 420 tools/microbench/vmm_regions ${REGION_GIB:-240} > $O/r05_vmm_regions.txt 2>&1; echo "rc=$?" >> $O/r05_vmm_regions.txt
      tail -5 $O/r05_vmm_regions.txt ;;
    pairs)     # which two half-block pieces make a fast block (tools/microbench/vmm_pairs.hip)
      timeout 420 tools/microbench/vmm_pairs ${PAIRS_GIB:-230} > $O/r05_vmm_pairs.txt 2>&1; echo "rc=$?" >> $O/r05_vmm_pairs.txt
      tail -12 $O/r05_vmm_pairs.txt ;;
    debugfs)   # can the box show where a buffer lies physically?
      (mount -t debugfs none /sys/kernel/debug 2>&1; ls /sys/kernel/debug/dri/ 2>&1 | head; ls /sys/class/kfd/kfd/topology/nodes/ 2>&1;
       cat /sys/module/amdgpu/version 2>&1; uname -r; cat /sys/class/drm/card*/device/mem_info_vram_total 2>&1 | head -3;
       ls /sys/class/drm/card*/device/ 2>&1 | tr '\n' ' ' | head -c 3000; echo;
       cat /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition 2>&1 | head) > $O/r05_debugfs.txt 2>&1
      head -40 $O/r05_debugfs.txt ;;
    suite)     # the whole -m gpu suite (RCCL test included when OPTILAND_TEST_RCCL=1)
      timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r05_suite_${TAG:-0}.log 2>&1
      echo "rc=$?" >> $O/r05_suite_${TAG:-0}.log; tail -4 $O/r05_suite_${TAG:-0}.log ;;
    rccl_in_suite)  # the suite with the RCCL test on, the worker's phase stamps + stacks logged
      export OPTILAND_TEST_RCCL=1 OPTILAND_TEST_RCCL_WAIT=${RCCL_WAIT:-150}
      export OPTILAND_RCCL_WORKER_LOG=$O/r05_rccl_worker_${TAG:-0}.log
      export NCCL_DEBUG=INFO NCCL_DEBUG_FILE=$O/r05_rccl_nccl_${TAG:-0}.%p.log
      timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r05_suite_rccl_${TAG:-0}.log 2>&1
      echo "rc=$?" >> $O/r05_suite_rccl_${TAG:-0}.log; tail -6 $O/r05_suite_rccl_${TAG:-0}.log
      unset NCCL_DEBUG NCCL_DEBUG_FILE OPTILAND_RCCL_WORKER_LOG ;;
    rccl_alone)
      OPTILAND_TEST_RCCL=1 OPTILAND_RCCL_WORKER_LOG=$O/r05_rccl_worker_alone.log timeout 400 \
        python -m pytest tests/test_gpu_rccl_one_rank.py -m gpu -q -p no:cacheprovider > $O/r05_rccl_alone.log 2>&1
      tail -3 $O/r05_rccl_alone.log ;;
    window_pmc)
      bash tools/gpu_window_pmc.sh > $O/r05_window_pmc.log 2>&1; tail -30 $O/r05_window_pmc.log ;;
    bench)
      python bench.py ${BENCH_ARGS:-} > $O/r05_bench_${TAG:-default}.json 2> $O/r05_bench_${TAG:-default}.err
      tail -c 1500 $O/r05_bench_${TAG:-default}.json ;;
    *) echo "unknown leg $leg" ;;
  esac
done
