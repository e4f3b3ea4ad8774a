#!/bin/bash
# Round 6: ONE parameterised runner for the GPU box (`gpurun -- 'bash tools/gpu_r06.sh <leg> ...'`).
# Every leg writes under gpurun_out/; what is judged is copied to profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; O=$R/gpurun_out; mkdir -p $O
export HSA_ENABLE_IPC_MODE_LEGACY=0
for leg in "$@"; do
  echo "=== leg $leg ($(date +%T))"
  case $leg in
    regions)   # the map behind fast record blocks (tools/microbench/vmm_regions.hip)
      timeout 420 tools/microbench/vmm_regions ${REGION_GIB:-240} > $O/r06_vmm_regions.txt 2>&1; echo "rc=$?" >> $O/r06_vmm_regions.txt
      tail -5 $O/r06_vmm_regions.txt ;;
    pairs)     # which two half-block pieces make a fast block (tools/microbench/vmm_pairs.hip)
      timeout 420 tools/microbench/vmm_pairs ${PAIRS_GIB:-230} > $O/r06_vmm_pairs.txt 2>&1; echo "rc=$?" >> $O/r06_vmm_pairs.txt
      tail -12 $O/r06_vmm_pairs.txt ;;
    junctions) # the smallest arena whose junction makes a fast block (tools/microbench/vmm_junctions.hip)
      timeout 300 tools/microbench/vmm_junctions > $O/r06_vmm_junctions.txt 2>&1; echo "rc=$?" >> $O/r06_vmm_junctions.txt
      cat $O/r06_vmm_junctions.txt | cut -c1-400 ;;
    split)     # a displacement between two groups of planes inside ONE allocation
      timeout 300 tools/microbench/split_offset > $O/r06_split_offset.txt 2>&1; echo "rc=$?" >> $O/r06_split_offset.txt
      grep -c . $O/r06_split_offset.txt ;;
    pace)      # evenly paced stores against rows of 8 back to back (tools/microbench/pace_probe.hip)
      timeout 300 tools/microbench/pace_probe > $O/r06_pace_probe.txt 2>&1; cat $O/r06_pace_probe.txt ;;
    lottery)   # how often a junction of two pieces makes a fast block, by the pieces' sizes
      timeout 600 tools/microbench/vmm_lottery ${LOTTERY_REPS:-6} > $O/r06_vmm_lottery.txt 2>&1; cat $O/r06_vmm_lottery.txt ;;
    debugfs)   # can the box show where a buffer lies physically?
      (mount -t debugfs none /sys/kernel/debug 2>&1; ls /sys/kernel/debug/dri/ 2>&1 | head; ls /sys/class/kfd/kfd/topology/nodes/ 2>&1;
       cat /sys/module/amdgpu/version 2>&1; uname -r; cat /sys/class/drm/card*/device/mem_info_vram_total 2>&1 | head -3;
       ls /sys/class/drm/card*/device/ 2>&1 | tr '\n' ' ' | head -c 3000; echo;
       cat /sys/class/drm/card*/device/current_memory_partition /sys/class/drm/card*/device/current_compute_partition 2>&1 | head) > $O/r06_debugfs.txt 2>&1
      head -40 $O/r06_debugfs.txt ;;
    suite)     # the whole -m gpu suite (RCCL test included when OPTILAND_TEST_RCCL=1)
      timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/r06_suite_${TAG:-0}.log 2>&1
      echo "rc=$?" >> $O/r06_suite_${TAG:-0}.log; tail -4 $O/r06_suite_${TAG:-0}.log ;;
    rccl_in_suite)  # the suite with the RCCL test on, the worker's phase stamps + stacks logged
      export OPTILAND_TEST_RCCL=1 OPTILAND_TEST_RCCL_WAIT=${RCCL_WAIT:-150}
      export OPTILAND_RCCL_WORKER_LOG=$O/r06_rccl_worker_${TAG:-0}.log
      export NCCL_DEBUG=INFO NCCL_DEBUG_FILE=$O/r06_rccl_nccl_${TAG:-0}.%p.log
      timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/r06_suite_rccl_${TAG:-0}.log 2>&1
      echo "rc=$?" >> $O/r06_suite_rccl_${TAG:-0}.log; tail -6 $O/r06_suite_rccl_${TAG:-0}.log
      unset NCCL_DEBUG NCCL_DEBUG_FILE OPTILAND_RCCL_WORKER_LOG ;;
    rccl_alone)
      OPTILAND_TEST_RCCL=1 OPTILAND_RCCL_WORKER_LOG=$O/r06_rccl_worker_alone.log timeout 400 \
        python -m pytest tests/test_gpu_rccl_one_rank.py -m gpu -q -p no:cacheprovider > $O/r06_rccl_alone.log 2>&1
      tail -3 $O/r06_rccl_alone.log ;;
    window_pmc)
      bash tools/gpu_window_pmc.sh > $O/r06_window_pmc.log 2>&1; tail -30 $O/r06_window_pmc.log ;;
    ab)        # in-process A/B: ARMS=product,arith_r04 CONFIGS=dg_f64_gen,zf_f32_gen [ROUNDS=2] [AB_EXTRA=--placed]
      timeout 600 python tools/ab_inproc.py --arms ${ARMS:-product} --configs ${CONFIGS:-dg_f32_gen} \
        --rounds ${ROUNDS:-2} ${AB_EXTRA:-} >> $O/r06_ab_${TAG:-0}.txt 2>&1; tail -${AB_TAIL:-12} $O/r06_ab_${TAG:-0}.txt ;;
    wgcap)     # resident-workgroup cap of the record launches: CAPS=0,2,3,4 CONFIGS=... [AB_EXTRA=--placed]
      timeout 600 python tools/ab_wgcap.py --caps ${CAPS:-0,2,3,4} --configs ${CONFIGS:-dg_f32_gen,dg_f64_gen} \
        --rounds ${ROUNDS:-2} ${AB_EXTRA:-} >> $O/r06_ab_wgcap_${TAG:-0}.txt 2>&1; tail -${AB_TAIL:-40} $O/r06_ab_wgcap_${TAG:-0}.txt ;;
    kernel_table) # rocprofv3 duration + SQ counters per ray of the dominant kernel, final library
      bash tools/gpu_kernel_table.sh $O/r06_kernel_table.txt > /dev/null 2>&1 <<'CFG'
dg_f32_gen  |
dg_f64_gen  | --dtype f64
rc_f32_gen  | --workload rc_asphere
zf_f32_gen  | --workload zernike_fresnel
zf_f64_gen  | --workload zernike_fresnel --dtype f64
dg_f64_spot | --mode spot --dtype f64
dg_opd      | --mode opd
CFG
      python - <<'PY'
import json
print(f"{'tag':<12} {'kernel':<58} {'us':>8} {'VALU/ray':>9} {'SALU/ray':>9} {'SMEM/ray':>8} {'issue_ms':>8} {'movedGB':>8} {'TB/s':>6} {'frac':>6}")
for ln in open("gpurun_out/r06_kernel_table.txt"):
    if not ln.startswith('{'): continue
    r=json.loads(ln)
    print(f"{r['tag']:<12} {r.get('kernel','?')[:58]:<58} {r.get('avg_us',0):8.1f} {r.get('VALU_per_ray',0):9.0f} {r.get('SALU_per_ray',0):9.0f} {r.get('SMEM_per_ray',0):8.0f} {r.get('valu_issue_ms',0):8.3f} {r.get('moved_GB',0):8.3f} {r.get('TBps_moved',0):6.2f} {r.get('frac',0):6.3f}")
PY
      ;;
    spotdiag)  # the reference's SpotDiagram through the seams: one launch per grid vs per cell
      timeout 600 python tools/gpu_spotdiag.py > $O/r06_spotdiag.txt 2>&1; tail -5 $O/r06_spotdiag.txt | cut -c1-600 ;;
    prof)      # rocprofv3 stats + PMC passes of the BASELINE configurations (tools/collect_profiles.py r06)
      rocm-smi --showserial > $O/r06_box.txt 2>&1
      bash tools/gpu_prof.sh r06_dg_f32_gen > $O/r06_prof.log 2>&1
      bash tools/gpu_prof.sh r06_dg_f64_gen --dtype f64 >> $O/r06_prof.log 2>&1
      bash tools/gpu_prof.sh r06_rc_f32_gen --workload rc_asphere >> $O/r06_prof.log 2>&1
      bash tools/gpu_prof.sh r06_zf_f32_gen --workload zernike_fresnel >> $O/r06_prof.log 2>&1
      grep -h "trace_kernel" $O/prof_r06_*/summary.txt | head -8 ;;
    benches)   # the other BASELINE configurations, steady state included (no baselines)
      for cfg in "dg_f64:--dtype f64" "c4_f32:--workload rc_asphere" "c4_f64:--workload rc_asphere --dtype f64" \
                 "c5_f32:--workload zernike_fresnel" "c5_f64:--workload zernike_fresnel --dtype f64" \
                 "dg_f64_spot:--mode spot --dtype f64" "dg_opd:--mode opd"; do
        tag=${cfg%%:*}; args=${cfg#*:}
        python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic committed $args \
          > $O/r06_bench_$tag.json 2> $O/r06_bench_$tag.err
        python - $O/r06_bench_$tag.json $tag <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    r = d["roofline"]; st = r.get("steady_state") or {}
    print(f"{sys.argv[2]:12s} value {d['value']:.4g} kernel_ms {r['kernel_ms']:.4f} ({r['kernel_us_minmax']}) frac {r['frac']:.3f}"
          f" steady {st.get('kernel_ms')} frac {st.get('frac')} placed {(r.get('record_placement') or {}).get('placed')}")
except Exception as e:
    print(sys.argv[2], "ERR", e)
PY
      done ;;
    smoke)     # what the driver runs before its bench
      timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" > $O/r06_smoke.log 2>&1; tail -4 $O/r06_smoke.log ;;
    exch)      # where the +0.05 ms per step of the reduce-first exchange go (one rank, RCCL)
      for dbg in none alwayswait nocoll; do
        OL_BENCH_EXCH_DEBUG=$dbg python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 \
          bench.py --gpus 1 --force-exchange --steps 40 --warmup 10 --settle 0 --no-cpu-baseline --no-ref-baselines --traffic committed \
          > $O/r06_exch_$dbg.json 2> $O/r06_exch_$dbg.err
        python - $O/r06_exch_$dbg.json $dbg <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1]); e = d["exchange"]
    print(f"{sys.argv[2]:8s} with {e['ms_per_step_with']:.4f} without {e['ms_per_step_without']:.4f} diff {e['exchange_ms_per_step']:.4f} kernel {d['roofline']['kernel_ms']:.4f}")
except Exception as ex:
    print(sys.argv[2], "ERR", ex)
PY
      done ;;
    bench1rank) # the N > 1 launch form with ONE rank: RCCL init, barrier, exchange legs, placement
      python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 \
        bench.py --gpus 1 --force-exchange --steps 20 --warmup 5 --no-cpu-baseline --no-ref-baselines --traffic committed \
        > $O/r06_bench_1rank_${TAG:-0}.json 2> $O/r06_bench_1rank_${TAG:-0}.err
      tail -c 1200 $O/r06_bench_1rank_${TAG:-0}.json; tail -3 $O/r06_bench_1rank_${TAG:-0}.err ;;
    c5_valu)   # SQ_INSTS_VALU per ray of the C5 launch against the Newton iteration cap (tools/gpu_c5_valu.py)
      OUTD=$O/prof_c5_valu; rm -rf $OUTD; mkdir -p $OUTD
      (cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUTD/a -o c5 -- \
        python $R/tools/gpu_c5_valu.py > $OUTD/a.log 2>&1)
      python - <<PY
import csv, glob
names = [l.split("VARIANT ", 1)[1].strip() for l in open("$O/prof_c5_valu/a.log") if l.startswith("VARIANT ")]
rows = []
for f in glob.glob("$O/prof_c5_valu/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0][9:], float(r["Counter_Value"])))
rows.sort()
L = 3
with open("$O/r06_c5_valu_${TAG:-0}.txt", "w") as out:
    for i, nm in enumerate(names):
        grp = rows[i * L:(i + 1) * L]
        if not grp:
            continue
        v = sum(g[2] for g in grp) / len(grp)
        line = f"{nm:52s} {grp[0][1]:56s} VALU/ray {v * 64 / 1e7:8.1f}"
        print(line)
        out.write(line + "\n")
PY
      tail -2 $OUTD/a.log ;;
    hot_loop)  # when the two-workgroup cap starts to pay in a loop (tools/gpu_hot_loop.py)
      timeout 600 python tools/gpu_hot_loop.py > $O/r06_hot_loop_${TAG:-0}.txt 2>&1; cat $O/r06_hot_loop_${TAG:-0}.txt | tail -20 ;;
    pool)      # the pool tests alone
      timeout 900 python -m pytest tests/test_gpu_fullsize.py -m gpu -q -p no:cacheprovider -k "pool or placed" > $O/r06_pool_${TAG:-0}.log 2>&1; tail -5 $O/r06_pool_${TAG:-0}.log ;;
    clock_transient)  # effective engine clock per launch (GRBM_GUI_ACTIVE / duration) + the part's own sysfs account
      OUTD=$O/clock_transient_${TAG:-0}; rm -rf $OUTD; mkdir -p $OUTD
      (cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUTD/a -o ct -- \
        python $R/tools/gpu_clock_transient.py > $OUTD/run.log 2>&1; echo "rc=$?" >> $OUTD/run.log)
      python $R/tools/clock_transient_report.py $OUTD > $O/r06_clock_transient_${TAG:-0}.txt 2>&1; head -60 $O/r06_clock_transient_${TAG:-0}.txt ;;
    ab6)       # in-process A/B of this round's kernel changes: ARMS, CONFIGS [AB_EXTRA=--placed]
      timeout 900 python tools/ab_inproc.py --arms ${ARMS:-product,o6_off} --configs ${CONFIGS:-zf_f32_gen,zf_f64_gen,rc_f32_gen} \
        --rounds ${ROUNDS:-2} ${AB_EXTRA:-} >> $O/r06_ab_${TAG:-0}.txt 2>&1; tail -${AB_TAIL:-40} $O/r06_ab_${TAG:-0}.txt ;;
    cycles_ab) # GRBM_GUI_ACTIVE per launch of the product against a variant library: cycles, not time (the clock moves)
      for arm in product ${VARIANTS:-o6_off}; do
        OUTD=$O/cycles_${arm}; rm -rf $OUTD; mkdir -p $OUTD
        if [ $arm != product ]; then export OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_${arm}.so; else unset OPTILAND_HIP_LIBRARY; fi
        (cd /tmp && TMPDIR=/tmp CONFIGS=${CONFIGS:-zf_f32,zf_f64} ROUNDS=2 LAUNCHES=40 timeout 600 \
          rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUTD/a -o ct -- \
          python $R/tools/gpu_clock_transient.py > $OUTD/run.log 2>&1; echo "rc=$?" >> $OUTD/run.log)
        python $R/tools/clock_transient_report.py $OUTD > $O/r06_cycles_${arm}.txt 2>&1
        python - $O/r06_cycles_${arm}.txt $arm <<'PY'
import re, sys
import numpy as np
cur, acc = None, {}
for ln in open(sys.argv[1]):
    m = re.match(r"## (\S+) round (\d+)", ln)
    if m:
        cur = m.group(1)
        continue
    m = re.match(r"\s+(\d+)\s+([\d.]+)\s+([\d.]+)\s+(\d+)\s+([\d.]+)", ln)
    if m and cur and int(m.group(1)) >= 10:
        acc.setdefault(cur, []).append((float(m.group(2)), float(m.group(4)), float(m.group(5))))
for k, v in acc.items():
    a = np.array(v)
    print(f"{sys.argv[2]:18s} {k:7s} launches 10-39: event_ms {a[:, 0].mean():.4f}  engine cycles per launch and XCD "
          f"{a[:, 1].mean() / 8:.4g}  clock {a[:, 2].mean() / 8:.0f} MHz")
PY
      done; unset OPTILAND_HIP_LIBRARY ;;
    c5_window) # the driver's window (5 + 20 launches) of C5 fp32, one ray per lane against pairs, arms alternating
      for rep in 1 2 3; do for arm in 1 3; do
        OL_TRACE_RPT=$arm python bench.py --steps 20 --warmup 5 --no-cpu-baseline --traffic committed --workload zernike_fresnel > $O/c5w_${arm}_${rep}.json 2>/dev/null
        python - $O/c5w_${arm}_${rep}.json $arm $rep <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
r = d["roofline"]; st = r.get("steady_state") or {}
print(f"rpt={sys.argv[2]} rep={sys.argv[3]} ms_per_step {d['ms_per_step']:.4f} kernel_ms {r['kernel_ms']:.4f} {r['kernel_us_minmax']} frac {r['frac']:.3f} steady {st.get('kernel_ms')} {st.get('frac')}")
PY
      done; done ;;
    polz_fuzz) # 1500 shaken C5 systems, pair form against one ray per lane, bits and status words
      timeout 1200 python tools/gpu_polz_fuzz.py 2>&1 | tail -5 ;;
    seam_profile) # wall time + cProfile of the reference's EE / OPD / FFTPSF / Optic.trace through the seams
      timeout 600 python tools/gpu_seam_profile.py 2>&1 | grep "ms per call" ;;
    valu_rate) # what a vector instruction costs (tools/microbench/valu_rate.hip, built beforehand)
      timeout 200 tools/microbench/valu_rate > $O/r06_valu_rate.txt 2>&1; grep "SIMD 8" $O/r06_valu_rate.txt ;;
    polz_table) # SQ counters per ray of the C5 fp32 launch on one ray per lane (OL_TRACE_RPT=1) and on pairs (3)
      for arm in 1 3; do
        OL_TRACE_RPT=$arm bash tools/gpu_kernel_table.sh $O/r06_polz_table_$arm.txt > /dev/null 2>&1 <<'CFG'
zf_f32_gen  | --workload zernike_fresnel
CFG
        cut -c1-900 $O/r06_polz_table_$arm.txt
      done ;;
    polz_pair) # C5 fp32 on two rays per lane: bits against the one-ray form, then arm against arm in one process
      for arm in product ${VARIANTS:-}; do
        if [ $arm != product ]; then export OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_${arm}.so; else unset OPTILAND_HIP_LIBRARY; fi
        timeout 900 python tools/gpu_polz_pair.py > $O/r06_polz_pair_${arm}.txt 2>&1; tail -${POLZ_TAIL:-24} $O/r06_polz_pair_${arm}.txt
      done; unset OPTILAND_HIP_LIBRARY ;;
    polz_cycles) # ... and in engine cycles per launch (GRBM_GUI_ACTIVE): OL_TRACE_RPT=1 (one ray per lane) against 3 (pair)
      for arm in 1 3; do
        OUTD=$O/polz_cycles_${arm}; rm -rf $OUTD; mkdir -p $OUTD
        (cd /tmp && TMPDIR=/tmp OL_TRACE_RPT=$arm CONFIGS=zf_f32 ROUNDS=2 LAUNCHES=40 timeout 600 \
          rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUTD/a -o ct -- \
          python $R/tools/gpu_clock_transient.py > $OUTD/run.log 2>&1; echo "rc=$?" >> $OUTD/run.log)
        python $R/tools/clock_transient_report.py $OUTD > $O/r06_polz_cycles_${arm}.txt 2>&1
        grep -A44 "## zf_f32 round 1" $O/r06_polz_cycles_${arm}.txt | awk 'NR>12 && NF>=5 {t+=$2; c+=$4; k+=1} END {if (k) printf "rpt=%s launches 10+: event_ms %.4f  cycles per launch %.4g\n", "'$arm'", t/k, c/k}'
      done ;;
    primed)    # is a block that has just been probed under sustained load "hot"? (tools/gpu_primed.py)
      timeout 600 python tools/gpu_primed.py > $O/r06_primed_${TAG:-0}.txt 2>&1; tail -12 $O/r06_primed_${TAG:-0}.txt ;;
    fuzz_tables) # kernel vs oracle on every table under fuzz_tables/ (tools/make_fuzz_tables.py LO HI, CPU, beforehand)
      timeout 1500 python tools/gpu_fuzz_tables.py > $O/r06_fuzz_tables_${TAG:-0}.txt 2>&1; tail -15 $O/r06_fuzz_tables_${TAG:-0}.txt | cut -c1-250 ;;
    bench)
      python bench.py ${BENCH_ARGS:-} > $O/r06_bench_${TAG:-default}.json 2> $O/r06_bench_${TAG:-default}.err
      tail -c 1500 $O/r06_bench_${TAG:-default}.json ;;
    *) echo "unknown leg $leg" ;;
  esac
done
