#!/bin/bash
# Final artefacts of round 2 in one gpurun call: kernel list of the LIVE drop-in alone
# (rocprofv3 --kernel-trace --stats around tools/gpu_dropin_kernels.py) and rocprofv3 stats
# of the C5 kernel.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out; mkdir -p $OUT/prof_zf_f32
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_live -o live -- \
  python $R/tools/gpu_dropin_kernels.py > $OUT/prof_live.log 2>&1
python - <<PY
import csv, glob
g = glob.glob("$OUT/prof_live/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(g[0]))) if g else []
with open("$OUT/live_dropin_kernels.txt", "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats -- python tools/gpu_dropin_kernels.py: 300 x Optic.trace_generic(1e6 rays) of a reference-built DoubleGauss under integration.enable(), MI355X\n")
    f.write("# " + open("$OUT/prof_live.log").read().strip().splitlines()[-1] + "\n")
    f.write("Name,Calls,TotalDurationNs,AverageNs,Percentage\n")
    for r in rows[:12]:
        nm = r["Name"]; nm = nm if len(nm) < 100 else nm[:97] + "..."
        f.write(f'"{nm}",{r["Calls"]},{r["TotalDurationNs"]},{float(r["AverageNs"]):.1f},{r["Percentage"]}\n')
print(open("$OUT/live_dropin_kernels.txt").read())
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_zf_f32/stats -o zf_f32 -- \
  python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload zernike_fresnel > $OUT/prof_zf_f32/stats.log 2>&1
grep '^{' $OUT/prof_zf_f32/stats.log | tail -c 200
