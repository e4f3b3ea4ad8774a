#!/bin/bash
# Round 4, closing pass on the ABI-9 library: tools/gpu_r04_final.sh + the reference's own
# consumer tests on cuda (stock vs drop-in) + the fit passes under rocprofv3.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
bash tools/gpu_r04_final.sh
timeout 900 python tools/gpu_ref_consumers.py > gpurun_out/r04_reference_consumers_on_device.txt 2>&1
tail -8 gpurun_out/r04_reference_consumers_on_device.txt | cut -c1-400
timeout 300 python tools/gpu_fit_timing.py > gpurun_out/r04_fit_timing.txt 2>&1
(cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04_fit -o fit -- python $R/tools/gpu_fit_timing.py > /dev/null 2>&1)
python - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/prof_r04_fit/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(f)))
    with open("gpurun_out/r04_fit_rocprof.txt", "w") as out:
        out.write("# rocprofv3 --kernel-trace --stats -- python tools/gpu_fit_timing.py\n")
        out.write("Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs\n")
        for r in rows[:24]:
            out.write(",".join('"%s"' % r[k] if k == "Name" else r[k] for k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")) + "\n")
    print(open("gpurun_out/r04_fit_rocprof.txt").read()[:1500])
PY
