#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_nr_iters; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES --output-format csv -d $OUT/a -o nr -- \
  python $R/tools/gpu_nr_iters.py > $OUT/a.log 2>&1
python - <<PY
import csv, glob
names = [l.split("VARIANT ", 1)[1].strip() for l in open("$OUT/a.log") if l.startswith("VARIANT ")]
rows = []
for f in glob.glob("$OUT/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"][:48], float(r["Counter_Value"])))
rows.sort()
L = 4
with open("$R/gpurun_out/nr_iters.txt", "w") as out:
    for i, nm in enumerate(names):
        grp = rows[i * L:(i + 1) * L]
        if not grp:
            continue
        v = sum(g[2] for g in grp) / len(grp)
        line = f"{nm:48s} {grp[0][1]:50s} VALU/ray {v / 156252:8.1f}"
        print(line)
        out.write(line + "\n")
PY
tail -3 $OUT/a.log
