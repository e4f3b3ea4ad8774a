#!/bin/bash
# VALU wave-instructions per launch of the C5 kernel with features stripped one by one
# (tools/gpu_zf_breakdown.py under rocprofv3 --pmc; own pass, kernel-trace only).
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_zf_valu
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/a -o zf -- \
  python $R/tools/gpu_zf_breakdown.py > $OUT/a.log 2>&1
python - <<'PY'
import csv, glob, os
out=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/prof_zf_valu'
for f in glob.glob(out+'/a/**/*counter_collection.csv', recursive=True):
    vals={}
    for r in csv.DictReader(open(f)):
        if 'trace_kernel' in r['Kernel_Name']:
            key=(r['Kernel_Name'][:60], r['Counter_Name'])
            vals.setdefault(key,[]).append(float(r['Counter_Value']))
    for (k,c),v in sorted(vals.items()):
        print(f"{k:60s} {c:16s} mean {sum(v)/len(v):14.1f} (n={len(v)}; per wave {sum(v)/len(v)/156252:8.1f})")
PY
grep ms $OUT/a.log
