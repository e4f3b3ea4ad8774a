#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_sq_zf
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="--workload zernike_fresnel ${ZF_ARGS}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/a -o sq -- \
  python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline $ARGS > $OUT/a.log 2>&1
python - <<'PY'
import csv, glob, os
out=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/prof_sq_zf'
for f in glob.glob(out+'/a/**/*counter_collection.csv', recursive=True):
    vals={}
    for r in csv.DictReader(open(f)):
        if 'trace_kernel' in r['Kernel_Name']:
            vals.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
    print({k: round(sum(v)/len(v),1) for k,v in vals.items()})
PY
cd $R
for m in record last; do python bench.py --workload zernike_fresnel --mode $m --no-cpu-baseline --steps 20 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['roofline']['kernel_ms'])"; done
