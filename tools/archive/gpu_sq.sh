#!/bin/bash
# SQ-level PMC passes for the dominant kernel (own runs; no trace domains besides kernel-trace)
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_sq
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
for mode in ${SQ_MODES:-record last}; do
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU --output-format csv -d $OUT/a_$mode -o sq -- \
  python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --mode $mode > $OUT/a_$mode.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/b_$mode -o sq -- \
  python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --mode $mode > $OUT/b_$mode.log 2>&1
done
python - <<'PY'
import csv, glob, os
out=os.environ.get('GRAFT_REPO_ROOT', os.getcwd())+'/gpurun_out/prof_sq'
for d in sorted(glob.glob(out+'/*_*/')):
    for f in glob.glob(d+'/**/*counter_collection.csv', recursive=True):
        vals={}
        for r in csv.DictReader(open(f)):
            if 'trace_kernel' in r['Kernel_Name']:
                vals.setdefault(r['Counter_Name'],[]).append(float(r['Counter_Value']))
        print(os.path.basename(d.rstrip('/')), {k: round(sum(v)/len(v),1) for k,v in vals.items()})
PY
