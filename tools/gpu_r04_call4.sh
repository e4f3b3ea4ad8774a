#!/bin/bash
# Round 4, GPU call 4: where is the write ceiling of the record-all pattern?  ol_stream_fill sweep
# (planes x width x stride alignment) and the trace kernels with other record-stride alignments.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 300 python tools/gpu_fill_sweep.py 2>&1 | tee gpurun_out/r04_fill_sweep.txt
OUT=$R/gpurun_out/r04_ab_stride.txt; : > $OUT
run() { local label=$1 align=$2; shift 2
  echo -n "$label   " >> $OUT
  OPTILAND_RECORD_ALIGN=$align timeout 120 python tools/ab_kernel.py --sustained --warmup 60 --steps 60 "$@" 2>/dev/null | tail -1 >> $OUT
  echo >> $OUT
}
for rep in 1 2; do
  for a in 2097152 256 4096 65536 1048576; do
    run "dg_f32_gen align$a" $a --mode gen
    run "dg_f64_gen align$a" $a --mode gen --dtype f64
    run "zf_f32_gen align$a" $a --mode gen --workload zernike_fresnel
  done
done
python - <<'PY'
import re, collections
d=collections.OrderedDict()
for ln in open("gpurun_out/r04_ab_stride.txt"):
    m=re.match(r"(\S+) (\S+)\s+kernel_ms=([\d.]+) min=([\d.]+) median=([\d.]+)", ln)
    if m: d.setdefault(m.group(1),collections.OrderedDict()).setdefault(m.group(2),[]).append(float(m.group(5)))
for tag,arms in d.items():
    print(tag, "  ".join(f"{a}={sum(v)/len(v):.4f}" for a,v in arms.items()))
PY
timeout 600 python -m pytest tests/test_pupil_points.py tests/test_gpu_math_probe.py -m gpu -q 2>&1 | tail -3
cat gpurun_out/pupil_points.json | tr -d '\n'
