#!/bin/bash
# Round 4, GPU call 5: plane stride of the record block.  Call 4 showed the 2 MiB-aligned stride of
# rounds 2-3 is CONSISTENTLY the slowest choice for the fp32 double-Gauss record-all kernel (0.733 ms
# against 0.60-0.68 for 256 B ... 1 MiB alignments) while every other alignment varies run to run
# with the physical placement of the block.  Sweep: 2 MiB alignment + a skew, three processes each.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
OUT=$R/gpurun_out/r04_ab_skew.txt; : > $OUT
run() { local label=$1 align=$2 skew=$3; shift 3
  echo -n "$label   " >> $OUT
  OPTILAND_RECORD_ALIGN=$align OPTILAND_RECORD_SKEW=$skew timeout 120 python tools/ab_kernel.py --sustained --warmup 40 --steps 40 "$@" 2>/dev/null | tail -1 >> $OUT
  echo >> $OUT
}
SKEWS="0 256 1024 4096 4352 12544 20480 36864 69632 135168 266240 528384 1296128"
for rep in 1 2 3; do
  for k in $SKEWS; do
    run "dg_f32_gen a2M+$k" 2097152 $k --mode gen
    run "dg_f64_gen a2M+$k" 2097152 $k --mode gen --dtype f64
  done
done
for rep in 1 2; do
  for k in 0 4352 20480 135168 1296128; do
    run "zf_f32_gen a2M+$k" 2097152 $k --mode gen --workload zernike_fresnel
    run "rc_f32_gen a2M+$k" 2097152 $k --mode gen --workload rc_asphere
    run "zf_f64_gen a2M+$k" 2097152 $k --mode gen --workload zernike_fresnel --dtype f64
    run "dg_f32_rec a2M+$k" 2097152 $k --mode record
  done
done
python - <<'PY'
import re, collections, statistics as st
d=collections.OrderedDict()
for ln in open("gpurun_out/r04_ab_skew.txt"):
    m=re.match(r"(\S+) (\S+)\s+kernel_ms=([\d.]+) min=([\d.]+) median=([\d.]+)", ln)
    if m: d.setdefault(m.group(1),collections.OrderedDict()).setdefault(m.group(2),[]).append(float(m.group(5)))
for tag,arms in d.items():
    print(tag)
    for a,v in arms.items(): print(f"   {a:<16} median {st.median(v):.4f}  min {min(v):.4f}  max {max(v):.4f}  n={len(v)}")
PY
