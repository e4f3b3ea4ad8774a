#!/bin/bash
# Round 4, GPU call 6: record-stride families on ONE box (call 4: small alignments 0.60-0.68 ms vs
# 0.733 for 2 MiB on its box; call 5: 2 MiB + any skew slower than 2 MiB + 0 on another box).
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
OUT=$R/gpurun_out/r04_ab_stride2.txt; : > $OUT
run() { local label=$1 align=$2 skew=$3; shift 3
  echo -n "$label   " >> $OUT
  OPTILAND_RECORD_ALIGN=$align OPTILAND_RECORD_SKEW=$skew timeout 120 python tools/ab_kernel.py --sustained --warmup 40 --steps 40 "$@" 2>/dev/null | tail -1 >> $OUT
  echo >> $OUT
}
for rep in 1 2 3; do
  for cfg in "2097152 0" "256 0" "4096 0" "65536 0" "1048576 0" "2097152 135168" "2097152 154112" "256 2097152" "65536 1900544"; do
    set -- $cfg
    run "dg_f32_gen a$1+$2" $1 $2 --mode gen
    run "dg_f64_gen a$1+$2" $1 $2 --mode gen --dtype f64
    run "dg_f32_rec a$1+$2" $1 $2 --mode record
  done
done
python - <<'PY'
import re, collections, statistics as st
d=collections.OrderedDict()
for ln in open("gpurun_out/r04_ab_stride2.txt"):
    m=re.match(r"(\S+) (\S+)\s+kernel_ms=([\d.]+) min=([\d.]+) median=([\d.]+)", ln)
    if m: d.setdefault(m.group(1),collections.OrderedDict()).setdefault(m.group(2),[]).append(float(m.group(5)))
for tag,arms in d.items():
    print(tag)
    for a,v in arms.items(): print(f"   {a:<18} " + " ".join(f"{x:.4f}" for x in v))
PY
