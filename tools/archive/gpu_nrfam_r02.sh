#!/bin/bash
# One gpurun call for the single-family Newton kernels:
#  1. parity subset (device vs host run, Zernike / Newton / asphere systems) with the family
#     kernels and again with OPTILAND_HIP_NR_FAMILY=0 (generic kernel)
#  2. interleaved A/B on C5 and C4: family kernels | generic kernel
#  3. SQ_INSTS_VALU of the C5 kernel
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
SEL="hostmath or zernike or newton or nr_family or asphere or aspheric or config4 or config5"
timeout 200 python -m pytest tests/test_gpu_hostmath.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py \
  -q -m gpu -p no:cacheprovider -k "$SEL" 2>&1 | tail -6 > $OUT/nrfam_pytest.txt
OPTILAND_HIP_NR_FAMILY=0 timeout 200 python -m pytest tests/test_gpu_hostmath.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py \
  -q -m gpu -p no:cacheprovider -k "$SEL" 2>&1 | tail -6 >> $OUT/nrfam_pytest.txt
cat $OUT/nrfam_pytest.txt
AB=$OUT/ab_nrfam.txt; : > $AB
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   kernel_ms=%.4f moved=%.0f GB/s frac=%.3f value=%.4g'%(r['kernel_ms'],r['achieved'],r['frac'],d['value']))"; }
arm() {
  local w=$1 v=$2
  echo -n "$w $v" >> $AB
  case $v in
    family) python bench.py --workload $w --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $AB ;;
    generic) OPTILAND_HIP_NR_FAMILY=0 python bench.py --workload $w --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $AB ;;
  esac
}
echo "# $(date -u) 1e7 rays fp32 record-all, 30 launches per arm, arm order alternating" >> $AB
for w in zernike_fresnel rc_asphere; do
  arm $w family; arm $w generic; arm $w generic; arm $w family; arm $w family; arm $w generic
done
cat $AB
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/prof_nrfam_valu -o zf -- \
  python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload zernike_fresnel > $OUT/prof_nrfam_valu.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof_nrfam_valu/**/*counter_collection.csv", recursive=True):
    vals = {}
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            vals.setdefault((r["Kernel_Name"][:46], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
    with open("$OUT/nrfam_valu.txt", "w") as o:
        for k, v in sorted(vals.items()):
            line = f"{k[0]} {k[1]} per wave {sum(v) / len(v) / 156252:.1f}"
            print(line); o.write(line + "\n")
PY
