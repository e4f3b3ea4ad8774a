#!/bin/bash
# One gpurun call for the one-polynomial Zernike evaluator:
#  1. device vs the host run of the same source + the Zernike / Newton parity tests
#  2. interleaved A/B on C5: product (one polynomial, split loads) | level form
#     (OPTILAND_HIP_ZERNIKE_MONO=0, same library) | variants zmono_nosplit, zmono_loops
#  3. rocprofv3 kernel stats and SQ_INSTS_VALU of the C5 kernel
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
OUT=$R/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_hostmath.py tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py \
  -q -m gpu -p no:cacheprovider -k "hostmath or zernike or newton or nr_family or c5 or C5" 2>&1 | tail -15 > $OUT/zmono_pytest.txt
cat $OUT/zmono_pytest.txt
AB=$OUT/ab_zmono.txt; : > $AB
show() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('   kernel_ms=%.4f moved=%.0f GB/s frac=%.3f value=%.4g'%(r['kernel_ms'],r['achieved'],r['frac'],d['value']))"; }
arm() {
  local v=$1
  echo -n "zf_f32_record $v" >> $AB
  case $v in
    product) python bench.py --workload zernike_fresnel --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $AB ;;
    levels) OPTILAND_HIP_ZERNIKE_MONO=0 python bench.py --workload zernike_fresnel --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $AB ;;
    *) OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$v.so python bench.py --workload zernike_fresnel --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | show >> $AB ;;
  esac
}
echo "# $(date -u) C5 (zernike_fresnel, 1e7 rays fp32 record-all), 30 launches per arm, arm order alternating" >> $AB
ARMS=(product levels zmono_nosplit zmono_loops)
for rep in 1 2 3 4; do
  if [ $((rep % 2)) -eq 1 ]; then for v in "${ARMS[@]}"; do arm "$v"; done
  else for ((i=${#ARMS[@]}-1; i>=0; i--)); do arm "${ARMS[$i]}"; done; fi
done
cat $AB
cd /tmp
mkdir -p $OUT/prof_zf_f32
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_zf_f32/stats -o zf_f32 -- \
  python $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --workload zernike_fresnel > $OUT/prof_zf_f32/stats.log 2>&1
grep '^{' $OUT/prof_zf_f32/stats.log | tail -c 300
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM --output-format csv -d $OUT/prof_zmono_valu -o zf -- \
  python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --workload zernike_fresnel > $OUT/prof_zmono_valu.log 2>&1
python - <<PY
import csv, glob
for f in glob.glob("$OUT/prof_zmono_valu/**/*counter_collection.csv", recursive=True):
    vals = {}
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    line = {k: round(sum(v) / len(v), 1) for k, v in vals.items()}
    print(line)
    open("$OUT/zmono_valu.txt", "w").write(repr(line) + "\n")
PY
