#!/bin/bash
# Round 4: device-resident wavefront reference -- parity suite, the reference's consumer tests, OPD
# latencies through the seams, and the OPD kernels with / without the launch-uniform device branch.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 | tee gpurun_out/r04_pytest_gpu_final.log
timeout 900 python tools/gpu_r04_dropin.py > gpurun_out/r04_dropin.log 2>&1
python - <<'PY'
import json
d=json.load(open("gpurun_out/r04_dropin.json")); print(json.dumps(d.get("reference_analyses_cooke_fp64_with_seams")))
PY
OPTILAND_HIP_DEVICE_REFERENCE=0 timeout 600 python - <<'PY' 2>&1 | grep -v amdgpu | tail -3
import sys, time, numpy as np, torch
sys.path.insert(0, ".")
from tests import _live
be = _live.import_reference()
be.set_backend("torch"); be.set_device("cuda"); be.set_precision("float64")
from optiland_amd import integration
from optiland.wavefront import OPD
integration.enable()
lens, w = _live.build_system("CookeTriplet")
def wall(fn, reps=7, warm=3):
    for _ in range(warm): fn()
    ts=[]
    for _ in range(reps):
        torch.cuda.synchronize(); t0=time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter()-t0)
    return round(float(np.median(ts))*1e3,4)
print("host-built reference (round 3 form): OPD_15 %.4f ms  OPD_256 %.4f ms" % (wall(lambda: OPD(lens,(0.0,1.0),w).rms()), wall(lambda: OPD(lens,(0.0,1.0),w,num_rays=256).rms())))
PY
OUT=$R/gpurun_out/r04_ab_opd_devref.txt; : > $OUT
run() { local label=$1 lib=$2; shift 2
  echo -n "$label   " >> $OUT
  if [ -n "$lib" ]; then
    OPTILAND_HIP_LIBRARY=$R/optiland_amd/lib/variant_$lib.so timeout 120 python tools/ab_kernel.py --sustained --warmup 100 --steps 60 "$@" 2>/dev/null | tail -1 >> $OUT
  else
    timeout 120 python tools/ab_kernel.py --sustained --warmup 100 --steps 60 "$@" 2>/dev/null | tail -1 >> $OUT
  fi
  echo >> $OUT
}
for rep in 1 2; do
  for v in product opd_nodevref; do
    run "dg_opd $v" "${v/product/}" --mode opd
    run "z_opd $v" "${v/product/}" --workload zernike --mode opd
    run "rc_opd $v" "${v/product/}" --workload rc_asphere --mode opd
  done
done
python tools/ab_summary.py $OUT
