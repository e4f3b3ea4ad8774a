#!/bin/bash
# rocprofv3 kernel statistics of the analysis paths (the kernels either side of the trace):
# SpotDiagram (fused ol_trace_spot), EncircledEnergy (ol_radial_energy), OPD / FFT PSF
# (ol_wavefront_opd), IncoherentIrradiance (ol_irradiance) on the shipped Cooke triplet.
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/prof_analyses
mkdir -p $OUT
cat > /tmp/_analyses.py <<PY
import sys, torch
sys.path.insert(0, "$R")
from optiland_amd import load_system
from optiland_amd.tracer import HipRayTracer
from optiland_amd.analysis import SpotDiagram, EncircledEnergy
from optiland_amd.wavefront import OPD, FFTPSF
t = HipRayTracer(load_system("cooke_generic"), "cuda:0", dtype=torch.float64)
for _ in range(5):
    SpotDiagram(t, num_rings=20)
    EncircledEnergy(t, num_rays=100000, distribution="random", num_points=256)
    OPD(t, (0.0, 1.0), 0.55, num_rays=30)
    OPD(t, (0.0, 1.0), 0.55, num_rays=30, strategy="best_fit_sphere", remove_tilt=True)
    FFTPSF(t, (0.0, 1.0), 0.55, num_rays=128)
torch.cuda.synchronize()
PY
export TMPDIR=/tmp
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o analyses -- python /tmp/_analyses.py > $OUT/stats.log 2>&1
f=$(find $OUT/stats -name "*kernel_stats.csv" | head -1)
{ echo "# r01 analyses (Cooke triplet, fp64): rocprofv3 --kernel-trace --stats -- 5 x {SpotDiagram 20 rings, EncircledEnergy 1e5 rays, OPD 30 rings (chief-ray, best-fit + remove_tilt), FFTPSF 128}"; head -25 "$f" | cut -c1-200; } | tee $OUT/summary.txt
