#!/bin/bash
# CPU, build container only: the reference's OWN consumer tests (spot diagram / encircled
# energy, wavefront, FFT PSF, MTF, optic) with the drop-in AND the analysis seams enabled
# (integration.enable(force=True, analyses=True)) on the host build of the kernel source.
# Prints pass / fail counts and how many calls each seam served.
#   tools/ref_consumers_seams.sh [workdir]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
W=${1:-/tmp/ol_ref_seams}
rm -rf "$W" && mkdir -p "$W" && cp -r /root/reference/tests "$W/tests"
cat > "$W/tests_fake.py" <<PY
import importlib.util, sys
sys.path.insert(0, "$R")
spec = importlib.util.spec_from_file_location("_ol_hostmath", "$R/tests/_hostmath.py")
mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
OracleEngine = mod.make_engine_class()
PY
python - "$W" "$R" <<'PY'
import sys
w, r = sys.argv[1:3]
p = w + "/tests/conftest.py"
s = open(p).read()
s = s.replace("import optiland.backend as be\n",
              "import optiland.backend as be\nimport sys\nsys.path.insert(0, %r)\n"
              "import optiland_amd.tracer as _tr\nfrom tests_fake import OracleEngine\n"
              "_tr._make_engine = lambda table, device: OracleEngine(table, device)\n"
              "from optiland_amd import integration as _integ\n"
              "_integ.enable(force=True, analyses=True)\n"
              # the distribution seams too: `ol_pupil_points` of the host build stands in
              # for the device
              "import optiland_amd.analysis_seams as _seams\n"
              "from optiland_amd import load_system as _ls\n"
              "_peng = OracleEngine(_ls('double_gauss'), 'cpu')\n"
              "_seams.POINTS_HOOK = lambda kind, num, dtype: _peng.pupil_points(kind, num, dtype)\n"
              % r, 1)
s = s.replace("be.grad_mode.enable()", "be.grad_mode.disable()")
s += """

def pytest_sessionfinish(session, exitstatus):
    import optiland_amd.analysis_seams as _a
    print("\\n[seams] " + " ".join("%s=%d" % kv for kv in _a.STATS.items()))
"""
open(p, "w").write(s)
PY
cd "$W"
OPTILAND_HIP_SEAM_LOG=$W/seam_log.txt PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=$R/tests/refshim:/root/reference:$W \
  python -m pytest -q -p no:cacheprovider -k "torch and not autodiff" \
  tests/test_analysis.py tests/test_analysis_extended.py tests/test_wavefront.py \
  tests/test_fft_psf.py tests/test_mtf.py tests/test_optic.py tests/test_zernike.py \
  > "$W/log.txt" 2>&1 || true
sort "$W/seam_log.txt" 2>/dev/null | uniq -c | sort -rn | head -20; grep -E "^\[seams\]|^FAILED|^ERROR| passed| failed" "$W/log.txt" | tail -30
