#!/usr/bin/env python
"""GPU box: the reference's OWN tests of the consumers of the path (test_optic.py,
test_analysis.py, test_wavefront.py, test_fft_psf.py ...) on the torch backend, device
cuda -- once with the stock reference, once with `integration.enable()` routing every
real-ray trace through the HIP kernels.  Prints both summaries and the tests that fail
ONLY with the drop-in (must be none).  Needs oracle/_ref (oracle/stage_reference.py).

    python tools/gpu_ref_consumers.py [test files ...]
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref")
FILES = ["tests/test_optic.py", "tests/test_analysis.py", "tests/test_wavefront.py",
         "tests/test_fft_psf.py", "tests/test_wavefront_strategy.py"]


def run(dropin: bool, files=None):
    files = list(files or FILES)
    tmp = tempfile.mkdtemp(prefix="ol_ref_gpu_")
    dst = os.path.join(tmp, "tests")
    shutil.copytree(os.path.join(REF, "tests"), dst)
    for _b, _d, _f in os.walk(dst):   # (the staged copy may be read-only; the box user is not root)
        for _n in _d + _f:
            os.chmod(os.path.join(_b, _n), os.stat(os.path.join(_b, _n)).st_mode | 0o200)
    conf = open(os.path.join(dst, "conftest.py")).read()
    conf = conf.replace('be.set_device("cpu")  # Use CPU for tests', 'be.set_device("cuda")')
    conf = conf.replace("be.grad_mode.enable()", "be.grad_mode.disable()")
    if dropin:
        conf = conf.replace(
            "import optiland.backend as be\n",
            "import optiland.backend as be\nimport atexit, sys\nsys.path.insert(0, %r)\n"
            "from optiland_amd import integration as _integ\n_integ.enable()\n"
            "import optiland_amd.tracer as _tr\n_made = [0]\n_orig = _tr._make_engine\n"
            "def _mk(table, device):\n    _made[0] += 1\n    return _orig(table, device)\n"
            "_tr._make_engine = _mk\n"
            "atexit.register(lambda: print('\\n[drop-in] device tables created:', _made[0], "
            "'SurfaceGroup seam launches:', _integ._SG['count']))\n"
            "import optiland_amd.analysis_seams as _seams\n"
            "atexit.register(lambda: print('\\n[seams] ' + ' '.join('%%s=%%d' %% kv for kv in "
            "_seams.STATS.items())))\n" % ROOT, 1)
    open(os.path.join(dst, "conftest.py"), "w").write(conf)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
               PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "refshim"), REF]))
    out = subprocess.run([sys.executable, "-m", "pytest", "-q", "-p", "no:cacheprovider",
                          "-k", "torch and not autodiff", "-rf", *files],
                         cwd=tmp, env=env, capture_output=True, text=True, timeout=3000)
    shutil.rmtree(tmp, ignore_errors=True)
    failed = set(re.findall(r"^FAILED (\S+)", out.stdout, flags=re.M))
    errors = set(re.findall(r"^ERROR (\S+)", out.stdout, flags=re.M))
    tail = [l for l in out.stdout.strip().splitlines()
            if " passed" in l or " failed" in l or "drop-in" in l or "[seams]" in l
            or "error" in l.lower()[:40]]
    tail.append(f"(pytest rc {out.returncode})")
    return failed | errors, tail, out


if __name__ == "__main__":
    base_f, base_tail, _ = run(False, sys.argv[1:])
    hip_f, hip_tail, out = run(True, sys.argv[1:])
    print("stock reference on cuda :", base_tail[-3:])
    print("with the drop-in        :", hip_tail[-5:])
    new = sorted(hip_f - base_f)
    print("failing on the stock torch backend (cuda):", sorted(base_f))
    print("failing with the drop-in                 :", sorted(hip_f))
    print("failing only with the drop-in:", new)
    for n in new[:10]:
        m = re.search(r"_{5,} %s _{5,}(.*?)(?=\n_{5,} |\n=+ )" % re.escape(n.split("::", 1)[1].replace("::", ".")), out.stdout, flags=re.S)
        if m:
            print("-" * 60, n)
            print(m.group(1)[-1500:])
    print("fixed by the drop-in (fail on the stock torch backend only):", len(base_f - hip_f))
    sys.exit(1 if new else 0)
