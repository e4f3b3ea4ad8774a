#!/usr/bin/env python
"""GPU box: where the time of the reference's own consumer tests goes WITH the drop-in
(tools/gpu_ref_consumers.py's second arm under cProfile): top entries by internal time, and
the share spent inside optiland_amd."""
import os
import pstats
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gpu_ref_consumers as G  # noqa: E402

tmp = tempfile.mkdtemp(prefix="ol_ref_prof_")
dst = os.path.join(tmp, "tests")
shutil.copytree(os.path.join(G.REF, "tests"), dst)
for _b, _d, _f in os.walk(dst):   # (the staged copy may be read-only; the box user is not root)
    for _n in _d + _f:
        os.chmod(os.path.join(_b, _n), os.stat(os.path.join(_b, _n)).st_mode | 0o200)
conf = open(os.path.join(dst, "conftest.py")).read()
conf = conf.replace('be.set_device("cpu")  # Use CPU for tests', 'be.set_device("cuda")')
conf = conf.replace("be.grad_mode.enable()", "be.grad_mode.disable()")
conf = conf.replace("import optiland.backend as be\n",
                    "import optiland.backend as be\nimport sys\nsys.path.insert(0, %r)\n"
                    "from optiland_amd import integration as _integ\n_integ.enable()\n" % ROOT, 1)
open(os.path.join(dst, "conftest.py"), "w").write(conf)
env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1",
           PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests", "refshim"), G.REF]))
prof = os.path.join(tmp, "prof.out")
subprocess.run([sys.executable, "-m", "cProfile", "-o", prof, "-m", "pytest", "-q", "-p",
                "no:cacheprovider", "-k", "torch and not autodiff", *G.FILES],
               cwd=tmp, env=env, capture_output=True, text=True, timeout=3000)
st = pstats.Stats(prof)
total = st.total_tt
ours = sum(v[2] for k, v in st.stats.items() if "optiland_amd" in k[0])
print(f"total internal time {total:.1f} s; inside optiland_amd/*.py {ours:.2f} s")
st.sort_stats("tottime").print_stats(45)
shutil.rmtree(tmp, ignore_errors=True)
