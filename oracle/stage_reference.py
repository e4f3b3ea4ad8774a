#!/usr/bin/env python
"""Stage the Python reference as TEST INFRASTRUCTURE under the git-ignored oracle/_ref/.

    python oracle/stage_reference.py            # /root/reference/optiland -> oracle/_ref/optiland

The reference is pure Python: there is nothing to compile, but the live-drop-in GPU
tests (tests/test_gpu_live_reference.py), `bench.py`'s `cpu_baseline.numpy` /
`gpu_baseline.torch` legs and `tools/gpu_live_e2e.py` need the real package next to the
HIP library on the GPU box, where /root/reference does not exist.  `oracle/_ref/` is
listed in .gitignore (never part of the history, never read by anything under
`optiland_amd/`) but NOT in .gpurunignore, so the staged copy travels with the snapshot
exactly like the built `.so` files do.  `__graft_entry__.build()` runs this whenever
/root/reference is present.

Nothing is modified: a plain file copy of `optiland/` (and of the reference's `tests/`, so
that its own consumer tests can run on the GPU through the drop-in) minus byte-code caches.  The
import stubs for the three packages the image lacks (numba / vtk / seaborn) are this
repo's own files under tests/refshim/.
"""

from __future__ import annotations

import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEST = os.path.join(ROOT, "oracle", "_ref")


def _newest(path: str) -> float:
    m = 0.0
    for d, _dirs, files in os.walk(path):
        if "__pycache__" in d:
            continue
        for f in files:
            m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def make_writable(tree: str) -> None:
    """u+w on every directory and file under `tree`.  /root/reference is read-only and
    `shutil.copytree` keeps the modes: as root that went unnoticed, but the GPU box runs the
    tests as an ordinary user, and a copy of the staged tests whose conftest.py is to be
    rewritten (tools/gpu_ref_consumers.py) could not be (round 6)."""
    for base, dirs, files in os.walk(tree):
        for name in dirs + files:
            path = os.path.join(base, name)
            try:
                os.chmod(path, os.stat(path).st_mode | 0o200)
            except OSError:
                pass
    try:
        os.chmod(tree, os.stat(tree).st_mode | 0o200)
    except OSError:
        pass


def stage(src: str = "/root/reference", force: bool = False, verbose: bool = True) -> str | None:
    """Copy `src`/optiland to oracle/_ref/optiland.  Returns the staged root, or None
    when the reference is absent (the GPU box: the copy made here is already there)."""
    pkg = os.path.join(src, "optiland")
    if not os.path.isdir(pkg):
        return DEST if os.path.isdir(os.path.join(DEST, "optiland")) else None
    out = os.path.join(DEST, "optiland")
    stamp = os.path.join(DEST, ".staged_from")
    if not force and os.path.isdir(out) and os.path.exists(stamp) \
            and os.path.isdir(os.path.join(DEST, "tests")) == os.path.isdir(os.path.join(src, "tests")) \
            and os.path.getmtime(stamp) >= _newest(pkg):
        make_writable(DEST)
        return DEST
    if os.path.isdir(out):
        shutil.rmtree(out)
    os.makedirs(DEST, exist_ok=True)
    shutil.copytree(pkg, out, ignore=shutil.ignore_patterns("__pycache__", "*.pyc"))
    # the reference's own test-suite (1.9 MB): run on the GPU box through the drop-in
    # (tests/test_gpu_live_reference.py::test_reference_consumer_suite_on_device)
    tests_src, tests_out = os.path.join(src, "tests"), os.path.join(DEST, "tests")
    if os.path.isdir(tests_src):
        if os.path.isdir(tests_out):
            shutil.rmtree(tests_out)
        shutil.copytree(tests_src, tests_out,
                        ignore=shutil.ignore_patterns("__pycache__", "*.pyc", "gui"))
    make_writable(DEST)
    with open(stamp, "w") as f:
        f.write(os.path.abspath(src) + "\n")
    if verbose:
        print(f"staged {pkg} -> {out}")
    return DEST


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    stage(*(args[:1] or ["/root/reference"]), force="--force" in sys.argv)
