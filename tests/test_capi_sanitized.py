"""ASAN + UBSAN pass over the host side of the C-ABI shim.

The sanitized library (tools/build_sanitized.py) is exercised in a subprocess with the
ASAN runtime preloaded (python itself is not instrumented):

* CPU: every golden / shipped surface table goes through `ol_system_create` -- all the
  host-side staging (Zernike regrouping, aperture trees, polygon tables, coefficient
  bounds checks) runs before the first HIP call, which then fails with OL_EHIP on a box
  without a GPU -- plus malformed inputs that must be REJECTED, not read out of bounds.
* GPU (`-m gpu`): the guard-band and ragged-tail cases of tests/test_gpu_edge_cases.py
  through the sanitized host library (launch wrappers, pointer arithmetic on the
  caller's buffers).

A sanitizer report aborts the subprocess (`-fno-sanitize-recover`, `halt_on_error=1`).
"""

import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def san():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import build_sanitized
        lib = build_sanitized.OUT
        if not os.path.exists(lib):
            lib = build_sanitized.build()
        rt = build_sanitized.asan_runtime()
    except Exception as exc:  # noqa: BLE001 - no hipcc / no runtime: nothing to test with
        pytest.skip(f"sanitized library unavailable: {exc}")
    finally:
        sys.path.pop(0)
    env = dict(os.environ, LD_PRELOAD=rt, OPTILAND_HIP_LIBRARY=lib,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:"
                            "protect_shadow_gap=0:detect_odr_violation=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", PYTHONPATH=ROOT)
    return env


CPU_SCRIPT = r'''
import ctypes as C, glob, os, sys
import numpy as np
from optiland_amd import _capi
from optiland_amd.system import SystemTable
lib = _capi.load()
assert "asan" in _capi.library_path()
tables = sorted(glob.glob(os.path.join(sys.argv[1], "tests", "golden", "*.json"))) + \
         sorted(glob.glob(os.path.join(sys.argv[1], "optiland_amd", "data", "*.json")))
codes = {}
def create(surf, coeffs, optics, n_coeffs=None, n_surf=None):
    h = C.c_void_p()
    rc = lib.ol_system_create(surf.ctypes.data, surf.shape[0] if n_surf is None else n_surf,
                              coeffs.ctypes.data if coeffs.size else None,
                              coeffs.size if n_coeffs is None else n_coeffs,
                              optics.ctypes.data, optics.shape[1], C.byref(h))
    if rc == 0:
        lib.ol_system_destroy(h)
    return rc
n_ok = 0
for path in tables:
    try:
        t = SystemTable.load(path)
    except KeyError:   # fp32_margins.json and friends: not a surface table
        continue
    surf = np.ascontiguousarray(t.surfaces)
    optics = np.ascontiguousarray(t.optics)
    coeffs = np.ascontiguousarray(t.coeffs, dtype=np.float64)
    rc = create(surf, coeffs, optics)
    codes[rc] = codes.get(rc, 0) + 1
    n_ok += 1
    # truncated coefficient buffer: must be refused by the bounds check, never read
    if coeffs.size:
        rc2 = create(surf, coeffs[: coeffs.size // 2].copy(), optics)
        assert rc2 != 0, path
        # a surface that claims more coefficients than exist
        bad = surf.copy()
        k = int(np.argmax(bad["n_coeff"]))
        bad["n_coeff"][k] = bad["n_coeff"][k] + 10_000
        assert create(bad, coeffs, optics) != 0, path
    bad = surf.copy()
    bad["geom_kind"][-1] = 99
    assert create(bad, coeffs, optics) != 0
    bad = surf.copy()
    bad["aperture_kind"][-1] = -3
    assert create(bad, coeffs, optics) != 0
assert create(surf, coeffs, optics, n_surf=0) != 0
h = C.c_void_p()
assert lib.ol_system_create(None, 3, None, 0, None, 1, C.byref(h)) != 0
assert lib.ol_system_create(surf.ctypes.data, surf.shape[0], None, 5, optics.ctypes.data, 1, C.byref(h)) != 0
assert lib.ol_set_tuning(99, 1) != 0
assert lib.ol_trace(None, 0, 10, (C.c_void_p * 8)(), 0, None, 0, None, 0, 0, 0, None, None) != 0
print("tables", n_ok, "return codes", codes, "last error:", lib.ol_last_error().decode()[:60])
'''


def test_table_staging_under_asan_and_ubsan(san):
    out = subprocess.run([sys.executable, "-c", CPU_SCRIPT, ROOT], env=san, capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    assert "tables" in out.stdout, out.stdout
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, \
        out.stderr[-4000:]
    n = int(out.stdout.split("tables")[1].split()[0])
    assert n >= 90


@pytest.mark.gpu
def test_guard_band_cases_under_the_sanitized_host_library(san):
    """Launch wrappers + the caller-buffer pointer arithmetic under ASAN / UBSAN on the
    GPU box: the no-write-outside guard-band test, the ragged / unaligned / empty sizes
    and the argument-validation tests, run against liboptiland_hip_asan.so."""
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
           os.path.join(ROOT, "tests", "test_gpu_edge_cases.py"),
           os.path.join(ROOT, "tests", "test_gpu_parity.py"),
           "-k", "no_write_outside or ragged or unaligned or empty or status or zero_copy"]
    out = subprocess.run(cmd, env=san, capture_output=True, text=True, timeout=1200, cwd=ROOT)
    tail = (out.stdout.strip().splitlines() or [""])[-1]
    assert out.returncode == 0, (out.stdout[-3000:], out.stderr[-3000:])
    assert " passed" in tail and "failed" not in tail, tail
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr
