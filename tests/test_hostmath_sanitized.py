"""The kernel arithmetic under AddressSanitizer + UndefinedBehaviorSanitizer.

Device-side ASAN is not available on this image (it needs an xnack+ target and ROCm's
instrumented runtime; tools/build_sanitized.py and tests/test_capi_sanitized.py cover the
HOST side of the product library).  The host build of the kernel source
(tests/hostmath) closes most of that gap: compiled with `-fsanitize=address,undefined`, every
read of the surface table / coefficient blocks / aperture token lists / polygon tables and
every read and write of the ray, record and PRT planes that `surface_math.h`,
`raygen_device.h`, `wavefront_device.h`, `wavefront_fit_device.h` and `epilogue_device.h`
perform is checked, on all
golden systems and on the randomised ones -- the whole of tests/test_hostmath.py and
tests/test_hostmath_fuzz.py re-run in a subprocess against the sanitized harness (python
itself is not instrumented: the ASAN runtime is LD_PRELOADed).  A report aborts the run
(`halt_on_error`, `-fno-sanitize-recover`).
"""

import os
import subprocess
import sys

import pytest

from tests import _hostmath as hm

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

pytestmark = pytest.mark.skipif(not hm.available(), reason="hipcc (used as host C++ compiler) missing")


def test_kernel_arithmetic_under_asan_and_ubsan():
    b = hm._builder()
    try:
        lib = b.build(sanitize=True)
        rt = b.asan_runtime()
    except Exception as exc:  # noqa: BLE001 - no sanitizer runtime in this toolchain
        pytest.skip(f"sanitized harness unavailable: {exc}")
    env = dict(os.environ, LD_PRELOAD=rt, OL_HOSTMATH_LIBRARY=lib, PYTHONPATH=ROOT,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:abort_on_error=1:"
                            "protect_shadow_gap=0:detect_odr_violation=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    out = subprocess.run(
        [sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
         os.path.join(ROOT, "tests", "test_hostmath.py"),
         os.path.join(ROOT, "tests", "test_hostmath_fuzz.py"),
         # round 4: the reduction passes of ol_wavefront_fit (wavefront_fit_device.h)
         os.path.join(ROOT, "tests", "test_wavefront_fit.py"), "-m", "not gpu"],
        cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    tail = (out.stdout + out.stderr)[-3000:]
    assert out.returncode == 0, tail
    assert " passed" in out.stdout and "failed" not in out.stdout, tail
    assert "AddressSanitizer" not in tail and "runtime error" not in tail, tail
