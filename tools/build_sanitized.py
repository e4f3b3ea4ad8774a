#!/usr/bin/env python
"""Sanitizer build of the C-ABI shim (SURVEY.md section 5 "sanitizer build of the C-ABI shim").

    python tools/build_sanitized.py     ->  optiland_amd/lib/liboptiland_hip_asan.so

The HOST side of the library -- argument validation, the surface-table staging of
`ol_system_create` (Zernike regrouping, aperture-tree flattening, polygon tables), the
launch wrappers -- is compiled with AddressSanitizer + UndefinedBehaviorSanitizer
(`-fsanitize=address,undefined`, host pass only); the gfx950 device code is the product's.
Device-side ASAN needs an `xnack+` target and ROCm's instrumented runtime
(/opt/rocm/lib/asan), which this image does not ship; out-of-bounds device writes are
covered by the guard-band tests instead (tests/test_gpu_edge_cases.py::
test_no_write_outside_the_callers_buffers and the ragged-tail cases).

Used by tests/test_capi_sanitized.py: python is not an ASAN binary, so the test
LD_PRELOADs the ASAN runtime and selects the library through OPTILAND_HIP_LIBRARY.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optiland_amd import build as B  # noqa: E402

OUT = os.path.join(B.LIBDIR, "liboptiland_hip_asan.so")
SAN = ["-Xarch_host", "-fsanitize=address,undefined", "-Xarch_host", "-fno-omit-frame-pointer",
       "-Xarch_host", "-fno-sanitize-recover=undefined", "-g1"]


def asan_runtime() -> str:
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not hits:
        raise RuntimeError("clang's ASAN runtime not found under /opt/rocm/lib/llvm")
    return hits[-1]


def build(force=False) -> str:
    srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
    hdrs = [os.path.join(B.CSRC, h) for h in B.HEADERS] + \
        [os.path.join(ROOT, "include", "optiland_hip.h")]
    if not force and os.path.exists(OUT) and \
            os.path.getmtime(OUT) >= max(os.path.getmtime(p) for p in srcs + hdrs):
        return OUT
    objdir = os.path.join(B.LIBDIR, "asan_obj")
    os.makedirs(objdir, exist_ok=True)
    flags = ["--offload-arch=" + B.ARCH, "-O1", "-std=c++17", "-fPIC", "-ffp-contract=on",
             "-fno-math-errno"] + SAN
    objs, jobs = [], []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s).replace(".hip", ".o"))
        jobs.append([B._hipcc(), *flags, "-c", s, "-o", o])
        objs.append(o)
    with ThreadPoolExecutor(max_workers=4) as pool:
        list(pool.map(subprocess.check_call, jobs))
    subprocess.check_call([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC",
                           "-fsanitize=address,undefined", "-shared-libsan", *objs, "-o", OUT])
    for o in objs:
        os.remove(o)
    os.rmdir(objdir)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(asan_runtime())
