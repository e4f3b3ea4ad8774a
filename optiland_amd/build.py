"""Compile the HIP kernels + C ABI into optiland_amd/lib/liboptiland_hip.so.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU, so this runs in the
build container; the resulting .so is git-ignored but travels with the tree to
the GPU box.
"""

from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIBNAME = "liboptiland_hip.so"
# trace_kernel.hip is compiled twice -- fp32 and fp64 instantiations in separate
# translation units -- so that the two halves (the bulk of the build) run in parallel
SOURCES = ("trace_kernel_f32.hip", "trace_kernel_f64.hip", "aux_kernels.hip", "capi.hip")
# every header under csrc/ (round 5: a hand-kept list had missed wavefront_fit_device.h since
# round 4 -- an edit of that file alone left a stale aux_kernels.o behind) + the included .hip
HEADERS = tuple(sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))) + ("trace_kernel.hip",)
ARCH = "gfx950"


def library_path() -> str:
    return os.path.join(LIBDIR, LIBNAME)


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found (need ROCm's hipcc to build the HIP extension)")
    return exe


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force: bool = False, verbose: bool = False) -> str:
    """Build (if stale) and return the path of the shared library."""
    os.makedirs(LIBDIR, exist_ok=True)
    pub = os.path.join(HERE, "..", "include", "optiland_hip.h")
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [pub]
    objs, jobs = [], []
    flags = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-ffp-contract=on",
             "-fno-math-errno", "-Wall"]
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + hdrs):
            jobs.append([_hipcc(), *flags, "-c", s, "-o", o])
        objs.append(o)
    if jobs:
        from concurrent.futures import ThreadPoolExecutor
        if verbose:
            for cmd in jobs:
                print(" ".join(cmd))
        with ThreadPoolExecutor(max_workers=len(jobs)) as pool:
            list(pool.map(subprocess.check_call, jobs))
    so = library_path()
    if force or _stale(so, objs):
        cmd = [_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", *objs, "-o", so]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return so


def fptoken_path() -> str:
    return os.path.join(LIBDIR, "_fptoken.so")


def build_fptoken(force: bool = False, verbose: bool = False) -> str:
    """gcc-build the CPython extension behind optiland_amd/fingerprint.py (the native inner
    loop of the drop-in's change detector; host runtime, no GPU code).  The package works
    without it (pure-Python walk, ~5x slower)."""
    import sysconfig
    os.makedirs(LIBDIR, exist_ok=True)
    src, out = os.path.join(CSRC, "fptoken.c"), fptoken_path()
    if force or _stale(out, [src]):
        cmd = ["gcc", "-O2", "-shared", "-fPIC", "-Wall", "-I", sysconfig.get_paths()["include"],
               src, "-o", out]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    print(build_library(force=False, verbose=True))
    print(build_fptoken(force=False, verbose=True))
