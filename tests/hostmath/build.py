"""Build tests/hostmath/_build/libol_hostmath.so: the kernel's per-surface arithmetic
(optiland_amd/csrc/surface_math.h) compiled for the HOST behind the C ABI -- a checker for
boxes without a GPU.  See harness.hip for what it is and is not.

    python tests/hostmath/build.py

hipcc is used as the C++ compiler only (`--offload-host-only`: no device code is
generated); the link is a plain g++ link without the HIP runtime.
"""

from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "optiland_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libol_hostmath.so")
# every header of the kernel source + the shim (a hand-kept list had missed
# wavefront_fit_device.h, as optiland_amd/build.py's had)
DEPS = tuple(sorted(f for f in os.listdir(CSRC) if f.endswith(".h"))) + ("capi.hip",)


def _cpu_has_fma() -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in line + " "
    except OSError:
        pass
    return False


def available() -> bool:
    return bool(shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc"))


LIB_SAN = os.path.join(OUT, "libol_hostmath_asan.so")


def asan_runtime() -> str:
    import glob
    hits = sorted(glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so"))
    if not hits:
        raise RuntimeError("clang's ASAN runtime not found under /opt/rocm/lib/llvm")
    return hits[-1]


def build(force: bool = False, verbose: bool = False, sanitize: bool = False) -> str:
    """sanitize=True: the same two translation units with AddressSanitizer +
    UndefinedBehaviorSanitizer -> libol_hostmath_asan.so.  On the host the surface table,
    the coefficient blocks and the ray planes are ordinary heap memory, so every read and
    write of the KERNEL ARITHMETIC is checked -- something the device build cannot offer
    on this image (tools/build_sanitized.py covers the host side of the product library)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    os.makedirs(OUT, exist_ok=True)
    lib = LIB_SAN if sanitize else LIB
    deps = [os.path.join(CSRC, d) for d in DEPS] + [
        os.path.join(HERE, "harness.hip"), os.path.join(ROOT, "include", "optiland_hip.h"),
        os.path.abspath(__file__)]
    if not force and not os.environ.get("OL_HOSTMATH_CXXFLAGS") and os.path.exists(lib) and \
            all(os.path.getmtime(d) <= os.path.getmtime(lib) for d in deps):
        return lib
    # fma() is written out everywhere it matters; -mfma additionally lets the host
    # contract a*b+c the way the device compiler does (-ffp-contract=on on both sides)
    flags = ["--offload-host-only", "-std=c++17", "-fPIC", "-ffp-contract=on",
             "-fno-math-errno", "-Wall"] + (["-mfma"] if _cpu_has_fma() else [])
    flags += ["-O1", "-g1", "-fsanitize=address,undefined", "-fno-omit-frame-pointer",
              "-fno-sanitize-recover=undefined"] if sanitize else ["-O2"]
    # extra compile-time knobs of the kernel source (A/B variants checked for accuracy here
    # before they cost GPU time); a variant build goes to its own file
    extra = os.environ.get("OL_HOSTMATH_CXXFLAGS", "").split()
    if extra:
        flags += extra
        lib = lib.replace(".so", "_variant.so")
    objs = []
    for src in (os.path.join(CSRC, "capi.hip"), os.path.join(HERE, "harness.hip")):
        o = os.path.join(OUT, os.path.basename(src).replace(
            ".hip", "_asan.o" if sanitize else ("_variant.o" if extra else ".o")))
        cmd = [hipcc, *flags, "-c", src, "-o", o]
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, stderr=None if verbose else subprocess.DEVNULL)
        objs.append(o)
    if sanitize:  # clang links its own shared sanitizer runtime (LD_PRELOADed by the test)
        clang = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "clang++")
        if not os.path.exists(clang):
            clang = "/opt/rocm/lib/llvm/bin/clang++"
        cmd = [clang, "-shared", "-Wl,-Bsymbolic", "-fsanitize=address,undefined", "-shared-libsan",
               *objs, "-o", lib]
    else:
        cmd = ["g++", "-shared", "-Wl,-Bsymbolic", *objs, "-o", lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    import sys
    print(build(force=True, verbose=True, sanitize="--sanitize" in sys.argv))
