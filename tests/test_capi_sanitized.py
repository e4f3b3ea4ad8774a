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
        import torch
        # here (build container, no GPU) a library older than the sources is rebuilt; the GPU
        # box takes the one that travelled with the snapshot (its copy has fresh mtimes)
        if not os.path.exists(lib) or not torch.cuda.is_available():
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


GPU_SCRIPT = r'''
# torch-free on purpose: only libamdhip64 + the sanitized shim live in this process
import ctypes as C, os, sys
import numpy as np
from optiland_amd import _capi
from optiland_amd.system import SystemTable
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipFree.argtypes = [C.c_void_p]
H2D, D2H = 1, 2
def chk(rc, what):
    assert rc == 0, (what, rc)
class Dev:
    """Device array with a poisoned guard band on both sides."""
    G = 256
    def __init__(self, n, dtype, fill=None):
        self.n, self.dtype = n, np.dtype(dtype)
        self.nbytes = n * self.dtype.itemsize
        self.base = C.c_void_p()
        chk(hip.hipMalloc(C.byref(self.base), self.nbytes + 2 * self.G), "hipMalloc")
        chk(hip.hipMemset(self.base, 0xA5, self.nbytes + 2 * self.G), "hipMemset")
        self.ptr = self.base.value + self.G
        if fill is not None:
            a = np.ascontiguousarray(fill, dtype=self.dtype)
            chk(hip.hipMemcpy(self.ptr, a.ctypes.data, self.nbytes, H2D), "H2D")
    def get(self):
        out = np.empty(self.n, dtype=self.dtype)
        if self.n:
            chk(hip.hipMemcpy(out.ctypes.data, self.ptr, self.nbytes, D2H), "D2H")
        return out
    def guards_intact(self):
        g = np.empty(2 * self.G, dtype=np.uint8)
        chk(hip.hipMemcpy(g.ctypes.data, self.base.value, self.G, D2H), "D2H")
        chk(hip.hipMemcpy(g.ctypes.data + self.G, self.ptr + self.nbytes, self.G, D2H), "D2H")
        return bool((g == 0xA5).all())
lib = _capi.load()
assert "asan" in _capi.library_path()
root = sys.argv[1]
checked = 0
for name in ("double_gauss", "rc_asphere", "zernike_fresnel_fringe"):
    t = SystemTable.load(os.path.join(root, "optiland_amd", "data", name + ".json"))
    surf, optics = np.ascontiguousarray(t.surfaces), np.ascontiguousarray(t.optics)
    coeffs = np.ascontiguousarray(t.coeffs, dtype=np.float64)
    h = C.c_void_p()
    chk(lib.ol_system_create(surf.ctypes.data, surf.shape[0], coeffs.ctypes.data if coeffs.size else None,
                             coeffs.size, optics.ctypes.data, optics.shape[1], C.byref(h)), "create")
    rg = t.raygen
    p = _capi.RaygenParams(int(rg["object_infinite"]), int(rg.get("field_kind", 0)), rg["EPL"], rg["EPD"],
                           float(rg.get("field_scale", rg["max_field"])), rg["offset"], rg["z_first"],
                           float(rg.get("tele_dz", 0.0)), 0.0, 0.0, 0, 0)
    pol = t.uses_polarization
    S = surf.shape[0]
    for dt, npdt in ((_capi.F32, np.float32), (_capi.F64, np.float64)):
        for n in (1, 2, 63, 64, 65, 254, 255, 1000, 4098, 4099):
            rng = np.random.default_rng(n)
            r, th = 0.9 * np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
            px, py = Dev(n, npdt, r * np.cos(th)), Dev(n, npdt, r * np.sin(th))
            stride = (n + 63) // 64 * 64
            rec = Dev(S * 8 * stride, npdt)
            rays = [Dev(n, npdt) for _ in range(8)]
            status = Dev(1, np.uint32, [0])
            prt = Dev(9 * n, npdt) if pol else None
            inp = _capi.RaygenInputs(None, None, px.ptr, py.ptr, None, None, 0.0, 0.5, 1.0, 1.0,
                                     _capi.RAYGEN_CHECK_PUPIL, 0)
            ptrs = (C.c_void_p * 8)(*[d.ptr for d in rays])
            chk(lib.ol_generate_rays(C.byref(p), dt, n, C.byref(inp), ptrs, status.ptr, None), "raygen")
            flags = 0x8 if pol else 0   # OL_TRACE_PRT_IDENTITY
            import optiland_amd.system as Sy
            flags = Sy.TRACE_PRT_IDENTITY if pol else 0
            chk(lib.ol_trace(h, dt, n, ptrs, 0, rec.ptr, stride, prt.ptr if pol else None, 0, S - 1,
                             flags, status.ptr, None), "trace")
            chk(hip.hipDeviceSynchronize(), "sync")
            last = rec.get().reshape(S, 8, stride)[-1, :, :n]
            assert np.isfinite(last[0]).any(), (name, n)
            assert int(status.get()[0]) == 0, (name, n, status.get())
            # ol_trace_generate into the same buffers (ABI 6 ...): generation + trace in one
            # launch -- for the polarised Zernike system in fp32 with an even ray count the launch
            # on PAIRS of rays (8-byte stores to the record, the PRT planes and the updated
            # intensity; OL_POLZ_PAIR), with the update_intensity epilogue
            upd = Dev(n, npdt) if pol else None
            ex = _capi.TraceExtras()
            st = _capi.PolarizationStateC(0, 0, 0.0, 0.0, 0.0, 0.0)
            if pol:
                ex.update_intensity_state = C.addressof(st)
                ex.updated_intensity = upd.ptr
            chk(lib.ol_trace_generate(h, dt, n, C.byref(p), C.byref(inp), 0, rec.ptr, stride, ptrs,
                                      prt.ptr if pol else None, 0, status.ptr, C.byref(ex), None),
                "trace_generate")
            chk(hip.hipDeviceSynchronize(), "sync")
            again = rec.get().reshape(S, 8, stride)[-1, :, :n]
            assert np.array_equal(np.isnan(again), np.isnan(last)), (name, n)
            assert np.allclose(np.nan_to_num(again), np.nan_to_num(last), rtol=0, atol=1e-4), (name, n)
            if pol:
                assert np.isfinite(upd.get()[:n]).any(), (name, n)
            for d in [px, py, rec, status] + rays + ([prt, upd] if pol else []):
                assert d.guards_intact(), (name, dt, n)
                hip.hipFree(d.base)
            checked += 1
    # bad arguments on a live system: refused with an error code
    assert lib.ol_trace(h, 7, 10, ptrs, 0, None, 0, None, 0, S - 1, 0, None, None) != 0
    assert lib.ol_trace(h, _capi.F32, 10, ptrs, 0, None, 0, None, 3, 1, 0, None, None) != 0
    assert lib.ol_trace(h, _capi.F32, 10, ptrs, 5, None, 0, None, 0, S - 1, 0, None, None) != 0
    lib.ol_system_destroy(h)
print("guard-band launches", checked)
'''


@pytest.mark.gpu
def test_guard_band_cases_under_the_sanitized_host_library(san):
    """Launch wrappers + the caller-buffer pointer arithmetic under ASAN / UBSAN on the GPU
    box.  The process holds nothing but libamdhip64 and the sanitized shim (no torch: its
    allocator and the ROCm runtime are not ASAN-clean under LD_PRELOAD): device buffers
    with poisoned guard bands, `ol_generate_rays` + `ol_trace` (record-all, write-only PRT
    on the polarised system) at ragged / sub-wave / unaligned sizes in both precisions,
    guard bands checked after every launch, bad arguments refused."""
    out = subprocess.run([sys.executable, "-c", GPU_SCRIPT, ROOT], env=san, capture_output=True,
                         text=True, timeout=1200, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[:3000], out.stderr[-2000:])
    assert "ERROR: AddressSanitizer" not in out.stderr and "runtime error" not in out.stderr, \
        out.stderr[:4000]
    assert int(out.stdout.split("guard-band launches")[1].split()[0]) == 60
