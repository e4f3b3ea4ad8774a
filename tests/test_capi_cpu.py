"""The C-ABI shared library on a machine WITHOUT a GPU: it must load, export
every symbol the public header declares, reject bad arguments with error codes,
and fail loudly (never fall back) when no HIP device exists."""

import ctypes as C
import os
import re
import subprocess
import tempfile

import numpy as np
import pytest
import torch

from optiland_amd import _capi, build, system as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "optiland_hip.h")


@pytest.fixture(scope="module")
def lib():
    build.build_library()
    return _capi.load()


def _declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ol_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported(lib):
    names = _declared_functions()
    assert set(names) == set(_capi.EXPORTS), (names, _capi.EXPORTS)
    out = subprocess.check_output(["nm", "-D", "--defined-only", _capi.library_path()], text=True)
    exported = set(re.findall(r" T (ol_[a-z0-9_]+)", out))
    assert set(names) <= exported
    for n in names:
        getattr(lib, n)


def test_abi_version(lib):
    assert lib.ol_abi_version() == _capi.ABI_VERSION == 11


def test_struct_layouts_agree_with_the_c_compiler():
    """numpy dtype == ctypes Structure == what gcc lays out for the header."""
    src = r'''
#include <stdio.h>
#include <stddef.h>
#include "optiland_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(ol_surface_desc),
         offsetof(ol_surface_desc, radius), offsetof(ol_surface_desc, origin),
         offsetof(ol_surface_desc, rot), offsetof(ol_surface_desc, aperture),
         offsetof(ol_surface_desc, coat), sizeof(ol_surface_optics), sizeof(ol_raygen_params));
  printf("%zu %zu\n", sizeof(ol_polarization_state), offsetof(ol_raygen_params, EPL));
  printf("%zu %zu %zu\n", sizeof(ol_wavefront_params), offsetof(ol_wavefront_params, wavelength_um),
         offsetof(ol_wavefront_params, nx));
  return 0;
}'''
    with tempfile.TemporaryDirectory() as d:
        c = os.path.join(d, "layout.c")
        open(c, "w").write(src)
        exe = os.path.join(d, "layout")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        nums = [int(v) for v in subprocess.check_output([exe], text=True).split()]
    dt = S.SURFACE_DESC_DTYPE
    assert nums[0] == dt.itemsize == C.sizeof(_capi.SurfaceDesc)
    for off, name in zip(nums[1:6], ("radius", "origin", "rot", "aperture", "coat")):
        assert off == dt.fields[name][1] == getattr(_capi.SurfaceDesc, name).offset, name
    assert nums[6] == S.SURFACE_OPTICS_DTYPE.itemsize == C.sizeof(_capi.SurfaceOptics)
    assert nums[7] == S.RAYGEN_DTYPE.itemsize == C.sizeof(_capi.RaygenParams)
    assert nums[8] == C.sizeof(_capi.PolarizationStateC)
    assert nums[9] == S.RAYGEN_DTYPE.fields["EPL"][1]
    from oracle.oracle import WavefrontParams as OracleWavefrontParams
    for cls in (_capi.WavefrontParams, OracleWavefrontParams):
        assert nums[10] == C.sizeof(cls)
        assert nums[11] == cls.wavelength_um.offset and nums[12] == cls.nx.offset


def test_argument_validation_without_a_device(lib):
    ptrs = (C.c_void_p * 8)()
    rc = lib.ol_trace(None, 0, 10, ptrs, 0, None, 0, None, 0, 0, 1, None, None)
    assert rc == -1 and b"system is NULL" in lib.ol_last_error()
    handle = C.c_void_p()
    rc = lib.ol_system_create(None, 0, None, 0, None, 0, C.byref(handle))
    assert rc == -1 and b"no surfaces" in lib.ol_last_error()
    rc = lib.ol_generate_rays(None, 0, 1, None, None, None, None)
    assert rc == -1
    # host-side part of the range validation (launch-uniform field scalars) and the
    # argument rules of ol_raygen_inputs need no device
    from optiland_amd._capi import RaygenInputs, RaygenParams
    par = RaygenParams(1, 0, 10.0, 5.0, 20.0, 5.0, 0.0)
    outp = (C.c_void_p * 8)(*([8] * 7 + [None]))  # never dereferenced: the call must fail first
    status = C.c_uint32(0)
    inp = RaygenInputs(None, None, 16, 16, None, None, 0.0, 1.5, 1.0, 1.0, 0x1, 0)
    rc = lib.ol_generate_rays(C.byref(par), 0, 4, C.byref(inp), outp, C.byref(status), None)
    assert rc == -1
    assert lib.ol_last_error() == b"Normalized field coordinates must be within (-1, 1)"
    inp = RaygenInputs(16, None, 16, 16, None, None, 0.0, 0.0, 1.0, 1.0, 0, 0)
    rc = lib.ol_generate_rays(C.byref(par), 0, 4, C.byref(inp), outp, None, None)
    assert rc == -1 and b"must be given together" in lib.ol_last_error()
    inp = RaygenInputs(None, None, 16, 16, None, None, 0.0, 0.0, 1.0, 1.0, 0x2, 0)
    rc = lib.ol_generate_rays(C.byref(par), 0, 4, C.byref(inp), outp, None, None)
    assert rc == -1 and b"needs a status word" in lib.ol_last_error()
    rc = lib.ol_spot_moments(0, 4, None, None, None, None, None)
    assert rc == -1
    rc = lib.ol_trace_spot(None, 0, 4, None, None, 0.0, 0.0, 0, None, None, None, None)
    assert rc == -1 and b"system is NULL" in lib.ol_last_error()
    # ABI 10: the batched spot refuses before it touches a device
    rc = lib.ol_trace_spot_batch(None, 0, 4, None, None, 1, None, None, 0, None, None, None)
    assert rc == -1 and b"system is NULL" in lib.ol_last_error()
    assert lib.ol_system_num_surfaces(None) == 0
    lib.ol_system_destroy(None)  # no-op
    # ABI 9: ol_wavefront_fit / ol_wavefront_opd_fitted refuse before they touch a device
    from optiland_amd._capi import WavefrontParams
    w = WavefrontParams(n_image=1.0, wavelength_um=0.55, half_epd=5.0)
    planes = (C.c_void_p * 8)(*([16] * 8))  # never dereferenced
    rc = lib.ol_wavefront_fit(0, None, 3.0, 0, 0, 4, planes, 16, 16, 16, 16, 16, None)
    assert rc == -1 and b"NULL argument" in lib.ol_last_error()
    rc = lib.ol_wavefront_fit(7, C.byref(w), 3.0, 0, 0, 4, planes, 16, 16, 16, 16, 16, None)
    assert rc == -1 and b"unknown kind" in lib.ol_last_error()
    rc = lib.ol_wavefront_fit(0, C.byref(w), 3.0, 0x10, 0, 4, planes, 16, 16, 16, 16, 16, None)
    assert rc == -1 and b"unknown flags" in lib.ol_last_error()
    rc = lib.ol_wavefront_fit(0, C.byref(w), 3.0, 0, 0, -1, planes, 16, 16, 16, 16, 16, None)
    assert rc == -1 and b"negative count" in lib.ol_last_error()
    bad = WavefrontParams(n_image=0.0, wavelength_um=0.55, half_epd=5.0)
    rc = lib.ol_wavefront_fit(1, C.byref(bad), 3.0, 0, 0, 4, planes, 16, 16, 16, 16, 16, None)
    assert rc == -1 and b"n_image" in lib.ol_last_error()
    hole = (C.c_void_p * 8)(*([16] * 5 + [None] + [16] * 2))
    rc = lib.ol_wavefront_fit(1, C.byref(w), 3.0, 0, 0, 4, hole, 16, 16, 16, 16, 16, None)
    assert rc == -1 and b"rays[5] is NULL" in lib.ol_last_error()
    rc = lib.ol_wavefront_opd_fitted(4, planes, 16, 16, None, 16, None, None)
    assert rc == -1 and b"NULL argument" in lib.ol_last_error()
    three = (C.c_void_p * 3)(16, None, 16)
    rc = lib.ol_wavefront_opd_fitted(4, planes, 16, 16, 16, 16, three, None)
    assert rc == -1 and b"pupil needs three planes" in lib.ol_last_error()
    assert lib.ol_set_tuning(2, 4096) == -1 and lib.ol_set_tuning(2, 0) == 0
    # OL_TUNE_RECORD_WG_CAP: 0 (default policy), 1 (never), 2 ... 8
    assert lib.ol_set_tuning(3, 9) == -1 and b"workgroup cap" in lib.ol_last_error()
    assert lib.ol_set_tuning(3, -1) == -1
    assert all(lib.ol_set_tuning(3, v) == 0 for v in (8, 2, 1, 0))


def test_unsupported_kinds_are_refused(lib):
    surf = np.zeros(1, dtype=S.SURFACE_DESC_DTYPE)
    surf[0]["geom_kind"] = 17
    optics = np.ones((1, 1), dtype=S.SURFACE_OPTICS_DTYPE)
    handle = C.c_void_p()
    rc = lib.ol_system_create(surf.ctypes.data, 1, None, 0, optics.ctypes.data, 1, C.byref(handle))
    assert rc == -2 and b"geometry kind 17" in lib.ol_last_error()


@pytest.mark.skipif(torch.cuda.is_available(), reason="only meaningful without a GPU")
def test_no_gpu_means_loud_failure_not_fallback(lib):
    from optiland_amd import load_system
    from optiland_amd.engine import HipSystem
    with pytest.raises(_capi.HipExtensionError, match="no HIP device|no CPU fallback"):
        HipSystem(load_system("double_gauss"))
    # and straight through the C ABI
    t = load_system("double_gauss")
    handle = C.c_void_p()
    surf, optics = np.ascontiguousarray(t.surfaces), np.ascontiguousarray(t.optics)
    rc = lib.ol_system_create(surf.ctypes.data, surf.shape[0], None, 0, optics.ctypes.data, 1,
                              C.byref(handle))
    assert rc == -3 and handle.value is None
    assert b"hip" in lib.ol_last_error().lower()


def test_missing_library_raises(monkeypatch, tmp_path):
    monkeypatch.setattr(_capi, "_LIB", None)
    monkeypatch.setenv("OPTILAND_HIP_LIBRARY", str(tmp_path / "nope.so"))
    with pytest.raises(_capi.HipExtensionError, match="HIP extension not built"):
        _capi.load()


def test_product_never_imports_the_oracle():
    """oracle/ is test infrastructure: nothing under optiland_amd/ may reference it."""
    pkg = os.path.join(ROOT, "optiland_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "liboracle" not in text, f


def test_system_table_json_roundtrip():
    from optiland_amd import available_systems, load_system
    for name in available_systems():
        t = load_system(name)
        t2 = S.SystemTable.from_json(t.to_json())
        assert t2.surfaces.tobytes() == t.surfaces.tobytes()
        assert np.array_equal(t2.coeffs, t.coeffs)
        assert t2.optics.tobytes() == t.optics.tobytes()
        assert t2.raygen == t.raygen and t2.polarization == t.polarization
        assert t.wavelength_index(float(t.wavelengths[0])) == 0
