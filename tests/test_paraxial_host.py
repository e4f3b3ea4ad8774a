"""optiland_amd/paraxial_host.py against the reference's `optic.paraxial` (build container).

The packer takes the generator's first-order scalars (EPL, EPD, XPL; via them the launch
offset and the exit-pupil position) from a plain-float restatement of the reference's
paraxial traces instead of ~400 backend array operations per re-pack.  Same recurrences in
the same order: the values must agree to rounding (1e-12 relative) on every sample lens,
for every system-aperture type, on random lenses -- and the packed generator scalars must be
IDENTICAL with the restatement switched off (OPTILAND_HIP_HOST_PARAXIAL=0).
"""

import inspect
import os

import numpy as np
import pytest

from tests import _live

pytestmark = pytest.mark.skipif(_live.reference_root() is None,
                                reason="reference package not present")


@pytest.fixture(params=["numpy", "torch"])
def be(request):
    be = _live.import_reference()
    be.set_backend(request.param)
    if request.param == "torch":
        be.set_device("cpu")
        be.set_precision("float64")
    yield be
    be.set_backend("numpy")


def _f(be, v):
    return float(np.asarray(be.to_numpy(v), dtype=np.float64).reshape(-1)[0])


def _sample_lenses():
    import importlib
    out = []
    for modname in ("objectives", "simple", "telescopes", "eyepieces", "infrared", "microscopes",
                    "lithography"):
        try:
            mod = importlib.import_module(f"optiland.samples.{modname}")
        except Exception:  # noqa: BLE001 - a samples module this version does not have
            continue
        for name, cls in inspect.getmembers(mod, inspect.isclass):
            if cls.__module__ == mod.__name__:
                out.append((f"{modname}.{name}", cls))
    return out


def _host(lens, be):
    from optiland_amd import packer
    from optiland_amd.packer import UnsupportedSystem, pack_surfaces
    w = _f(be, lens.primary_wavelength)
    try:
        table = pack_surfaces(lens.surfaces, [w])
    except UnsupportedSystem:
        return None
    pos = np.asarray(table.surfaces["origin"][:, 2], dtype=np.float64)
    return packer._host_first_order(lens, table, pos, bool(lens.object_surface.is_infinite))


def test_host_first_order_equals_optic_paraxial_on_the_samples(be):
    checked = 0
    for label, cls in _sample_lenses():
        try:
            lens = cls()
        except Exception:  # noqa: BLE001 - samples that need files / extras
            continue
        fo = _host(lens, be)
        if fo is None:
            continue
        want = {"EPL": lens.paraxial.EPL(), "EPD": lens.paraxial.EPD(), "XPL": lens.paraxial.XPL(),
                "f2": lens.paraxial.f2()}
        for k, v in want.items():
            np.testing.assert_allclose(fo[k], _f(be, v), rtol=1e-12, atol=1e-12,
                                       err_msg=f"{label}: {k}")
        checked += 1
    assert checked >= 10, checked


@pytest.mark.parametrize("ap_type,value", [("EPD", 8.0), ("imageFNO", 4.0), ("objectNA", 0.08),
                                           ("float_by_stop_size", 3.5)])
@pytest.mark.parametrize("finite", [False, True])
def test_every_system_aperture_type(be, ap_type, value, finite):
    from optiland.samples.objectives import CookeTriplet
    lens = CookeTriplet()
    if ap_type == "objectNA" and not finite:
        pytest.skip("objectNA needs a finite object")
    if finite:
        lens.surfaces[0].geometry.cs.z = be.array(-250.0) if be.get_backend() == "torch" else -250.0
        lens.surfaces[0].thickness = 250.0
        lens.fields.set_type("object_height")
    lens.set_aperture(aperture_type=ap_type, value=value)
    fo = _host(lens, be)
    assert fo is not None
    for k, v in (("EPL", lens.paraxial.EPL()), ("EPD", lens.paraxial.EPD()),
                 ("XPL", lens.paraxial.XPL())):
        np.testing.assert_allclose(fo[k], _f(be, v), rtol=1e-12, atol=1e-12, err_msg=k)


@pytest.mark.parametrize("seed", range(40))
def test_random_lenses_pack_identically_with_and_without_the_restatement(be, seed, monkeypatch):
    from optiland_amd import packer
    from optiland_amd.packer import UnsupportedSystem, pack_optic
    from tests.test_reference_fuzz import build_random_lens
    lens, _rng = build_random_lens(seed, be)
    w = _f(be, lens.primary_wavelength)
    try:
        monkeypatch.setenv("OPTILAND_HIP_HOST_PARAXIAL", "0")
        packer._RAYGEN_CACHE.pop(lens, None)
        a = dict(pack_optic(lens, wavelengths=[w]).raygen)
        monkeypatch.setenv("OPTILAND_HIP_HOST_PARAXIAL", "1")
        packer._RAYGEN_CACHE.pop(lens, None)
        b = dict(pack_optic(lens, wavelengths=[w]).raygen)
    except UnsupportedSystem:
        pytest.skip("outside the fused path")
    assert a.keys() == b.keys()
    for k in a:
        np.testing.assert_allclose(b[k], a[k], rtol=1e-12, atol=1e-12, err_msg=k)


def test_uncovered_systems_fall_back_to_the_reference(be):
    from optiland.samples.objectives import CookeTriplet
    lens = CookeTriplet()
    lens.set_aperture(aperture_type="EPD", value=10.0)
    assert _host(lens, be) is not None
    for s in lens.surfaces:          # no stop surface: the reference raises, the host declines
        s.is_stop = False
    assert _host(lens, be) is None
