"""`updater.set_radius / set_conic` on a biconic or toroidal surface (round 6, found by
tools/seam_fuzz.py `edit_loop`).

Those geometries hand `radius_x` / `radius_y` to `NewtonRaphsonGeometry` as the BASE radius
(biconic.py:56-66, toroidal.py:67-82) and keep their own Rx / R_yz, cx / c_yz for the sag.
`Optic.set_radius` writes `geometry.radius` alone (optic_updater.py:38-54): the sag does not
move, but `SurfaceGroup.radii` -- what the paraxial tracer reads -- does, and with it the
entrance pupil the rays are aimed at.  The packed table holds the profile's radius, so the
memo of the first-order scalars, keyed on the packed bytes, served the OLD pupil: 4e-4 mm on
every recorded row, until the next edit that re-packed.
"""

import os
import sys

import numpy as np
import pytest

REF = os.environ.get("OPTILAND_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "optiland")),
                                reason="reference package not present")


@pytest.fixture(scope="module")
def ref():
    shim = os.path.join(os.path.dirname(os.path.abspath(__file__)), "refshim")
    sys.dont_write_bytecode = True
    added = [p for p in (shim, REF) if p not in sys.path]
    sys.path[:0] = added
    import optiland.backend as be
    yield be
    be.set_backend("numpy")
    for p in added:
        sys.path.remove(p)


@pytest.fixture
def host_engine(ref, monkeypatch):
    import optiland_amd.tracer as tr
    from optiland_amd import system as S
    from tests import _hostmath as hm
    if not hm.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    cls = hm.make_engine_class()
    monkeypatch.setattr(tr, "_make_engine", lambda table, device: cls(table, device))
    be = ref
    be.set_backend("torch")
    be.set_device("cpu")
    be.set_precision("float64")
    before = dict(S.OPTIONS)
    yield be
    S.OPTIONS.update(before)
    be.set_backend("numpy")


def _lens(be, kind):
    from optiland import optic as optic_mod
    lens = optic_mod.Optic(name=f"{kind} singlet")
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, radius=40.0, thickness=4.0, material="N-BK7")
    if kind == "toroidal":
        lens.surfaces.add(index=2, surface_type="toroidal", radius_x=-55.0, radius_y=-43.0,
                          thickness=3.0)
    else:
        lens.surfaces.add(index=2, surface_type="biconic", radius_x=-55.0, radius_y=-43.0,
                          conic_x=-0.2, conic_y=0.1, thickness=3.0)
    lens.surfaces.add(index=3, radius=-80.0, thickness=60.0, material="N-SF5", is_stop=True)
    lens.surfaces.add(index=4)
    lens.set_aperture(aperture_type="EPD", value=8.0)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=4.0)
    lens.wavelengths.add(value=0.55, is_primary=True)
    return lens


def _steps(be, lens):
    """trace; move the base radius; trace; move the base conic; trace -- the recorded rows."""
    out = []

    def rows():
        lens.trace(0.2, 0.6, 0.55, 4, "hexapolar")
        return np.stack([np.asarray(be.to_numpy(getattr(lens.surfaces, k)), dtype=np.float64)
                         for k in ("x", "y", "z", "L", "M", "N")])
    out.append(rows())
    g = lens.surfaces.surfaces[2].geometry
    lens.updater.set_radius(float(be.to_numpy(g.radius)) * 1.002, 2)
    out.append(rows())
    lens.updater.set_conic(float(be.to_numpy(g.k)) + 2e-3, 2)
    out.append(rows())
    return out


@pytest.mark.parametrize("reference_newton", [False, True])
@pytest.mark.parametrize("kind", ["toroidal", "biconic"])
def test_base_radius_edit_moves_the_pupil_as_in_the_reference(host_engine, kind,
                                                              reference_newton):
    be = host_engine
    from optiland_amd import integration
    be.set_backend("numpy")
    want = _steps(be, _lens(be, kind))
    assert np.nanmax(np.abs(want[1][:, 0] - want[0][:, 0])) > 1e-6   # the pupil DID move
    be.set_backend("torch")
    be.set_device("cpu")
    be.set_precision("float64")
    lens = _lens(be, kind)
    tracer = integration.install(lens, force=True, reference_newton=reference_newton)
    got = _steps(be, lens)
    # (per-ray stop rule: within 1e-7 of the reference's iterate; its own rule: rounding)
    tol = 1e-11 if reference_newton else 5e-7
    for step, (a, b) in enumerate(zip(got, want)):
        assert np.array_equal(np.isnan(a), np.isnan(b)), step
        assert np.nanmax(np.abs(a - b)) <= tol, (step, float(np.nanmax(np.abs(a - b))))
    # with the option a base conic that left the profile's sends the surface to the
    # reference's loop (the iteration count depends on the start); without it the kernel serves
    assert tracer.last_path == ("reference" if reference_newton else "hip")
