"""`OL_SURF_REFERENCE_NEWTON` (ABI 11, opt-in: `integration.enable(reference_newton=True)` /
OPTILAND_HIP_REFERENCE_NEWTON=1): the reference's OWN stop rule on Newton-Raphson surfaces.

`NewtonRaphsonGeometry.distance` (geometries/newton_raphson.py:119-168) iterates the whole batch
of a trace call in lockstep and leaves when `max_j |f_j| < tol`: every ray takes the same number
K of updates -- `max_iter` when any ray of the batch is NaN (`be.max` of an array with a NaN is
NaN and `NaN < tol` is False) -- and the normal is evaluated at the end point.  The default
kernels stop per ray and converge further.  With the option the drop-in gives the reference's
NUMBERS: to rounding with the factory settings, and also for a user-set loose `tol` / small
`max_iter`, where the default leaves the 1e-6 contract ("what's weak" 1a:
`tol = 1e-2, max_iter = 1` -> 8.4e-6 relative).

The engine under test is the product's own `HipSystem` on the host build of the kernel source
(tests/_hostmath.make_engine_class); `tests/test_gpu_live_reference.py` repeats the core of it on
the MI355X.
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


def _np(be, a):
    return np.asarray(be.to_numpy(a), dtype=np.float64)


def _numpy_reference(be, build, call):
    """`call(lens)` on the reference's NumPy backend (the parity target), then back to torch."""
    be.set_backend("numpy")
    try:
        lens = build()
        out = call(lens)
        rays = {k: np.array(getattr(out, k), dtype=np.float64)
                for k in ("x", "y", "z", "L", "M", "N", "i", "opd")}
        surf = {k: np.array(getattr(lens.surfaces, k), dtype=np.float64)
                for k in ("x", "y", "z", "L", "M", "N", "intensity", "opd")}
    finally:
        be.set_backend("torch")
        be.set_device("cpu")
        be.set_precision("float64")
    return rays, surf


def _worst(be, got_rays, lens, want):
    rays, surf = want
    scale = max(np.nanmax(np.abs(surf["x"])), np.nanmax(np.abs(surf["y"])), 1.0)
    worst = 0.0
    for k, v in rays.items():
        g = _np(be, getattr(got_rays, k))
        assert np.array_equal(np.isnan(g), np.isnan(v)), k
        s = scale if k in ("x", "y", "z", "opd") else 1.0
        if np.isfinite(v).any():
            worst = max(worst, float(np.nanmax(np.abs(g - v))) / s)
    for k, v in surf.items():
        g = _np(be, getattr(lens.surfaces, k))
        assert g.shape == v.shape, k
        assert np.array_equal(np.isnan(g), np.isnan(v)), k
        s = scale if k in ("x", "y", "z", "opd") else 1.0
        if np.isfinite(v).any():
            worst = max(worst, float(np.nanmax(np.abs(g - v))) / s)
    return worst


def _asphere(tol=None, max_iter=None):
    def build():
        from optiland.samples.simple import AsphericSinglet
        lens = AsphericSinglet()
        lens.fields.add(y=8.0)   # (axis-parallel rays meet f(t) LINEAR in t: one step is exact)
        for s in lens.surfaces.surfaces:
            g = s.geometry
            if hasattr(g, "max_iter"):
                if tol is not None:
                    g.tol = tol
                if max_iter is not None:
                    g.max_iter = max_iter
        return lens
    return build


def _call(lens):
    return lens.trace(0.0, 1.0, 0.55, 8, "hexapolar")


def test_loose_tolerance_follows_the_reference_only_with_the_option(host_engine):
    """The verdict's case: `tol = 1e-2, max_iter = 1`.  Per-ray stop: every ray takes its one
    update and the kernel is 1e-5 from the reference, whose batch stops where the WORST ray is
    below 1e-2 (K = 0 or 1 for everybody).  With the option: rounding."""
    be = host_engine
    from optiland_amd.integration import install
    build = _asphere(tol=1e-2, max_iter=1)
    want = _numpy_reference(be, build, _call)
    lens = build()
    tracer = install(lens, force=True, reference_newton=False)
    plain = _worst(be, _call(lens), lens, want)
    assert tracer.last_path == "hip"
    lens = build()
    tracer = install(lens, force=True, reference_newton=True)
    exact = _worst(be, _call(lens), lens, want)
    assert tracer.last_path == "hip"
    assert exact <= 1e-12, exact
    assert plain > 1e-8, plain   # (the case is one where the two rules differ at all)


@pytest.mark.parametrize("tol,max_iter", [(None, None), (1e-3, 3), (1e-6, 2), (1e-12, 100),
                                          (1e-2, 0)])
def test_counts_of_the_batch(host_engine, tol, max_iter):
    be = host_engine
    from optiland_amd.integration import install
    build = _asphere(tol=tol, max_iter=max_iter)
    want = _numpy_reference(be, build, _call)
    lens = build()
    install(lens, force=True, reference_newton=True)
    assert _worst(be, _call(lens), lens, want) <= 1e-12


def test_a_nan_ray_sends_the_batch_to_max_iter(host_engine):
    """One ray that misses the base conic of the asphere: `be.max(|f|)` is NaN from the first
    iteration on, the reference's loop never breaks, every OTHER ray takes `max_iter` updates.
    With `max_iter = 2` and a loose tolerance that is visible; the NaN masks agree too."""
    be = host_engine
    from optiland_amd.integration import install
    build = _asphere(tol=1e-1, max_iter=2)

    def call(lens):
        # a pupil point far outside the lens: the ray misses the front surface's base conic
        px = be.array([0.0, 0.3, -0.5, 40.0])
        py = be.array([0.0, 0.4, 0.2, 0.0])
        return lens.surfaces.trace(_rays(be, lens, px, py))

    def _rays(be, lens, px, py):
        from optiland.rays import RealRays
        n = len(px)
        M = be.full_like(px, 0.12)   # (tilted: along the axis f(t) is linear, one step exact)
        return RealRays(px * 6.0, py * 6.0, be.full_like(px, -5.0), be.zeros_like(px),
                        M, be.sqrt(1.0 - M * M), be.ones_like(px), be.full_like(px, 0.55))

    want = _numpy_reference(be, build, call)
    assert np.isnan(want[0]["x"]).any() and np.isfinite(want[0]["x"]).any()
    lens = build()
    install(lens, force=True, reference_newton=True)
    got = call(lens)
    assert _worst(be, got, lens, want) <= 1e-12
    lens = build()
    install(lens, force=True, reference_newton=False)
    assert _worst(be, call(lens), lens, want) > 1e-9   # per-ray rule: the finite rays stop early


def test_the_count_is_per_trace_call(host_engine):
    """K belongs to the batch: the same ray traced alone and traced next to a slower one gets
    different iteration counts in the reference -- and here."""
    be = host_engine
    from optiland_amd.integration import install
    build = _asphere(tol=1e-4, max_iter=50)

    def alone(lens):
        return lens.trace_generic(0.0, 0.0, 0.05, 0.0, 0.55)

    def together(lens):
        return lens.trace_generic(0.0, 0.0, be.array([0.05, 0.98]), be.array([0.0, 0.0]), 0.55)

    for call in (alone, together):
        want = _numpy_reference(be, build, call)
        lens = build()
        install(lens, force=True, reference_newton=True)
        assert _worst(be, call(lens), lens, want) <= 1e-12


def test_zernike_and_polarised(host_engine):
    """C5's system (Zernike freeform + Fresnel coatings, polarised): the counting launches run
    the unpolarised kernel (geometry does not depend on the polarisation), the final launch the
    polarised one with the counts."""
    be = host_engine
    from optiland_amd.integration import install
    from tests import _live

    def build():
        lens = _live.zernike_fresnel("elliptical")
        g = lens.surfaces.surfaces[1].geometry
        g.tol, g.max_iter = 1e-3, 4
        return lens

    def call(lens):
        return lens.trace(0.0, 1.0, 0.55, 6, "hexapolar")

    want = _numpy_reference(be, build, call)
    lens = build()
    tracer = install(lens, force=True, reference_newton=True)
    got = call(lens)
    assert tracer.last_path == "hip"
    assert _worst(be, got, lens, want) <= 1e-12


def test_two_newton_surfaces_and_a_segment(host_engine):
    """Two aspheric faces (each count depends on the one before it), the RC telescope of C4, and
    the `SurfaceGroup.trace(rays, skip)` seam with caller-made rays entering behind the first
    Newton surface."""
    be = host_engine
    from optiland_amd.integration import install
    from tests import _live

    def two_faces():
        from optiland import optic as optic_mod
        lens = optic_mod.Optic(name="two aspheric faces")
        lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
        lens.surfaces.add(index=1, thickness=6.0, radius=25.0, is_stop=True, material="N-SF11",
                          surface_type="even_asphere", conic=-0.3,
                          coefficients=[-1.1e-4, -3.0e-6, 2.0e-8])
        lens.surfaces.add(index=2, thickness=24.0, radius=-60.0, surface_type="even_asphere",
                          conic=0.2, coefficients=[2.0e-4, -1.0e-6])
        lens.surfaces.add(index=3)
        lens.set_aperture(aperture_type="EPD", value=16.0)
        lens.fields.set_type(field_type="angle")
        lens.fields.add(y=0)
        lens.fields.add(y=10.0)
        lens.wavelengths.add(value=0.55, is_primary=True)
        for k, (tol, it) in ((1, (1e-3, 4)), (2, (1e-5, 6))):
            lens.surfaces.surfaces[k].geometry.tol = tol
            lens.surfaces.surfaces[k].geometry.max_iter = it
        return lens

    def rc():
        lens = _live.rc_asphere()
        lens.surfaces.surfaces[4].geometry.tol = 1e-5
        lens.surfaces.surfaces[4].geometry.max_iter = 3
        return lens

    def call(lens):
        return lens.trace(0.0, 1.0, 0.55, 5, "hexapolar")

    def segment(lens):
        from optiland.rays import RealRays
        g = np.random.default_rng(7)
        x, y = (be.array(v) for v in g.uniform(-5, 5, (2, 9)))
        M = be.array(g.uniform(-0.1, 0.1, 9))
        rays = RealRays(x, y, be.full_like(x, 3.0), be.zeros_like(x), M, be.sqrt(1.0 - M * M),
                        be.ones_like(x), be.full_like(x, 0.55))
        return lens.surfaces.trace(rays, skip=2)

    # (the RC mirrors are nearly parabolic, |1 + k| = 1e-3: the reference's own quadratic is 2e-9 mm
    # off there, tests/test_reference_root.py -- both of its forms, then)
    for build, run, root in ((two_faces, call, False), (two_faces, segment, False),
                             (rc, call, True)):
        want = _numpy_reference(be, build, run)
        lens = build()
        install(lens, force=True, reference_newton=True, reference_root=root)
        assert _worst(be, run(lens), lens, want) <= 1e-12, (build.__name__, run.__name__)


def test_option_is_part_of_the_change_detector(host_engine):
    """Switching the option re-packs: a table packed without the flag is not reused."""
    be = host_engine
    from optiland_amd import integration as ig
    build = _asphere(tol=1e-2, max_iter=1)
    want = _numpy_reference(be, build, _call)
    lens = build()
    ig.install(lens, force=True, reference_newton=False)
    off = _worst(be, _call(lens), lens, want)
    ig._set_reference_newton(True)
    on = _worst(be, _call(lens), lens, want)
    ig._set_reference_newton(False)
    off2 = _worst(be, _call(lens), lens, want)
    assert on <= 1e-12 < off and off2 == off


def test_fused_seams_stand_back(host_engine):
    """SpotDiagram on an optic with reference-rule surfaces: the fused spot kernel declines (its
    Newton loop is the per-ray one), the reference's own loop runs on the drop-in's traces and
    the data are the NumPy backend's."""
    be = host_engine
    from optiland import analysis
    from optiland_amd import analysis_seams
    from optiland_amd.integration import install
    build = _asphere(tol=1e-3, max_iter=2)
    be.set_backend("numpy")
    want = analysis.SpotDiagram(build(), num_rings=4).rms_spot_radius()
    want = [[float(v) for v in row] for row in want]
    be.set_backend("torch")
    be.set_device("cpu")
    be.set_precision("float64")
    lens = build()
    install(lens, force=True, reference_newton=True)
    before = analysis_seams.STATS["spot_grid"]
    got = analysis.SpotDiagram(lens, num_rings=4).rms_spot_radius()
    assert analysis_seams.STATS["spot_grid"] == before
    for a, b in zip(got, want):
        for x, y in zip(a, b):
            assert abs(float(be.to_numpy(x)) - y) <= 1e-12 * max(abs(y), 1e-3)
