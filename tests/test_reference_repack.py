"""An optic that is edited between traces (optimisers, tolerancing loops): the drop-in must
(a) always trace the CURRENT prescription, (b) re-read only what changed and patch the device
table in place instead of rebuilding everything.

Live reference on CPU; the engine is the product's `HipSystem` class on the host build of the
kernel source (it has `ol_system_update`), the oracle-backed stand-in where noted.
"""

import os

import numpy as np
import pytest

from tests.test_reference_integration import REF, hip_on_cpu, ref  # noqa: F401 (fixtures)

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "optiland")),
                                reason="reference package not present")
PLANES = ("x", "y", "z", "L", "M", "N", "intensity", "opd")


def _np(be, a):
    return np.asarray(be.to_numpy(a), dtype=np.float64)


def _fresh_numpy_trace(be, build, edit, call):
    be.set_backend("numpy")
    try:
        lens = build()
        edit(lens)
        call(lens)
        return {k: np.asarray(getattr(lens.surfaces, k), dtype=np.float64) for k in PLANES}
    finally:
        be.set_backend("torch")
        be.set_device("cpu")
        be.set_precision("float64")


EDITS = {
    "radius": lambda lens, v: lens.updater.set_radius(50.0 + v, 2),
    "conic": lambda lens, v: lens.updater.set_conic(-0.1 * v, 3),
    "thickness": lambda lens, v: lens.updater.set_thickness(6.0 + 0.1 * v, 2),
    "index": lambda lens, v: lens.updater.set_index(1.60 + 0.01 * v, 1),
}


@pytest.mark.parametrize("what", list(EDITS))
def test_edit_then_trace_follows_the_prescription(hip_on_cpu, what):
    be = hip_on_cpu
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import integration
    lens = CookeTriplet()
    t = integration.install(lens, force=True)
    call = lambda o: o.trace(0.0, 0.7, 0.55, 4, "hexapolar")  # noqa: E731
    try:
        call(lens)
        for step in (1.0, 2.0, 3.0):
            EDITS[what](lens, step)
            call(lens)
            assert t.last_path == "hip"
            want = _fresh_numpy_trace(be, CookeTriplet, lambda o: EDITS[what](o, step), call)
            for k in PLANES:
                np.testing.assert_allclose(_np(be, getattr(lens.surfaces, k)), want[k],
                                           rtol=1e-9, atol=1e-9, err_msg=f"{what} {step} {k}")
    finally:
        integration.uninstall(lens)


def test_a_repack_reads_only_the_edited_surface_and_patches_the_table(hip_on_cpu, request):
    be = hip_on_cpu
    from optiland.samples.objectives import DoubleGauss
    from optiland_amd import integration, packer
    lens = DoubleGauss()
    t = integration.install(lens, force=True)
    reads = {"n": 0}
    orig = packer._pack_surface_local

    def counting(*a, **k):
        reads["n"] += 1
        return orig(*a, **k)

    packer._pack_surface_local = counting
    try:
        px, py = be.array([0.0, 0.3, -0.5]), be.array([0.1, -0.2, 0.4])
        lens.trace_generic(0.0, 0.7, px, py, 0.5876)
        n_surf = len(lens.surfaces.surfaces)
        assert reads["n"] == n_surf and t.pack_count == 1
        created = t._hip_engine
        for k in range(1, 6):
            reads["n"] = 0
            lens.updater.set_radius(55.0 + 0.01 * k, 2)
            lens.trace_generic(0.0, 0.7, px, py, 0.5876)
            assert t.pack_count == 1 + k
            # the edited surface (its neighbours share no object with it) -- not all 13
            assert 1 <= reads["n"] <= 2, reads["n"]
            if "kernel-source" in request.node.name:   # the engine with ol_system_update
                assert t._hip_engine is created and t.engine_updates == k
                assert len(t._hip_engines) == 1
        # after a miss the next call validates before it launches; a hit re-arms speculation
        lens.trace_generic(0.0, 0.7, px, py, 0.5876)
        misses = t.speculative_misses
        lens.trace_generic(0.0, 0.7, px, py, 0.5876)
        lens.trace_generic(0.0, 0.7, px, py, 0.5876)
        assert t.speculative_misses == misses and t.speculative_hits >= 1
    finally:
        packer._pack_surface_local = orig
        integration.uninstall(lens)


def test_update_that_does_not_fit_falls_back_to_a_new_system(hip_on_cpu, request):
    """More Zernike / asphere coefficients than the allocated block has room for, another
    surface count: `ol_system_update` declines, a new system is created, results follow."""
    be = hip_on_cpu
    if "kernel-source" not in request.node.name:
        pytest.skip("needs the engine with ol_system_update")
    from optiland.samples.simple import AsphericSinglet
    from optiland_amd import integration
    lens = AsphericSinglet()
    t = integration.install(lens, force=True)
    try:
        lens.trace(0.0, 0.0, 0.587, 3, "hexapolar")
        eng0 = t._hip_engine
        g = lens.surfaces[1].geometry
        g.coefficients = list(g.coefficients) + [0.0 for k in range(100)]  # > head room
        lens.trace(0.0, 0.0, 0.587, 3, "hexapolar")
        assert t._hip_engine is not eng0 and t.last_path == "hip"
        want = _fresh_numpy_trace(
            be, AsphericSinglet,
            lambda o: setattr(o.surfaces[1].geometry, "coefficients",
                              list(o.surfaces[1].geometry.coefficients)
                              + [0.0 for k in range(100)]),
            lambda o: o.trace(0.0, 0.0, 0.587, 3, "hexapolar"))
        np.testing.assert_allclose(_np(be, lens.surfaces.z), want["z"], rtol=0, atol=1e-7)
    finally:
        integration.uninstall(lens)


@pytest.mark.parametrize("seed", range(12))
def test_incremental_pack_equals_full_pack_on_random_lenses(ref, seed):
    """pack_optic with the per-surface cache == pack_optic without, byte for byte, after
    random edits (boolean / polygon apertures and coatings relocate their blocks)."""
    be = ref
    be.set_backend("numpy")
    from optiland_amd import fingerprint as fp
    from optiland_amd.packer import UnsupportedSystem, pack_optic
    from tests.test_reference_fuzz import build_random_lens
    lens, rng = build_random_lens(seed, be)
    w = float(lens.primary_wavelength)
    cache = {}
    handed_out = []   # (table, its bytes when it was returned): later packs patch COPIES
    try:
        for step in range(5):
            if step:
                i = int(rng.integers(1, len(lens.surfaces.surfaces) - 1))
                g = lens.surfaces[i].geometry
                coef = getattr(g, "coefficients", None)
                if step == 4 and coef is not None and np.ndim(coef) == 1 and len(coef):
                    # a coefficient block that changes its LENGTH: every later block moves
                    g.coefficients = list(np.asarray(coef, dtype=float)) + [0.0, 0.0]
                elif hasattr(g, "radius") and np.isfinite(float(g.radius)):
                    g.radius = float(g.radius) * (1.0 + 1e-3 * step)
                else:
                    lens.surfaces[i].geometry.cs.x = float(lens.surfaces[i].geometry.cs.x) + 1e-3
            tok, _keep = fp.optic_token(lens, w)
            a = pack_optic(lens, wavelengths=[w], tokens=tok[1], cache=cache)
            b = pack_optic(lens, wavelengths=[w])
            assert a.surfaces.tobytes() == b.surfaces.tobytes()
            assert a.coeffs.tobytes() == b.coeffs.tobytes()
            assert a.optics.tobytes() == b.optics.tobytes()
            assert a.raygen == b.raygen
            for t_old, bytes_old in handed_out:
                assert (t_old.surfaces.tobytes(), t_old.coeffs.tobytes(),
                        t_old.optics.tobytes()) == bytes_old
            handed_out.append((a, (a.surfaces.tobytes(), a.coeffs.tobytes(),
                                   a.optics.tobytes())))
    except UnsupportedSystem:
        pytest.skip("outside the fused path")
