"""Host-side logic of the tracer mirror, on CPU: with the oracle-backed engine, and with
the product's engine class on the host build of the kernel source.

Checks that `HipRayTracer.trace / trace_generic` reproduce what the reference's
`Optic.trace / trace_generic` produced for the same arguments (goldens), i.e. the
field x pupil expansion order, vignetting handling, the polarised epilogue
semantics (trace only) and the error texts.
"""

import os

import numpy as np
import pytest
import torch

from optiland_amd import tracer as tr
from tests._fake_engine import OracleEngine
from tests._util import assert_close_planes, load_case


@pytest.fixture(autouse=True, params=["oracle", "kernel-source"])
def engine_kind(monkeypatch, request):
    """Every test runs twice: on the oracle-backed stand-in, and on the product's own
    `HipSystem` class driving the host build of the kernel source through the real C ABI
    (tests/_hostmath.make_engine_class)."""
    if request.param == "oracle":
        monkeypatch.setattr(tr, "_make_engine", lambda table, device: OracleEngine(table, device))
    else:
        from tests import _hostmath as hm
        if not hm.available():
            pytest.skip("hipcc (used as host C++ compiler) missing")
        cls = hm.make_engine_class()
        monkeypatch.setattr(tr, "_make_engine", lambda table, device: cls(table, device))
    return request.param


def _stack(surfaces):
    return torch.stack([getattr(surfaces, k) for k in
                        ("x", "y", "z", "L", "M", "N", "intensity", "opd")], dim=1).numpy()


def test_trace_hexapolar_matches_reference_trace():
    table, data = load_case("cooke_trace_hexapolar6")
    t = tr.HipRayTracer(table, dtype=torch.float64)
    rays = t.trace([0.0, 0.0, 0.0], [0.0, 0.7, 1.0], 0.55, num_rays=6, distribution="hexapolar")
    assert len(rays) == 3 * 127
    assert_close_planes(_stack(t.surfaces), data["record"], 1e-10, 1e-11, "cooke trace()")
    np.testing.assert_allclose(rays.y.numpy(), data["final"][1], rtol=1e-10, atol=1e-10)
    assert t.surfaces.x.shape == (8, 381)


def test_trace_generic_matches_reference():
    table, data = load_case("double_gauss_multifield")
    t = tr.HipRayTracer(table, dtype=torch.float64)
    rays = t.trace_generic(data["Hx"], data["Hy"], data["Px"], data["Py"], 0.4861)
    assert_close_planes(_stack(t.surfaces), data["record"], 1e-10, 1e-11, "dg trace_generic()")
    np.testing.assert_allclose(rays.opd.numpy(), data["final"][7], rtol=1e-12)
    np.testing.assert_allclose(rays.w.numpy(), 0.4861)
    # L0/M0/N0 = pre-interaction cosines at the image surface
    np.testing.assert_allclose(rays.L0.numpy(), data["pre_dir"][0], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(rays.N0.numpy(), data["pre_dir"][2], rtol=1e-10, atol=1e-12)


def test_scalar_field_broadcast():
    table, data = load_case("double_gauss")
    t = tr.HipRayTracer(table, dtype=torch.float64)
    t.trace_generic(0.0, 0.7, data["Px"], data["Py"], 0.5876)
    assert_close_planes(_stack(t.surfaces), data["record"], 1e-10, 1e-11, "scalar H")


def test_polarized_trace_applies_update_intensity_but_generic_does_not():
    table, data = load_case("zernike_fresnel_fringe")
    t = tr.HipRayTracer(table, dtype=torch.float64)
    rays = t.trace([0.0, 0.0], [0.0, 1.0], 0.55, num_rays=24, distribution="uniform")
    np.testing.assert_allclose(rays.i.numpy(), data["i_updated"], rtol=1e-10)
    # recorded last-row intensity stays the pre-update value
    np.testing.assert_allclose(t.surfaces.intensity[-1].numpy(), data["record"][-1, 6], rtol=1e-12)
    assert rays.p.shape == (816, 3, 3) and rays.p.is_complex()
    np.testing.assert_allclose(rays.p.numpy().real, data["prt"].real, rtol=1e-9, atol=1e-12)
    table, data = load_case("zernike_fresnel_polarized")
    t = tr.HipRayTracer(table, dtype=torch.float64)
    rays = t.trace_generic(data["Hx"], data["Hy"], data["Px"], data["Py"], 0.55)
    np.testing.assert_allclose(rays.i.numpy(), data["i_before_update"], rtol=1e-10)
    rays.update_intensity(table.polarization)
    np.testing.assert_allclose(rays.i.numpy(), data["i_updated"], rtol=1e-10)


def test_validation_errors_match_reference_text():
    table, _ = load_case("double_gauss")
    t = tr.HipRayTracer(table, dtype=torch.float64)
    with pytest.raises(ValueError, match=r"Normalized field coordinates must be within \(-1, 1\)"):
        t.trace(0.0, 1.5, 0.5876, 4)
    with pytest.raises(ValueError, match=r"Normalized pupil coordinates must be within \(-1, 1\)"):
        t.trace_generic(0.0, 0.0, 1.2, 0.0, 0.5876)
    with pytest.raises(ValueError, match="Invalid distribution type."):
        t.trace(0.0, 0.0, 0.5876, 4, "nonsense")
    with pytest.raises(KeyError):
        t.trace(0.0, 0.0, 0.123, 4)
    with pytest.raises(NotImplementedError):
        t.set_aiming("iterative")


def test_fresnel_without_polarization_raises():
    table, _ = load_case("zernike_fresnel_fringe")
    table.polarization = None
    t = tr.HipRayTracer(table, dtype=torch.float64)
    with pytest.raises(ValueError, match="Polarization must be set"):
        t.trace(0.0, 0.0, 0.55, 4)


def test_record_last_mode_returns_final_state_only():
    table, data = load_case("double_gauss")
    t = tr.HipRayTracer(table, dtype=torch.float64)
    t.record_all = False
    rays = t.trace_generic(0.0, 0.7, data["Px"], data["Py"], 0.5876)
    np.testing.assert_allclose(rays.x.numpy(), data["final"][0], rtol=1e-10, atol=1e-10)
    assert t.surfaces.x.numel() == 0


def test_distributions_match_reference_counts_and_order():
    from optiland_amd.distribution import create_distribution
    d = create_distribution("hexapolar").generate_points(64)
    assert d.x.size == 1 + 3 * 64 * 65  # 12 481
    table, data = load_case("cooke_trace_hexapolar6")
    d = create_distribution("hexapolar").generate_points(6)
    np.testing.assert_allclose(d.x, data["Px"][:127], atol=1e-15)
    np.testing.assert_allclose(d.y, data["Py"][:127], atol=1e-15)
    table, data = load_case("zernike_fresnel_fringe")
    d = create_distribution("uniform").generate_points(24)
    np.testing.assert_allclose(d.x, data["Px"][:408], atol=1e-15)
    assert create_distribution("cross").generate_points(5).x.size == 9
    assert create_distribution("cross").generate_points(4).x.size == 8
    assert create_distribution("ring").generate_points(7).x.size == 7
    r = create_distribution("random").generate_points(1000)
    assert np.all(r.x**2 + r.y**2 <= 1)
    # the one-pass hexapolar sampler IS the per-ring definition (distribution.py:201-220:
    # centre, then `r_k (cos, sin)(linspace(0, 2 pi, 6 k + 1)[:-1])` ring by ring), bit for bit
    for rings in list(range(1, 24)) + [64, 257]:
        xs, ys = [np.zeros(1)], [np.zeros(1)]
        radii = np.linspace(0.0, 1.0, rings + 1)
        for i in range(rings):
            theta = np.linspace(0.0, 2.0 * np.pi, 6 * (i + 1) + 1)[:-1]
            xs.append(radii[i + 1] * np.cos(theta))
            ys.append(radii[i + 1] * np.sin(theta))
        d = create_distribution("hexapolar").generate_points(rings)
        assert np.array_equal(d.x, np.concatenate(xs)), rings
        assert np.array_equal(d.y, np.concatenate(ys)), rings


def test_uniform_sampler_is_the_masked_meshgrid():
    """distribution.py:161-186 (`UniformDistribution`): the row-by-row sampler gives the
    masked n x n meshgrid, value for value and in its order."""
    from optiland_amd.distribution import create_distribution
    for n in (1, 2, 3, 4, 5, 8, 24, 25, 64, 101, 300):
        g = np.linspace(-1.0, 1.0, n)
        x, y = np.meshgrid(g, g)
        keep = x**2 + y**2 <= 1
        d = create_distribution("uniform").generate_points(n)
        assert np.array_equal(d.x, x[keep]) and np.array_equal(d.y, y[keep]), n


def test_pupil_planes_are_shared_between_tracers_of_a_process():
    """A spot diagram over three wavelengths runs on three tracers: the deterministic pupil
    planes are sampled and uploaded once (`tracer._shared_pupil_planes`); stochastic samplers
    draw afresh on every call."""
    from optiland_amd import tracer as tr
    made = []
    to_dev = lambda a: (made.append(1), torch.as_tensor(a, dtype=torch.float64))[1]  # noqa: E731
    tr._PUPIL_PLANES.clear()
    a = tr._shared_pupil_planes(("hexapolar", 5), torch.float64, "cpu", to_dev)
    b = tr._shared_pupil_planes(("hexapolar", 5), torch.float64, "cpu", to_dev)
    assert a[0] is b[0] and len(made) == 2
    c = tr._shared_pupil_planes(("hexapolar", 5), torch.float32, "cpu", to_dev)
    assert c[0] is not a[0] and len(made) == 4          # other precision: other planes
    r1 = tr._shared_pupil_planes(("random", 50), torch.float64, "cpu", to_dev)
    r2 = tr._shared_pupil_planes(("random", 50), torch.float64, "cpu", to_dev)
    assert not torch.equal(r1[0], r2[0])
    cap = tr._PUPIL_PLANES_CAP
    tr._PUPIL_PLANES_CAP = 3 * 2 * 91 * 8                 # room for ~3 five-ring samplers
    try:
        for n in range(6, 12):
            tr._shared_pupil_planes(("hexapolar", n), torch.float64, "cpu", to_dev)
        assert 1 <= len(tr._PUPIL_PLANES) <= 3             # least recently used went
        assert (("hexapolar", 11), torch.float64, "cpu") in tr._PUPIL_PLANES
    finally:
        tr._PUPIL_PLANES_CAP = cap
        tr._PUPIL_PLANES.clear()


# reference goldens: tests/test_analysis.py:76-102 (CookeTriplet, fields 0/14/20 deg,
# wavelengths 0.48/0.55/0.65 um, 6 hexapolar rings, chief-ray centred)
COOKE_GEO = [[0.00597244087781, 0.00628645771124, 0.00931911440064],
             [0.03928464835617618, 0.04075295155639047, 0.04772194200606705],
             [0.018909146395329878, 0.022501847359635008, 0.036545592330568866]]
COOKE_RMS = [[0.003791335461448, 0.004293689564257, 0.006195618755672],
             [0.01582480029344623, 0.016918412809703662, 0.019221165873836682],
             [0.013236232767092956, 0.012116688566406967, 0.013648684944411313]]


def test_spot_diagram_reproduces_reference_goldens_host_logic():
    from optiland_amd import load_system
    from optiland_amd.analysis import SpotDiagram
    table = load_system("cooke_generic")
    t = tr.HipRayTracer(table, dtype=torch.float64)
    spot = SpotDiagram(t)
    np.testing.assert_allclose(spot.rms_spot_radius(), COOKE_RMS, rtol=1e-5)
    np.testing.assert_allclose(spot.geometric_spot_radius(), COOKE_GEO, rtol=1e-5)
    c = SpotDiagram(t, reference="centroid")
    assert np.all(np.array(c.rms_spot_radius()) <= np.array(spot.rms_spot_radius()) + 1e-12)


def test_polarizer_and_retarder_system_through_the_tracer():
    """Host logic with a complex PRT (18 planes): trace() applies update_intensity."""
    table, data = load_case("polarizer_retarder")
    assert table.needs_complex_prt
    t = tr.HipRayTracer(table, dtype=torch.float64)
    rays = t.trace([0.0, 0.0], [0.0, 1.0], 0.55, num_rays=20, distribution="uniform")
    np.testing.assert_allclose(rays.p.numpy(), data["prt"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(rays.i.numpy(), data["i_updated"], rtol=1e-10)


def test_vignetting_factors_trace_and_trace_generic():
    """Nearest-field vignetting (fields/field_group.py:93-122) through both entry
    points; trace_generic applies (1 - v) twice like the reference
    (real_ray_tracer.py:134-137 + ray_aiming/paraxial.py:60-62,90-91)."""
    table, data = load_case("vignetted_trace")
    t = tr.HipRayTracer(table, dtype=torch.float64)
    t.trace([0.0, 0.0, 0.25], [0.0, 0.7, 1.0], 0.55, num_rays=5, distribution="hexapolar")
    assert_close_planes(_stack(t.surfaces), data["record"], 1e-10, 1e-11, "vignetted trace()")
    table, data = load_case("vignetted_generic")
    t = tr.HipRayTracer(table, dtype=torch.float64)
    t.trace_generic(data["Hx"], data["Hy"], data["Px"], data["Py"], 0.55)
    assert_close_planes(_stack(t.surfaces), data["record"], 1e-10, 1e-11,
                        "vignetted trace_generic()")


def test_encircled_energy_host_logic():
    """EncircledEnergy on the oracle-backed engine == the reference's definition
    (analysis/encircled_energy.py:147-160) evaluated with numpy on the same hits."""
    from optiland_amd import load_system
    from optiland_amd.analysis import EncircledEnergy
    table = load_system("cooke_generic")
    t = tr.HipRayTracer(table, dtype=torch.float64)
    ee = EncircledEnergy(t, wavelength=0.55, num_rays=8, distribution="hexapolar", num_points=64)
    assert ee.ee.shape == (len(ee.fields), 64) and ee.r_step[0] == 0.0
    for k, (h, c) in enumerate(zip(ee._hits, ee._centers)):
        x, y, e = (v.double().numpy() for v in h)
        r = np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2)
        want = np.array([np.nansum(e[r <= v]) for v in ee.r_step])
        np.testing.assert_allclose(ee.ee[k], want, rtol=1e-12, atol=1e-12)
        assert ee.ee[k][-1] == pytest.approx(np.nansum(e))  # everything inside 1.2 r_max
    assert np.all(np.diff(ee.ee, axis=1) >= 0)


def test_analysis_signatures_follow_the_reference(monkeypatch):
    """Argument names and error behaviour of the stand-alone analyses mirror the
    reference's: SpotDiagram(coordinates=, reference=) (analysis/spot_diagram/core.py:
    69-121), OPD(num_rays=, distribution=, strategy=, remove_tilt=) (wavefront/opd.py:72-93),
    FFTPSF(num_rays=, grid_size=, strategy=, remove_tilt=) (psf/fft.py:87-110)."""
    import inspect
    import torch
    import optiland_amd.tracer as tr
    from optiland_amd import load_system
    from optiland_amd.analysis import SpotDiagram
    from optiland_amd.wavefront import FFTPSF, OPD
    from tests._fake_engine import OracleEngine
    monkeypatch.setattr(tr, "_make_engine", lambda table, device: OracleEngine(table, device))
    names = lambda f: list(inspect.signature(f).parameters)[1:]  # noqa: E731
    assert names(SpotDiagram.__init__)[:7] == ["tracer", "fields", "wavelengths", "num_rings",
                                               "distribution", "coordinates", "reference"]
    assert names(OPD.__init__)[:7] == ["tracer", "field", "wavelength", "num_rays", "distribution",
                                       "strategy", "remove_tilt"]
    assert names(FFTPSF.__init__)[:7] == ["tracer", "field", "wavelength", "num_rays", "grid_size",
                                          "strategy", "remove_tilt"]
    t = tr.HipRayTracer(load_system("cooke_generic"), dtype=torch.float64)
    with pytest.raises(ValueError, match="Coordinates must be 'global' or 'local'"):
        SpotDiagram(t, coordinates="polar")
    with pytest.raises(ValueError, match="Invalid reference"):
        SpotDiagram(t, reference="nowhere")
    with pytest.raises(ValueError, match="Unknown wavefront strategy"):
        OPD(t, (0, 0), 0.55, strategy="nearest_sphere")
    flat = OPD(t, (0, 1), 0.55, num_rays=4, remove_tilt=True)
    raw = OPD(t, (0, 1), 0.55, num_rays=4)
    assert flat.rms() <= raw.rms() + 1e-12
    loc, glo = SpotDiagram(t, num_rings=3), SpotDiagram(t, num_rings=3, coordinates="global")
    np.testing.assert_allclose(loc.rms_spot_radius(), glo.rms_spot_radius(), rtol=1e-12)
    oz = np.asarray(t.table.surfaces[-1]["origin"], dtype=np.float64)
    for (a, b), (c, d) in zip(loc.centroid(), glo.centroid()):
        np.testing.assert_allclose([c - a, d - b], oz[:2], atol=1e-12)


def test_record_pool_window_choice():
    """`RecordPool._pick`: the fastest windows under the limit that do not overlap (offsets of
    one arena closer than a block are the same boundary seen twice); windows of other arenas
    never overlap."""
    from optiland_amd.engine import RecordPool
    need = 4
    cand = [(0.60, 0, 30), (0.61, 0, 31), (0.62, 0, 14), (0.70, 0, 0), (0.59, 1, 5),
            (0.595, 1, 8)]
    assert RecordPool._pick(cand, 0.65, need, 2) == [(0.59, 1, 5), (0.60, 0, 30)]
    assert RecordPool._pick(cand, 0.65, need, 4) == [(0.59, 1, 5), (0.60, 0, 30), (0.62, 0, 14)]
    assert RecordPool._pick(cand, 0.50, need, 2) == []
    assert RecordPool._pick([], 1.0, need, 2) == []


def test_record_pool_configuration(monkeypatch):
    """`integration.enable(placed_records=...)` / OPTILAND_HIP_PLACED_RECORDS only set the
    process-wide pool configuration ("auto" by default since round 5); on a CPU engine
    `alloc_record` never consults it."""
    from optiland_amd import engine as E
    from optiland_amd import integration as ig
    monkeypatch.delenv("OPTILAND_HIP_PLACED_RECORDS", raising=False)
    E.HipSystem.reset_record_pool()
    assert E._POOL_CONFIG["slots"] == "auto"
    try:
        ig._set_record_pool(True)
        assert E._POOL_CONFIG["slots"] == 2
        ig._set_record_pool(3)
        assert E._POOL_CONFIG["slots"] == 3
        ig._set_record_pool(None)                      # None: leave as is ...
        assert E._POOL_CONFIG["slots"] == 3
        monkeypatch.setenv("OPTILAND_HIP_PLACED_RECORDS", "1")
        ig._set_record_pool(None)                      # ... unless the environment says
        assert E._POOL_CONFIG["slots"] == 1
        monkeypatch.setenv("OPTILAND_HIP_PLACED_RECORDS", "auto")
        ig._set_record_pool(None)
        assert E._POOL_CONFIG["slots"] == "auto"
        ig._set_record_pool(False)
        assert E._POOL_CONFIG["slots"] == 0 and not E._RECORD_POOLS
        ig._set_record_pool("auto")
        assert E._POOL_CONFIG["slots"] == "auto"
    finally:
        monkeypatch.delenv("OPTILAND_HIP_PLACED_RECORDS", raising=False)
        E.HipSystem.reset_record_pool()


def test_record_pool_policy_auto_lru_and_cooldown(monkeypatch):
    """Which shapes get a pool (`HipSystem._pool_for`), on a stand-in that builds no arenas:
    "auto" waits for the SECOND request of a shape and for spare device memory; the least
    recently USED shape is evicted; after an eviction the device is served plain for a while
    (a workload cycling through three shapes does not rebuild a pool per trace); a shape whose
    probe found no window is not probed again."""
    import torch

    from optiland_amd import engine as E

    built = []

    class FakePool:
        def __init__(self, hip, n, dtype, rows, slots, arena_bytes=None, max_arenas=3):
            built.append((n, slots, arena_bytes, max_arenas))
            self.windows = []

        def acquire(self):
            return None

    class Dev:
        type, index = "cuda", 0

    class Lib:
        ol_stream_fill = True

    class Sys:
        device, lib = Dev(), Lib()
        _pool_for = E.HipSystem._pool_for
        _pool_for_locked = E.HipSystem._pool_for_locked
        free = 200 << 30

        def _auto_arena_bytes(self, need):
            return None if self.free < (144 << 30) else min(max(3 * need, 40 << 30), self.free // 4)

    monkeypatch.setattr(E, "RecordPool", FakePool)
    monkeypatch.delenv("OPTILAND_HIP_PLACED_RECORDS", raising=False)
    E.HipSystem.reset_record_pool()
    hip, f32, big = Sys(), torch.float32, 1 << 30
    try:
        assert hip._pool_for(1000, f32, 13, 1 << 20) is None            # small: never
        assert hip._pool_for(10, f32, 13, big) is None and not built    # first request: plain
        assert hip._pool_for(10, f32, 13, big) is not None              # second: a pool
        assert built == [(10, 2, 40 << 30, 3)]
        assert hip._pool_for(10, f32, 13, big) is not None and len(built) == 1   # (kept, empty)
        hip.free = 100 << 30                                            # memory is short
        assert hip._pool_for(11, f32, 13, big) is None
        assert hip._pool_for(11, f32, 13, big) is None and len(built) == 1
        hip.free = 200 << 30
        assert hip._pool_for(11, f32, 13, big) is not None and len(built) == 2
        hip._pool_for(10, f32, 13, big)                                 # 10 is used again ...
        hip._pool_for(12, f32, 13, big)
        assert hip._pool_for(12, f32, 13, big) is not None              # ... so 11 goes
        assert [k[1] for k in E._RECORD_POOLS] == [10, 12]
        # cool-down: shape 11 coming back right away is served plain, not re-probed
        n_built = len(built)
        for _ in range(5):
            assert hip._pool_for(11, f32, 13, big) is None
        assert len(built) == n_built and [k[1] for k in E._RECORD_POOLS] == [10, 12]
        # an explicit count: from the first request on, three arenas at most
        E.HipSystem.enable_record_pool(3)
        assert hip._pool_for(20, f32, 13, big) is not None
        assert built[-1] == (20, 3, None, 3)
        E.HipSystem.enable_record_pool(0)
        assert hip._pool_for(20, f32, 13, big) is None and not E._RECORD_POOLS
    finally:
        E.HipSystem.reset_record_pool()


def test_few_waves_hint_follows_the_placed_windows():
    """Round 5: `TRACE_FEW_WAVES` goes with record blocks of >= 256 MB that lie in no placed
    window; windows are remembered per device, newest last, at most `_PLACED_MAX` of them."""
    from optiland_amd import engine as E
    from optiland_amd import system as S

    class Block:
        def __init__(self, ptr, nbytes, index=0):
            self._ptr, self._n = ptr, nbytes
            self.device = torch.device("cuda", index)

        def numel(self):
            return self._n

        def element_size(self):
            return 1

        def data_ptr(self):
            return self._ptr

    saved = E._PLACED_WINDOWS.copy()
    E._PLACED_WINDOWS.clear()
    try:
        big, small = 1 << 30, 1 << 20
        assert E._few_waves_flag(None) == 0
        assert E._few_waves_flag(Block(0x1000, small)) == 0
        assert E._few_waves_flag(Block(0x1000, big)) == S.TRACE_FEW_WAVES
        E._note_placed(torch.device("cuda", 0), 0x7000_0000_0000, big)
        assert E._few_waves_flag(Block(0x7000_0000_0000, big)) == 0
        assert E._few_waves_flag(Block(0x7000_0000_0000 + big - 1, big)) == 0
        assert E._few_waves_flag(Block(0x7000_0000_0000 + big, big)) == S.TRACE_FEW_WAVES
        assert E._few_waves_flag(Block(0x7000_0000_0000, big, index=1)) == S.TRACE_FEW_WAVES
        for k in range(E._PLACED_MAX + 5):
            E._note_placed(torch.device("cuda", 0), 0x1_0000_0000 * (k + 1), big)
        assert len(E._PLACED_WINDOWS) == E._PLACED_MAX
        assert E._few_waves_flag(Block(0x7000_0000_0000, big)) == S.TRACE_FEW_WAVES  # forgotten
    finally:
        E._PLACED_WINDOWS.clear()
        E._PLACED_WINDOWS.update(saved)


def test_polarised_trace_rays_refuses_unnormalised_directions():
    """Round 5: the kernels' PRT update equals the reference's only for unit direction cosines
    (DESIGN 0a); the stand-alone `trace_rays` says so instead of returning a matrix 1e-3 off."""
    from optiland_amd import load_system
    from optiland_amd.tracer import HipRayTracer
    from tests._fake_engine import OracleEngine
    table = load_system("zernike_fresnel_fringe")
    assert table.polarization is not None
    t = HipRayTracer(table, "cpu", dtype=torch.float64, engine=OracleEngine(table, "cpu"))
    n = 5
    z = torch.full((n,), -10.0, dtype=torch.float64)
    zero, one = torch.zeros(n, dtype=torch.float64), torch.ones(n, dtype=torch.float64)
    t.trace_rays([zero, zero, z, zero, zero, one, one], 0.55)                    # unit: fine
    with pytest.raises(ValueError, match="unit direction cosines"):
        t.trace_rays([zero, zero, z, zero, zero, one * 1.0005, one], 0.55)


def test_hot_loop_hint_counts_launches_in_a_row(monkeypatch):
    """Round 6: a big record block that is the target of launch after launch, enqueued without
    a pause, is traced with `TRACE_FEW_WAVES` from the `after`-th launch on (two workgroups per
    CU: faster once the clocks have settled, slower right after an idle gap --
    profiles/r05_ab_wgcap.txt); an idle gap starts the count again; small blocks and
    `after = 0` never."""
    from optiland_amd import engine as E
    from optiland_amd import system as S

    class Block:
        def __init__(self, ptr, nbytes):
            self._ptr, self._n = ptr, nbytes
            self.device = torch.device("cuda", 0)

        def numel(self):
            return self._n

        def element_size(self):
            return 1

        def data_ptr(self):
            return self._ptr

    now = [100.0]
    monkeypatch.setattr(E.time, "perf_counter", lambda: now[0])
    monkeypatch.setitem(E._HOT_LOOP, "after", 4)
    E._HOT_BLOCKS.clear()
    try:
        big, other, small = Block(1 << 40, 1 << 30), Block(2 << 40, 1 << 30), Block(3 << 40, 1 << 20)
        got = []
        for _ in range(7):
            now[0] += 0.001
            got.append(E._hot_loop_flag(big))
        assert got == [0, 0, 0, 0] + [S.TRACE_FEW_WAVES] * 3      # launches 0..3 cold, then hot
        assert E._hot_loop_flag(other) == 0 and E._hot_loop_flag(small) == 0
        now[0] += 1.0                                              # the device may have gone idle
        assert E._hot_loop_flag(big) == 0
        for _ in range(4):
            now[0] += 0.001
            last = E._hot_loop_flag(big)
        assert last == S.TRACE_FEW_WAVES
        monkeypatch.setitem(E._HOT_LOOP, "after", 0)
        assert E._hot_loop_flag(big) == 0
        assert E._hot_loop_flag(None) == 0
    finally:
        E._HOT_BLOCKS.clear()


def test_idle_record_pools_give_their_arenas_back():
    """Round 6: a pool with no block lent out that nobody has asked for `idle_s` seconds is
    dropped (`release_record_pools(only_idle=True)`, what the sweeper thread calls); the shape's
    next request counts as "the second" again, so a loop that resumes gets its pool back at
    once.  A pool with a block in a user's hands stays; an explicit release takes all."""
    from optiland_amd import engine as E

    class Pool:
        def __init__(self, lent, age):
            self.windows = [(None, 1), (None, 2)]
            self.free = [0] if lent else [0, 1]
            self.last_used = E.time.monotonic() - age
            self.info = {"arena_bytes_kept": 7}

        def idle(self):
            return len(self.free) == len(self.windows)

    E.HipSystem.reset_record_pool()
    keep = E._POOL_CONFIG["idle_s"]
    try:
        E._POOL_CONFIG["idle_s"] = 10.0
        E._RECORD_POOLS[(0, 1, torch.float32, 13)] = Pool(lent=False, age=100.0)   # idle: goes
        E._RECORD_POOLS[(0, 2, torch.float32, 13)] = Pool(lent=True, age=100.0)    # lent: stays
        E._RECORD_POOLS[(0, 3, torch.float32, 13)] = Pool(lent=False, age=1.0)     # recent: stays
        stats = E.record_pool_stats()
        assert [p["lent"] for p in stats["pools"]] == [0, 1, 0] and stats["idle_s"] == 10.0
        assert E.release_record_pools(only_idle=True) == 1
        assert [k[1] for k in E._RECORD_POOLS] == [2, 3]
        assert E._SHAPE_SEEN[(0, 1, torch.float32, 13)] == 1
        assert E.release_record_pools() == 2 and not E._RECORD_POOLS
    finally:
        E._POOL_CONFIG["idle_s"] = keep
        E.HipSystem.reset_record_pool()


def test_the_package_never_empties_the_users_allocator_cache():
    """Round 6: arenas are the library's own hipMalloc blocks
    (`ol_arena_alloc`); nothing under optiland_amd/ calls `torch.cuda.empty_cache()`."""
    import re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "optiland_amd")
    for base, _dirs, files in os.walk(root):
        for name in files:
            if name.endswith(".py"):
                text = open(os.path.join(base, name)).read()
                code = "\n".join(ln.split("#")[0] for ln in text.splitlines())
                code = re.sub(r'"""[\s\S]*?"""', "", code)
                assert "empty_cache(" not in code, os.path.join(base, name)
