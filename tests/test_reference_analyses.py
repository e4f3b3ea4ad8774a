"""The fused kernels behind the reference's OWN analysis classes (build container only).

`integration.enable()` / `install()` patch `SpotDiagram` / `EncircledEnergy`
(`_generate_field_data`), the chief-ray wavefront strategy (`compute_wavefront_data`) and
`ScalarFFTPSF` (`_generate_pupils`, `_pad_pupils`) so that they call `ol_trace_spot`,
`ol_trace_opd` and `ol_pupil_fill` (optiland_amd/analysis_seams.py).  Here the reference's
classes are run three ways on the same lens -- NumPy backend (the yardstick), torch backend
with the drop-in but WITHOUT the analysis seams (record-all trace + the reference's own
reductions), and with the seams -- on the CPU stand-ins for the device engine (the
oracle-backed engine and the product engine class on the host build of the kernel source).

Skipped where /root/reference does not exist (the GPU box runs the same checks against the
real library in tests/test_gpu_live_reference.py).
"""

import pickle

import numpy as np
import pytest

from tests.test_reference_integration import REF, hip_on_cpu, ref  # noqa: F401 (fixtures)

import os

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "optiland")),
                                reason="reference package not present")


def _np(be, a):
    return np.asarray(be.to_numpy(a), dtype=np.float64)


@pytest.fixture
def seams(hip_on_cpu):
    from optiland_amd import analysis_seams, integration
    be = hip_on_cpu
    integration.enable(force=True, analyses=True)
    for k in analysis_seams.STATS:
        analysis_seams.STATS[k] = 0
    yield be, analysis_seams.STATS
    integration.disable()


def _numpy_reference(be, build, run):
    """`run(lens)` under the NumPy backend, results as numpy."""
    from optiland_amd import analysis_seams
    was = be.get_backend()
    be.set_backend("numpy")
    keep = dict(analysis_seams.STATS)  # NumPy-backend calls decline the seams: not counted
    try:
        return run(build())
    finally:
        analysis_seams.STATS.update(keep)
        be.set_backend(was)
        if was == "torch":
            be.set_device("cpu")
            be.set_precision("float64")


def _cooke():
    from optiland.samples.objectives import CookeTriplet
    return CookeTriplet()


def _singlet_asphere():
    from optiland.samples.simple import AsphericSinglet
    return AsphericSinglet()


@pytest.mark.parametrize("build", [_cooke, _singlet_asphere])
@pytest.mark.parametrize("reference", ["chief_ray", "centroid"])
def test_spot_diagram_through_the_fused_seam(seams, build, reference, request):
    be, stats = seams
    from optiland import analysis

    def run(lens, coords="local"):
        s = analysis.SpotDiagram(lens, num_rings=5, reference=reference, coordinates=coords)
        return ([[(_np(be, d.x), _np(be, d.y), _np(be, d.intensity)) for d in f] for f in s.data],
                [[float(_np(be, v)) for v in f] for f in s.rms_spot_radius()],
                [[float(_np(be, v)) for v in f] for f in s.geometric_spot_radius()],
                [(float(_np(be, a)), float(_np(be, b))) for a, b in s.centroid()])

    want = _numpy_reference(be, build, run)
    got = run(build())
    assert stats["spot"] > 0 and stats["spot_fallback"] == 0
    if "kernel-source" in request.node.name:
        # round 5: the whole fields x wavelengths grid is ONE `ol_trace_spot_batch` launch (the
        # oracle-backed stand-in has no such entry point and keeps the per-cell seam)
        assert stats["spot_grid"] >= 1
    for fg, fw in zip(got[0], want[0]):
        for (x, y, i), (xw, yw, iw) in zip(fg, fw):
            assert x.shape == xw.shape
            np.testing.assert_allclose(x, xw, rtol=0, atol=1e-8)
            np.testing.assert_allclose(y, yw, rtol=0, atol=1e-8)
            np.testing.assert_allclose(i, iw, rtol=1e-12, atol=0)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-7)
    np.testing.assert_allclose(got[2], want[2], rtol=1e-7)
    np.testing.assert_allclose(got[3], want[3], rtol=0, atol=1e-8)
    # global coordinates too
    want_g = _numpy_reference(be, build, lambda lens: run(lens, "global"))
    got_g = run(build(), "global")
    np.testing.assert_allclose(got_g[1], want_g[1], rtol=1e-7)
    np.testing.assert_allclose(got_g[3], want_g[3], rtol=0, atol=1e-8)


def test_spot_grid_equals_the_per_cell_seam(seams, request, monkeypatch):
    """`_spot_generate_data` as ONE batched launch against the same seam cell by cell (the batch
    switched off): the same `SpotData`, bit for bit, for a spot diagram in local and global
    coordinates and for an encircled energy (whose cells are unmasked and global); "random"
    pupils -- a fresh draw per cell in the reference's loop -- keep the per-cell path; and what
    `Optic.trace` would have left on the surfaces is the LAST cell's trace."""
    if "kernel-source" not in request.node.name:
        pytest.skip("the oracle-backed stand-in has no ol_trace_spot_batch")
    be, stats = seams
    from optiland import analysis
    from optiland_amd import analysis_seams

    def run(lens, cls, **kw):
        a = cls(lens, **kw)
        out = [[(_np(be, d.x), _np(be, d.y), _np(be, d.intensity)) for d in f] for f in a.data]
        return out, _np(be, lens.surfaces.y)[-1]

    for cls, kw in ((analysis.SpotDiagram, dict(num_rings=4, coordinates="local")),
                    (analysis.SpotDiagram, dict(num_rings=4, coordinates="global")),
                    (analysis.EncircledEnergy, dict(num_rays=5, distribution="hexapolar",
                                                    num_points=16))):
        n0 = stats["spot_grid"]
        got, last_y = run(_cooke(), cls, **kw)
        assert stats["spot_grid"] == n0 + 1
        with monkeypatch.context() as m:
            m.setattr(analysis_seams, "_spot_grid", lambda self: None)
            want, last_y_w = run(_cooke(), cls, **kw)
        assert stats["spot_grid"] == n0 + 1
        for fg, fw in zip(got, want):
            for a, b in zip(fg, fw):
                for u, v in zip(a, b):
                    np.testing.assert_array_equal(u, v)
        np.testing.assert_array_equal(last_y, last_y_w)
    n0 = stats["spot_grid"]
    analysis.SpotDiagram(_cooke(), num_rings=50, distribution="random")
    assert stats["spot_grid"] == n0


@pytest.mark.parametrize("state", ["unpolarized", "elliptical"])
def test_spot_seams_serve_polarised_optics(seams, request, state):
    """A POLARISED optic (Fresnel coatings + a polarization state; BASELINE C5's system): what
    `SpotDiagram` / `EncircledEnergy` read is the recorded last row -- positions and the
    geometric intensity; the PRT matrix and `update_intensity` never reach it (SURVEY.md
    Appendix D) -- so the fused spot kernel serves it (`OL_SPOT_POLARIZED_OK`, ABI 10; round 4
    declined).  Against the NumPy backend, which traces `PolarizedRays`."""
    if "kernel-source" not in request.node.name:
        pytest.skip("the oracle-backed stand-in refuses polarised spots")
    be, stats = seams
    from optiland import analysis
    from tests import _live

    def build():
        return _live.zernike_fresnel(polarization=state)

    def run(lens):
        s = analysis.SpotDiagram(lens, num_rings=5)
        e = analysis.EncircledEnergy(lens, num_rays=6, distribution="hexapolar", num_points=16)
        return ([[(_np(be, d.x), _np(be, d.y), _np(be, d.intensity)) for d in f] for f in s.data],
                [[(_np(be, d.x), _np(be, d.y), _np(be, d.intensity)) for d in f] for f in e.data],
                _np(be, lens.surfaces.y)[-1])

    want = _numpy_reference(be, build, run)
    got = run(build())
    assert stats["spot"] > 0 and stats["spot_fallback"] == 0 and stats["spot_grid"] >= 1
    assert stats["ee"] > 0 and stats["ee_fallback"] == 0
    for G, W in ((got[0], want[0]), (got[1], want[1])):
        for fg, fw in zip(G, W):
            for (x, y, i), (xw, yw, iw) in zip(fg, fw):
                assert x.shape == xw.shape
                np.testing.assert_allclose(x, xw, rtol=0, atol=1e-7)   # Newton stop tolerance
                np.testing.assert_allclose(y, yw, rtol=0, atol=1e-7)
                np.testing.assert_allclose(i, iw, rtol=1e-12, atol=0)
    # what Optic.trace would have left on the surfaces (a polarised record-all re-run)
    np.testing.assert_allclose(got[2], want[2], rtol=0, atol=1e-7)


def test_spot_seam_on_a_tilted_image_surface_in_local_coordinates(seams, request):
    """`SpotDiagram(coordinates="local")` localises the image-plane hits into the image
    surface's own frame (visualization/system/utils.py:17-47); with a TILTED image surface the
    kernel leaves the hits in that frame itself (`OL_SPOT_HITS_LOCAL`, ABI 10; round 4 declined),
    batched and cell by cell."""
    if "kernel-source" not in request.node.name:
        pytest.skip("the oracle-backed stand-in has no local-frame hits")
    be, stats = seams
    from optiland import analysis
    from optiland_amd import analysis_seams

    def build():
        lens = _cooke()
        lens.surfaces[-1].geometry.cs.rx = be.array(0.05) if be.get_backend() == "torch" else 0.05
        lens.surfaces[-1].geometry.cs.ry = be.array(-0.03) if be.get_backend() == "torch" else -0.03
        return lens

    def run(lens, coords):
        s = analysis.SpotDiagram(lens, num_rings=4, coordinates=coords)
        return [[(_np(be, d.x), _np(be, d.y)) for d in f] for f in s.data]

    for coords in ("local", "global"):
        want = _numpy_reference(be, build, lambda lens: run(lens, coords))
        n0, g0 = stats["spot"], stats["spot_grid"]
        got = run(build(), coords)
        assert stats["spot"] > n0 and stats["spot_fallback"] == 0 and stats["spot_grid"] == g0 + 1
        with __import__("contextlib").ExitStack() as st:
            mp = pytest.MonkeyPatch()
            st.callback(mp.undo)
            mp.setattr(analysis_seams, "_spot_grid", lambda self: None)
            cell = run(build(), coords)                     # the per-cell seam
        assert stats["spot_fallback"] == 0
        for G in (got, cell):
            for fg, fw in zip(G, want):
                for (x, y), (xw, yw) in zip(fg, fw):
                    np.testing.assert_allclose(x, xw, rtol=0, atol=1e-9)
                    np.testing.assert_allclose(y, yw, rtol=0, atol=1e-9)


def test_spot_seam_masks_clipped_rays_like_the_reference(seams):
    """core.py:470-476: rays with zero intensity are dropped -- a system that vignettes."""
    be, stats = seams
    from optiland import analysis
    from tests import _live

    def run(lens):
        s = analysis.SpotDiagram(lens, num_rings=8)
        return [[_np(be, d.x) for d in f] for f in s.data]

    want = _numpy_reference(be, _live.rc_asphere, run)
    got = run(_live.rc_asphere())
    assert stats["spot"] > 0
    n_all = 1 + 3 * 8 * 9
    assert any(x.size < n_all for f in want for x in f), "no ray was clipped: weak test"
    for fg, fw in zip(got, want):
        for x, xw in zip(fg, fw):
            assert x.shape == xw.shape
            np.testing.assert_allclose(x, xw, rtol=0, atol=1e-6)


def test_encircled_energy_through_the_fused_seam(seams):
    be, stats = seams
    from optiland import analysis

    def run(lens):
        e = analysis.EncircledEnergy(lens, num_rays=7, distribution="hexapolar", num_points=32)
        return ([[(_np(be, d.x), _np(be, d.y), _np(be, d.intensity)) for d in f] for f in e.data],
                [(float(_np(be, a)), float(_np(be, b))) for a, b in e.centroid()])

    want = _numpy_reference(be, _cooke, run)
    got = run(_cooke())
    assert stats["ee"] > 0 and stats["ee_fallback"] == 0
    for fg, fw in zip(got[0], want[0]):
        for a, b in zip(fg, fw):
            for u, v in zip(a, b):
                np.testing.assert_allclose(u, v, rtol=0, atol=1e-8)
    np.testing.assert_allclose(got[1], want[1], rtol=0, atol=1e-8)


@pytest.mark.parametrize("build", [_cooke, _singlet_asphere])
@pytest.mark.parametrize("remove_tilt", [False, True])
def test_wavefront_opd_through_the_fused_seam(seams, build, remove_tilt):
    be, stats = seams
    from optiland.wavefront import OPD, Wavefront

    def run(lens):
        w = Wavefront(lens, fields=[(0.0, 0.7)], wavelengths="primary", num_rays=9,
                      distribution="hexapolar", remove_tilt=remove_tilt)
        d = w.get_data((0.0, 0.7), lens.primary_wavelength)
        o = OPD(lens, (0.0, 1.0), lens.primary_wavelength, num_rings=7, remove_tilt=remove_tilt)
        return ([_np(be, getattr(d, k)) for k in ("opd", "intensity", "pupil_x", "pupil_y",
                                                   "pupil_z")], float(_np(be, d.radius)),
                float(_np(be, o.rms())))

    want = _numpy_reference(be, build, run)
    got = run(build())
    assert stats["opd"] >= 2 and stats["opd_fallback"] == 0
    newton = build is _singlet_asphere  # the reference stops Newton at 1e-6 mm = 2e-3 waves
    np.testing.assert_allclose(got[0][0], want[0][0], rtol=0, atol=5e-3 if newton else 1e-6)
    for a, b in zip(got[0][1:], want[0][1:]):
        np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-7)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-10)
    np.testing.assert_allclose(got[2], want[2], rtol=0, atol=5e-3 if newton else 1e-6)


@pytest.mark.parametrize("build", [_cooke, _singlet_asphere])
def test_chief_ray_strategy_constructor_takes_the_exit_pupil_from_the_packed_table(seams, build):
    """wavefront/strategy.py:156-160: `pupil_z` (= XPL + last vertex z) comes from the packed
    table instead of three backend walks over the surfaces -- same value, shape and dtype as
    the reference's constructor; optics the drop-in does not serve keep that constructor."""
    be, stats = seams
    from optiland.distribution import create_distribution
    from optiland.wavefront.strategy import ChiefRayStrategy
    from optiland_amd import analysis_seams

    dist = create_distribution("hexapolar")
    dist.generate_points(3)
    lens = build()
    s = ChiefRayStrategy(lens, dist, reference_type="plane")
    assert stats["opd_init"] == 1 and stats["opd_init_fallback"] == 0
    ref_s = analysis_seams._ORIG["chief_init"]
    t = ChiefRayStrategy.__new__(ChiefRayStrategy)
    ref_s(t, build(), dist, reference_type="plane")
    assert type(s.pupil_z) is type(t.pupil_z) and s.pupil_z.shape == t.pupil_z.shape
    assert s.pupil_z.dtype == t.pupil_z.dtype
    np.testing.assert_allclose(_np(be, s.pupil_z), _np(be, t.pupil_z), rtol=1e-12)
    assert float(_np(be, s.n_image)) == float(_np(be, t.n_image))
    assert s.reference_type == t.reference_type == "plane" and s._chief_ray is None
    assert s.optic is lens and s.distribution is dist
    # NumPy backend: the drop-in declines, the reference's constructor runs
    was = dict(stats)
    be.set_backend("numpy")
    try:
        ChiefRayStrategy(build(), dist)
    finally:
        be.set_backend("torch")
        be.set_device("cpu")
        be.set_precision("float64")
    assert stats["opd_init"] == was["opd_init"]
    assert stats["opd_init_fallback"] == was["opd_init_fallback"] + 1


def test_fft_psf_through_the_fused_seams(seams):
    be, stats = seams
    from optiland.psf import FFTPSF

    def run(lens):
        p = FFTPSF(lens, (0.0, 0.7), lens.primary_wavelength, num_rays=64, grid_size=128)
        return _np(be, p.psf), float(_np(be, p.strehl_ratio())), \
            np.asarray(be.to_numpy(p.pupils[0]))

    want = _numpy_reference(be, _cooke, run)
    got = run(_cooke())
    assert stats["opd"] >= 1 and stats["pupil"] >= 1 and stats["pupil_fallback"] == 0
    np.testing.assert_allclose(got[2], want[2], rtol=0, atol=1e-5)
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-4 * want[0].max())
    np.testing.assert_allclose(got[1], want[1], rtol=1e-5)
    # default grid (num_rays -> OpticStudio-like sampling, odd pupil size)
    want2 = _numpy_reference(be, _cooke,
                             lambda lens: _np(be, FFTPSF(lens, (0.0, 0.0), 0.55, num_rays=40).psf))
    got2 = _np(be, FFTPSF(_cooke(), (0.0, 0.0), 0.55, num_rays=40).psf)
    np.testing.assert_allclose(got2, want2, rtol=0, atol=1e-4 * want2.max())


def test_seams_fall_back_where_the_fused_path_does_not_apply(seams, request):
    """Where a seam does not apply the reference's own method runs (and still traces through
    the drop-in's Optic.trace); results equal the NumPy backend's: a centroid strategy asked
    for by name, and -- on an engine without the ABI-10 spot flags (the oracle-backed
    stand-in; the product serves both since round 5, see the two tests above) -- a polarised
    system and a tilted image surface in local coordinates."""
    be, stats = seams
    old_engine = "oracle" in request.node.name
    from optiland import analysis
    from optiland.wavefront import Wavefront
    from tests import _live

    def spot_rms(lens):
        return [[float(_np(be, v)) for v in f]
                for f in analysis.SpotDiagram(lens, num_rings=4).rms_spot_radius()]

    # polarised: ol_trace_spot refuses -> original method
    build = lambda: _live.zernike_fresnel("unpolarized")  # noqa: E731
    want = _numpy_reference(be, build, spot_rms)
    got = spot_rms(build())
    if old_engine:
        assert stats["spot"] == 0 and stats["spot_fallback"] > 0
    else:
        assert stats["spot"] > 0 and stats["spot_fallback"] == 0
    np.testing.assert_allclose(got, want, rtol=1e-6)

    # centroid wavefront strategy is not patched at all; chief-ray on fp32 falls back
    def opd_centroid(lens):
        w = Wavefront(lens, fields=[(0.0, 0.0)], wavelengths="primary", num_rays=6,
                      strategy="centroid_sphere")
        return _np(be, w.get_data((0.0, 0.0), lens.primary_wavelength).opd)

    want = _numpy_reference(be, _cooke, opd_centroid)
    before = stats["opd"]
    got = opd_centroid(_cooke())
    assert stats["opd"] == before
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-6)

    # tilted image surface, local coordinates -> original method
    def tilted():
        lens = _cooke()
        lens.surfaces[-1].geometry.cs.rx = 0.05
        return lens

    f0 = stats["spot_fallback"]
    want = _numpy_reference(be, tilted, spot_rms)
    got = spot_rms(tilted())
    assert (stats["spot_fallback"] > f0) == old_engine
    np.testing.assert_allclose(got, want, rtol=1e-6)


def test_seams_are_inert_without_the_drop_in(hip_on_cpu):
    """The patches are class-wide; an optic the drop-in does not serve runs the original
    methods (and `disable()` restores the classes)."""
    be = hip_on_cpu
    from optiland import analysis
    from optiland.analysis.spot_diagram.core import SpotDiagram
    from optiland_amd import analysis_seams, integration
    orig = SpotDiagram._generate_field_data
    lens = _cooke()
    integration.install(lens, force=True)       # patches the classes, serves THIS optic only
    assert SpotDiagram._generate_field_data is not orig
    for k in analysis_seams.STATS:
        analysis_seams.STATS[k] = 0
    other = _cooke()                              # not installed, enable() not active
    a = analysis.SpotDiagram(other, num_rings=3).rms_spot_radius()
    assert analysis_seams.STATS["spot"] == 0 and analysis_seams.STATS["spot_fallback"] > 0
    b = analysis.SpotDiagram(lens, num_rings=3).rms_spot_radius()
    assert analysis_seams.STATS["spot"] > 0
    np.testing.assert_allclose([[float(_np(be, v)) for v in f] for f in a],
                               [[float(_np(be, v)) for v in f] for f in b], rtol=1e-7)
    integration.uninstall(lens)
    analysis_seams.disable()
    assert SpotDiagram._generate_field_data is orig


def test_spot_diagram_validates_each_wavelength_once(seams, monkeypatch):
    """`SpotDiagram._generate_data` (core.py:420-438) traces fields x wavelengths and edits
    nothing: inside `integration.unchanged(optic)` the change detector validates every
    wavelength's table once, not once per (field, wavelength) -- with ONE walk over the optic
    (round 5: the tokens of one optic differ between wavelengths in their first element only) --
    and again on the next analysis."""
    be, stats = seams
    from optiland import analysis
    from optiland_amd import fingerprint as fp
    from optiland_amd import integration as ig
    lens = _cooke()                                  # 3 fields x 3 wavelengths
    walks = {"n": 0}
    orig = fp.optic_token

    def counting(*a, **k):
        walks["n"] += 1
        return orig(*a, **k)

    monkeypatch.setattr(fp, "optic_token", counting)
    analysis.SpotDiagram(lens, num_rings=3)
    first = walks["n"]
    n_w = len(lens.wavelengths.wavelengths)
    assert stats["spot"] == 3 * n_w
    walks["n"] = 0
    analysis.SpotDiagram(lens, num_rings=3)
    assert walks["n"] == 1, (first, walks["n"], n_w)  # validated again: one walk, three tables
    comp = ig.hip_tracer_of(lens)
    assert comp._hip_trusted is None and comp._hip_trust_depth == 0
    # an edit between two analyses is seen
    before = [[float(_np(be, v)) for v in f]
              for f in analysis.SpotDiagram(lens, num_rings=3).rms_spot_radius()]
    lens.updater.set_radius(float(lens.surfaces[1].geometry.radius) * 1.05, 1)
    after = [[float(_np(be, v)) for v in f]
             for f in analysis.SpotDiagram(lens, num_rings=3).rms_spot_radius()]
    assert abs(after[0][0] - before[0][0]) > 1e-6 * abs(before[0][0])


def test_seams_are_left_off_when_the_reference_signature_changed(ref, monkeypatch):
    """Version guard: the seams replace private methods of the reference;
    one whose target no longer has the signature it was written against is NOT installed (and
    the seams that depend on it go with it), the others are, and `disable()` restores
    exactly what was patched."""
    from optiland.analysis.spot_diagram.core import SpotDiagram
    from optiland.psf.fft import ScalarFFTPSF
    from optiland.wavefront.strategy import ChiefRayStrategy
    from optiland_amd import analysis_seams as seams
    seams.disable()
    stock = {"spot": SpotDiagram._generate_field_data, "pupils": ScalarFFTPSF._generate_pupils,
             "pad": ScalarFFTPSF._pad_pupils, "opd": ChiefRayStrategy.compute_wavefront_data}

    def renamed(self, field, wavelength, num_rays, distribution, coords):  # one name differs
        return stock["spot"](self, field, wavelength, num_rays, distribution, coords)

    def extra(self, field, wavelength, mode="x"):                          # one parameter more
        return stock["opd"](self, field, wavelength)

    monkeypatch.setattr(SpotDiagram, "_generate_field_data", renamed)
    monkeypatch.setattr(ChiefRayStrategy, "compute_wavefront_data", extra)
    try:
        seams.enable()
        assert set(seams.SKIPPED) == {"spot", "opd", "pupils", "pad"}, seams.SKIPPED
        assert "coords" in seams.SKIPPED["spot"] and "needs opd" in seams.SKIPPED["pupils"]
        assert SpotDiagram._generate_field_data is renamed          # untouched
        assert ChiefRayStrategy.compute_wavefront_data is extra
        assert ScalarFFTPSF._generate_pupils is stock["pupils"]
        assert SpotDiagram._generate_data is seams._spot_generate_data   # the others are on
        assert ChiefRayStrategy.__init__ is seams._chief_init
    finally:
        seams.disable()
    assert ScalarFFTPSF._pad_pupils is stock["pad"]
    assert SpotDiagram._generate_data is not seams._spot_generate_data
    monkeypatch.undo()
    seams.enable()
    try:
        assert seams.SKIPPED == {}
        assert SpotDiagram._generate_field_data is seams._spot_generate_field_data
    finally:
        seams.disable()


def test_distribution_seams_sample_the_reference_grids(ref, monkeypatch):
    """`HexagonalDistribution.generate_points` through
    `ol_pupil_points` (here: the host build of the same source): the reference's own points in
    the reference's own order, within a few ulps of the radius (another libm) -- and an `OPD` built on top of them equals the stock one."""
    import torch
    from tests import _hostmath as hm
    if not hm.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    be = ref
    from optiland.distribution import create_distribution
    from optiland_amd import analysis_seams as seams, load_system
    eng = hm.make_engine_class()(load_system("double_gauss"))
    try:
        # the truth: the reference's NumPy backend (its torch backend's own `linspace` walks
        # the second half of an interval backwards from the end point and lands a few ulps
        # away from it)
        be.set_backend("numpy")
        stock = {}
        for name, num in (("hexapolar", 7), ("hexapolar", 0), ("hexapolar", 1), ("hexapolar", 40)):
            d = create_distribution(name)
            d.generate_points(num)
            stock[(name, num)] = (_np(be, d.x), _np(be, d.y))
        be.set_backend("torch")
        be.set_device("cpu")
        be.set_precision("float64")
        monkeypatch.setattr(seams, "POINTS_HOOK",
                            lambda kind, num, dtype: eng.pupil_points(kind, num, dtype))
        seams.enable()
        n0 = seams.STATS["dist"]
        for (name, num), (wx, wy) in stock.items():
            d = create_distribution(name)
            d.generate_points(num)
            gx, gy = _np(be, d.x), _np(be, d.y)
            assert gx.shape == wx.shape
            np.testing.assert_allclose(gx, wx, rtol=0, atol=7e-16)
            np.testing.assert_allclose(gy, wy, rtol=0, atol=7e-16)
        assert seams.STATS["dist"] == n0 + 4
        # the uniform grid is NOT seamed: its consumers rebuild it with the backend's own
        # linspace and need the two masks to agree (psf/fft.py:140-155)
        d = create_distribution("uniform")
        d.generate_points(33)
        assert seams.STATS["dist"] == n0 + 4
        # autograd on: the reference's own sampler
        with be.grad_mode.temporary_enable():
            d = create_distribution("hexapolar")
            d.generate_points(3)
        assert seams.STATS["dist"] == n0 + 4 and seams.STATS["dist_fallback"] >= 1
        # NumPy backend: untouched
        be.set_backend("numpy")
        d = create_distribution("hexapolar")
        d.generate_points(3)
        assert isinstance(d.x, np.ndarray)
    finally:
        seams.disable()
        be.set_backend("numpy")
        eng.close()


@pytest.mark.parametrize("reference_type", ["sphere", "plane"])
def test_device_resident_reference_equals_the_host_built_one(seams, reference_type, monkeypatch,
                                                             request):
    """`ol_wavefront_reference` + `ol_trace_opd_dev` (round 4: the chief ray traced on the
    device, the reference sphere / plane left there, no read-back between the two launches)
    against round 3's form of the same seam (one-ray `Optic.trace_generic`, seven scalars read
    back, the sphere built on the host, `ol_trace_opd`): same OPD map, pupil points, radius; the
    strategy's `_chief_ray` is the chief ray, built on first use."""
    be, stats = seams
    if "oracle" in request.node.name:
        pytest.skip("the oracle-backed stand-in has no ol_wavefront_reference")
    from optiland.wavefront import Wavefront

    def run(lens):
        w = Wavefront(lens, fields=[(0.0, 0.7)], wavelengths="primary", num_rays=8,
                      distribution="hexapolar", afocal=(reference_type == "plane"))
        d = w.get_data((0.0, 0.7), lens.primary_wavelength)
        chief = w.strategy._chief_ray
        # wavefront_data.py:36 and the reference's own test_wavefront_strategy.py:191: a float
        assert isinstance(d.radius, float)
        from optiland.wavefront.wavefront_data import WavefrontData
        back = pickle.loads(pickle.dumps(d))
        assert type(back) is WavefrontData and back.radius == d.radius
        if type(chief).__name__ == "_LazyChiefRay":
            # (advisor, round 4: deepcopy / pickle of the lazy chief ray recursed for ever --
            # and with it every object that holds the strategy)
            import copy
            for dup in (copy.deepcopy(chief), pickle.loads(pickle.dumps(chief)),
                        copy.deepcopy(w.strategy)._chief_ray):
                assert type(dup).__name__ == "_LazyChiefRay"
                assert float(_np(be, dup.y).reshape(-1)[0]) == \
                    float(_np(be, chief.y).reshape(-1)[0])
        return ([_np(be, getattr(d, k)) for k in ("opd", "intensity", "pupil_x", "pupil_y",
                                                   "pupil_z")], d.radius,
                [float(_np(be, getattr(chief, k)).reshape(-1)[0]) for k in
                 ("x", "y", "z", "L", "M", "N", "opd")], type(chief).__name__)

    got = run(_cooke())
    assert got[3] == "_LazyChiefRay" and stats["opd"] >= 1 and stats["opd_fallback"] == 0
    monkeypatch.setenv("OPTILAND_HIP_DEVICE_REFERENCE", "0")
    want = run(_cooke())
    assert want[3] == "RealRays"
    for a, b in zip(got[0], want[0]):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-9)   # OPD in waves; mm elsewhere
    if reference_type == "sphere":
        np.testing.assert_allclose(got[1], want[1], rtol=1e-14)
    else:
        assert got[1] == want[1] == float("inf")
    np.testing.assert_allclose(got[2], want[2], rtol=1e-13, atol=1e-13)


@pytest.mark.parametrize("absorbing", [False, True])
def test_an_optic_whose_last_surface_has_a_thickness(seams, request, absorbing, monkeypatch):
    """`Optic.trace` ends with a propagation by the last surface's thickness
    (real_ray_tracer.py:104-110); the fused kernels end AT the last surface.  An optic without an
    image plane (the reference's own `test_finite_conjugate_angle_field_opd` builds one: object,
    two lens surfaces, 95 mm to nowhere): round 4 made the seams DECLINE it (the OPD seam had
    been off by exactly those 95 mm -- 172 727 waves of piston -- invisibly to that test, which
    compares two such optics with each other); round 5 (ABI 10) hands the propagation to the
    kernels (`ol_wavefront_params.last_thickness / last_absorb`), host-built and
    device-resident reference alike, through an absorbing last medium too; the spot data are the
    RECORDED last row and never saw it.  The oracle-backed stand-in does not propagate and
    still declines."""
    be, stats = seams
    from optiland import analysis
    from optiland.materials import IdealMaterial
    from optiland.optic import Optic
    from optiland.wavefront import OPD
    served = "kernel-source" in request.node.name

    def build():
        optic = Optic()
        optic.surfaces.add(index=0, thickness=100.0)
        optic.surfaces.add(index=1, radius=50.0, thickness=5.0, material="BK7", is_stop=True)
        if absorbing:
            optic.surfaces.add(index=2, radius=-50.0, thickness=95.0,
                               material=IdealMaterial(n=1.1, k=2e-7))
        else:
            optic.surfaces.add(index=2, radius=-50.0, thickness=95.0)
        optic.set_aperture("EPD", 10.0)
        optic.wavelengths.add(0.55, is_primary=True)
        optic.fields.set_type("angle")
        optic.fields.add(y=5.0)
        return optic

    def run(lens):
        o = OPD(lens, field=(0, 1), wavelength="primary", num_rays=10, distribution="line_y")
        d = o.get_data((0, 1), lens.primary_wavelength)
        s = analysis.SpotDiagram(lens, num_rings=4)
        return (_np(be, d.opd), _np(be, d.intensity), _np(be, d.pupil_z), float(d.radius),
                _np(be, s.data[0][0].x), _np(be, s.data[0][0].y), _np(be, s.data[0][0].intensity))

    want = _numpy_reference(be, build, run)
    for device_reference in ("1", "0"):
        monkeypatch.setenv("OPTILAND_HIP_DEVICE_REFERENCE", device_reference)
        before, spots = stats["opd"], stats["spot"]
        got = run(build())
        assert stats["opd"] == before + (1 if served else 0)
        assert stats["spot"] == spots + 1
        np.testing.assert_allclose(got[0], want[0], rtol=0, atol=1e-6)     # waves
        # (the REPORTED intensity is the recorded last row: the 95 mm of absorbing medium
        # behind it -- exp(-0.43) -- do not show, in the reference or here)
        np.testing.assert_allclose(got[1], want[1], rtol=1e-12, atol=0)
        assert want[1].min() > 0.99
        np.testing.assert_allclose(got[2], want[2], rtol=0, atol=1e-8)
        np.testing.assert_allclose(got[3], want[3], rtol=1e-12)
        for a, b in zip(got[4:], want[4:]):
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-9)


# ----------------------------------------------------------------------------------
# CentroidStrategy / BestFitStrategy behind the reference's own classes (round 4)
# ----------------------------------------------------------------------------------
@pytest.mark.parametrize("reference_type", ["sphere", "plane"])
@pytest.mark.parametrize("strategy", ["centroid", "best_fit"])
@pytest.mark.parametrize("build", [_cooke, _singlet_asphere])
def test_fitted_strategies_through_the_fit_seam(seams, build, strategy, reference_type, request):
    """`CentroidStrategy.compute_wavefront_data` (strategy.py:307-364, inherited by
    `BestFitStrategy`) as generating launch + `ol_wavefront_fit` + `ol_wavefront_opd_fitted`,
    against the reference's own method run on the same strategy object (torch backend, through
    the drop-in's `Optic.trace`: the same rays): reference centre / radius, OPD map, pupil
    points, intensity, the `center` attribute BestFitStrategy keeps."""
    be, stats = seams
    if "oracle" in request.node.name:
        pytest.skip("the fit seam needs the generating launch of the product's engine")
    from optiland.wavefront import Wavefront
    from optiland_amd import analysis_seams

    lens = build()
    field, wl = (0.0, 0.7), lens.primary_wavelength
    w = Wavefront(lens, fields=[field], wavelengths="primary", num_rays=8,
                  distribution="hexapolar", strategy=strategy,
                  afocal=(reference_type == "plane"))
    assert stats["opd_fit"] == 1 and stats["opd_fit_fallback"] == 0
    got = w.get_data(field, wl)
    assert isinstance(got.radius, float)
    centre = getattr(w.strategy, "center", None)
    want = analysis_seams._ORIG["opd_fit"](w.strategy, field, wl)
    for k in ("opd", "intensity", "pupil_x", "pupil_y", "pupil_z"):
        a, b = _np(be, getattr(got, k)), _np(be, getattr(want, k))
        assert np.array_equal(np.isnan(a), np.isnan(b)), k
        # OPD in waves: 1e-8 (the two backends of the reference differ by 1.5e-10 on the golden
        # bundles, tools/make_golden_fitted.py); positions in mm
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-8 if k == "opd" else 1e-9,
                                   equal_nan=True)
    if reference_type == "sphere":
        np.testing.assert_allclose(got.radius, want.radius, rtol=1e-12)
        if strategy == "best_fit":
            np.testing.assert_allclose(centre, w.strategy.center, rtol=0, atol=1e-9)
    else:
        assert got.radius == want.radius == float("inf")
    # what Optic.trace would have left on the surfaces is there when somebody reads it
    assert lens.surfaces.x.shape[-1] == got.opd.shape[0]


def _window():
    """A plane-parallel plate in a collimated beam: no power anywhere."""
    from optiland.optic import Optic
    lens = Optic()
    lens.add_surface(index=0, radius=np.inf, thickness=np.inf)
    lens.add_surface(index=1, radius=np.inf, thickness=4.0, material="N-BK7", is_stop=True)
    lens.add_surface(index=2, radius=np.inf, thickness=30.0)
    lens.add_surface(index=3)
    lens.set_aperture(aperture_type="EPD", value=8.0)
    lens.set_field_type(field_type="angle")
    lens.add_field(y=0.0)
    lens.add_field(y=3.0)
    lens.add_wavelength(value=0.55, is_primary=True)
    return lens


def test_fit_seam_hands_a_flat_wavefront_to_the_reference(seams, request):
    """Round 5.  The wavefront points of a collimated beam lie in one plane: the sphere through
    them is not determined, the device fit says so (FIT_SINGULAR) and the seam declines -- the
    reference's own `BestFitStrategy` code runs (its backend's `lstsq`, whatever that makes of
    a rank-3 system) on the drop-in's trace.  The centroid sphere and the best-fit PLANE of the
    same beam stay on the device."""
    be, stats = seams
    if "oracle" in request.node.name:
        pytest.skip("the fit seam needs the generating launch of the product's engine")
    from optiland.wavefront import Wavefront
    from optiland_amd import analysis_seams

    lens = _window()
    field, wl = (0.0, 1.0), lens.primary_wavelength
    w = Wavefront(lens, fields=[field], wavelengths="primary", num_rays=6,
                  distribution="hexapolar", strategy="best_fit")
    assert stats["opd_fit"] == 0 and stats["opd_fit_fallback"] == 1
    got = w.get_data(field, wl)
    want = analysis_seams._ORIG["opd_fit"](w.strategy, field, wl)
    np.testing.assert_allclose(_np(be, got.opd), _np(be, want.opd), rtol=0, atol=1e-9,
                               equal_nan=True)
    for kw in (dict(strategy="best_fit", afocal=True), dict(strategy="centroid")):
        before = stats["opd_fit"]
        w = Wavefront(lens, fields=[field], wavelengths="primary", num_rays=6,
                      distribution="hexapolar", **kw)
        assert stats["opd_fit"] == before + 1 and stats["opd_fit_fallback"] == 1
        got = w.get_data(field, wl)
        want = analysis_seams._ORIG["opd_fit"](w.strategy, field, wl)
        np.testing.assert_allclose(_np(be, got.opd), _np(be, want.opd), rtol=0, atol=1e-8)


def test_fit_seam_declines_a_subclass_with_its_own_geometry(seams, request):
    be, stats = seams
    if "oracle" in request.node.name:
        pytest.skip("the fit seam needs the generating launch of the product's engine")
    from optiland.distribution import create_distribution
    from optiland.wavefront.strategy import CentroidStrategy

    class Mine(CentroidStrategy):
        def _calculate_weights(self, rays, image_points, valid_mask):
            return be.ones_like(rays.i[valid_mask])

    dist = create_distribution("hexapolar")
    dist.generate_points(4)
    lens = _cooke()
    d = Mine(lens, dist).compute_wavefront_data((0.0, 0.0), lens.primary_wavelength)
    assert stats["opd_fit"] == 0 and stats["opd_fit_fallback"] == 1
    assert np.isfinite(_np(be, d.opd)).all()


def test_fit_seam_raises_the_reference_errors(seams, request):
    """A bundle that is vignetted entirely: strategy.py:387 `No valid ray samples found`."""
    be, stats = seams
    if "oracle" in request.node.name:
        pytest.skip("the fit seam needs the generating launch of the product's engine")
    from optiland import physical_apertures
    from optiland.distribution import create_distribution
    from optiland.wavefront.strategy import BestFitStrategy, CentroidStrategy

    lens = _cooke()
    lens.surfaces.surfaces[2].aperture = physical_apertures.RadialAperture(r_max=1e-9,
                                                                                r_min=0.0)
    dist = create_distribution("hexapolar")
    dist.generate_points(3)
    dist.x, dist.y = dist.x[1:], dist.y[1:]   # without the chief ray: every ray is clipped
    for cls in (CentroidStrategy, BestFitStrategy):
        with pytest.raises(ValueError, match="No valid ray samples"):
            cls(lens, dist).compute_wavefront_data((0.0, 0.0), lens.primary_wavelength)
    assert stats["opd_fit"] == 0 and stats["opd_fit_fallback"] == 0


def _sample_ids():
    from tests.test_reference_protocol import SAMPLES
    return SAMPLES


@pytest.mark.parametrize("mod,name", _sample_ids(), ids=[n for _, n in _sample_ids()])
def test_fit_seam_over_every_sample_lens(seams, mod, name, request):
    """The fitted strategies on EVERY `optiland.samples` lens (photographic objectives,
    microscopes, eyepieces, telescopes, an eye model, a lithography lens ...: vignetted bundles,
    finite and infinite conjugates, steep fields): where the seam serves the optic its OPD map,
    pupil points, radius and piston equal the reference's own method on the same strategy
    object; where it declines (iterative ray aiming, unsupported surfaces, a trailing
    thickness) the reference's method runs and the result is the reference's by construction."""
    be, stats = seams
    if "oracle" in request.node.name:
        pytest.skip("the fit seam needs the generating launch of the product's engine")
    import importlib
    import time

    from optiland.wavefront import Wavefront
    from optiland_amd import analysis_seams

    lens = getattr(importlib.import_module(f"optiland.samples.{mod}"), name)()
    wl = lens.primary_wavelength
    field = (0.0, 0.7)
    t0 = time.perf_counter()
    for strategy in ("centroid", "best_fit"):
        before = stats["opd_fit"]
        try:
            w = Wavefront(lens, fields=[field], wavelengths="primary", num_rays=5,
                          distribution="hexapolar", strategy=strategy)
        except ValueError as exc:   # a bundle the reference itself refuses (fully vignetted)
            with pytest.raises(ValueError, match=str(exc)[:20]):
                analysis_seams._ORIG["opd_fit"](w.strategy if "w" in dir() else None, field, wl)
            continue
        if stats["opd_fit"] == before:
            continue  # declined: the reference's own code produced the data
        got = w.get_data(field, wl)
        want = analysis_seams._ORIG["opd_fit"](w.strategy, field, wl)
        a, b = _np(be, got.opd), _np(be, want.opd)
        assert np.array_equal(np.isnan(a), np.isnan(b)), (name, strategy)
        # waves; a few of the samples are far from diffraction limited (|OPD| ~ 1e3 waves)
        scale = max(1.0, float(np.nanmax(np.abs(b))) if np.isfinite(b).any() else 1.0)
        # ... and an OPD is a difference of optical paths of the size of the reference radius:
        # a few ulps of THAT, in waves (the Hubble sample: 57.6 m, 1 ulp = 2.6e-8 waves)
        ulps = 8 * np.finfo(np.float64).eps * abs(float(want.radius)) / (float(wl) * 1e-3)
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-9 * scale + ulps, equal_nan=True,
                                   err_msg=f"{name} {strategy}")
        np.testing.assert_allclose(got.radius, want.radius, rtol=1e-10)
        for k in ("pupil_x", "pupil_y", "pupil_z", "intensity"):
            np.testing.assert_allclose(_np(be, getattr(got, k)), _np(be, getattr(want, k)),
                                       rtol=0, atol=1e-8, equal_nan=True)
        if time.perf_counter() - t0 > 20.0:
            break  # (a lens with iterative aiming: seconds per trace)


@pytest.mark.parametrize("strategy", ["chief_ray", "centroid_sphere", "best_fit_sphere"])
def test_float32_backend_wavefronts_are_served_by_the_fp64_kernels(seams, strategy, request):
    """Round 6 (VERDICT r5 missing 4): a float32 backend used to keep the reference's own fp32
    wavefront chain ("fp32 wavefront" in the seam log).  The fused wavefront kernels compute in
    fp64 -- an OPD in waves is a difference of path lengths 2e5 waves long -- and the seams now
    serve such an optic with them and hand the maps over in the backend's precision: every
    array of the result is float32 (what the reference's own code would have returned), and its
    values are the fp64 NumPy reference's to float32 rounding -- closer to it than the
    reference's own float32 chain gets."""
    be, stats = seams
    if "oracle" in request.node.name:
        pytest.skip("the device-resident reference and the fit kernels have no oracle-backed "
                    "stand-in")
    import torch
    from optiland.psf import FFTPSF
    from optiland.wavefront import Wavefront

    def run(lens):
        w = Wavefront(lens, fields=[(0.0, 0.7)], wavelengths="primary", num_rays=9,
                      distribution="hexapolar", strategy=strategy)
        d = w.get_data((0.0, 0.7), lens.primary_wavelength)
        psf = FFTPSF(lens, (0.0, 0.7), lens.primary_wavelength, num_rays=32, grid_size=64,
                     strategy=strategy)
        return d, psf.psf

    want_d, want_psf = _numpy_reference(be, _cooke, run)
    be.set_precision("float32")
    try:
        for k in stats:
            stats[k] = 0
        d, psf = run(_cooke())
        served = stats["opd"] + stats["opd_fit"]
        assert served >= 2 and stats["opd_fallback"] == 0 and stats["opd_fit_fallback"] == 0
        assert stats["pupil"] >= 1 and stats["pupil_fallback"] == 0
        # (the PRESCRIPTION of a float32 backend is float32 too -- radii, thicknesses and
        # indices rounded to 6e-8 -- which alone moves an OPD by ~1e-4 waves: the reference's own
        # float32 chain adds the rounding of the 2e5-wave path lengths, 2e-2 waves, on top)
        for k in ("opd", "intensity", "pupil_x", "pupil_y", "pupil_z"):
            got = getattr(d, k)
            assert got.dtype == torch.float32, k
            np.testing.assert_allclose(_np(be, got), _np(be, getattr(want_d, k)), rtol=1e-5,
                                       atol=1e-3 if k == "opd" else 1e-5, err_msg=k)
        assert psf.dtype == torch.float32
        np.testing.assert_allclose(_np(be, psf), _np(be, want_psf), rtol=0,
                                   atol=2e-3 * float(np.max(_np(be, want_psf))))
    finally:
        be.set_precision("float64")


@pytest.mark.parametrize("absorbing", [False, True], ids=["clear", "absorbing"])
@pytest.mark.parametrize("strategy", ["centroid_sphere", "best_fit_sphere"])
def test_fitted_strategies_on_an_optic_whose_last_surface_has_a_thickness(seams, request, strategy,
                                                                          absorbing):
    """Round 6 (VERDICT r5 missing 4).  The fitted strategies read the rays `Optic.trace`
    RETURNS (strategy.py:319) -- moved on by the last surface's thickness, through its medium
    (real_ray_tracer.py:104-110) -- where the chief-ray kernels take the propagation as a launch
    parameter.  The fit seam applies it to the recorded row it fits (three elementwise
    operations) instead of declining the optic."""
    be, stats = seams
    if "oracle" in request.node.name:
        pytest.skip("the fit kernels have no oracle-backed stand-in")
    from optiland.materials import IdealMaterial
    from optiland.optic import Optic
    from optiland.wavefront import Wavefront

    def build():
        optic = Optic()
        optic.surfaces.add(index=0, thickness=100.0)
        optic.surfaces.add(index=1, radius=50.0, thickness=5.0, material="BK7", is_stop=True)
        if absorbing:
            optic.surfaces.add(index=2, radius=-50.0, thickness=95.0,
                               material=IdealMaterial(n=1.1, k=2e-7))
        else:
            optic.surfaces.add(index=2, radius=-50.0, thickness=95.0)
        optic.set_aperture("EPD", 10.0)
        optic.wavelengths.add(0.55, is_primary=True)
        optic.fields.set_type("angle")
        optic.fields.add(y=5.0)
        return optic

    def run(lens):
        w = Wavefront(lens, fields=[(0.0, 1.0)], wavelengths="primary", num_rays=8,
                      distribution="hexapolar", strategy=strategy)
        d = w.get_data((0.0, 1.0), lens.primary_wavelength)
        return [_np(be, getattr(d, k)) for k in ("opd", "intensity", "pupil_x", "pupil_y",
                                                  "pupil_z")] + [float(_np(be, d.radius))]

    want = _numpy_reference(be, build, run)
    before = stats["opd_fit"]
    got = run(build())
    assert stats["opd_fit"] == before + 1 and stats["opd_fit_fallback"] == 0
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=2e-6)       # waves
    if absorbing:
        assert want[1].max() < 0.7        # exp(-0.43): the returned rays DID cross the medium
    np.testing.assert_allclose(got[1], want[1], rtol=1e-12, atol=0)
    for a, b in zip(got[2:5], want[2:5]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-7)
    np.testing.assert_allclose(got[5], want[5], rtol=1e-9)


@pytest.mark.parametrize("reference", ["chief_ray", "centroid"])
def test_spot_radii_over_the_whole_grid_at_once(seams, reference, request):
    """Round 6: `rms_spot_radius` / `geometric_spot_radius` of a
    `SpotDiagram` whose data still are the blocks the grid launch wrote: one pass over the
    (cells, n) blocks with the reference's own centres, instead of a deep copy and five
    elementwise launches per cell.  Same numbers; a cell somebody replaced sends the call back
    to the reference's own method."""
    be, stats = seams
    if "oracle" in request.node.name:
        pytest.skip("the oracle-backed stand-in has no grid launch")
    from optiland import analysis

    def run(lens):
        s = analysis.SpotDiagram(lens, num_rings=5, reference=reference)
        return (np.array([[float(_np(be, v)) for v in f] for f in s.rms_spot_radius()]),
                np.array([[float(_np(be, v)) for v in f] for f in s.geometric_spot_radius()]))

    want = _numpy_reference(be, _cooke, run)
    before = stats["spot_radius"]
    got = run(_cooke())
    assert stats["spot_radius"] == before + 2 and stats["spot_grid"] >= 1
    np.testing.assert_allclose(got[0], want[0], rtol=1e-11)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-11)
    # a caller that edits the data gets the reference's own arithmetic on what it left there
    s = analysis.SpotDiagram(_cooke(), num_rings=5, reference=reference)
    cell = s.data[1][0]
    s.data[1][0] = type(cell)(x=cell.x * 2.0, y=cell.y, intensity=cell.intensity)
    before = stats["spot_radius"]
    edited = s.rms_spot_radius()
    assert stats["spot_radius"] == before
    assert float(_np(be, edited[1][0])) > 1.2 * got[0][1][0]


@pytest.mark.parametrize("state", ["unpolarized", "elliptical"])
@pytest.mark.parametrize("reference_type", ["sphere", "plane"])
def test_polarised_opd_map_through_the_seam(seams, state, reference_type, request):
    """Round 6 (VERDICT r5 missing 4, the last f4 decline): `Wavefront` / `OPD` of a POLARISED
    optic (C5's system).  The two traces are the drop-in's polarised launches; the reference's
    chain over the bundle is one `ol_wavefront_opd` launch; `prt_matrix` and `E_exits` are the
    returned rays' own.  Everything against the NumPy backend."""
    be, stats = seams
    if "oracle" in request.node.name:
        pytest.skip("the oracle-backed stand-in has no ol_wavefront_opd")
    from optiland.wavefront import Wavefront
    from tests import _live

    def build():
        return _live.zernike_fresnel(state)

    def run(lens):
        w = Wavefront(lens, fields=[(0.0, 1.0)], wavelengths="primary", num_rays=8,
                      distribution="hexapolar",
                      afocal=reference_type == "plane") if reference_type == "plane" else \
            Wavefront(lens, fields=[(0.0, 1.0)], wavelengths="primary", num_rays=8,
                      distribution="hexapolar")
        d = w.get_data((0.0, 1.0), lens.primary_wavelength)
        out = [_np(be, getattr(d, k)) for k in ("opd", "intensity", "pupil_x", "pupil_y",
                                                 "pupil_z")]
        out.append(np.asarray(be.to_numpy(d.prt_matrix)))
        out.append([np.asarray(be.to_numpy(e)) for e in d.E_exits])
        out.append(float(_np(be, d.radius)))
        return out

    try:
        want = _numpy_reference(be, build, run)
    except TypeError:
        pytest.skip("this reference has no `afocal` argument")
    before, declined = stats["opd"], stats["opd_fallback"]
    got = run(build())
    assert stats["opd"] == before + 1 and stats["opd_fallback"] == declined
    # (a Zernike surface: the per-ray Newton rule is within 1e-6 mm = 2e-3 waves of the reference's)
    np.testing.assert_allclose(got[0], want[0], rtol=0, atol=5e-3)
    np.testing.assert_allclose(got[1], want[1], rtol=1e-6, atol=1e-9)
    for a, b in zip(got[2:5], want[2:5]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)
    np.testing.assert_allclose(got[5], want[5], rtol=0, atol=1e-6)
    assert len(got[6]) == len(want[6])
    for a, b in zip(got[6], want[6]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)
    np.testing.assert_allclose(got[7], want[7], rtol=1e-9)


def test_field_coordinates_are_remembered_until_a_field_or_the_backend_changes(seams):
    """`FieldGroup.get_field_coords()` -- asked for by every analysis constructor, a dozen device
    launches and 2 F + 1 read-backs on the torch backend -- is computed once by the reference's own
    code and handed out again while the fields' (x, y) and the backend's arithmetic are what they
    were (`analysis_seams._memo_get_field_coords`)."""
    be, stats = seams
    from optiland_amd import analysis_seams
    original = analysis_seams._ORIG["field_coords"]
    lens = _cooke()
    fg = lens.fields
    first = fg.get_field_coords()
    assert first == original(fg) and stats.get("field_coords_memo", 0) == 0
    again = fg.get_field_coords()
    assert again == first and again is not first and stats["field_coords_memo"] == 1
    again.append("mine")                                   # the caller's list is the caller's
    assert fg.get_field_coords() == first
    fg.fields[1].y = fg.fields[1].y * 0.5                  # an edited field
    assert fg.get_field_coords() == original(fg) != first
    n = stats["field_coords_memo"]
    be.set_precision("float32")                            # other arithmetic: computed again
    try:
        assert fg.get_field_coords() == original(fg)
        assert stats["field_coords_memo"] == n
    finally:
        be.set_precision("float64")
    fg.fields[2].y = be.array(fg.fields[2].y)              # a tensor (a variable): no memo
    n = stats["field_coords_memo"]
    assert fg.get_field_coords() == original(fg) and fg.get_field_coords() == original(fg)
    assert stats["field_coords_memo"] == n
    import copy
    assert copy.deepcopy(lens).fields.get_field_coords() == original(fg)


def test_encircled_energy_reads_the_status_once_per_grid_and_still_raises(seams):
    """The encircled-energy cells need nothing of their launch on the host: inside the grid loop
    no cell reads the status word back -- it accumulates in the engine and is read once when the
    loop is over (`analysis_seams._image_hits` / `_spot_generate_data`).  A Zernike surface whose
    normalisation radius the beam overfills still raises the reference's own error
    (geometries/zernike.py:262-266), and a good lens after it is served as before."""
    be, stats = seams
    from optiland import analysis
    from tests import _live

    lens = _live.zernike_fresnel(polarization=None)
    e = analysis.EncircledEnergy(lens, num_rays=7, distribution="random", num_points=16)
    assert stats["ee"] == len(e.fields) * len(e.wavelengths) and stats["ee_fallback"] == 0
    assert stats["spot_grid"] == 0      # ("random": a draw per cell, the reference's own loop)
    bad = _live.zernike_fresnel(polarization=None)
    bad.surfaces[1].geometry.norm_radius = 5.0          # EPD 20: rays leave the unit disc
    with pytest.raises(ValueError, match="Zernike coordinates must be normalized"):
        analysis.EncircledEnergy(bad, num_rays=7, distribution="random", num_points=16)
    n = stats["ee"]
    analysis.EncircledEnergy(lens, num_rays=7, distribution="random", num_points=16)
    assert stats["ee"] > n and stats["ee_fallback"] == 0


def test_wavefront_and_fft_psf_constructors_walk_the_optic_once(seams, monkeypatch):
    """`Wavefront.__init__` / `ScalarFFTPSF.__init__` only read their optic: they run inside
    `integration.unchanged(optic)`, so the strategy constructor, the wavefront data and the pupil
    fill share ONE walk of the change detector -- and an edit between two constructors is seen."""
    be, stats = seams
    from optiland import wavefront
    from optiland.psf import FFTPSF
    from optiland_amd import fingerprint as fp
    from optiland_amd import integration as ig
    lens = _cooke()
    walks = {"n": 0}
    orig = fp.optic_token

    def counting(*a, **k):
        walks["n"] += 1
        return orig(*a, **k)

    monkeypatch.setattr(fp, "optic_token", counting)
    wavefront.OPD(lens, (0, 1), 0.55, num_rays=16)       # (packs the optic: any number of walks)
    for build in (lambda: wavefront.OPD(lens, (0, 1), 0.55, num_rays=16),
                  lambda: wavefront.OPD(lens, (0, 1), 0.55, num_rays=16, strategy="best_fit_sphere"),
                  lambda: FFTPSF(lens, (0, 1), 0.55, num_rays=32, grid_size=64)):
        build()
        walks["n"] = 0
        build()
        assert walks["n"] == 1, walks["n"]
    comp = ig.hip_tracer_of(lens)
    assert comp._hip_trusted is None and comp._hip_trust_depth == 0
    before = _np(be, FFTPSF(lens, (0, 1), 0.55, num_rays=32, grid_size=64).psf)
    lens.updater.set_radius(float(lens.surfaces[1].geometry.radius) * 1.05, 1)
    after = _np(be, FFTPSF(lens, (0, 1), 0.55, num_rays=32, grid_size=64).psf)
    assert np.abs(after - before).max() > 1e-6 * np.abs(before).max()


def test_analyses_follow_field_and_prescription_edits(seams):
    """Field coordinates edited in place, a field added, a radius and a thickness changed --
    between analyses of ONE optic: the remembered field coordinates, the one-walk constructor
    scopes and the whole-grid seams all have to notice.  Spot radii (local chief-ray and global
    centroid references), an OPD map and an FFT PSF after every edit, against the NumPy backend."""
    be, stats = seams
    from optiland import analysis, wavefront
    from optiland.psf import FFTPSF

    def measure(lens):
        out = []
        s = analysis.SpotDiagram(lens, num_rings=4)
        out += [_np(be, v) for f in s.rms_spot_radius() for v in f]
        out += [_np(be, v) for f in s.geometric_spot_radius() for v in f]
        s2 = analysis.SpotDiagram(lens, num_rings=4, coordinates="global", reference="centroid")
        out += [_np(be, v) for f in s2.rms_spot_radius() for v in f]
        o = wavefront.OPD(lens, (0, 1), 0.55, num_rays=16)
        out.append(_np(be, list(o.data.values())[0].opd))
        out.append(_np(be, FFTPSF(lens, (0, 0.7), 0.55, num_rays=32, grid_size=64).psf))
        return out

    def edit(lens, k):
        if k == 0:
            lens.fields.fields[1].y = 9.0
        elif k == 1:
            lens.fields.fields[2].x = 3.0
        elif k == 2:
            lens.fields.add(y=5.0)
        elif k == 3:
            lens.updater.set_radius(float(lens.surfaces[1].geometry.radius) * 1.03, 1)
        elif k == 4:
            lens.fields.fields[0].y = 1.0
        else:
            lens.updater.set_thickness(float(lens.surfaces[2].thickness) * 1.02, 2)

    def run(lens):
        states = [measure(lens)]
        for k in range(6):
            edit(lens, k)
            states.append(measure(lens))
        return states

    want = _numpy_reference(be, _cooke, run)
    got = run(_cooke())
    assert stats["spot_fallback"] == stats["opd_fallback"] == stats["pupil_fallback"] == 0
    assert stats["field_coords_memo"] > 0
    assert stats["spot_radius"] > 0 or stats["spot_grid"] == 0   # (the oracle stand-in: per cell)
    for step, (a, b) in enumerate(zip(got, want)):
        for u, v in zip(a, b):
            scale = max(float(np.abs(v).max()), 1e-30)
            assert float(np.abs(u - v).max()) <= 1e-8 * scale, step


def test_uniform_pupil_grid_is_remembered_per_size_and_backend_arithmetic(seams):
    """`UniformDistribution.generate_points` (distribution.py:176-186): the backend's OWN grid --
    the one its consumers' masks have to agree with -- computed once per (num_points, precision,
    device) and handed out as copies; autograd on: the reference's code."""
    be, stats = seams
    from optiland import distribution
    from optiland_amd import analysis_seams
    analysis_seams._UNIFORM_MEMO.clear()
    original = analysis_seams._ORIG["dist_uniform"]
    a = distribution.create_distribution("uniform")
    a.generate_points(33)
    assert stats.get("uniform_memo", 0) == 0
    b = distribution.create_distribution("uniform")
    b.generate_points(33)
    assert stats["uniform_memo"] == 1
    ref = distribution.create_distribution("uniform")
    original(ref, 33)
    for d in (a, b):
        assert np.array_equal(_np(be, d.x), _np(be, ref.x)) and np.array_equal(_np(be, d.y), _np(be, ref.y))
    b.x *= 2.0                                  # a caller's copy is the caller's
    c = distribution.create_distribution("uniform")
    c.generate_points(33)
    assert np.array_equal(_np(be, c.x), _np(be, ref.x))
    c.generate_points(34)                       # another size: computed
    assert c.x.shape != ref.x.shape and stats["uniform_memo"] == 2
    be.set_precision("float32")
    try:
        n = stats["uniform_memo"]
        d32 = distribution.create_distribution("uniform")
        d32.generate_points(33)
        assert stats["uniform_memo"] == n and d32.x.dtype != ref.x.dtype
    finally:
        be.set_precision("float64")
    be.grad_mode.enable()
    try:
        n = stats["uniform_memo"]
        distribution.create_distribution("uniform").generate_points(33)
        assert stats["uniform_memo"] == n
    finally:
        be.grad_mode.disable()
