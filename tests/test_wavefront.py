"""f4: wavefront OPD + scalar FFT PSF against the reference's own numbers.

CPU part: host logic with the oracle-backed engine (incl. the reference's hard-coded
golden `OPD(CookeTriplet(), (0, 1), 0.55).rms() == 0.9709788038168692`,
tests/test_wavefront.py:138-142).  GPU part: the same through the HIP kernels.
"""

import os

import numpy as np
import pytest
import torch

from optiland_amd import load_system, tracer as tr
from optiland_amd.wavefront import FFTPSF, OPD, Wavefront, calculate_grid_size
from tests._util import GOLDEN

GOLD = dict(np.load(os.path.join(GOLDEN, "wavefront.npz")))
CASES = {"cooke": ("cooke_generic", (0.0, 1.0), 0.55),
         "dgauss": ("double_gauss", (0.0, 0.7), 0.5876)}


# Where the "real" engine of a test runs: on the MI355X, or -- the product's engine class on
# the host build of the kernel source through the real C ABI (tests/_hostmath.py) -- on the
# CPU of a box without a GPU.  `-m gpu` selects the first, `-m "not gpu"` the second.
WHERE = [pytest.param("cuda", marks=pytest.mark.gpu), "host"]


def _real_tracer(table, where):
    if where == "cuda":
        return tr.HipRayTracer(table, "cuda:0", dtype=torch.float64)
    from tests import _hostmath as hm
    if not hm.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    return tr.HipRayTracer(table, "cpu", dtype=torch.float64, engine=hm.make_engine_class()(table))


@pytest.fixture(params=["oracle", "kernel-source"])
def cpu_engine(monkeypatch, request):
    """Host-logic tests run twice: oracle-backed stand-in, and the product's engine class on
    the host build of the kernel source."""
    if request.param == "oracle":
        from tests._fake_engine import OracleEngine
        monkeypatch.setattr(tr, "_make_engine", lambda table, device: OracleEngine(table, device))
    else:
        from tests import _hostmath as hm
        if not hm.available():
            pytest.skip("hipcc (used as host C++ compiler) missing")
        cls = hm.make_engine_class()
        monkeypatch.setattr(tr, "_make_engine", lambda table, device: cls(table, device))
    return request.param


def _check(tracer, tag, field, wl, rtol_opd, rtol_psf):
    opd = OPD(tracer, field, wl)
    d = opd.data
    np.testing.assert_allclose(d.radius, GOLD[f"{tag}_radius"], rtol=1e-12)
    np.testing.assert_allclose(d.opd.cpu().numpy(), GOLD[f"{tag}_opd"], rtol=rtol_opd,
                               atol=rtol_opd)
    got_p = torch.stack([d.pupil_x, d.pupil_y, d.pupil_z]).cpu().numpy()
    np.testing.assert_allclose(got_p, GOLD[f"{tag}_pupil"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(opd.rms(), GOLD[f"{tag}_rms"], rtol=1e-7)
    psf = FFTPSF(tracer, field, wl, num_rays=64)
    assert [psf.num_rays, psf.grid_size] == list(GOLD[f"{tag}_grid"])
    full = psf.psf.cpu().numpy()
    c = full.shape[0] // 2
    np.testing.assert_allclose(full[c - 16:c + 16, c - 16:c + 16], GOLD[f"{tag}_psf_center"],
                               rtol=rtol_psf, atol=rtol_psf * GOLD[f"{tag}_psf_center"].max())
    np.testing.assert_allclose(full.sum(), GOLD[f"{tag}_psf_sum"], rtol=rtol_psf)
    np.testing.assert_allclose(psf.strehl_ratio(), GOLD[f"{tag}_strehl"], rtol=rtol_psf)
    return opd


@pytest.mark.parametrize("tag", list(CASES))
def test_wavefront_and_psf_host_logic(tag, cpu_engine):
    name, field, wl = CASES[tag]
    t = tr.HipRayTracer(load_system(name), dtype=torch.float64)
    opd = _check(t, tag, field, wl, 1e-7, 1e-6)
    if tag == "cooke":  # the reference's own hard-coded golden
        np.testing.assert_allclose(opd.rms(), 0.9709788038168692, rtol=1e-5)


def test_wavefront_with_record_all_switched_off(cpu_engine):
    """Wavefront reads the recorded image-plane intensity; with the tracer in
    record-last mode (IncoherentIrradiance toggles it, users may) that used to raise an
    opaque IndexError -- the trace now forces record-all and restores the flag."""
    t = tr.HipRayTracer(load_system("cooke_generic"), dtype=torch.float64)
    want = OPD(t, (0.0, 1.0), 0.55).rms()
    t.record_all = False
    got = OPD(t, (0.0, 1.0), 0.55).rms()
    assert t.record_all is False
    np.testing.assert_allclose(got, want, rtol=1e-12)


def test_grid_size_rule():
    # psf/fft.py:20-39 (values from the reference's tests/test_fft_psf.py:60-72 table)
    assert calculate_grid_size(32) == (32, 64)
    assert calculate_grid_size(64) == (45, 128)
    assert calculate_grid_size(128) == (64, 256)
    with pytest.raises(ValueError):
        FFTPSF(None, (0, 0), 0.55, num_rays=16)


@pytest.mark.parametrize("n", [11, 21, 31, 75])
def test_fftpsf_odd_num_rays_with_explicit_grid(n, monkeypatch):
    """the pupil mask must select exactly the cells the 'uniform' sampler
    traced -- with a torch.linspace grid 41 odd sizes in 3..299 disagreed by boundary
    points (n=11: 81 cells vs 77 samples) and the scatter raised a shape mismatch."""
    from tests._fake_engine import OracleEngine
    monkeypatch.setattr(tr, "_make_engine", lambda table, device: OracleEngine(table, device))
    t = tr.HipRayTracer(load_system("cooke_generic"), dtype=torch.float64)
    psf = FFTPSF(t, (0.0, 0.0), 0.55, num_rays=n, grid_size=2 * n)
    assert psf.pupil.shape == (n, n) and psf.psf.shape == (2 * n, 2 * n)
    assert int((psf.pupil.abs() > 0).sum()) <= psf.wavefront.data.opd.numel()
    assert 0.0 < psf.strehl_ratio() <= 1.0 + 1e-9


def test_fp32_tracer_is_refused(monkeypatch):
    from tests._fake_engine import OracleEngine
    monkeypatch.setattr(tr, "_make_engine", lambda table, device: OracleEngine(table, device))
    t = tr.HipRayTracer(load_system("cooke_generic"), dtype=torch.float32)
    with pytest.raises(ValueError, match="fp64"):
        OPD(t, (0.0, 1.0), 0.55)


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("tag", list(CASES))
def test_wavefront_and_psf_on_device(tag, where):
    name, field, wl = CASES[tag]
    t = _real_tracer(load_system(name), where)
    opd = _check(t, tag, field, wl, 1e-6, 1e-5)
    if tag == "cooke":
        np.testing.assert_allclose(opd.rms(), 0.9709788038168692, rtol=1e-5)
    t.engine.close()


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("strategy", ["centroid_sphere", "best_fit_sphere"])
@pytest.mark.parametrize("tag", list(CASES))
def test_fitted_reference_spheres_on_device(tag, strategy, where):
    """Centroid / best-fit reference spheres (wavefront/strategy.py:287-582) and the
    weighted tilt removal on the real engine against the same host code on the
    oracle-backed engine (itself held to the live reference by tests/test_reference_fuzz.py)."""
    from tests._fake_engine import OracleEngine
    name, field, wl = CASES[tag]
    table = load_system(name)
    real = _real_tracer(table, where)
    fake = tr.HipRayTracer(table, "cpu", dtype=torch.float64, engine=OracleEngine(table, "cpu"))
    try:
        for detrend, afocal in ((False, False), (True, False), (False, True)):
            a = OPD(real, field, wl, num_rays=8, strategy=strategy, remove_tilt=detrend,
                    afocal=afocal)
            b = OPD(fake, field, wl, num_rays=8, strategy=strategy, remove_tilt=detrend,
                    afocal=afocal)
            if afocal:   # planar reference (reference_geometry.py:87-128)
                assert a.data.radius == b.data.radius == float("inf")
            else:
                np.testing.assert_allclose(a.data.radius, b.data.radius, rtol=1e-6)
            np.testing.assert_allclose(a.data.opd.cpu().numpy(), b.data.opd.numpy(), rtol=0,
                                       atol=1e-6 * max(1.0, float(b.data.opd.abs().max())))
            np.testing.assert_allclose(a.rms(), b.rms(), rtol=1e-5, atol=1e-7)
        c = OPD(real, field, wl, num_rays=8, afocal=True)     # chief-ray plane
        d = OPD(fake, field, wl, num_rays=8, afocal=True)
        np.testing.assert_allclose(c.data.opd.cpu().numpy(), d.data.opd.numpy(), rtol=0,
                                   atol=1e-6 * max(1.0, float(d.data.opd.abs().max())))
        psf = FFTPSF(real, field, wl, num_rays=32, strategy=strategy)
        assert 0.0 < psf.strehl_ratio() <= 1.0 + 1e-9
    finally:
        real.engine.close()


# ----------------------------------------------------------------------------------
# fused generate -> trace -> OPD (`ol_trace_opd`) and the pupil-function scatter
# (`ol_pupil_fill`): SURVEY.md 8 f4, VERDICT r1 "next" #9
# ----------------------------------------------------------------------------------
FUSED_SYSTEMS = ["cooke_generic", "double_gauss", "rc_asphere", "aspheric_singlet",
                 "apodized_gaussian_trace", "vignetted_trace", "finite_object_height_trace"]


def _golden_or_data(name):
    from tests._util import load_case_table
    try:
        return load_system(name)
    except KeyError:
        return load_case_table(name)


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("name", FUSED_SYSTEMS)
def test_fused_opd_kernel_vs_oracle_composition(name, where):
    """ol_trace_opd == oracle generate -> oracle trace -> oracle OPD on the same pupil
    points (conic, Newton, apodized, vignetted systems; hexapolar + a ragged random set),
    and its 12 device sums == the numpy reductions of that map."""
    from tests._fake_engine import OracleEngine
    table = _golden_or_data(name)
    if not table.raygen or "pupil_z" not in table.raygen:
        pytest.skip("no ray-generation / exit-pupil scalars in this table")
    import copy
    real = _real_tracer(table, where)
    # the oracle with its Newton loops fully converged: the reference (and the oracle as its
    # restatement) stops the whole batch at max|f| < tol = 1e-6 mm, which is 2e-3 WAVES of
    # path; the kernel's per-ray rule converges each ray further, so the
    # converged oracle is the yardstick, as in test_record_matches_reference
    conv = copy.deepcopy(table)
    newton = bool(np.any(conv.surfaces["max_iter"] > 0))
    conv.surfaces["tol"] = np.where(conv.surfaces["max_iter"] > 0, 1e-13, conv.surfaces["tol"])
    fake = tr.HipRayTracer(conv, "cpu", dtype=torch.float64, engine=OracleEngine(conv, "cpu"))
    w = float(table.wavelengths[0])
    try:
        for field in ((0.0, 0.0), (0.0, 0.7)):
            a = OPD(real, field, w, num_rays=9, fused=True)
            b = OPD(fake, field, w, num_rays=9, fused=True)
            assert a.fused and a.data.moments is not None
            scale = max(1.0, float(b.data.opd.abs().max()))
            np.testing.assert_allclose(a.data.opd.cpu().numpy(), b.data.opd.numpy(), rtol=0,
                                       atol=(2e-5 if newton else 2e-7) * scale)
            np.testing.assert_allclose(a.data.intensity.cpu().numpy(), b.data.intensity.numpy(),
                                       rtol=1e-9, atol=1e-12)
            for k in ("pupil_x", "pupil_y", "pupil_z"):
                np.testing.assert_allclose(getattr(a.data, k).cpu().numpy(),
                                           getattr(b.data, k).numpy(), rtol=1e-8, atol=1e-8)
            ma, mb = a.data.moments.cpu().numpy(), b.data.moments.numpy()
            mt = 1e-4 if newton else 1e-6
            np.testing.assert_allclose(ma, mb, rtol=mt, atol=mt * np.abs(mb).max())
            np.testing.assert_allclose(a.rms(), b.rms(), rtol=mt, atol=1e-6 if newton else 1e-9)
            # the device sums really are the reductions of the device map
            o, wi = a.data.opd.cpu().numpy(), a.data.intensity.cpu().numpy()
            alive = wi > 0
            np.testing.assert_allclose(ma[9:], [alive.sum(), o[alive].sum(), (o[alive] ** 2).sum()],
                                       rtol=1e-10, atol=1e-9)
            X = a.data.pupil_x.cpu().numpy()
            np.testing.assert_allclose(ma[7], (wi * o * X).sum(), rtol=1e-9, atol=1e-9)
    finally:
        real.engine.close()


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("tag", list(CASES))
def test_fused_equals_unfused_on_device(tag, where):
    """Same device, same kernels' arithmetic: the fused launch reproduces the chain
    ol_generate_rays -> ol_trace (record-all) -> ol_wavefront_opd -> torch reductions,
    with and without tilt removal, and so does the PSF built by ol_pupil_fill."""
    name, field, wl = CASES[tag]
    t = _real_tracer(load_system(name), where)
    try:
        for detrend in (False, True):
            a = OPD(t, field, wl, num_rays=12, remove_tilt=detrend, fused=True)
            b = OPD(t, field, wl, num_rays=12, remove_tilt=detrend, fused=False)
            assert a.fused and not b.fused and b.data.moments is None
            scale = max(1.0, float(b.data.opd.abs().max()))
            np.testing.assert_allclose(a.data.opd.cpu().numpy(), b.data.opd.cpu().numpy(), rtol=0,
                                       atol=1e-9 * scale)
            assert torch.equal(a.data.intensity, b.data.intensity)
            np.testing.assert_allclose(a.rms(), b.rms(), rtol=1e-9)
        pa = FFTPSF(t, field, wl, num_rays=64, fused=True)
        pb = FFTPSF(t, field, wl, num_rays=64, fused=False)
        np.testing.assert_allclose(pa.psf.cpu().numpy(), pb.psf.cpu().numpy(), rtol=1e-7,
                                   atol=1e-9 * float(pb.psf.max()))
        np.testing.assert_allclose(pa.strehl_ratio(), pb.strehl_ratio(), rtol=1e-9)
    finally:
        t.engine.close()


@pytest.mark.parametrize("where", WHERE)
def test_fit_and_remove_tilt_twice_on_fused_data(where):
    """The public static `fit_and_remove_tilt(data)` always fits the data it is given
    (wavefront/wavefront.py:103-148): the fused kernel's device sums describe the OPD as first
    produced, so a second call on detrended data -- or a call after the caller edited
    `data.opd` -- must not subtract the original plane again."""
    name, field, wl = CASES["cooke"]
    t = _real_tracer(load_system(name), where)
    try:
        a = OPD(t, field, wl, num_rays=10, remove_tilt=True, fused=True)
        b = OPD(t, field, wl, num_rays=10, remove_tilt=True, fused=False)
        assert a.fused and a.data.moments is None  # stale sums are dropped with the detrend
        again = Wavefront.fit_and_remove_tilt(a.data)
        ref = Wavefront.fit_and_remove_tilt(b.data)
        scale = max(1.0, float(b.data.opd.abs().max()))
        # idempotent: a detrended map has no plane left to take out
        np.testing.assert_allclose(again.cpu().numpy(), a.data.opd.cpu().numpy(), rtol=0,
                                   atol=1e-9 * scale)
        np.testing.assert_allclose(again.cpu().numpy(), ref.cpu().numpy(), rtol=0, atol=1e-9 * scale)
        np.testing.assert_allclose(a.rms(), b.rms(), rtol=1e-9)
        # piston removal on the detrended data == the un-fused path's
        pa = Wavefront.fit_and_remove_tilt(a.data, remove_piston=True)
        pb = Wavefront.fit_and_remove_tilt(b.data, remove_piston=True)
        np.testing.assert_allclose(pa.cpu().numpy(), pb.cpu().numpy(), rtol=0, atol=1e-9 * scale)
        # a caller that edits data.opd of a FUSED, un-detrended map gets a fit of the edited map
        c = OPD(t, field, wl, num_rays=10, remove_tilt=False, fused=True)
        assert c.data.moments is not None
        c.data.opd = c.data.opd + 0.25 * c.data.pupil_x
        d = OPD(t, field, wl, num_rays=10, remove_tilt=False, fused=False)
        d.data.opd = d.data.opd + 0.25 * d.data.pupil_x
        np.testing.assert_allclose(Wavefront.fit_and_remove_tilt(c.data).cpu().numpy(),
                                   Wavefront.fit_and_remove_tilt(d.data).cpu().numpy(), rtol=0,
                                   atol=1e-9 * scale)
        np.testing.assert_allclose(c.rms(), d.rms(), rtol=1e-9)
    finally:
        t.engine.close()


@pytest.mark.gpu
def test_pupil_fill_kernel_equals_definition():
    from optiland_amd.engine import HipSystem
    hip = HipSystem(load_system("cooke_generic"), "cuda:0")
    try:
        for n, gsz in ((11, 22), (32, 64), (45, 128), (75, 151)):
            g = np.linspace(-1.0, 1.0, n)
            xg, yg = np.meshgrid(g, g)
            cells = np.flatnonzero((xg**2 + yg**2 <= 1).reshape(-1)).astype(np.int32)
            m = cells.size
            gen = torch.Generator(device="cuda:0").manual_seed(n)
            opd = torch.randn(m, generator=gen, device="cuda:0", dtype=torch.float64) * 3
            inten = torch.rand(m, generator=gen, device="cuda:0", dtype=torch.float64)
            inten[::7] = 0
            X = torch.randn(m, generator=gen, device="cuda:0", dtype=torch.float64)
            Y = torch.randn(m, generator=gen, device="cuda:0", dtype=torch.float64)
            cell = torch.from_numpy(cells).cuda()
            for plane in (None, (0.3, -0.2, 0.15)):
                got = hip.pupil_fill(opd, inten, cell, n, gsz,
                                     pupil_xy=None if plane is None else (X, Y), plane=plane)
                o = opd if plane is None else opd - (plane[0] + plane[1] * X + plane[2] * Y)
                want = torch.zeros(n * n, dtype=torch.complex128, device="cuda:0")
                want[cell.long()] = torch.sqrt(inten) * torch.exp(-2j * np.pi * o)
                before = (gsz - n) // 2
                after = before + (gsz - n) % 2
                want = torch.nn.functional.pad(want.reshape(n, n), (before, after, before, after))
                assert got.shape == (gsz, gsz)
                np.testing.assert_allclose(got.cpu().numpy(), want.cpu().numpy(), rtol=0, atol=1e-12)
    finally:
        hip.close()


@pytest.mark.gpu
def test_fused_opd_refusals():
    from optiland_amd.engine import HipSystem
    hip = HipSystem(load_system("zernike_fresnel_fringe"), "cuda:0")   # polarised coatings
    px = torch.zeros(8, dtype=torch.float64, device="cuda:0")
    params = dict(xc=0.0, yc=0.0, zc=80.0, R=50.0, n_image=1.0, opd_ref=0.0, ux=0.0, uy=0.0,
                  half_epd=10.0, wavelength_um=0.55)
    with pytest.raises(ValueError, match="Polarization must be set"):
        hip.trace_opd(params, px, px, 0, field=(0.0, 0.0))
    hip.close()
    hip = HipSystem(load_system("cooke_generic"), "cuda:0")
    with pytest.raises(ValueError, match="fp64"):
        hip.trace_opd(params, px.float(), px.float(), 0, field=(0.0, 0.0))
    opd, inten, pupil, mom = hip.trace_opd(params, px[:0], px[:0], 0, field=(0.0, 0.0))
    assert opd.numel() == 0 and float(mom.abs().sum()) == 0.0
    t = tr.HipRayTracer(load_system("zernike_fresnel_fringe"), "cuda:0", dtype=torch.float64)
    with pytest.raises(ValueError, match="unpolarised"):
        OPD(t, (0.0, 0.0), 0.55, fused=True)
    hip.close()
    t.engine.close()


def _opd_capable_goldens():
    from tests._util import golden_cases, load_case_table
    out = []
    for c in golden_cases():
        if not (c.startswith("sample_") or c.startswith("fuzz_")):
            continue
        t = load_case_table(c)
        if t.raygen and "pupil_z" in t.raygen and t.polarization is None \
                and not t.uses_polarization:
            out.append(c)
    return out


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("case", _opd_capable_goldens())
def test_fused_opd_on_every_sample_and_fuzz_lens(case, where):
    """Breadth: the fused kernel on every unpolarised lens of optiland.samples and of the
    frozen fuzz set that carries exit-pupil data (real prescriptions, stops in odd places,
    aspheres, tilts, vignetting, object-height fields ...), two fields each, against the
    converged oracle composition."""
    import copy
    from tests._fake_engine import OracleEngine
    from tests._util import load_case_table
    table = load_case_table(case)
    conv = copy.deepcopy(table)
    newton = bool(np.any(conv.surfaces["max_iter"] > 0))
    conv.surfaces["tol"] = np.where(conv.surfaces["max_iter"] > 0, 1e-13, conv.surfaces["tol"])
    real = _real_tracer(table, where)
    fake = tr.HipRayTracer(conv, "cpu", dtype=torch.float64, engine=OracleEngine(conv, "cpu"))
    w = float(table.wavelengths[0])
    try:
        for field in ((0.0, 0.0), (0.0, 0.6)):
            try:
                b = OPD(fake, field, w, num_rays=6, fused=True)
            except ValueError as exc:   # e.g. Zernike range error: must be raised on both sides
                with pytest.raises(ValueError, match=str(exc)[:30]):
                    OPD(real, field, w, num_rays=6, fused=True)
                continue
            a = OPD(real, field, w, num_rays=6, fused=True)
            ob, oa = b.data.opd.numpy(), a.data.opd.cpu().numpy()
            assert np.array_equal(np.isnan(oa), np.isnan(ob)), case
            fin = np.isfinite(ob)
            scale = max(1.0, float(np.abs(ob[fin]).max())) if fin.any() else 1.0
            # an OPD is a difference of two path lengths: rounding at 1e-12 of the PATH
            # (57 m in the Hubble sample = 1e8 waves) is its floor, whatever its own size
            z = table.surfaces["origin"][:, 2]
            z = z[np.isfinite(z)]
            path_waves = float(np.abs(np.diff(z)).sum()) / (w * 1e-3)
            np.testing.assert_allclose(oa[fin], ob[fin], rtol=0,
                                       atol=(5e-5 if newton else 5e-7) * scale + 3e-12 * path_waves,
                                       err_msg=case)
            ia, ib = a.data.intensity.cpu().numpy(), b.data.intensity.numpy()
            assert np.array_equal(ia == 0, ib == 0), case
            np.testing.assert_allclose(ia, ib, rtol=1e-8, atol=1e-12, equal_nan=True)
    finally:
        real.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["double_gauss", "apodized_gaussian_trace"])
def test_fused_opd_two_rays_per_lane_is_bit_identical(name):
    """Round 5: without a Newton surface a launch of >= 2^20 rays runs `opd_trace_kernel` with
    TWO rays per lane (16-byte loads / stores, two independent chains); a ray's arithmetic is
    the same, so the map, the intensities and the pupil points equal the one-ray form's bit
    for bit (`OL_TUNE_RAYS_PER_THREAD = 1`), the 12 sums to their rounding.  An odd count
    (the last lane holds one ray) and an unaligned plane (one ray per lane again) included."""
    from optiland_amd import _capi
    from optiland_amd.engine import HipSystem
    table = _golden_or_data(name)
    hip = HipSystem(table, "cuda:0")
    t = tr.HipRayTracer(table, "cuda:0", dtype=torch.float64, engine=hip)
    wf = Wavefront(t, (0.0, 0.7), float(table.wavelengths[0]), num_rays=3)
    params = wf.chief_reference()[0]
    g = torch.Generator(device="cuda:0").manual_seed(5)
    n = (1 << 20) + 4097
    r = torch.rand(n + 1, generator=g, device="cuda:0", dtype=torch.float64).sqrt()
    th = 2 * np.pi * torch.rand(n + 1, generator=g, device="cuda:0", dtype=torch.float64)
    px_all, py_all = r * th.cos(), r * th.sin()
    try:
        for lo, count, want_pupil in ((0, n - 1, True), (0, n, False), (1, n - 1, False)):
            px, py = px_all[lo:lo + count], py_all[lo:lo + count]   # lo = 1: 8-byte aligned only
            out = {}
            for rpt in (1, 0):
                assert hip.lib.ol_set_tuning(_capi.TUNE_RAYS_PER_THREAD, rpt) == 0
                out[rpt] = hip.trace_opd(params, px, py, 0, field=(0.0, 0.7),
                                         want_pupil=want_pupil)
            a, b = out[1], out[0]
            assert torch.isfinite(a[0]).float().mean() > 0.5
            assert torch.equal(a[0].nan_to_num(), b[0].nan_to_num())
            assert torch.equal(torch.isnan(a[0]), torch.isnan(b[0]))
            assert torch.equal(a[1], b[1])
            if want_pupil:
                assert torch.equal(a[2].nan_to_num(), b[2].nan_to_num())
            np.testing.assert_allclose(b[3].cpu().numpy(), a[3].cpu().numpy(), rtol=1e-11,
                                       atol=1e-9 * float(a[3].abs().max()))
    finally:
        hip.lib.ol_set_tuning(_capi.TUNE_RAYS_PER_THREAD, 0)
        hip.close()


# ----------------------------------------------------------------------------------
# round 5: wavefront points that do not span space (found by the analyses fuzz on the fitted
# strategies, tools/gpu_fuzz_analyses.py)
# ----------------------------------------------------------------------------------
@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("name", ["fuzz_r05_flat_wavefront", "fuzz_r05_window",
                                  "fuzz_r05_flat_wavefront_b", "fuzz_r05_flat_wavefront_c"])
def test_best_fit_sphere_of_a_collimated_beam_is_the_references(name, where):
    """A system without power (air / air surfaces; a plane-parallel plate): the wavefront
    points `p - (opd / n) d` of a field lie in ONE plane (z spread 1e-14 mm) and the sphere
    through them is not determined.  The reference's NumPy backend returns the rank-3
    minimum-norm solution of `lstsq([x y z 1], |p|^2)` (strategy.py:556-582): radius 4.68 /
    4.72 mm, hundreds of waves of "OPD" -- an artefact, but its answer.  The device fit used to
    scale the flat axis by the square root of its cancellation noise and fit a sphere to that
    (radius 1e2 ... 1e13) or stop with "singular normal equations"; now it reports the cloud
    as singular and the host follows the reference (`Wavefront._best_fit_rank_deficient`)."""
    from tests._fake_engine import OracleEngine
    from tests._util import load_case_table
    table = load_case_table(name)
    real = _real_tracer(table, where)
    fake = tr.HipRayTracer(table, "cpu", dtype=torch.float64, engine=OracleEngine(table, "cpu"))
    w = float(table.wavelengths[0])
    try:
        a = OPD(real, (0.0, 0.7), w, num_rays=5, strategy="best_fit_sphere")
        b = OPD(fake, (0.0, 0.7), w, num_rays=5, strategy="best_fit_sphere")
        assert 4.0 < b.data.radius < 5.0                       # the minimum-norm artefact
        # (_b / _c: two more of the eleven lenses of the fuzz -- the ones the MI355X run still
        # showed wrong while a stale aux_kernels.o kept the device on the old fit; the build's
        # header list had missed wavefront_fit_device.h)
        np.testing.assert_allclose(a.data.radius, b.data.radius, rtol=1e-9)
        scale = float(b.data.opd.abs().max())
        assert scale > 100.0
        np.testing.assert_allclose(a.data.opd.cpu().numpy(), b.data.opd.numpy(), rtol=0,
                                   atol=1e-8 * scale)
        np.testing.assert_allclose(a.rms(), b.rms(), rtol=1e-8)
        # the fits that such a beam wants are untouched: a plane, and the centroid sphere
        for kw in (dict(strategy="best_fit_sphere", afocal=True), dict(strategy="centroid_sphere")):
            a = OPD(real, (0.0, 0.7), w, num_rays=5, **kw)
            b = OPD(fake, (0.0, 0.7), w, num_rays=5, **kw)
            np.testing.assert_allclose(a.data.opd.cpu().numpy(), b.data.opd.numpy(), rtol=0,
                                       atol=1e-7 * max(1.0, float(b.data.opd.abs().max())))
    finally:
        real.engine.close()


@pytest.mark.parametrize("where", WHERE)
def test_wavefront_fit_reports_a_tilted_plane_of_points_as_singular(where):
    """`ol_wavefront_fit` (best fit, sphere) on points of a TILTED plane -- no axis is flat, the
    scaled Gram matrix simply has no fourth pivot -- and on a cloud with one flat axis: the
    FIT_SINGULAR bit; a proper spherical cap of the same size: no bit, centre and radius to
    1e-9."""
    from optiland_amd import _capi
    table = load_system("cooke_generic")
    real = _real_tracer(table, where)
    eng, dev = real.engine, real.device
    rng = np.random.default_rng(3)
    n = 500
    u, v = rng.uniform(-2, 2, n), rng.uniform(-2, 2, n)

    def fit(points):
        # rays at the points themselves, zero path: pts = p
        planes = [torch.as_tensor(np.ascontiguousarray(c), dtype=torch.float64, device=dev)
                  for c in (points[:, 0], points[:, 1], points[:, 2], np.zeros(n), np.zeros(n),
                            np.ones(n), np.zeros(n), np.ones(n))]
        zero = torch.zeros(n, dtype=torch.float64, device=dev)
        ref = eng.wavefront_fit("best_fit", dict(n_image=1.0, wavelength_um=0.55, ux=0.0, uy=0.0,
                                                 half_epd=1.0), planes, zero, zero,
                                flavour="numpy")
        host = ref.cpu()
        return host[:4].numpy(), int(host[-1:].view(torch.int32)[0])

    try:
        origin = np.array([3.0, -2.0, 40.0])
        e1, e2 = np.array([1.0, 0.2, 0.3]), np.array([-0.1, 1.0, 0.5])
        tilted = origin + u[:, None] * e1 + v[:, None] * e2
        assert fit(tilted)[1] & _capi.FIT_SINGULAR
        flat = np.stack([u + 3.0, v - 2.0, np.full(n, 40.0)], axis=1)
        assert fit(flat)[1] & _capi.FIT_SINGULAR
        centre, R = np.array([3.0, -2.0, -60.0]), 100.0
        cap = np.stack([u + 3.0, v - 2.0,
                        centre[2] + np.sqrt(R * R - u * u - v * v)], axis=1)
        got, bits = fit(cap)
        assert bits == 0
        np.testing.assert_allclose(got[:3], centre, rtol=0, atol=1e-7)
        np.testing.assert_allclose(got[3], R, rtol=1e-9)
    finally:
        eng.close()
