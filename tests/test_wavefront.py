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
from optiland_amd.wavefront import FFTPSF, OPD, calculate_grid_size
from tests._util import GOLDEN

GOLD = dict(np.load(os.path.join(GOLDEN, "wavefront.npz")))
CASES = {"cooke": ("cooke_generic", (0.0, 1.0), 0.55),
         "dgauss": ("double_gauss", (0.0, 0.7), 0.5876)}


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
def test_wavefront_and_psf_host_logic(tag, monkeypatch):
    from tests._fake_engine import OracleEngine
    monkeypatch.setattr(tr, "_make_engine", lambda table, device: OracleEngine(table, device))
    name, field, wl = CASES[tag]
    t = tr.HipRayTracer(load_system(name), dtype=torch.float64)
    opd = _check(t, tag, field, wl, 1e-7, 1e-6)
    if tag == "cooke":  # the reference's own hard-coded golden
        np.testing.assert_allclose(opd.rms(), 0.9709788038168692, rtol=1e-5)


def test_wavefront_with_record_all_switched_off(monkeypatch):
    """Wavefront reads the recorded image-plane intensity; with the tracer in
    record-last mode (IncoherentIrradiance toggles it, users may) that used to raise an
    opaque IndexError -- the trace now forces record-all and restores the flag."""
    from tests._fake_engine import OracleEngine
    monkeypatch.setattr(tr, "_make_engine", lambda table, device: OracleEngine(table, device))
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


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_wavefront_and_psf_on_device(tag):
    name, field, wl = CASES[tag]
    t = tr.HipRayTracer(load_system(name), "cuda:0", dtype=torch.float64)
    opd = _check(t, tag, field, wl, 1e-6, 1e-5)
    if tag == "cooke":
        np.testing.assert_allclose(opd.rms(), 0.9709788038168692, rtol=1e-5)
    t.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("strategy", ["centroid_sphere", "best_fit_sphere"])
@pytest.mark.parametrize("tag", list(CASES))
def test_fitted_reference_spheres_on_device(tag, strategy):
    """Centroid / best-fit reference spheres (wavefront/strategy.py:287-582) and the
    weighted tilt removal on the real engine against the same host code on the
    oracle-backed engine (itself held to the live reference by tests/test_reference_fuzz.py)."""
    from tests._fake_engine import OracleEngine
    name, field, wl = CASES[tag]
    table = load_system(name)
    real = tr.HipRayTracer(table, "cuda:0", dtype=torch.float64)
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
