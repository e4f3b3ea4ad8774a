"""`ol_wavefront_fit` + `ol_wavefront_opd_fitted` (ABI 9): the fitted reference sphere / plane of
CentroidStrategy / BestFitStrategy (wavefront/strategy.py:287-620) as a chain of device
reductions.

Checked against (a) the reference's OWN results on bundles it traced itself
(tests/golden/wavefront_fitted.npz, tools/make_golden_fitted.py: both backends' flavours, clean
and "dirty" bundles -- NaN rays, unlit rays, negative weights, outliers the 3-sigma rule
removes), (b) `oracle.wavefront_fit`, the NumPy restatement pinned by the same file, on random
bundles and on the edge cases the reference's code distinguishes.  `-m gpu`: the kernels on the
MI355X through the C ABI; otherwise the same source on the host (tests/hostmath).
"""

import math
import os

import numpy as np
import pytest
import torch

from optiland_amd import _capi, load_system
from oracle import oracle
from tests._util import GOLDEN

GOLD = dict(np.load(os.path.join(GOLDEN, "wavefront_fitted.npz")))
LENSES = {"cooke": "cooke_generic", "dgauss": "double_gauss"}
WHERE = [pytest.param("cuda", marks=pytest.mark.gpu), "host"]
KINDS = ["centroid", "best_fit"]


@pytest.fixture(scope="module")
def engines():
    made = {}

    def get(where, name="cooke_generic"):
        key = (where, name)
        if key not in made:
            table = load_system(name)
            if where == "cuda":
                from optiland_amd.engine import HipSystem
                made[key] = HipSystem(table, torch.device("cuda", 0))
            else:
                from tests import _hostmath as hm
                if not hm.available():
                    pytest.skip("hipcc (used as host C++ compiler) missing")
                made[key] = hm.make_engine_class()(table)
        return made[key]

    yield get
    for e in made.values():
        e.close()


def _params(tag):
    rg = load_system(LENSES[tag]).raygen
    hx, hy = GOLD[f"{tag}_field"]
    tx = math.tan(math.radians(hx * rg["max_field"]))
    ty = math.tan(math.radians(hy * rg["max_field"]))
    uz = 1.0 / math.sqrt(1.0 + tx * tx + ty * ty)
    return dict(n_image=rg["n_image"], wavelength_um=float(GOLD[f"{tag}_wl"]), ux=tx * uz,
                uy=ty * uz, half_epd=rg["EPD"] / 2.0)


def _dev(eng, a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float64).to(eng.device)


def _fit(eng, kind, params, rays, px, py, **kw):
    r8 = [_dev(eng, r) for r in rays]
    px, py = _dev(eng, px), _dev(eng, py)
    ref = eng.wavefront_fit(kind, params, r8, px, py, **kw)
    opd, pupil = eng.wavefront_opd_fitted(ref, r8[:7], px, py)
    host = ref.cpu()
    return (host[:14].numpy(), int(host[-1:].view(torch.int32)[0]), opd.cpu().numpy(),
            pupil.cpu().numpy())


def _same(a, b, atol, what):
    assert np.array_equal(np.isnan(a), np.isnan(b)), f"{what}: NaN masks differ"
    np.testing.assert_allclose(a, b, rtol=0, atol=atol, equal_nan=True, err_msg=what)


# ---------------------------------------------------------------- the oracle, pinned
@pytest.mark.parametrize("flavour", ["numpy", "torch"])
@pytest.mark.parametrize("rtype", ["sphere", "plane"])
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("variant", ["clean", "dirty"])
@pytest.mark.parametrize("tag", list(LENSES))
def test_oracle_restatement_against_the_reference(tag, variant, kind, rtype, flavour):
    """Operation for operation the NumPy backend's code: bit-identical to it; the torch
    backend's results differ from it by the torch backend's own rounding (<= 2e-10 waves)."""
    k = f"{tag}_{variant}_{kind}_{rtype}_{flavour}"
    got = oracle.wavefront_fit(kind, _params(tag), GOLD[f"{tag}_{variant}_rays"],
                               GOLD[f"{tag}_px"], GOLD[f"{tag}_py"], flavour=flavour,
                               planar=rtype == "plane")
    exact = flavour == "numpy"
    _same(got["center"], GOLD[k + "_center"], 0.0 if exact else 1e-11, "centre")
    _same(got["opd"], GOLD[k + "_opd"], 0.0 if exact else 1e-9, "opd")
    _same(got["pupil"], GOLD[k + "_pupil"], 0.0 if exact else 1e-12, "pupil")
    if rtype == "sphere":
        np.testing.assert_allclose(got["radius"], GOLD[k + "_radius"], rtol=0 if exact else 1e-13)
    else:
        n, w = got["normal"], GOLD[k + "_normal"]
        assert min(np.abs(n - w).max(), np.abs(n + w).max()) <= (0.0 if exact else 1e-14)


# ---------------------------------------------------------------- the kernels
@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("flavour", ["numpy", "torch"])
@pytest.mark.parametrize("rtype", ["sphere", "plane"])
@pytest.mark.parametrize("kind", KINDS)
@pytest.mark.parametrize("variant", ["clean", "dirty"])
@pytest.mark.parametrize("tag", list(LENSES))
def test_fit_kernels_against_the_reference(engines, tag, variant, kind, rtype, flavour, where):
    eng = engines(where)
    k = f"{tag}_{variant}_{kind}_{rtype}_{flavour}"
    ref, bits, opd, pupil = _fit(eng, kind, _params(tag), GOLD[f"{tag}_{variant}_rays"],
                                 GOLD[f"{tag}_px"], GOLD[f"{tag}_py"], flavour=flavour,
                                 planar=rtype == "plane")
    assert bits == 0
    # centre: the least-squares sphere is solved from centred, scaled normal equations where
    # the reference calls LAPACK on the raw columns -- both are right to ~1e-11 mm (the
    # reference's two backends differ from each other by 1.5e-12)
    _same(ref[0:3], GOLD[k + "_center"], 5e-11, "centre")
    _same(opd, GOLD[k + "_opd"], 2e-9, "opd (waves)")
    _same(pupil, GOLD[k + "_pupil"], 1e-11, "pupil points (mm)")
    if rtype == "sphere":
        np.testing.assert_allclose(ref[3], GOLD[k + "_radius"], rtol=1e-12)
    else:
        n, w = ref[10:13], GOLD[k + "_normal"]
        assert min(np.abs(n - w).max(), np.abs(n + w).max()) <= 1e-14


def _random_bundle(rng, n, spread=0.02, dead=0.1):
    """A converging bundle around an image point ~100 mm behind a 20 mm pupil, with noise."""
    px = rng.uniform(-1, 1, n)
    py = rng.uniform(-1, 1, n)
    img = np.array([0.3, -2.0, 55.0])
    start = np.stack([10 * px, 10 * py + 4.0, np.full(n, -45.0)], axis=1)
    hit = img + rng.normal(0, spread, (n, 3)) * np.array([1, 1, 0])
    d = hit - start
    length = np.linalg.norm(d, axis=1)
    d /= length[:, None]
    opd = 12.0 + length * 1.0003 + rng.normal(0, 2e-4, n)
    inten = rng.uniform(0.3, 1.0, n)
    inten[rng.random(n) < dead] = 0.0
    rays = np.stack([hit[:, 0], hit[:, 1], hit[:, 2], d[:, 0], d[:, 1], d[:, 2], opd, inten])
    return rays, px, py


PARAMS = dict(n_image=1.0003, wavelength_um=0.6328, ux=0.0123, uy=-0.2, half_epd=10.0)


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("n", [4, 5, 257, 70001])
@pytest.mark.parametrize("planar", [False, True])
@pytest.mark.parametrize("kind", KINDS)
def test_fit_kernels_against_the_oracle_on_random_bundles(engines, kind, planar, n, where):
    """Sizes on both sides of one block and of one grid (70001 > 256 x 256: several blocks, the
    last one partly filled); trimming with other factors, and switched off."""
    eng = engines(where)
    rng = np.random.default_rng(n * 7 + planar)
    rays, px, py = _random_bundle(rng, n, dead=0.0 if n < 10 else 0.1)
    if n > 100:
        rays[1, :7] += 0.5  # outliers for the trimming to find
    for trim, flavour in ((3.0, "torch"), (1.5, "numpy"), (0.0, "torch")):
        want = oracle.wavefront_fit(kind, PARAMS, rays, px, py, trim_std=trim, flavour=flavour,
                                    planar=planar)
        ref, bits, opd, pupil = _fit(eng, kind, PARAMS, rays, px, py, trim_std=trim,
                                     flavour=flavour, planar=planar)
        assert bits == 0
        if kind == "best_fit" and planar and n < 10:
            continue  # a plane through 4-5 noisy points of a near-degenerate cloud: see below
        # (a least-squares sphere through a shallow noisy cap is ill-conditioned along its axis:
        # the centre is held relative to the radius, the OPD map -- which is what such a shift
        # cannot change -- absolutely)
        _same(ref[0:3], want["center"], 1e-9 if kind == "centroid" or planar
              else 1e-9 * want["radius"], "centre")
        if planar:
            m, w = ref[10:13], want["normal"]
            assert min(np.abs(m - w).max(), np.abs(m + w).max()) <= 1e-11
        else:
            np.testing.assert_allclose(ref[3], want["radius"], rtol=1e-9)
        np.testing.assert_allclose(ref[9], want["opd_ref"], rtol=0, atol=1e-10)  # mm, of ~110
        _same(opd, want["opd"], 5e-8, "opd")
        _same(pupil, want["pupil"], 1e-9, "pupil")


@pytest.mark.gpu
@pytest.mark.parametrize("kind", KINDS)
def test_fit_kernels_beyond_the_grid_cap(engines, kind):
    """3e6 rays: more than 2048 blocks x 4 rays per lane -- the grid-stride loop, every row of
    the workspace in the finishing sum."""
    eng = engines("cuda")
    rng = np.random.default_rng(17)
    rays, px, py = _random_bundle(rng, 3_000_001)
    rays[1, :50] += 0.5
    want = oracle.wavefront_fit(kind, PARAMS, rays, px, py, flavour="torch")
    ref, bits, opd, pupil = _fit(eng, kind, PARAMS, rays, px, py, flavour="torch")
    assert bits == 0
    _same(ref[0:3], want["center"], 1e-9 if kind == "centroid" else 1e-9 * want["radius"], "c")
    np.testing.assert_allclose(ref[3], want["radius"], rtol=1e-9)
    _same(opd, want["opd"], 5e-8, "opd")
    _same(pupil, want["pupil"], 1e-9, "pupil")


@pytest.mark.parametrize("where", WHERE)
def test_trimming_follows_the_flavour_of_the_standard_deviation(engines, where):
    """Eight rays, one far out, k = 2.55: a single outlier among n has a z-score of
    sqrt(n - 1) = 2.65 with np.std (ddof 0) and (n - 1) / sqrt(n) = 2.47 with torch.std (ddof 1)
    -- the NumPy flavour trims it, the torch flavour cannot.  (Found by the mutation test of round 4: no
    other case told the two divisors apart.)"""
    eng = engines(where)
    rng = np.random.default_rng(8)
    rays, px, py = _random_bundle(rng, 8, spread=1e-4, dead=0.0)
    rays[0, 0] += 5.0
    got = {}
    for flavour in ("numpy", "torch"):
        want = oracle.wavefront_fit("centroid", PARAMS, rays, px, py, trim_std=2.55,
                                    flavour=flavour)
        ref, bits, opd, _ = _fit(eng, "centroid", PARAMS, rays, px, py, trim_std=2.55,
                                 flavour=flavour)
        assert bits == 0
        _same(ref[0:3], want["center"], 1e-10, flavour)
        np.testing.assert_allclose(ref[3], want["radius"], rtol=1e-12)
        _same(opd, want["opd"], 5e-8, flavour)
        got[flavour] = ref[0:3].copy()
    assert abs(got["numpy"][0] - got["torch"][0]) > 0.1   # trimmed vs kept: 5 mm / 8 apart


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("axes", [(0, 1, 2), (2, 0, 1), (1, 2, 0)])
def test_best_fit_plane_whatever_axis_the_normal_is_along(engines, axes, where):
    """The least-squares plane's normal is the eigenvector of the SMALLEST eigenvalue wherever
    it ends up on the diagonal: the same bundle with its coordinate axes permuted.  (Mutation
    test of round 4: with the beam along z the smallest eigenvalue is always the third.)"""
    eng = engines(where)
    rng = np.random.default_rng(21)
    rays, px, py = _random_bundle(rng, 500, dead=0.05)
    perm = rays.copy()
    for k, a in enumerate(axes):
        perm[k], perm[3 + k] = rays[a], rays[3 + a]
    want = oracle.wavefront_fit("best_fit", PARAMS, perm, px, py, planar=True)
    ref, bits, opd, pupil = _fit(eng, "best_fit", PARAMS, perm, px, py, planar=True)
    assert bits == 0
    n, w = ref[10:13], want["normal"]
    assert min(np.abs(n - w).max(), np.abs(n + w).max()) <= 1e-10
    assert int(np.argmax(np.abs(n))) == axes.index(2)  # the beam axis, wherever it went
    _same(ref[0:3], want["center"], 1e-9, "plane point")
    _same(opd, want["opd"], 5e-8, "opd")
    _same(pupil, want["pupil"], 1e-9, "pupil")


@pytest.mark.parametrize("where", WHERE)
def test_fit_status_bits_are_the_reference_errors(engines, where):
    eng = engines(where)
    rng = np.random.default_rng(5)
    rays, px, py = _random_bundle(rng, 64, dead=0.0)
    from optiland_amd.engine import HipSystem

    def bits_of(r, kind="centroid"):
        return _fit(eng, kind, PARAMS, r, px, py)[1]

    dark = rays.copy()
    dark[7] = 0.0                                  # nothing lit: no valid sample
    for kind in KINDS:
        b = bits_of(dark, kind)
        assert b & _capi.FIT_NO_VALID
        with pytest.raises(ValueError, match="No valid ray samples"):
            HipSystem.raise_for_fit_status(b)
        with pytest.raises(ValueError, match="No valid ray samples"):
            oracle.wavefront_fit(kind, PARAMS, dark, px, py)
    few = rays.copy()
    few[7, 3:] = 0.0                               # three lit rays
    b = bits_of(few, "best_fit")
    assert b & _capi.FIT_TOO_FEW and not b & _capi.FIT_NO_VALID
    with pytest.raises(ValueError, match="at least 4"):
        HipSystem.raise_for_fit_status(b)
    with pytest.raises(ValueError, match="at least 4"):
        oracle.wavefront_fit("best_fit", PARAMS, few, px, py)
    assert bits_of(few, "centroid") == 0           # the centroid needs no minimum
    neg = rays.copy()
    neg[7] = -1.0                                  # valid (i != 0) but none with i > 0
    b = bits_of(neg, "centroid")
    assert b == _capi.FIT_NO_ALIVE
    with pytest.raises(ValueError, match="non-zero intensity"):
        HipSystem.raise_for_fit_status(b)
    with pytest.raises(ValueError, match="non-zero intensity"):
        oracle.wavefront_fit("centroid", PARAMS, neg, px, py)


@pytest.mark.parametrize("where", WHERE)
def test_unit_weights_when_every_weight_clamps_to_zero(engines, where):
    """strategy.py:406-414: negative intensities are clamped to 0; if that leaves no weight at
    all, every valid ray weighs 1.  (A few positive rays keep the piston defined.)"""
    eng = engines(where)
    rng = np.random.default_rng(11)
    rays, px, py = _random_bundle(rng, 300, dead=0.0)
    rays[7] = -rays[7]
    with pytest.raises(ValueError, match="non-zero intensity"):
        oracle.wavefront_fit("centroid", PARAMS, rays, px, py, flavour="numpy")
    ref, bits, opd, _ = _fit(eng, "centroid", PARAMS, rays, px, py, flavour="numpy")
    assert bits == _capi.FIT_NO_ALIVE  # ... which the reference reports AFTER the geometry
    # the geometry itself: compare with a bundle whose weights are all 1
    ones = rays.copy()
    ones[7] = 1.0
    ref1, bits1, _, _ = _fit(eng, "centroid", PARAMS, ones, px, py, flavour="numpy")
    assert bits1 == 0
    np.testing.assert_allclose(ref[0:4], ref1[0:4], rtol=0, atol=1e-12)


@pytest.mark.parametrize("where", WHERE)
def test_fit_is_reproducible_and_leaves_the_bundle_alone(engines, where):
    eng = engines(where)
    rng = np.random.default_rng(3)
    rays, px, py = _random_bundle(rng, 200_000)
    r8 = [_dev(eng, r) for r in rays]
    keep = [t.clone() for t in r8]
    dpx, dpy = _dev(eng, px), _dev(eng, py)
    a = eng.wavefront_fit("centroid", PARAMS, r8, dpx, dpy).cpu()
    b = eng.wavefront_fit("centroid", PARAMS, r8, dpx, dpy).cpu()
    assert torch.equal(a[:13], b[:13])   # sums in a fixed order: bit-identical
    c = eng.wavefront_fit("best_fit", PARAMS, r8, dpx, dpy).cpu()
    d = eng.wavefront_fit("best_fit", PARAMS, r8, dpx, dpy).cpu()
    assert torch.equal(c[:13], d[:13])
    for t, k in zip(r8, keep):
        assert torch.equal(t, k)


@pytest.mark.parametrize("where", WHERE)
def test_empty_bundle(engines, where):
    eng = engines(where)
    e = [torch.empty(0, dtype=torch.float64, device=eng.device) for _ in range(10)]
    ref = eng.wavefront_fit("centroid", PARAMS, e[:8], e[8], e[9])
    _r, bits = eng.fit_result(ref)
    assert bits == _capi.FIT_NO_VALID
    opd, pupil = eng.wavefront_opd_fitted(ref, e[:7], e[8], e[9])
    assert opd.numel() == 0 and pupil.shape == (3, 0)


def test_argument_checks():
    from tests import _hostmath as hm
    if not hm.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    eng = hm.make_engine_class()(load_system("cooke_generic"))
    rng = np.random.default_rng(1)
    rays, px, py = _random_bundle(rng, 16)
    r8 = [torch.as_tensor(r) for r in rays]
    with pytest.raises(ValueError, match="float64"):
        eng.wavefront_fit("centroid", PARAMS, [t.float() for t in r8], torch.as_tensor(px),
                          torch.as_tensor(py))
    with pytest.raises(KeyError):
        eng.wavefront_fit("median", PARAMS, r8, torch.as_tensor(px), torch.as_tensor(py))
    bad = dict(PARAMS, n_image=0.0)
    with pytest.raises(Exception, match="n_image"):
        eng.wavefront_fit("centroid", bad, r8, torch.as_tensor(px), torch.as_tensor(py))
    eng.close()
