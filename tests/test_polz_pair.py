"""The polarised Zernike fp32 generating launch on PAIRS of rays (configuration C5's kernel,
`trace_kernel<float, 2, true, 1, kNrZernike, ..., kGenUniform>`: two rays per lane traced as one
f32x2 through `surface_step<f32x2, 1, 1, kNrZernike>`, csrc/surface_math.h "pair forms") against
the one-ray-per-lane form of the SAME launch: record block, PRT planes, updated intensity and the
status word, bit for bit -- the pair forms state the scalar code's operations element for element.

Here on the host build of the kernel source behind the unmodified C ABI (tests/hostmath);
`tests/test_gpu_fuzz.py::test_polarised_zernike_pair_equals_the_one_ray_form` is the device twin.
What the reference computes for these systems: the goldens / oracle tests of the one-ray form.
"""

import copy

import numpy as np
import pytest
import torch

from optiland_amd import _capi, load_system
from optiland_amd import system as S
from tests import _hostmath as hm

pytestmark = pytest.mark.skipif(not hm.available(), reason="hipcc (host C++ compiler) missing")

STATE = {"is_polarized": False, "Ex": None, "Ey": None, "phase_x": None, "phase_y": None}
POLARISED = {"is_polarized": True, "Ex": 0.8, "Ey": 0.6, "phase_x": 0.3, "phase_y": -0.4}


def random_c5_table(seed: int):
    """The packaged Zernike freeform singlet, shaken: coefficients, curvatures, conics, coatings
    (Fresnel / none / simple), apertures, decentres and tilts, an absorbing glass, a mirror on
    some seeds, the normalisation radius small enough on some that rays leave the unit disc."""
    rng = np.random.default_rng(77_000 + seed)
    table = copy.deepcopy(load_system("zernike_fresnel_fringe"))
    s = table.surfaces
    c = table.coeffs
    off, nterm = int(s[1]["coeff_offset"]), int(s[1]["n_coeff"])
    for k in range(nterm):
        c[off + 4 * k] = rng.uniform(-6e-4, 6e-4)
    s[1]["radius"] = rng.choice([-1, 1]) * rng.uniform(35.0, 200.0)
    s[1]["conic"] = rng.choice([0.0, -1.0, rng.uniform(-1.5, 0.6)])
    s[1]["norm_radius"] = rng.choice([15.0, 15.0, 9.5])   # 9.5: OL_STATUS_ZERNIKE_RANGE
    s[2]["radius"] = rng.choice([-1, 1]) * rng.uniform(60.0, 400.0) if seed % 5 else np.inf
    s[2]["conic"] = rng.choice([0.0, rng.uniform(-1.2, 0.5)])
    for i in (1, 2):
        s[i]["coating_kind"] = rng.choice([S.COAT_FRESNEL, S.COAT_FRESNEL, S.COAT_NONE,
                                           S.COAT_SIMPLE])
        if s[i]["coating_kind"] == S.COAT_SIMPLE:
            s[i]["coat"][:2] = (rng.uniform(0.6, 1.0), rng.uniform(0.0, 0.4))
        if rng.random() < 0.5:
            s[i]["aperture_kind"] = S.AP_RADIAL
            s[i]["aperture"][:2] = (0.0, rng.uniform(6.0, 11.0))   # (r_min, r_max)
        if rng.random() < 0.4:
            s[i]["origin"][:2] += rng.uniform(-0.5, 0.5, 2)
    if seed % 4 == 3:   # a tilted last surface
        a = rng.uniform(-0.05, 0.05)
        s[3]["rot"] = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)],
                                [0, np.sin(a), np.cos(a)]]).reshape(-1)
    if seed % 6 == 5:   # a Fresnel mirror: the second surface reflects into the glass again
        s[2]["interaction"] = S.INTERACT_REFLECT
        s[2]["coating_kind"] = S.COAT_FRESNEL
    if seed % 7 == 6:   # dense glass, steep surface: total internal reflection for the rim
        table.optics["n2"][1, 0] = table.optics["n1"][2, 0] = 1.9
        s[2]["radius"] = -14.0
    table.__dict__.pop("_ref_newton", None)
    return table


def _launch(eng, px, py, wl, rays_per_thread, **kw):
    assert eng.lib.ol_set_tuning(_capi.TUNE_RAYS_PER_THREAD, rays_per_thread) == 0
    try:
        n = px.numel()
        before = eng.lib.ol_hostmath_pair_launches() if hasattr(eng.lib, "ol_hostmath_pair_launches") \
            else 0
        prt = torch.full((9, n), float("nan"), dtype=px.dtype, device=px.device)
        res = eng.trace_generate(px, py, wl, record=True, prt=prt, defer_status=True, **kw)
        upd = None if res.updated_intensity is None else res.updated_intensity.clone()
        pairs = (eng.lib.ol_hostmath_pair_launches() - before) \
            if hasattr(eng.lib, "ol_hostmath_pair_launches") else None
        # (defer_status: the engine raises nothing; the word the kernel OR-ed its bits into)
        return res.record[:, :, :n].clone(), prt, upd, int(eng._status.item()), pairs
    finally:
        eng.lib.ol_set_tuning(_capi.TUNE_RAYS_PER_THREAD, 0)


def _bits(t):
    return t.contiguous().view(torch.int32)


def assert_same_bits(one, pair, what):
    for name, a, b in (("record", one[0], pair[0]), ("prt", one[1], pair[1]),
                       ("updated intensity", one[2], pair[2])):
        if a is None:
            assert b is None
            continue
        if not torch.equal(_bits(a), _bits(b)):
            d = (a.double() - b.double()).abs().nan_to_num()
            raise AssertionError(f"{what}: {name} differs, max |d| = {float(d.max()):.3g}, "
                                 f"NaN {int(torch.isnan(a).sum())} / {int(torch.isnan(b).sum())}")
    assert one[3] == pair[3], f"{what}: status {one[3]:#x} != {pair[3]:#x}"


def _pupil(n, rng, reach=1.0):
    r, th = np.sqrt(rng.random(n)) * reach, 2 * np.pi * rng.random(n)
    px = torch.tensor(r * np.cos(th), dtype=torch.float32)
    py = torch.tensor(r * np.sin(th), dtype=torch.float32)
    k = min(n, 4)
    px[:k] = 0.0
    py[:k] = torch.tensor([0.0, 1e-6, 1.0, -1.0])[:k]   # chief ray, a near-vertex ray, the rim
    return px, py


@pytest.fixture(scope="module")
def engine_class():
    return hm.make_engine_class()


@pytest.mark.parametrize("seed", range(36))
def test_pair_form_equals_the_one_ray_form_on_random_systems(engine_class, seed):
    table = random_c5_table(seed)
    rng = np.random.default_rng(seed)
    eng = engine_class(table)
    try:
        n = int(rng.choice([2, 254, 1000, 4098]))
        px, py = _pupil(n, rng, reach=1.0 if seed % 3 else 1.08)
        state = POLARISED if seed % 2 else STATE
        kw = dict(field=(float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))),
                  update_intensity=state if seed % 3 != 1 else None)
        one = _launch(eng, px, py, 0, 1, **kw)
        pair = _launch(eng, px, py, 0, 3, **kw)
        assert one[4] == 0 and pair[4] == 1      # (which form ran)
        assert_same_bits(one, pair, f"seed {seed}")
        if float(table.surfaces[1]["norm_radius"]) < 10.0 and n >= 254:
            assert one[3] & S.STATUS_ZERNIKE_RANGE     # (the status comparison is not vacuous)
        if seed % 7 == 6 and n >= 254:
            assert torch.isnan(one[0][-1, 3]).any()   # the TIR case really loses rays
    finally:
        eng.close()


def field_plane_cases(make_engine, device="cpu"):
    """Per-ray field planes, vignetting planes and an apodized pupil through both forms."""
    table = copy.deepcopy(load_system("zernike_fresnel_fringe"))
    table.raygen["apod_kind"], table.raygen["apod_a"] = 1.0, 0.8   # gaussian apodization
    rng = np.random.default_rng(5)
    n = 3000
    px, py = (t.to(device) for t in _pupil(n, rng))
    hx = torch.tensor(rng.uniform(-1, 1, n), dtype=torch.float32, device=device)
    hy = torch.tensor(rng.uniform(-1, 1, n), dtype=torch.float32, device=device)
    vx = torch.tensor(rng.uniform(0.7, 1.0, n), dtype=torch.float32, device=device)
    vy = torch.tensor(rng.uniform(0.7, 1.0, n), dtype=torch.float32, device=device)
    eng = make_engine(table)
    try:
        for kw in (dict(field=(hx, hy), vig=(vx, vy), update_intensity=POLARISED),
                   dict(field=(hx, hy), vig=None, update_intensity=None),
                   dict(field=(0.3, -0.8), vig=(0.9, 0.8), update_intensity=STATE)):
            one = _launch(eng, px, py, 0, 1, **kw)
            pair = _launch(eng, px, py, 0, 3, **kw)
            assert pair[4] in (1, None)
            assert_same_bits(one, pair, str(sorted(kw)))
            # (the two rays of a lane really have their own field points)
            assert not torch.equal(one[0][-1, 0, 0::2], one[0][-1, 0, 1::2])
    finally:
        eng.close()


def test_pair_form_with_field_planes_vignetting_planes_and_an_apodized_pupil(engine_class):
    field_plane_cases(engine_class)


def test_launches_that_stay_on_the_one_ray_form(engine_class, monkeypatch):
    """An odd number of rays (the PRT planes are written with 8-byte lane accesses), a Zernike
    surface in the level form, a polarizer coating: the launcher keeps them off the pair."""
    rng = np.random.default_rng(9)
    table = load_system("zernike_fresnel_fringe")
    eng = engine_class(table)
    try:
        px, py = _pupil(1001, rng)
        assert _launch(eng, px, py, 0, 3, field=(0.0, 0.7))[4] == 0
        px, py = _pupil(1000, rng)
        assert _launch(eng, px, py, 0, 3, field=(0.0, 0.7))[4] == 1
    finally:
        eng.close()
    pol = copy.deepcopy(table)
    pol.surfaces[2]["coating_kind"] = S.COAT_POLARIZER
    pol.surfaces[2]["coat"][0] = len(pol.coeffs)     # (offset of the axis block)
    pol.coeffs = np.concatenate([pol.coeffs, [1.0, 0.0, 0.0]])
    eng = engine_class(pol)
    try:
        assert _launch(eng, px, py, 0, 3, field=(0.0, 0.7))[4] == 0
    finally:
        eng.close()
    monkeypatch.setenv("OPTILAND_HIP_ZERNIKE_MONO", "0")   # read at ol_system_create
    eng = engine_class(table)
    try:
        assert _launch(eng, px, py, 0, 3, field=(0.0, 0.7))[4] == 0
    finally:
        eng.close()
