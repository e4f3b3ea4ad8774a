"""Fused generate -> trace -> reduce spot kernel (`ol_trace_spot`, SURVEY.md 8 f1+f2)
against the CPU oracle (ray generation + trace + numpy reductions) and against the
un-fused HIP path (ol_generate_rays -> ol_trace -> ol_spot_moments / ol_spot_max_r2).

Tolerances: BASELINE.json's bar for the hit coordinates is fp32 1e-4 / fp64 1e-6 of
the position scale; the tests hold the hits AND the per-ray means / rms / max radius
derived from the moments to 1e-5 (fp32) and 1e-10 (fp64) of that scale.
"""

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

SYSTEMS = [("double_gauss", 0.5876, (0.0, 0.7)), ("cooke_generic", 0.55, (0.0, 1.0)),
           ("rc_asphere", 0.55, (0.0, 0.5))]


@pytest.fixture(scope="module", params=SYSTEMS, ids=[s[0] for s in SYSTEMS])
def system(request):
    from optiland_amd import load_system
    from optiland_amd.engine import HipSystem
    name, wavelength, field = request.param
    table = load_system(name)
    hip = HipSystem(table, DEV)
    yield hip, table, table.wavelength_index(wavelength), field
    hip.close()


def _pupil(n, seed, dtype):
    rng = np.random.default_rng(seed)
    r = np.sqrt(rng.random(n))
    th = 2 * np.pi * rng.random(n)
    px, py = r * np.cos(th), r * np.sin(th)
    return (torch.as_tensor(px, dtype=dtype, device=DEV), torch.as_tensor(py, dtype=dtype, device=DEV))


def _oracle_spot(table, wl, hx, hy, px, py, vx, vy, center):
    from oracle import oracle
    f = lambda t: t.double().cpu().numpy()  # noqa: E731
    g = oracle.generate_rays(table.raygen, f(hx), f(hy), f(px), f(py),
                             None if vx is None else f(vx), None if vy is None else f(vy))
    g["opd"] = np.zeros_like(g["x"])
    out = oracle.trace(table, g, wl, record=False)
    x, y, i = out["x"], out["y"], out["i"]
    m = i > 0
    dx, dy = x[m] - center[0], y[m] - center[1]
    r2 = dx * dx + dy * dy
    r2 = r2[~np.isnan(r2)]
    mom = np.array([m.sum(), dx.sum(), dy.sum(), (dx * dx).sum(), (dy * dy).sum(), i[m].sum(),
                    r2.max() if r2.size else 0.0])
    return mom, (x, y, i)


def _scale(table, wx, wy):
    """Position scale of the trace (tests/_util.py: all positions share one scale):
    the largest coordinate the rays take anywhere in the system."""
    o = np.abs(np.asarray(table.surfaces["origin"], dtype=np.float64))
    z = o[np.isfinite(o)].max()
    return max(1.0, float(z), table.raygen["EPD"] / 2, float(np.nanmax(np.abs(wy))),
               float(np.nanmax(np.abs(wx))))


def _check_moments(got, want, scale, tol):
    """Counts exact; means / rms / max radius to `tol` of the position scale."""
    assert got[0] == want[0]
    n = max(want[0], 1.0)
    np.testing.assert_allclose(got[1:3] / n, want[1:3] / n, atol=tol * scale)
    np.testing.assert_allclose(np.sqrt(got[3:5] / n), np.sqrt(want[3:5] / n), atol=tol * scale)
    np.testing.assert_allclose(got[5], want[5], rtol=max(tol, 1e-12))
    np.testing.assert_allclose(np.sqrt(got[6]), np.sqrt(want[6]), atol=tol * scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("n", [1, 3, 1025, 100_003])
def test_fused_spot_matches_oracle_uniform_field(system, dtype, n):
    hip, table, wl, field = system
    px, py = _pupil(n, 11 + n, dtype)
    hx = torch.full((n,), field[0], dtype=dtype, device=DEV)
    hy = torch.full((n,), field[1], dtype=dtype, device=DEV)
    want0, _ = _oracle_spot(table, wl, hx, hy, px, py, None, None, (0.0, 0.0))
    center = (want0[1] / max(want0[0], 1), want0[2] / max(want0[0], 1))  # about the centroid
    want, (wx, wy, wi) = _oracle_spot(table, wl, hx, hy, px, py, None, None, center)
    hits = [torch.empty(n, dtype=dtype, device=DEV) for _ in range(3)]
    got = hip.trace_spot(px, py, wl, field=field, center=center, hits=hits).cpu().numpy()
    scale = _scale(table, wx, wy)
    tol = 1e-5 if dtype == torch.float32 else 1e-10
    _check_moments(got, want, scale, tol)
    for h, w in zip(hits, (wx, wy, wi)):
        hv = h.double().cpu().numpy()
        assert np.array_equal(np.isnan(hv), np.isnan(w))
        np.testing.assert_allclose(hv, w, atol=tol * scale, rtol=tol, equal_nan=True)
    # without the hit planes: same moments up to the order of the fp64 atomics
    again = hip.trace_spot(px, py, wl, field=field, center=center).cpu().numpy()
    np.testing.assert_allclose(again, got, rtol=1e-11, atol=1e-13 * scale * n)
    assert again[0] == got[0] and again[6] == got[6]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_fused_spot_per_ray_field_and_vignetting_planes(system, dtype):
    hip, table, wl, field = system
    n = 20_011
    px, py = _pupil(n, 5, dtype)
    rng = np.random.default_rng(2)
    hx = torch.as_tensor(rng.uniform(-0.3, 0.3, n), dtype=dtype, device=DEV)
    hy = torch.as_tensor(rng.uniform(0.0, field[1], n), dtype=dtype, device=DEV)
    vx = torch.as_tensor(rng.uniform(0.7, 1.0, n), dtype=dtype, device=DEV)
    vy = torch.as_tensor(rng.uniform(0.7, 1.0, n), dtype=dtype, device=DEV)
    want, (wx, wy, wi) = _oracle_spot(table, wl, hx, hy, px, py, vx, vy, (0.0, 0.0))
    got = hip.trace_spot(px, py, wl, hx=hx, hy=hy, vx=vx, vy=vy).cpu().numpy()
    _check_moments(got, want, _scale(table, wx, wy), 1e-5 if dtype == torch.float32 else 1e-10)
    # launch-uniform vignetting scalars == constant planes
    a = hip.trace_spot(px, py, wl, field=field, vig=(0.9, 0.8)).cpu().numpy()
    c = torch.full((n,), 1.0, dtype=dtype, device=DEV)
    b = hip.trace_spot(px, py, wl, field=field, vx=c * 0.9, vy=c * 0.8).cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=1e-6 if dtype == torch.float32 else 1e-12)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
def test_fused_spot_equals_unfused_hip_pipeline(system, dtype):
    """Same device arithmetic either way: hits agree to rounding of the FMA schedule,
    reductions to the fp64 summation order."""
    hip, table, wl, field = system
    n = 300_007
    px, py = _pupil(n, 9, dtype)
    hx = torch.full((n,), field[0], dtype=dtype, device=DEV)
    hy = torch.full((n,), field[1], dtype=dtype, device=DEV)
    planes = [p.contiguous().clone() for p in hip.generate_rays(hx, hy, px, py)]
    planes.append(torch.zeros(n, dtype=dtype, device=DEV))
    hip.trace(planes, wl, record=False)
    x, y, i = planes[0], planes[1], planes[6]
    hits = [torch.empty(n, dtype=dtype, device=DEV) for _ in range(3)]
    got = hip.trace_spot(px, py, wl, field=field, hits=hits).cpu().numpy()
    eps = 1e-6 if dtype == torch.float32 else 1e-14
    scale = float(y.abs().nan_to_num().max()) + 1.0
    for h, w in zip(hits, (x, y, i)):
        assert torch.equal(torch.isnan(h), torch.isnan(w))
        assert float((h - w).abs().nan_to_num().max()) <= 50 * eps * scale
    mom = hip.spot_moments(x, y, i).cpu().numpy()
    assert got[0] == mom[0]
    np.testing.assert_allclose(got[1:5], mom[1:5], rtol=1e-5 if dtype == torch.float32 else 1e-11)
    r2 = float(hip.spot_max_r2(x, y, i, 0.0, 0.0).item())
    np.testing.assert_allclose(got[6], r2, rtol=1e-5 if dtype == torch.float32 else 1e-11)


def test_fused_spot_accumulates_and_shards(system):
    """out7 is accumulated: two half launches into one buffer == one full launch
    (the multi-GPU reduction is an all-reduce of exactly these seven doubles)."""
    hip, table, wl, field = system
    dtype = torch.float64
    n = 50_001
    px, py = _pupil(n, 4, dtype)
    full = hip.trace_spot(px, py, wl, field=field, center=(0.0, 1.0)).cpu().numpy()
    out = torch.zeros(7, dtype=torch.float64, device=DEV)
    h = n // 2 + 1
    hip.trace_spot(px[:h].clone(), py[:h].clone(), wl, field=field, center=(0.0, 1.0), out=out)
    hip.trace_spot(px[h:].clone(), py[h:].clone(), wl, field=field, center=(0.0, 1.0), out=out)
    np.testing.assert_allclose(out.cpu().numpy(), full, rtol=1e-11)
    assert out.cpu().numpy()[6] == full[6]  # max is exact


def test_fused_spot_empty_and_all_clipped():
    from optiland_amd import load_system
    from optiland_amd.engine import HipSystem
    table = load_system("double_gauss")
    hip = HipSystem(table, DEV)
    try:
        z = torch.zeros(0, dtype=torch.float32, device=DEV)
        assert float(hip.trace_spot(z, z, 0, field=(0.0, 0.0)).abs().sum()) == 0.0
        # pupil points far outside the stop: every ray is clipped (i = 0) or misses
        n = 777
        px = torch.full((n,), 5.0, dtype=torch.float32, device=DEV)
        out = hip.trace_spot(px, px, 0, field=(0.0, 0.0)).cpu().numpy()
        assert out[0] == 0.0 and out[6] == 0.0 and np.all(out[1:6] == 0.0)
    finally:
        hip.close()


def test_fused_spot_refuses_polarised_systems():
    from optiland_amd import load_system
    from optiland_amd.engine import HipSystem
    table = load_system("zernike_fresnel_fringe")
    hip = HipSystem(table, DEV)
    try:
        z = torch.zeros(8, dtype=torch.float32, device=DEV)
        with pytest.raises(ValueError, match="Polarization must be set"):
            hip.trace_spot(z, z, 0, field=(0.0, 0.0))
    finally:
        hip.close()


@pytest.mark.parametrize("dtype,n", [(torch.float32, 10_000_000), (torch.float64, 5_000_000)],
                         ids=["f32-1e7", "f64-5e6"])
def test_fused_spot_fullsize_against_planes(dtype, n):
    """BASELINE size: fused reduction == reductions of the traced planes."""
    from optiland_amd import load_system
    from optiland_amd.engine import HipSystem
    table = load_system("double_gauss")
    hip = HipSystem(table, DEV)
    try:
        g = torch.Generator(device=DEV).manual_seed(8)
        r = torch.rand(n, generator=g, device=DEV, dtype=torch.float64).sqrt()
        th = 2 * np.pi * torch.rand(n, generator=g, device=DEV, dtype=torch.float64)
        px, py = (r * th.cos()).to(dtype), (r * th.sin()).to(dtype)
        hx = torch.zeros(n, dtype=dtype, device=DEV)
        hy = torch.full((n,), 0.7, dtype=dtype, device=DEV)
        planes = [p.contiguous().clone() for p in hip.generate_rays(hx, hy, px, py)]
        planes.append(torch.zeros(n, dtype=dtype, device=DEV))
        hip.trace(planes, 0, record=False)
        x, y, i = planes[0], planes[1], planes[6]
        m = i > 0
        cx, cy = float(x[m].double().mean()), float(y[m].double().mean())
        got = hip.trace_spot(px, py, 0, field=(0.0, 0.7), center=(cx, cy)).cpu().numpy()
        dx, dy = x[m].double() - cx, y[m].double() - cy
        assert got[0] == float(m.sum())
        tol = 2e-5 if dtype == torch.float32 else 1e-10
        rms_want = float(torch.sqrt((dx * dx + dy * dy).mean()))
        rms_got = float(np.sqrt((got[3] + got[4]) / got[0]))
        np.testing.assert_allclose(rms_got, rms_want, rtol=tol * 100)
        np.testing.assert_allclose(np.sqrt(got[6]), float(torch.sqrt((dx * dx + dy * dy).max())),
                                   rtol=tol * 100)
        assert abs(got[1] / got[0]) < tol * 20 and abs(got[2] / got[0]) < tol * 20
    finally:
        hip.close()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("mode", ["record", "last"])
@pytest.mark.parametrize("n", [1, 255, 100_003])
def test_trace_epilogue_moments_equal_plane_reductions(system, dtype, mode, n):
    """`ol_trace_ex` spot epilogue (slotted atomics inside the trace kernel) == the
    separate reductions of the traced image-plane planes; the traced state itself is
    unchanged by asking for it (ragged sizes: lanes past the end stay for the barrier)."""
    hip, table, wl, field = system
    px, py = _pupil(n, 17, dtype)
    planes = [p.contiguous().clone() for p in hip.generate_rays(field[0], field[1], px, py)]
    planes.append(torch.zeros(n, dtype=dtype, device=DEV))
    ref_in = [p.clone() for p in planes]
    slots = hip.alloc_spot_slots()
    cx, cy = 0.0, 0.25
    res = hip.trace(planes, wl, record=(mode == "record"), spot=(slots, cx, cy))
    got = hip.reduce_spot_slots(slots).cpu().numpy()
    plain = hip.trace(ref_in, wl, record=(mode == "record"))
    if mode == "record":
        assert torch.equal(res.record[:, :, :n].nan_to_num(), plain.record[:, :, :n].nan_to_num())
        x, y, i = (plain.row(plain.last, k) for k in (0, 1, 6))
    else:
        for a, b in zip(planes, ref_in):
            assert torch.equal(a.nan_to_num(), b.nan_to_num())
        x, y, i = ref_in[0], ref_in[1], ref_in[6]
    m = i > 0
    dx, dy = x[m].double() - cx, y[m].double() - cy
    ok = ~(torch.isnan(dx) | torch.isnan(dy))
    want = np.array([float(m.sum()), float(dx.sum()), float(dy.sum()), float((dx * dx).sum()),
                     float((dy * dy).sum()), float(i[m].double().sum()),
                     float((dx * dx + dy * dy)[ok].max()) if bool(ok.any()) else 0.0])
    assert got[0] == want[0]
    np.testing.assert_allclose(got[1:6], want[1:6], rtol=1e-10, atol=1e-9)
    assert got[6] == want[6]


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("case,field", [("apodized_tukey_trace", (0.0, 0.7)),
                                        ("apodized_hann_trace", (0.0, 0.0)),
                                        ("finite_object_height_generic", (1 / 3, 1.0)),
                                        ("finite_object_height_telecentric_generic", (0.1, 0.6)),
                                        ("finite_angle_generic", (-0.2, 0.9)),
                                        ("sample_UVProjectionLens", (0.0, 1.0))])
def test_fused_spot_field_kinds_and_apodization(case, field, dtype):
    """The fused kernel shares the ray generator: object-height / telecentric / finite
    angle fields and pupil apodization (initial intensities feed the i > 0 mask and
    sum i) against the oracle pipeline."""
    from optiland_amd.engine import HipSystem
    from tests._util import load_case
    table, _ = load_case(case)
    hip = HipSystem(table, DEV)
    try:
        n = 5003
        px, py = _pupil(n, 23, dtype)
        hx = torch.full((n,), field[0], dtype=dtype, device=DEV)
        hy = torch.full((n,), field[1], dtype=dtype, device=DEV)
        vig = (0.95, 0.9)
        vx = torch.full((n,), vig[0], dtype=dtype, device=DEV)
        vy = torch.full((n,), vig[1], dtype=dtype, device=DEV)
        want, (wx, wy, wi) = _oracle_spot(table, 0, hx, hy, px, py, vx, vy, (0.0, 0.0))
        got = hip.trace_spot(px, py, 0, field=field, vig=vig).cpu().numpy()
        # a pupil point within rounding of an apodization edge (r = R) may fall on either
        # side in fp32: allow the count to differ by what sits inside 1e-6 of an edge
        if dtype == torch.float64:
            _check_moments(got, want, _scale(table, wx, wy), 1e-10)
        else:
            assert abs(got[0] - want[0]) <= 2
            np.testing.assert_allclose(got[5], want[5], rtol=1e-4)
            nrm = max(want[0], 1.0)
            sc = _scale(table, wx, wy)
            np.testing.assert_allclose(got[1:3] / nrm, want[1:3] / nrm, atol=2e-5 * sc)
    finally:
        hip.close()


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("n", [1, 37, 1025, 200_003])
def test_spot_batch_equals_single_launches(system, dtype, n):
    """`ol_trace_spot_batch` (ABI 10: blockIdx.y = (field, wavelength) cell) against one
    `ol_trace_spot` per cell: the hits bit for bit (the same per-ray arithmetic), the counts
    exactly, the sums to the rounding of a different summation order; more cells than one
    launch takes (32) and every wavelength row of the table."""
    hip, table, wl0, field = system
    px, py = _pupil(n, 7 + n, dtype)
    rng = np.random.default_rng(5)
    n_wl = int(table.optics.shape[1]) if table.optics.ndim > 1 else 1
    cells = []
    for k in range(35):
        hx, hy = (0.0, 0.0) if k == 0 else tuple(rng.uniform(-1, 1, 2) * field[1])
        cells.append((hx, hy, 1.0 - 0.1 * (k % 3), 1.0 - 0.05 * (k % 2), rng.normal(), rng.normal(),
                      k % n_wl))
    mom, hits = hip.trace_spot_batch(px, py, cells, hits=True)
    assert mom.shape == (35, 8) and hits.shape[:2] == (35, 3)
    mom = mom.cpu().numpy()
    for k, (hx, hy, vx, vy, cx, cy, wl) in enumerate(cells):
        one = [torch.empty(n, dtype=dtype, device=DEV) for _ in range(3)]
        want = hip.trace_spot(px, py, wl, field=(hx, hy), vig=(vx, vy), center=(cx, cy),
                              hits=one).cpu().numpy()
        for q in range(3):
            assert torch.equal(hits[k, q, :n].nan_to_num(), one[q].nan_to_num()), (k, q)
        assert mom[k, 0] == want[0]
        np.testing.assert_allclose(mom[k, 1:6], want[1:6], rtol=1e-9 if dtype == torch.float64
                                   else 1e-6, atol=1e-9 * max(n, 1))
        assert mom[k, 6] == want[6]                   # a maximum: no rounding
    # accumulation into a caller's block, no hits, empty cell list
    again, none = hip.trace_spot_batch(px, py, cells[:3], out=torch.zeros((3, 8), dtype=torch.float64,
                                                                            device=DEV))
    assert none is None
    np.testing.assert_allclose(again.cpu().numpy()[:, :6], mom[:3, :6], rtol=1e-9 if
                               dtype == torch.float64 else 1e-6, atol=1e-9 * max(n, 1))
    empty, _ = hip.trace_spot_batch(px, py, [])
    assert empty.shape == (0, 8)
