"""Edge cases on hand-built systems, HIP vs the pinned CPU oracle (fp64 tight,
fp32 contract): degenerate quadratics, rays that start on the surface, grazing and
backward rays, misses, hemispheric limits, TIR, clipping at the exact rim, NaN/inf
inputs, write-only PRT, polarised partial traces."""

import math

import numpy as np
import pytest
import torch

from optiland_amd import system as S
from optiland_amd.system import SystemTable
from tests._util import PLANES, assert_close_planes, load_case

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def table_of(surfs, n2s):
    """surfs: list of dict(kind, radius, conic, z, interaction, aperture...)."""
    n = len(surfs) + 1
    desc = np.zeros(n, dtype=S.SURFACE_DESC_DTYPE)
    optics = np.zeros((n, 1), dtype=S.SURFACE_OPTICS_DTYPE)
    desc["rot"] = np.eye(3).reshape(-1)
    desc["norm_radius"] = 1.0
    desc[0]["geom_kind"] = S.GEOM_PLANE
    desc[0]["interaction"] = S.INTERACT_RECORD_ONLY
    desc[0]["origin"] = (0, 0, -math.inf)
    optics[0, 0] = (1.0, 1.0, 0.0)
    n_prev = 1.0
    for i, (sf, n2) in enumerate(zip(surfs, n2s), start=1):
        d = desc[i]
        d["geom_kind"] = sf.get("kind", S.GEOM_STANDARD)
        d["interaction"] = sf.get("interaction", S.INTERACT_REFRACT)
        d["radius"] = sf.get("radius", math.inf)
        d["conic"] = sf.get("conic", 0.0)
        d["origin"] = (0.0, 0.0, sf.get("z", 0.0))
        d["aperture_kind"] = sf.get("ap_kind", S.AP_NONE)
        d["aperture"] = sf.get("ap", (0, 0, 0, 0))
        optics[i, 0] = (n_prev, n2, sf.get("absorb", 0.0))
        n_prev = n2 if d["interaction"] == S.INTERACT_REFRACT else n_prev
    return SystemTable(surfaces=desc, coeffs=np.zeros(0), optics=optics,
                       wavelengths=np.array([0.55]))


def run_both(table, rays, dtype, polarized=False):
    from optiland_amd.engine import HipSystem
    from oracle import oracle
    hip = HipSystem(table, DEV)
    n = len(rays["x"])
    planes = [torch.tensor(np.asarray(rays[k], dtype=np.float64), dtype=dtype, device=DEV)
              for k in PLANES[:7]]
    planes.append(torch.zeros(n, dtype=dtype, device=DEV))
    prt = torch.empty((9, n), dtype=dtype, device=DEV) if polarized else None
    res = hip.trace(planes, 0, record=True, prt=prt, prt_identity=polarized)
    got = res.record[:, :, :n].double().cpu().numpy()
    want = oracle.trace(table, rays, 0, record=True, polarized=polarized)
    hip.close()
    return got, want, prt


def bundle(n, seed, z0=-5.0, spread=0.3, radius=3.0):
    rng = np.random.default_rng(seed)
    x, y = rng.uniform(-radius, radius, n), rng.uniform(-radius, radius, n)
    L, M = rng.uniform(-spread, spread, n), rng.uniform(-spread, spread, n)
    N = np.sqrt(1 - L * L - M * M)
    return dict(x=x, y=y, z=np.full(n, z0), L=L, M=M, N=N, i=np.ones(n))


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4)],
                         ids=["f64", "f32"])
@pytest.mark.parametrize("name,surfs,n2s", [
    ("paraboloid-A=0", [dict(radius=20.0, conic=-1.0)], [1.5]),
    ("hyperboloid", [dict(radius=-15.0, conic=-2.5)], [1.7]),
    ("oblate-ellipsoid", [dict(radius=12.0, conic=0.8)], [1.5]),
    ("near-flat-R=1e7", [dict(radius=1e7)], [1.5]),
    ("steep-sphere-misses", [dict(radius=2.5)], [1.5]),
    ("mirror-then-backward", [dict(radius=-30.0, interaction=S.INTERACT_REFLECT),
                              dict(radius=15.0, z=-8.0)], [1.0, 1.4]),
    ("glass-to-air-TIR", [dict(radius=math.inf), dict(radius=-4.0, z=3.0)], [1.8, 1.0]),
    ("absorbing-slab", [dict(radius=math.inf), dict(radius=math.inf, z=10.0, absorb=0.03)],
     [1.5, 1.0]),
    ("rim-aperture", [dict(radius=40.0, ap_kind=S.AP_RADIAL, ap=(0.5, 2.0, 0, 0))], [1.5]),
])
def test_conic_edge_systems(name, surfs, n2s, dtype, tol):
    table = table_of(surfs, n2s)
    rays = bundle(4096, 7)
    if name == "paraboloid-A=0":      # axis-parallel rays: quadratic degenerates (a == 0)
        rays["L"][:] = 0.0
        rays["M"][:] = 0.0
        rays["N"][:] = 1.0
    if name == "rim-aperture":        # rays landing exactly on r_max and r_min
        # (the two off-rim probes sit 1e-7 away in fp64 and 1e-3 away in fp32, whose
        # resolution at 2.0 is 2.4e-7: closer than that the clip decision is rounding)
        d = 1e-7 if dtype == torch.float64 else 1e-3
        rays["x"][:8] = [2.0, 0.5, 0.0, 0.0, 2.0 + d, 0.5 - d, 1.0, 2.0 - d]
        rays["y"][:8] = 0.0
        rays["L"][:8] = 0.0
        rays["M"][:8] = 0.0
        rays["N"][:8] = 1.0
    got, want, _ = run_both(table, rays, dtype)
    if name == "near-flat-R=1e7" and dtype == torch.float64:
        # here the ORACLE is the inaccurate side: the reference's (-b +- sqrt(d)) / 2a
        # (standard.py:128-146) cancels catastrophically for near-flat surfaces --
        # 1.7e-9 mm off a long-double evaluation on this very bundle, against 8e-13
        # for the kernel's t = C / q form (see test_stable_root_beats_reference_formula)
        tol = 2e-8
    assert_close_planes(got, want["record"], tol, tol, f"{name}:{dtype}")
    if dtype == torch.float64:
        assert np.array_equal(got[:, 6] == 0, want["record"][:, 6] == 0)


def test_stable_root_beats_reference_formula():
    """fp64 kernel vs a long-double evaluation of the conic intersection on the
    near-flat sphere: the kernel must be CLOSER to it than the reference formula is."""
    table = table_of([dict(radius=1e7)], [1.5])
    rays = bundle(4096, 7)
    got, want, _ = run_both(table, rays, torch.float64)
    ld = np.longdouble
    x, y, z, L, M, N = (rays[k].astype(ld) for k in ("x", "y", "z", "L", "M", "N"))
    R = ld(1e7)
    a = L * L + M * M + N * N
    b = 2 * L * x + 2 * M * y - 2 * N * R + 2 * N * z
    c = -2 * R * z + x * x + y * y + z * z
    t = (-b - np.sqrt(b * b - 4 * a * c)) / (2 * a)   # near root for R > 0, N > 0
    t_alt = (2 * c) / (-b + np.sqrt(b * b - 4 * a * c))  # same root, stable form
    zhit = np.asarray(z + t_alt * N, dtype=np.float64)
    err_kernel = np.max(np.abs(got[1, 2] - zhit))
    err_oracle = np.max(np.abs(want["record"][1, 2] - zhit))
    assert err_kernel < 1e-11, err_kernel
    assert err_oracle > 20 * err_kernel, (err_oracle, err_kernel)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_special_rays(dtype):
    """Rays starting on the vertex plane, perpendicular to the axis (N = 0), NaN and
    inf inputs, zero intensity: NaN/inf masks must match the oracle exactly."""
    table = table_of([dict(radius=25.0), dict(radius=math.inf, z=5.0)], [1.5, 1.0])
    x = np.array([0.0, 1.0, 0.0, 2.0, np.nan, 1.0, 0.5, 0.0])
    y = np.array([0.0, 0.0, 1.0, 0.0, 0.0, np.inf, 0.5, 0.0])
    z = np.array([0.0, -1.0, -2.0, -3.0, -1.0, -1.0, -1e-9, -1.0])
    L = np.array([0.0, 1.0, 0.0, 0.6, 0.0, 0.0, 0.0, 0.0])
    M = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0])
    N = np.array([1.0, 0.0, -1.0, 0.8, 1.0, 1.0, 1.0, 1.0])
    i = np.array([1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0])
    rays = dict(x=x, y=y, z=z, L=L, M=M, N=N, i=i)
    got, want, _ = run_both(table, rays, dtype)
    w = want["record"]
    assert np.array_equal(np.isnan(got), np.isnan(w))
    assert np.array_equal(np.isinf(got), np.isinf(w))
    fin = np.isfinite(w)
    tol = 1e-10 if dtype == torch.float64 else 1e-4
    np.testing.assert_allclose(got[fin], w[fin], rtol=tol, atol=tol * 30)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
def test_write_only_prt_equals_identity_start(dtype):
    from optiland_amd.engine import HipSystem
    from optiland_amd.rays import new_prt
    table, data = load_case("zernike_fresnel_fringe")
    hip = HipSystem(table, DEV)
    r = data["rays_in"]
    n = r.shape[1]
    mk = lambda: [torch.tensor(r[k], dtype=dtype, device=DEV) for k in range(7)] + \
        [torch.zeros(n, dtype=dtype, device=DEV)]  # noqa: E731
    a = new_prt(n, dtype, DEV, False)
    hip.trace(mk(), 0, record=False, prt=a)
    b = torch.full((9, n), float("nan"), dtype=dtype, device=DEV)  # garbage in
    hip.trace(mk(), 0, record=False, prt=b, prt_identity=True)
    assert torch.equal(a, b)
    hip.close()


@pytest.mark.parametrize("case", ["zernike_fresnel_fringe", "polarizer_retarder"])
def test_polarised_partial_traces_continue_the_prt(case):
    """[0, k] then [k+1, S] with the PRT carried over == one full trace."""
    from optiland_amd.engine import HipSystem
    from optiland_amd.rays import new_prt
    table, data = load_case(case)
    hip = HipSystem(table, DEV)
    dtype = torch.float64
    r = data["rays_in"]
    n = r.shape[1]
    mk = lambda: [torch.tensor(r[k], dtype=dtype, device=DEV) for k in range(7)] + \
        [torch.zeros(n, dtype=dtype, device=DEV)]  # noqa: E731
    cplx = table.needs_complex_prt
    full = new_prt(n, dtype, DEV, cplx)
    rays_full = mk()
    hip.trace(rays_full, 0, record=False, prt=full)
    S_ = table.num_surfaces - 1
    k = max(1, S_ // 2)
    part = new_prt(n, dtype, DEV, cplx)
    rays = mk()
    hip.trace(rays, 0, record=False, prt=part, first=0, last=k)
    hip.trace(rays, 0, record=False, prt=part, first=k + 1, last=S_)
    np.testing.assert_allclose(part.cpu().numpy(), full.cpu().numpy(), rtol=1e-9, atol=1e-12)
    for a, b in zip(rays, rays_full):
        np.testing.assert_allclose(a.cpu().numpy(), b.cpu().numpy(), rtol=1e-9, atol=1e-9)
    hip.close()


@pytest.mark.parametrize("dtype,tol", [(torch.float64, 1e-10), (torch.float32, 1e-4)],
                         ids=["f64", "f32"])
def test_record_only_surface_in_the_middle(dtype, tol):
    """A RECORD_ONLY (dummy) surface between tilted traced surfaces: it must record the
    state without moving it, and the next surface must start from the right frame."""
    table, data = load_case("tilted_fold")
    import copy
    t2 = copy.deepcopy(table)
    # insert a copy of surface 2 as a record-only dummy after it (same frame)
    surf = np.insert(t2.surfaces, 3, t2.surfaces[2])
    surf[3]["interaction"] = S.INTERACT_RECORD_ONLY
    surf[3]["origin"] = (0.3, -0.4, 11.0)          # its own frame is irrelevant
    opt = np.insert(t2.optics, 3, t2.optics[2], axis=0)
    t2.surfaces, t2.optics = surf, opt
    rays = {k: data["rays_in"][j] for j, k in enumerate(PLANES[:7])}
    got, want, _ = run_both(t2, rays, dtype)
    w = want["record"]
    assert_close_planes(got, w, tol, tol, "dummy surface")
    # the dummy row repeats the previous row; later rows equal the original system's
    assert np.array_equal(np.nan_to_num(got[3]), np.nan_to_num(got[2]))
    assert_close_planes(np.delete(got, 3, axis=0), data["record"], max(tol, 1e-9), max(tol, 1e-9),
                        "rows after the dummy")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("n", [1, 63, 257, 1000, 4099])
def test_no_write_outside_the_callers_buffers(dtype, n):
    """Memory safety: sentinels around the ray planes, in the stride padding of every
    record plane, in a spare record row and around the PRT must survive the trace
    (vector path, ragged tail and scalar path all covered by the sizes)."""
    from optiland_amd.engine import HipSystem
    table, data = load_case("zernike_fresnel_fringe")
    hip = HipSystem(table, DEV)
    reps = -(-n // data["rays_in"].shape[1])
    src = np.tile(data["rays_in"], (1, reps))[:, :n]
    SENT = 12345.0
    pad = 8
    big = torch.full((8, n + 2 * pad), SENT, dtype=dtype, device=DEV)
    rays = [big[k, pad:pad + n] for k in range(8)]
    for k in range(7):
        rays[k].copy_(torch.tensor(src[k], dtype=dtype))
    rays[7].zero_()
    rows = table.num_surfaces
    stride = n + 64
    rec = torch.full((rows + 1, 8, stride), SENT, dtype=dtype, device=DEV)
    prt_big = torch.full((9, n), SENT, dtype=dtype, device=DEV)
    guard = torch.full((2, 64), SENT, dtype=dtype, device=DEV)  # neighbours of small allocs
    res = hip.trace(rays, 0, record=rec, prt=prt_big, prt_identity=True, write_rays=True)
    torch.cuda.synchronize()
    assert bool((big[:, :pad] == SENT).all()) and bool((big[:, pad + n:] == SENT).all())
    assert bool((rec[:rows, :, n:] == SENT).all()), "stride padding was written"
    assert bool((rec[rows] == SENT).all()), "spare record row was written"
    assert bool((guard == SENT).all())
    assert not bool((rec[:rows, :, :n] == SENT).any())
    assert not bool((prt_big == SENT).any())
    # final state written back == last recorded row
    for k in range(8):
        assert torch.equal(rays[k].nan_to_num(nan=-1.0), res.row(rows - 1, k).nan_to_num(nan=-1.0))
    hip.close()


def test_range_validation_happens_in_the_raygen_kernel():
    """Device-resident Hx/Hy/Px/Py are validated by the ray-generation kernel
    (OL_RAYGEN_CHECK_* -> status bits) with the reference's messages
    (real_ray_tracer.py:156-173); host scalars are validated on the host."""
    from optiland_amd import load_system, tracer as tr
    t = tr.HipRayTracer(load_system("double_gauss"), DEV, dtype=torch.float32)
    n = 1000
    ok = torch.linspace(-1, 1, n, device=DEV)
    bad = ok.clone()
    bad[n // 2] = 1.0001
    nan = ok.clone()
    nan[3] = float("nan")
    t.trace_generic(ok * 0.5, ok, ok * 0.1, ok * 0.1, 0.5876)  # passes
    with pytest.raises(ValueError, match="Normalized field coordinates"):
        t.trace_generic(bad, ok, ok, ok, 0.5876)
    with pytest.raises(ValueError, match="Normalized field coordinates"):
        t.trace_generic(ok, nan, ok, ok, 0.5876)
    with pytest.raises(ValueError, match="Normalized pupil coordinates"):
        t.trace_generic(ok, ok, ok, bad, 0.5876)
    with pytest.raises(ValueError, match="Normalized pupil coordinates"):
        t.trace_generic(0.0, 0.5, bad, ok, 0.5876)
    with pytest.raises(ValueError, match="Normalized field coordinates"):  # field first
        t.trace_generic(bad, ok, bad, ok, 0.5876)
    with pytest.raises(ValueError, match="Normalized field coordinates"):
        t.trace_generic(0.0, 1.5, ok, ok, 0.5876)
    with pytest.raises(ValueError, match="Normalized field coordinates"):
        t.trace(torch.tensor([0.0, 2.0], device=DEV), torch.tensor([0.0, 0.0], device=DEV),
                0.5876, 3, "hexapolar")
    # trace() does not validate the pupil (a distribution may exceed the unit disc)
    r = t.trace_generic(0.0, 0.0, ok * 0.2, ok * 0.2, 0.5876)
    assert len(r) == n
    t.engine.close()


def test_uniform_field_scalars_equal_field_planes():
    """Launch-uniform (Hx, Hy) scalars and constant per-ray planes give the same rays,
    with and without vignetting (trace and trace_generic)."""
    from optiland_amd import load_system, tracer as tr
    from tests._util import load_case
    for dtype in (torch.float32, torch.float64):
        table, _ = load_case("vignetted_trace")
        t = tr.HipRayTracer(table, DEV, dtype=dtype)
        n = 515
        g = torch.Generator(device=DEV).manual_seed(3)
        px = (torch.rand(n, generator=g, device=DEV, dtype=torch.float64) - 0.5).to(dtype)
        py = (torch.rand(n, generator=g, device=DEV, dtype=torch.float64) - 0.5).to(dtype)
        mf = table.raygen["max_field"]
        hy0 = float(np.asarray(table.fields)[-1][1] / mf)
        a = t.trace_generic(0.0, hy0, px, py, float(table.wavelengths[0]))
        ax = [v.clone() for v in (a.x, a.y, a.L, a.i, a.opd)]
        hx = torch.zeros(n, dtype=dtype, device=DEV)
        hy = torch.full((n,), hy0, dtype=dtype, device=DEV)
        b = t.trace_generic(hx, hy, px, py, float(table.wavelengths[0]))
        eps = 1e-6 if dtype == torch.float32 else 1e-14
        for u, v in zip(ax, (b.x, b.y, b.L, b.i, b.opd)):
            assert torch.equal(torch.isnan(u), torch.isnan(v))
            scale = float(v.abs().nan_to_num().max()) + 1.0
            assert float((u - v).abs().nan_to_num().max()) <= 20 * eps * scale
        t.engine.close()


def test_graphed_trace_replays_the_eager_kernels_bit_for_bit():
    """hipGraph capture of (status zero -> ol_generate_rays -> ol_trace): replays with
    new inputs equal eager traces bit for bit; range errors still surface."""
    from optiland_amd import load_system
    from optiland_amd.engine import HipSystem
    from optiland_amd.graph import GraphedTrace
    table = load_system("double_gauss")
    hip = HipSystem(table, DEV)
    try:
        for dtype, record_all in ((torch.float32, True), (torch.float64, True),
                                  (torch.float32, False)):
            n = 300
            g = GraphedTrace(hip, n, dtype, wavelength_index=1, record_all=record_all)
            gen = torch.Generator(device=DEV).manual_seed(5)
            for rep in range(3):
                c = [(torch.rand(n, generator=gen, device=DEV, dtype=torch.float64) * 1.2 - 0.6)
                     .to(dtype) for _ in range(4)]
                for dst, src in zip((g.hx, g.hy, g.px, g.py), c):
                    dst.copy_(src)
                res = g.replay()
                planes = hip.generate_rays(*c)
                eager = [p.contiguous().clone() for p in planes] + \
                    [torch.zeros(n, dtype=dtype, device=DEV)]
                er = hip.trace(eager, 1, record=record_all)
                if record_all:
                    assert torch.equal(res.record[:, :, :n].nan_to_num(), er.record[:, :, :n].nan_to_num())
                else:
                    for a, b in zip(g.rays, eager):
                        assert torch.equal(a.nan_to_num(), b.nan_to_num())
            g.px[7] = 1.5
            with pytest.raises(ValueError, match="Normalized pupil coordinates"):
                g.replay()
            g.px[7] = 0.0
            g.replay()  # status word is re-zeroed inside the graph
    finally:
        hip.close()


@pytest.mark.parametrize("case", ["tilted_fold", "double_gauss", "rc_asphere"])
def test_segments_around_a_placeholder_row(case):
    """What the SurfaceGroup seam does around a surface the fused path does not implement
    (integration._hip_surface_group_trace): its row is a never-traced placeholder, the
    runs [0, k-1] and [k+1, S] are separate launches and the caller moves the rays across
    surface k itself.  Here the 'foreign' surface is traced by a second handle holding the
    ORIGINAL table, so the chain must reproduce the one-launch trace: the run after the
    gap starts from the global frame, not from a frame relative to the placeholder."""
    import copy
    from optiland_amd.engine import HipSystem
    table, data = load_case(case)
    S_ = table.num_surfaces - 1
    k = max(2, S_ // 2)
    holed = copy.deepcopy(table)
    holed.surfaces[k] = np.zeros((), dtype=S.SURFACE_DESC_DTYPE)
    holed.surfaces[k]["rot"] = np.eye(3).reshape(-1)
    holed.surfaces[k]["interaction"] = S.INTERACT_RECORD_ONLY
    holed.optics[k, :] = (1.0, 1.0, 0.0)
    dtype = torch.float64
    r = data["rays_in"]
    n = r.shape[1]
    mk = lambda: [torch.tensor(r[j], dtype=dtype, device=DEV) for j in range(7)] + \
        [torch.zeros(n, dtype=dtype, device=DEV)]  # noqa: E731
    whole, gap = HipSystem(table, DEV), HipSystem(holed, DEV)
    try:
        full = whole.trace(mk(), 0, record=True)
        rays = mk()
        a = gap.trace(rays, 0, record=True, first=0, last=k - 1, write_rays=True)
        b = whole.trace(rays, 0, record=True, first=k, last=k, write_rays=True)
        c = gap.trace(rays, 0, record=True, first=k + 1, last=S_, write_rays=True)
        got = torch.cat([a.record[:, :, :n], b.record[:, :, :n], c.record[:, :, :n]])
        assert_close_planes(got.cpu().numpy(), full.record[:, :, :n].cpu().numpy(),
                            1e-10, 1e-10, case)
        for u, v in zip(rays, full.rows(S_)):
            np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-10, atol=1e-10)
    finally:
        whole.close(), gap.close()


def test_engine_survives_deepcopy_and_pickle():
    """The drop-in caches engines on reference objects the reference deep-copies; a
    HipSystem copy is a fresh handle on the same table, and traces identically."""
    import copy
    import pickle
    from optiland_amd.engine import HipSystem
    table, data = load_case("double_gauss")
    r = data["rays_in"]
    n = r.shape[1]
    mk = lambda: [torch.tensor(r[j], dtype=torch.float64, device=DEV) for j in range(7)] + \
        [torch.zeros(n, dtype=torch.float64, device=DEV)]  # noqa: E731
    a = HipSystem(table, DEV)
    holder = {"engines": {"k": (a, table)}, "tag": 3}
    b = copy.deepcopy(holder)["engines"]["k"][0]
    c = pickle.loads(pickle.dumps(a))
    assert b is not a and len({a._handle.value, b._handle.value, c._handle.value}) == 3
    want = a.trace(mk(), 0, record=True).record[:, :, :n].clone()
    a.close()                                   # the copies do not depend on the original
    for other in (b, c):
        assert torch.equal(other.trace(mk(), 0, record=True).record[:, :, :n], want)
        other.close()


@pytest.mark.parametrize("case", ["zernike_fresnel_fringe", "coated_mirror_polarised"])
def test_complex_prt_planes_on_a_real_prt_system(case):
    """The SurfaceGroup seam always hands over 18 PRT planes (a caller's PolarizedRays.p
    is complex whatever the coatings).  On a system whose own PRT is real the complex
    kernel must (a) reproduce the 9-plane result with a zero imaginary part from the
    identity, and (b) advance an arbitrary complex start matrix p0 to P @ p0."""
    from optiland_amd.engine import HipSystem
    from optiland_amd.rays import prt_to_complex
    table, data = load_case(case)
    assert not table.needs_complex_prt
    dtype = torch.float64
    r = data["rays_in"]
    n = r.shape[1]
    mk = lambda: [torch.tensor(r[j], dtype=dtype, device=DEV) for j in range(7)] + \
        [torch.zeros(n, dtype=dtype, device=DEV)]  # noqa: E731
    hip = HipSystem(table, DEV)
    try:
        p9 = torch.empty((9, n), dtype=dtype, device=DEV)
        hip.trace(mk(), 0, record=False, prt=p9, prt_identity=True)
        p18 = torch.empty((18, n), dtype=dtype, device=DEV)
        hip.trace(mk(), 0, record=False, prt=p18, prt_identity=True)
        ok = ~torch.isnan(p9).any(0)
        assert ok.float().mean() > 0.5
        assert torch.equal(p18[:9, ok], p9[:, ok]) and float(p18[9:, ok].abs().max()) == 0.0
        g = torch.Generator(device="cpu").manual_seed(5)
        p0 = torch.complex(torch.randn(n, 3, 3, generator=g, dtype=dtype),
                           torch.randn(n, 3, 3, generator=g, dtype=dtype)).to(DEV)
        start = torch.cat([p0.real.reshape(n, 9).t(), p0.imag.reshape(n, 9).t()]).contiguous()
        hip.trace(mk(), 0, record=False, prt=start)
        got = prt_to_complex(start)
        want = prt_to_complex(p18) @ p0
        np.testing.assert_allclose(got[ok].cpu().numpy(), want[ok].cpu().numpy(),
                                   rtol=1e-10, atol=1e-11)
    finally:
        hip.close()


def test_random_pupil_is_drawn_on_the_device():
    """distribution.py:132-158 (`RandomDistribution`): uniform over the unit disc, a fresh
    sample per call -- drawn on the device like the reference's torch backend does (the host
    sampler + upload cost 31 ms per 1e6 points), and traced like any other pupil."""
    from optiland_amd import load_system
    from optiland_amd.tracer import HipRayTracer
    t = HipRayTracer(load_system("double_gauss"), "cuda:0", dtype=torch.float32)
    n = 400_000
    px, py = t._pupil_planes("random", n)
    assert px.is_cuda and px.dtype == torch.float32 and px.numel() == n == py.numel()
    r2 = (px * px + py * py).double()
    assert float(r2.max()) <= 1.0 + 1e-6
    assert abs(float(r2.mean()) - 0.5) < 5e-3            # E[r^2] of the uniform disc
    assert abs(float(px.double().mean())) < 5e-3 and abs(float(py.double().mean())) < 5e-3
    quadrant = float(((px > 0) & (py > 0)).double().mean())
    assert abs(quadrant - 0.25) < 5e-3
    px2, _ = t._pupil_planes("random", n)
    assert not torch.equal(px, px2)
    rays = t.trace(0.0, 0.7, 0.5876, 5000, "random")
    assert rays.x.numel() == 5000 and bool(torch.isfinite(rays.x).any())
