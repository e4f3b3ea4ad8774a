"""The kernel's per-surface arithmetic, executed on the HOST, against the golden vectors and
the oracle -- the checks tests/test_gpu_parity.py makes on the MI355X, made here without a
GPU.

`optiland_amd/csrc/surface_math.h` is compiled a second time for the host by
tests/hostmath (see harness.hip: test infrastructure, not a fallback -- the product
library contains this arithmetic as device code only and the package never loads the
harness).  What is checked here is the SOURCE of the kernel arithmetic: formulas, branch
structure, table layout and the C ABI's table conversion.  What is not: the launch glue
(plane addressing, vector / packed variants, wave-level code), the 1-ulp hardware
rcp / sqrt / rsq / exp, and anything about speed -- those stay with the `-m gpu` tests.

Tolerances are the GPU suite's own (tests/test_gpu_parity.py).
"""

import copy

import numpy as np
import pytest

from tests import _hostmath as hm
from tests._util import (PLANES, assert_close_planes, fp32_group_tolerances, fp32_image_tolerance,
                         golden_cases, image_plane_error_over_spot, load_case)

pytestmark = pytest.mark.skipif(not hm.available(), reason="hipcc (used as host C++ compiler) missing")

TOL = {np.float64: 1e-6, np.float32: 1e-4}
TIGHT64 = 1e-9
DTYPES = [np.float64, np.float32]
IDS = ["f64", "f32"]


@pytest.fixture(scope="module")
def host():
    cache = {}

    def get(case):
        if case not in cache:
            table, data = load_case(case)
            cache[case] = (hm.HostMathSystem(table), table, data)
        return cache[case]

    yield get
    for sysm, _, _ in cache.values():
        sysm.close()


def _rays(data, dtype, n=None):
    r = data["rays_in"] if n is None else data["rays_in"][:, :n]
    planes = [np.array(r[k], dtype=dtype, order="C", copy=True) for k in range(7)]  # (write-back!)
    planes.append(np.zeros(planes[0].size, dtype=dtype))
    return planes


def test_harness_is_not_the_product_library():
    """The package refuses to take the harness for the HIP extension."""
    import subprocess
    import sys
    lib = hm._builder().build()
    code = ("import os, sys; os.environ['OPTILAND_HIP_LIBRARY'] = sys.argv[1]\n"
            "from optiland_amd import _capi\n"
            "try:\n    _capi.load()\nexcept _capi.HipExtensionError as e:\n"
            "    print('refused:', e); sys.exit(0)\nsys.exit(1)\n")
    out = subprocess.run([sys.executable, "-c", code, lib], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "TEST harness" in out.stdout


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", golden_cases())
def test_record_matches_reference(host, case, dtype):
    sysm, table, data = host(case)
    polarized = "prt" in data
    rays = _rays(data, dtype)
    n = rays[0].size
    prt = hm.new_prt(n, dtype, table.needs_complex_prt) if polarized else None
    rec, status = sysm.trace(rays, 0, record=True, prt=prt)
    assert (status & ~0x20) == 0  # (0x20: OL_STATUS_NAN_DIRECTION, informational)
    got = rec.astype(np.float64)
    tol = TOL[dtype]
    assert_close_planes(got, data["record"], tol, tol, f"{case}:{dtype.__name__}")
    if dtype == np.float32:
        gt = fp32_group_tolerances(case)
        assert gt is not None
        assert_close_planes(got, data["record"], tol, tol, f"{case}:fp32 tight", group_tol=gt)
        img = image_plane_error_over_spot(got, data["record"], data)
        assert img <= fp32_image_tolerance(case), (case, img, fp32_image_tolerance(case))
    else:
        from oracle import oracle
        tt = copy.deepcopy(table)
        tt.surfaces["tol"] = np.where(tt.surfaces["max_iter"] > 0, 1e-13, tt.surfaces["tol"])
        rin = {k: data["rays_in"][j] for j, k in enumerate(PLANES[:7])}
        conv = oracle.trace(tt, rin, 0, record=True, polarized=polarized)["record"]
        has_nr = bool(np.any(tt.surfaces["max_iter"] > 0))
        tight = 1e-7 if has_nr else TIGHT64
        assert_close_planes(got, conv, tight, tight, f"{case}:tight-vs-converged-oracle")
        assert np.array_equal(got[:, 6, :] == 0, data["record"][:, 6, :] == 0)
    if polarized:
        p = hm.prt_to_complex(prt)
        want = data["prt"]
        ok = ~np.isnan(want.real)
        np.testing.assert_allclose(p.real[ok], want.real[ok], rtol=tol, atol=tol)
        np.testing.assert_allclose(np.where(np.isnan(p.imag), 0, p.imag)[ok],
                                   np.where(np.isnan(want.imag), 0, want.imag)[ok], rtol=tol, atol=tol)
        assert np.array_equal(np.isnan(p.real), np.isnan(want.real))


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", ["double_gauss", "rc_asphere", "tilted_fold", "zernike_nopol"])
def test_writeback_and_partial_ranges(host, case, dtype):
    sysm, table, data = host(case)
    full, _ = sysm.trace(_rays(data, dtype), 0, record=True)
    rays = _rays(data, dtype)
    sysm.trace(rays, 0, record=False)
    for k in range(8):
        assert np.array_equal(np.nan_to_num(rays[k], nan=-7.0), np.nan_to_num(full[-1, k], nan=-7.0))
    S = table.num_surfaces - 1
    k = S // 2
    rays = _rays(data, dtype)
    sysm.trace(rays, 0, record=False, first=0, last=k)
    sysm.trace(rays, 0, record=False, first=k + 1, last=S)
    tol = TOL[dtype] * 1e-2
    assert_close_planes(np.stack(rays).astype(np.float64)[None], full[-1].astype(np.float64)[None],
                        tol, tol, f"{case}:split")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_fresh_prt_equals_identity_prt(host, dtype):
    """OL_TRACE_PRT_IDENTITY (write-only PRT, first update specialised) == reading an
    identity matrix and multiplying."""
    for case in ("zernike_fresnel_fringe", "coated_mirror_polarised", "polarizer_only"):
        sysm, table, data = host(case)
        n = data["rays_in"].shape[1]
        a = hm.new_prt(n, dtype, table.needs_complex_prt)
        b = np.full_like(a, 123.0)  # garbage: never read
        ra, _ = sysm.trace(_rays(data, dtype), 0, record=True, prt=a)
        rb, _ = sysm.trace(_rays(data, dtype), 0, record=True, prt=b, prt_identity=True)
        assert np.array_equal(np.nan_to_num(ra), np.nan_to_num(rb))
        eps = np.finfo(dtype).eps
        np.testing.assert_allclose(np.nan_to_num(b), np.nan_to_num(a), rtol=0, atol=8 * eps)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("case", golden_cases())
def test_ray_generation(host, case, dtype):
    sysm, table, data = host(case)
    if not table.raygen:
        pytest.skip("no ray-generation scalars")
    generic = not bool(data["via_trace"])
    vx, vy = 1.0 - data["vx"], 1.0 - data["vy"]
    px = data["Px"] * (vx if generic else 1.0)
    py = data["Py"] * (vy if generic else 1.0)
    a = lambda v: np.ascontiguousarray(v, dtype=dtype)
    out, status = sysm.generate_rays(a(data["Hx"]), a(data["Hy"]), a(px), a(py), a(vx), a(vy))
    assert (status & ~0x20) == 0  # (0x20: OL_STATUS_NAN_DIRECTION, informational)
    tol = 1e-12 if dtype == np.float64 else 2e-6
    for j in range(7):
        want = data["rays_in"][j]
        scale = max(1.0, np.abs(data["rays_in"][:3]).max()) if j < 3 else 1.0
        np.testing.assert_allclose(out[j].astype(np.float64), want, rtol=tol, atol=tol * scale,
                                   err_msg=f"{case}:{PLANES[j]}")


def test_status_bits(host):
    sysm, table, data = host("zernike_nopol")
    rays = _rays(data, np.float64)
    rays[0] += 40.0  # far outside norm_radius
    _, status = sysm.trace(rays, 0, record=False)
    assert status & 0x1  # OL_STATUS_ZERNIKE_RANGE
    sysm, table, data = host("double_gauss")
    _, st = sysm.generate_rays(0.0, 0.5, np.array([0.0, 1.5]), np.array([0.0, 0.0]), flags=0x2)
    assert st & 0x10  # OL_STATUS_PUPIL_RANGE


def test_c_abi_errors_are_the_product_ones(host):
    """Same capi.hip: the validation errors of the boundary come out of the harness too."""
    sysm, table, data = host("zernike_fresnel_fringe")
    with pytest.raises(RuntimeError, match="Polarization must be set"):
        sysm.trace(_rays(data, np.float64), 0, record=False)
    sysm, table, data = host("polarizer_retarder")
    n = data["rays_in"].shape[1]
    with pytest.raises(RuntimeError, match="retarder"):
        sysm.trace(_rays(data, np.float64), 0, record=False, prt=hm.new_prt(n, np.float64, False))


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_oracle_side_by_side_large(host, dtype):
    """Seeded 1e5-ray bundle on the double Gauss: host-executed kernel arithmetic vs the
    oracle directly."""
    from oracle import oracle
    sysm, table, data = host("double_gauss")
    rng = np.random.default_rng(11)
    n = 100_000
    r, th = np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    rays = oracle.generate_rays(table.raygen, np.zeros(n), rng.uniform(-1, 1, n),
                                r * np.cos(th), r * np.sin(th))
    want = oracle.trace(table, rays, 0, record=True)["record"]
    planes = [np.ascontiguousarray(rays[k], dtype=dtype) for k in PLANES[:7]]
    planes.append(np.zeros(n, dtype=dtype))
    rec, _ = sysm.trace(planes, 0, record=True)
    tol = TOL[dtype]
    assert_close_planes(rec.astype(np.float64), want, tol, tol, "oracle-1e5")


def _zernike_singlet(seed, degree, scheme_norm):
    """A singlet whose first surface is a Zernike freeform with EVERY term of radial order
    <= degree (random coefficients), second surface spherical, then the image plane."""
    from optiland_amd import system as S
    from optiland_amd.system import SystemTable
    rng = np.random.default_rng(77_000 + 31 * seed + degree)
    surf = np.zeros(4, dtype=S.SURFACE_DESC_DTYPE)
    optics = np.zeros((4, 1), dtype=S.SURFACE_OPTICS_DTYPE)
    surf["rot"] = np.eye(3).reshape(-1)
    surf["norm_radius"] = 1.0
    surf[0]["geom_kind"], surf[0]["interaction"] = S.GEOM_PLANE, S.INTERACT_RECORD_ONLY
    surf[0]["origin"] = (0.0, 0.0, -10.0)
    optics[0, 0] = (1.0, 1.0, 0.0)
    terms = []
    for n in range(degree + 1):
        for m in range(-n, n + 1, 2):
            if n == 0:
                continue
            c = rng.uniform(-1, 1) * 4e-3 / (1 + n * n)
            if n < degree and rng.random() < 0.25:
                c = 0.0  # holes, like a hand-written coefficient list
            N = np.sqrt((2.0 if m else 1.0) * (n + 1)) if scheme_norm else 1.0
            terms.extend([c, float(n), float(m), N])
    surf[1]["geom_kind"], surf[1]["radius"], surf[1]["conic"] = S.GEOM_ZERNIKE, 60.0, -0.3
    surf[1]["norm_radius"] = 9.0
    surf[1]["max_iter"], surf[1]["tol"] = 100, 1e-12
    surf[1]["coeff_offset"], surf[1]["n_coeff"] = 0, len(terms) // 4
    surf[1]["interaction"] = S.INTERACT_REFRACT
    surf[1]["origin"] = (0.0, 0.0, 0.0)
    optics[1, 0] = (1.0, 1.5168, 0.0)
    surf[2]["geom_kind"], surf[2]["radius"] = S.GEOM_STANDARD, -150.0
    surf[2]["interaction"] = S.INTERACT_REFRACT
    surf[2]["origin"] = (0.0, 0.0, 5.0)
    optics[2, 0] = (1.5168, 1.0, 0.0)
    surf[3]["geom_kind"], surf[3]["interaction"] = S.GEOM_PLANE, S.INTERACT_REFRACT
    surf[3]["origin"] = (0.0, 0.0, 90.0)
    optics[3, 0] = (1.0, 1.0, 0.0)
    table = SystemTable(surfaces=surf, coeffs=np.array(terms, dtype=np.float64), optics=optics,
                        wavelengths=np.array([0.55]), name=f"zern_deg{degree}_{seed}")
    n = 1500
    r, th = 7.5 * np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    rays = {"x": r * np.cos(th), "y": r * np.sin(th), "z": np.full(n, -10.0)}
    L, M = rng.uniform(-0.05, 0.05, n), rng.uniform(-0.05, 0.05, n)
    rays.update(L=L, M=M, N=np.sqrt(1 - L * L - M * M), i=np.ones(n))
    # ray 0: down the axis through the vertex (the eps-regularised gradient, zernike.py:206-231);
    # rays 1, 2: within 1e-6 / 1e-3 of it
    for k, d in enumerate((0.0, 1e-6 * 9.0, 1e-3 * 9.0)):
        rays["x"][k], rays["y"][k], rays["L"][k], rays["M"][k], rays["N"][k] = d, -d / 2, 0, 0, 1
    return table, rays


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("scheme_norm", [False, True], ids=["fringe-like", "normalised"])
@pytest.mark.parametrize("degree", range(1, 11))
def test_zernike_every_degree(degree, scheme_norm, dtype):
    """Degrees 2..6 run the unrolled one-polynomial evaluator, 1, 7, 8 its loop form, 9 and 10
    the per-|m| level form (capi.hip: kZernMonoMaxDegree): each against the oracle (the
    reference's polar formulas), including the vertex ray; and the two forms against each
    other on the degrees where both exist."""
    from oracle import oracle
    from tests.test_hostmath_fuzz import _planes, _through_fp32
    table, rays = _zernike_singlet(0, degree, scheme_norm)
    if dtype == np.float32:
        rays = _through_fp32(rays)
    want = oracle.trace(table, rays, 0, record=True)
    assert want["status"] == 0
    sysm = hm.HostMathSystem(table)
    got, status = sysm.trace(_planes(rays, dtype), 0, record=True)
    sysm.close()
    assert (status & ~0x20) == 0  # (0x20: OL_STATUS_NAN_DIRECTION, informational)
    tol = 1e-7 if dtype == np.float64 else 1e-4
    assert_close_planes(got.astype(np.float64), want["record"], tol, tol, f"zern{degree}")
    if dtype == np.float64:  # fp64: far inside the tolerance -- 1e-11 of the system size
        assert_close_planes(got, want["record"], 1e-11, 1e-11, f"zern{degree}:tight")
    # the level form of the same surface (OPTILAND_HIP_ZERNIKE_MONO=0 at ol_system_create)
    import os
    os.environ["OPTILAND_HIP_ZERNIKE_MONO"] = "0"
    try:
        lev = hm.HostMathSystem(table)
    finally:
        del os.environ["OPTILAND_HIP_ZERNIKE_MONO"]
    got_lev, _ = lev.trace(_planes(rays, dtype), 0, record=True)
    lev.close()
    assert_close_planes(got_lev.astype(np.float64), want["record"], tol, tol, f"zern{degree}:levels")
    same = np.array_equal(np.nan_to_num(got), np.nan_to_num(got_lev))
    # degree <= 8 -> the two builds really are different evaluators (they differ in the last
    # bits); above the cap both are the level form
    assert same == (degree > 8), (degree, same)


def _plane_system(aperture_kind=0, aperture=(0, 0, 0, 0), geom=None, n2=1.0, polarization=None):
    """object plane -> one plane surface (optional aperture) -> image plane"""
    from optiland_amd import system as S
    from optiland_amd.system import SystemTable
    surf = np.zeros(3, dtype=S.SURFACE_DESC_DTYPE)
    optics = np.zeros((3, 1), dtype=S.SURFACE_OPTICS_DTYPE)
    surf["rot"] = np.eye(3).reshape(-1)
    surf["norm_radius"] = 1.0
    surf[0]["interaction"] = S.INTERACT_RECORD_ONLY
    surf[0]["origin"] = (0.0, 0.0, -10.0)
    surf[1]["geom_kind"] = S.GEOM_PLANE if geom is None else geom
    surf[1]["radius"] = np.inf
    surf[1]["interaction"] = S.INTERACT_REFRACT
    surf[1]["aperture_kind"], surf[1]["aperture"] = aperture_kind, aperture
    surf[2]["geom_kind"], surf[2]["interaction"] = S.GEOM_PLANE, S.INTERACT_REFRACT
    surf[2]["radius"] = np.inf
    surf[2]["origin"] = (0.0, 0.0, 20.0)
    optics[0, 0] = (1.0, 1.0, 0.0)
    optics[1, 0] = (1.0, n2, 0.0)
    optics[2, 0] = (n2, n2, 0.0)
    t = SystemTable(surfaces=surf, coeffs=np.zeros(0), optics=optics, wavelengths=np.array([0.55]),
                    name="plane_system")
    t.polarization = polarization
    return t


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_aperture_rims_are_inside(dtype):
    """Rays EXACTLY on an aperture's edge (physical_apertures/radial.py:56-70 and friends use
    <= / >=): the rim belongs to the aperture, one ulp beyond does not.  (The mutation test
    of the kernel source -- tools/host_mutation_test.py -- found the `<` / `<=` mutants of
    these comparisons alive: no golden has a ray exactly on a rim.)"""
    from optiland_amd import system as S
    from oracle import oracle
    up = lambda v: np.nextafter(dtype(v), dtype(np.inf))    # noqa: E731
    dn = lambda v: np.nextafter(dtype(v), dtype(-np.inf))   # noqa: E731
    cases = [
        # (kind, aperture, [(x, y, inside?) ...])
        (S.AP_RADIAL, (2.0, 5.0, 0, 0), [(3, 4, True), (up(3), 4, False), (0, 2, True), (0, dn(2), False),
                                         (5, 0, True), (0, -5, True)]),
        (S.AP_OFFSET_RADIAL, (0.0, 5.0, 1.0, -2.0), [(4, 2, True), (up(4), 2, False), (1, 3, True)]),
        (S.AP_RECTANGULAR, (-2.0, 3.0, -1.0, 4.0), [(3, 4, True), (up(3), 0, False), (0, up(4), False),
                                                    (-2, -1, True), (dn(-2), 0, False), (0, dn(-1), False)]),
        (S.AP_ELLIPTICAL, (4.0, 2.0, 0.0, 0.0), [(4, 0, True), (0, 2, True), (up(4), 0, False),
                                                 (0, up(2), False)]),
    ]
    for kind, ap, pts in cases:
        table = _plane_system(kind, ap)
        n = len(pts)
        rays = {"x": np.array([p[0] for p in pts], dtype=np.float64),
                "y": np.array([p[1] for p in pts], dtype=np.float64), "z": np.full(n, -10.0),
                "L": np.zeros(n), "M": np.zeros(n), "N": np.ones(n), "i": np.ones(n)}
        want = oracle.trace(table, rays, 0, record=True)["record"]
        sysm = hm.HostMathSystem(table)
        planes = [np.ascontiguousarray(rays[k], dtype=dtype) for k in PLANES[:7]] + [np.zeros(n, dtype=dtype)]
        got, _ = sysm.trace(planes, 0, record=True)
        sysm.close()
        inside = np.array([p[2] for p in pts])
        assert np.array_equal(got[-1, 6] > 0, inside), (kind, got[-1, 6])
        assert np.array_equal(want[-1, 6] > 0, inside), (kind, "oracle", want[-1, 6])


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_sign_of_zero_dot_zeroes_the_normal(dtype):
    """real_rays.py:535-571: the normal is multiplied by `sign(dot)`, and sign(0) = 0 -- a ray
    travelling exactly IN the surface (N = 0 at a flat one) keeps u * k0 as its new direction
    (SURVEY.md Appendix D).  Mutant found alive by tools/host_mutation_test.py."""
    from optiland_amd import system as S
    from oracle import oracle
    table = _plane_system(geom=S.GEOM_STANDARD, n2=1.5)
    rays = {"x": np.zeros(2), "y": np.zeros(2), "z": np.array([-10.0, -10.0]),
            "L": np.array([1.0, 0.6]), "M": np.zeros(2), "N": np.array([0.0, 0.8]), "i": np.ones(2)}
    want = oracle.trace(table, rays, 0, record=True)["record"]
    sysm = hm.HostMathSystem(table)
    planes = [np.ascontiguousarray(rays[k], dtype=dtype) for k in PLANES[:7]] + [np.zeros(2, dtype=dtype)]
    got, _ = sysm.trace(planes, 0, record=True)
    sysm.close()
    # direction after the surface: u * k0 for the in-plane ray (no normal component added)
    np.testing.assert_allclose(got[1, 3:6, 0], [1.0 / 1.5, 0.0, 0.0], atol=1e-7)
    np.testing.assert_allclose(got[1, 3:6, 0], want[1, 3:6, 0], atol=1e-7)
    np.testing.assert_allclose(got[1, 3:6, 1], want[1, 3:6, 1], atol=1e-6)


def test_zernike_range_bit_just_outside_the_unit_square(host):
    """zernike.py:254-266 raises when |x / norm| > 1 or |y / norm| > 1: a ray at 1.2 norm radii
    must set the bit, one at 0.99 must not."""
    sysm, table, data = host("zernike_nopol")
    norm = float(table.surfaces["norm_radius"][table.surfaces["geom_kind"] == 3][0])
    for frac, flagged in ((0.99, False), (1.2, True)):
        rays = [np.array([frac * norm]), np.zeros(1), np.array([-10.0]), np.zeros(1), np.zeros(1),
                np.ones(1), np.ones(1), np.zeros(1)]
        _, status = sysm.trace(rays, 0, record=False)
        assert bool(status & 0x1) == flagged, (frac, status)


def test_polygon_aperture_points_on_edges_and_vertices():
    """physical_apertures/polygon.py delegates to matplotlib's `Path.contains_points`, whose
    crossings test decides points exactly on an edge or a vertex by its tie rule (`>=`); the
    oracle's restatement is pinned to matplotlib on such points (tests/test_oracle_polygon.py),
    and the kernel source is held to the oracle here -- convex and concave polygons, points on
    vertices, on horizontal / vertical / slanted edges, just inside and just outside.  (The
    `>` for `>=` mutant of the crossing test survived tools/host_mutation_test.py before.)"""
    from optiland_amd import system as S
    from oracle import oracle
    polys = [
        [(-2, -2), (2, -2), (2, 2), (-2, 2)],
        [(0, 0), (4, 0), (4, 4), (2, 1.5), (0, 4)],            # concave
        [(-3, 0), (0, -3), (3, 0), (0, 3)],                    # diamond: slanted edges
    ]
    pts = [(2, 0), (0, 2), (-2, 0), (0, -2), (2, 2), (-2, -2), (2, -2), (-2, 2), (0, 0), (4, 2), (2, 1.5),
           (3, 0), (0, 3), (1.5, 1.5), (-1.5, 1.5), (1.5, -1.5), (-1.5, -1.5), (1, 0.75), (3, 2.75),
           (0, 4), (4, 4), (4, 0), (2, 0.5), (1.9999999, 0), (2.0000001, 0), (5, 5)]
    n = len(pts)
    for verts in polys:
        table = _plane_system(S.AP_POLYGON, (0.0, float(len(verts)), 0.0, 0.0))
        table.coeffs = np.array([c for v in verts for c in v], dtype=np.float64)
        rays = {"x": np.array([p[0] for p in pts], dtype=np.float64),
                "y": np.array([p[1] for p in pts], dtype=np.float64), "z": np.full(n, -10.0),
                "L": np.zeros(n), "M": np.zeros(n), "N": np.ones(n), "i": np.ones(n)}
        want = oracle.trace(table, rays, 0, record=True)["record"][-1, 6] > 0
        sysm = hm.HostMathSystem(table)
        planes = [np.ascontiguousarray(rays[k]) for k in PLANES[:7]] + [np.zeros(n)]
        got, _ = sysm.trace(planes, 0, record=True)
        sysm.close()
        assert np.array_equal(got[-1, 6] > 0, want), (verts, list(zip(pts, got[-1, 6] > 0, want)))
        assert want.any() and not want.all()
