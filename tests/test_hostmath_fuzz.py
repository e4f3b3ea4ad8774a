"""tests/test_gpu_fuzz.py's randomised systems, run through the HOST build of the kernel
arithmetic (tests/hostmath) against the oracle -- same generators, same seeds, same
tolerances, no GPU.  See tests/test_hostmath.py for what this does and does not cover.
"""

import numpy as np
import pytest

from optiland_amd import system as S
from tests import _hostmath as hm
from tests._util import PLANES, assert_close_planes
from tests.test_gpu_fuzz import (random_nr_system, random_polarised_system, random_system)

pytestmark = pytest.mark.skipif(not hm.available(), reason="hipcc (used as host C++ compiler) missing")
DTYPES = [np.float64, np.float32]
IDS = ["f64", "f32"]


def _planes(rays, dtype):
    p = [np.array(rays[k], dtype=dtype, order="C", copy=True) for k in PLANES[:7]]
    p.append(np.zeros(p[0].size, dtype=dtype))
    return p


def _through_fp32(rays):
    return {k: v.astype(np.float32).astype(np.float64) for k, v in rays.items()}


@pytest.mark.parametrize("seed", range(40))
def test_random_system_fp64(seed):
    from oracle import oracle
    table, rays, has_nr = random_system(seed)
    want = oracle.trace(table, rays, 0, record=True)["record"]
    sysm = hm.HostMathSystem(table)
    got, _ = sysm.trace(_planes(rays, np.float64), 0, record=True)
    sysm.close()
    tol = 1e-7 if has_nr else 1e-9
    assert_close_planes(got, want, tol, tol, f"fuzz{seed}")
    assert np.array_equal(got[:, 6, :] == 0, want[:, 6, :] == 0)


@pytest.mark.parametrize("seed", range(40))
def test_random_system_fp32_on_well_conditioned_rays(seed):
    from oracle import oracle
    table, rays, _ = random_system(seed)
    n = rays["x"].size
    r32 = _through_fp32(rays)
    want = oracle.trace(table, r32, 0, record=True)["record"]
    stable = np.ones(n, dtype=bool)
    rng = np.random.default_rng(1000 + seed)
    for _ in range(10):
        pert = {k: v.copy() for k, v in r32.items()}
        for k in ("x", "y"):
            pert[k] += rng.uniform(-1e-3, 1e-3, n)
        for k in ("L", "M"):
            pert[k] += rng.uniform(-1e-4, 1e-4, n)
        pert["N"] = np.sqrt(1 - pert["L"] ** 2 - pert["M"] ** 2)
        alt = oracle.trace(table, pert, 0, record=True)["record"]
        stable &= np.all(np.isnan(alt) == np.isnan(want), axis=(0, 1))
        stable &= np.all((alt[:, 6, :] == 0) == (want[:, 6, :] == 0), axis=0)
        with np.errstate(invalid="ignore"):
            stable &= np.all(np.nan_to_num(np.abs(alt[:, 3:6, :] - want[:, 3:6, :])) < 1e-2,
                             axis=(0, 1))
            # (host only) the nudge is ~600 fp32 roundings of the launch state: a ray whose
            # hit moves by more than 5e-3 of the system size under it -- a clipped ray
            # refracted to within 2 degrees of grazing and then flung 900 mm, seed 32 --
            # lands on either side of the 1e-4 contract depending on how rcp / sqrt round,
            # which is exactly what differs between the host build and the device
            pos = want[:, :3, :]
            scale = np.max(np.abs(pos[np.isfinite(pos)]))
            stable &= np.all(np.nan_to_num(np.abs(alt[:, :3, :] - pos)) < 5e-3 * scale, axis=(0, 1))
    assert stable.sum() > 0.3 * n
    sysm = hm.HostMathSystem(table)
    got, _ = sysm.trace(_planes(r32, np.float32), 0, record=True)
    sysm.close()
    got = got.astype(np.float64)
    assert_close_planes(got[:, :, stable], want[:, :, stable], 1e-4, 1e-4, f"fuzz{seed}:f32")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("seed", range(25))
def test_random_polarised_system(seed, dtype):
    from oracle import oracle
    table, rays = random_polarised_system(seed)
    if dtype == np.float32:
        rays = _through_fp32(rays)
    n = rays["x"].size
    out = oracle.trace(table, rays, 0, record=True, polarized=True)
    sysm = hm.HostMathSystem(table)
    prt = np.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype)
    got, _ = sysm.trace(_planes(rays, dtype), 0, record=True, prt=prt, prt_identity=True)
    sysm.close()
    p = hm.prt_to_complex(prt)
    tol = 1e-9 if dtype == np.float64 else 1e-4
    assert_close_planes(got.astype(np.float64), out["record"], tol, tol, f"polfuzz{seed}")
    assert np.array_equal(np.isnan(p.real), np.isnan(out["prt"].real))
    np.testing.assert_allclose(np.nan_to_num(p.real), np.nan_to_num(out["prt"].real), rtol=0,
                               atol=tol * 10)
    if table.needs_complex_prt:
        np.testing.assert_allclose(np.nan_to_num(p.imag), np.nan_to_num(out["prt"].imag), rtol=0,
                                   atol=tol * 10)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("seed", range(25))
def test_random_polarised_system_with_directions_that_are_not_unit_vectors(seed, dtype):
    """`OL_TRACE_NONUNIT_K` (ABI 11): the bundles of the reference's iterative / robust aimers,
    |k|^2 - 1 ~ 1e-3 (rays/ray_aiming/iterative.py:339-366).  The oracle restates
    polarized_rays.py:136-202 literally -- k as it comes, triads that are not orthonormal --
    and the kernels' rank-2 update on the normalised directions with the amplitudes scaled by
    |k0| |k1| is the same matrix; without the flag the two differ by ~|k|^2 - 1.  Every
    coating kind (the polarizer / retarder forms carry the lengths in p0 and p1).  The image
    plane -- the one equal-index surface of these systems, where the reference's s is rounding
    noise -- is left out of the range, as integration.py leaves it to the reference."""
    from oracle import oracle
    table, rays = random_polarised_system(seed)
    g = np.random.default_rng(99 + seed)
    scale = 1.0 + 1e-3 * g.uniform(-1.0, 1.0, rays["x"].size)
    rays = dict(rays)
    for k in ("L", "M", "N"):
        rays[k] = rays[k] * scale
    if dtype == np.float32:
        rays = _through_fp32(rays)
    n = rays["x"].size
    last = table.num_surfaces - 2
    out = oracle.trace(table, rays, 0, record=True, polarized=True, last=last)
    sysm = hm.HostMathSystem(table)
    shape = (18 if table.needs_complex_prt else 9, n)
    prt, plain = np.empty(shape, dtype=dtype), np.empty(shape, dtype=dtype)
    got, _ = sysm.trace(_planes(rays, dtype), 0, record=True, prt=prt, prt_identity=True,
                        last=last, nonunit_directions=True)
    sysm.trace(_planes(rays, dtype), 0, record=True, prt=plain, prt_identity=True, last=last)
    sysm.close()
    p, q = hm.prt_to_complex(prt), hm.prt_to_complex(plain)
    tol = 1e-9 if dtype == np.float64 else 1e-4
    assert_close_planes(got.astype(np.float64), out["record"], tol, tol, f"polfuzz{seed}")
    assert np.array_equal(np.isnan(p.real), np.isnan(out["prt"].real))
    np.testing.assert_allclose(np.nan_to_num(p), np.nan_to_num(out["prt"]), rtol=0, atol=tol * 10)
    kinds = table.surfaces["coating_kind"][1:last + 1]
    # the rank-2 form with an s-amplitude that is not zero (a Fresnel "mirror" between equal
    # indices has j0 = j1 = 0, and the p and k terms are right with or without the flag)
    coated = np.any(((kinds == S.COAT_FRESNEL) | (kinds == S.COAT_NONE))
                    & (table.surfaces["interaction"][1:last + 1] == S.INTERACT_REFRACT))
    # (... of a matrix that is not zero: a Fresnel "mirror" between equal indices reflects nothing)
    alive = np.isfinite(out["prt"].real).any() and np.nanmin(np.abs(out["prt"]).max(axis=(1, 2))) > 0.1
    if dtype == np.float64 and coated and alive:
        # (the case shows something: without the flag the matrices are NOT the reference's)
        assert np.nanmax(np.abs(q - out["prt"])) > 1e-5


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("seed", range(28))
def test_random_newton_raphson_system(seed, dtype):
    from oracle import oracle
    table, rays = random_nr_system(seed)
    if dtype == np.float32:
        rays = _through_fp32(rays)
    want = oracle.trace(table, rays, 0, record=True)
    assert want["status"] == 0
    sysm = hm.HostMathSystem(table)
    got, status = sysm.trace(_planes(rays, dtype), 0, record=True)
    sysm.close()
    assert (status & ~0x20) == 0  # (0x20: OL_STATUS_NAN_DIRECTION, informational)
    tol = 1e-7 if dtype == np.float64 else 1e-4
    assert_close_planes(got.astype(np.float64), want["record"], tol, tol, f"nrfuzz{seed}")


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("seed", range(28))
def test_random_newton_raphson_system_polarised(seed, dtype):
    from oracle import oracle
    table, rays = random_nr_system(seed)
    rng = np.random.default_rng(40_000 + seed)
    for i in range(1, table.num_surfaces - 1):
        ck = rng.choice([S.COAT_FRESNEL, S.COAT_FRESNEL, S.COAT_SIMPLE, S.COAT_NONE])
        table.surfaces[i]["coating_kind"] = ck
        if ck == S.COAT_SIMPLE:
            table.surfaces[i]["coat"] = (rng.uniform(0.6, 1.0), rng.uniform(0.0, 0.4))
    table.polarization = {"is_polarized": bool(seed % 2), "Ex": 0.8, "Ey": 0.6, "phase_x": 0.3,
                          "phase_y": -0.4}
    if dtype == np.float32:
        rays = _through_fp32(rays)
    n = rays["x"].size
    out = oracle.trace(table, rays, 0, record=True, polarized=True)
    assert (out["status"] & ~0x20) == 0
    sysm2 = hm.HostMathSystem(table)
    prt = np.empty((9, n), dtype=dtype)
    got, _ = sysm2.trace(_planes(rays, dtype), 0, record=True, prt=prt, prt_identity=True)
    planes0 = _planes(rays, dtype)
    iu, ist = sysm2.polarized_intensity(prt, planes0[3:6], planes0[6], table.polarization)
    sysm2.close()
    p = hm.prt_to_complex(prt)
    tol = 1e-7 if dtype == np.float64 else 1e-4
    assert_close_planes(got.astype(np.float64), out["record"], tol, tol, f"nrpol{seed}")
    assert np.array_equal(np.isnan(p.real), np.isnan(out["prt"].real))
    np.testing.assert_allclose(np.nan_to_num(p.real), np.nan_to_num(out["prt"].real), rtol=0,
                               atol=tol * 10)
    # update_intensity: polarised (odd seeds) and UNPOLARISED states (mean of two, even seeds)
    want_i, wst = oracle.polarized_intensity(out["prt"], rays["L"], rays["M"], rays["N"], rays["i"],
                                             table.polarization)
    assert ist == 0 and wst == 0
    np.testing.assert_allclose(np.nan_to_num(iu.astype(np.float64)), np.nan_to_num(want_i), rtol=0,
                               atol=tol * 10)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("seed", range(30))
def test_random_ray_generation(seed, dtype):
    from oracle import oracle
    from optiland_amd import _capi, load_system
    rng = np.random.default_rng(30_000 + seed)
    table = load_system("double_gauss")
    kind = int(rng.integers(0, 3))
    infinite = bool(rng.random() < 0.5) if kind != 1 else False
    rg = {"object_infinite": 1.0 if infinite else 0.0, "field_kind": float(kind),
          "EPL": float(rng.uniform(5, 60)), "EPD": float(rng.uniform(4, 25)),
          "max_field": float(rng.uniform(2, 25)), "offset": float(rng.uniform(5, 30)) if infinite else 0.0,
          "z_first": float(rng.uniform(-200, -20)) if not infinite else 0.0,
          "tele_dz": float(rng.uniform(5, 40)) if (kind != 0 and not infinite and rng.random() < 0.4) else 0.0,
          "apod_kind": float(rng.integers(0, 7)), "apod_a": float(rng.uniform(0.6, 1.2)),
          "apod_b": float(rng.uniform(0.3, 0.9))}
    if kind == 2:
        rg["field_scale"] = float(rng.uniform(0.05, 0.6))
    if int(rg["apod_kind"]) == 5:
        rg["apod_b"] = float(rng.uniform(2.0, 6.0))
    if int(rg["apod_kind"]) == 4:
        rg["apod_b"] = float(rng.uniform(0.5, 3.0))
    table.raygen = rg
    sysm = hm.HostMathSystem(table)
    n = 3001
    r, th = np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    px, py = r * np.cos(th), r * np.sin(th)
    hx, hy = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    vx, vy = rng.uniform(0.7, 1.0, n), rng.uniform(0.7, 1.0, n)
    dev = lambda a: np.ascontiguousarray(a, dtype=dtype)  # noqa: E731
    seen = lambda a: dev(a).astype(np.float64)  # noqa: E731
    prescale = bool(rng.random() < 0.5)
    flags = _capi.RAYGEN_PRESCALE_PUPIL if prescale else 0
    got, _ = sysm.generate_rays(dev(hx), dev(hy), dev(px), dev(py), dev(vx), dev(vy), flags=flags)
    pxs, pys = (seen(px) * seen(vx), seen(py) * seen(vy)) if prescale else (seen(px), seen(py))
    want = oracle.generate_rays(rg, seen(hx), seen(hy), pxs, pys, seen(vx), seen(vy))
    scale = max(1.0, abs(rg["z_first"]), rg["EPD"], rg["offset"] + rg["EPL"])
    tol = 1e-12 if dtype == np.float64 else 2e-6
    for k, g in zip(("x", "y", "z", "L", "M", "N", "i"), got):
        gv = g.astype(np.float64)
        s_ = scale if k in "xyz" else 1.0
        if dtype == np.float32 and k == "i":
            ok = np.abs(gv - want[k]) < 1e-3
            assert ok.mean() > 0.995
            continue
        np.testing.assert_allclose(gv, want[k], rtol=0, atol=tol * s_ * 10, err_msg=k)
    sysm.close()


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("seed", range(25))
def test_random_polarised_system_update_intensity(seed, dtype):
    """trace + the `update_intensity` epilogue (csrc/epilogue_device.h) of random coated
    systems, host-run, against the oracle's -- real and complex PRT planes."""
    from oracle import oracle
    table, rays = random_polarised_system(seed)
    if dtype == np.float32:
        rays = _through_fp32(rays)
    n = rays["x"].size
    out = oracle.trace(table, rays, 0, record=True, polarized=True)
    sysm = hm.HostMathSystem(table)
    planes = _planes(rays, dtype)
    k0 = [planes[3].copy(), planes[4].copy(), planes[5].copy()]
    i0 = planes[6].copy()
    prt = np.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype)
    sysm.trace(planes, 0, record=True, prt=prt, prt_identity=True)
    iu, status = sysm.polarized_intensity(prt, k0, i0, table.polarization)
    sysm.close()
    assert (status & ~0x20) == 0  # (0x20: OL_STATUS_NAN_DIRECTION, informational)
    want_i, wstatus = oracle.polarized_intensity(out["prt"], rays["L"], rays["M"], rays["N"],
                                                 rays["i"], table.polarization)
    assert wstatus == 0
    tol = 1e-9 if dtype == np.float64 else 1e-4
    np.testing.assert_allclose(np.nan_to_num(iu.astype(np.float64)), np.nan_to_num(want_i), rtol=0,
                               atol=tol * 10)


def test_update_intensity_k_parallel_to_x_sets_the_status_bit():
    table, rays = random_polarised_system(0)
    sysm = hm.HostMathSystem(table)
    n = 4
    prt = hm.new_prt(n, np.float64, table.needs_complex_prt)
    k0 = [np.ones(n), np.zeros(n), np.zeros(n)]
    _, status = sysm.polarized_intensity(prt, k0, np.ones(n), table.polarization)
    sysm.close()
    assert status & S.STATUS_K_PARALLEL_X


@pytest.mark.parametrize("seed", range(20))
def test_random_wavefront_opd(seed, dtype=np.float64):
    """`ol_wavefront_opd` (csrc/wavefront_device.h), host-run, against the oracle: random
    image-plane bundles, spherical and planar references, with and without tilt terms.
    fp64 like the GPU test (wavefront work is fp64 throughout the package; the generator
    starts its rays ON the reference sphere, where the reference's root choice `t1 < 0`
    is decided by the last bits -- fine in fp64, a coin toss in fp32)."""
    from oracle import oracle
    from optiland_amd import load_system
    rng = np.random.default_rng(60_000 + seed)
    n = 5003
    R = float(rng.uniform(30, 400)) * (1 if rng.random() < 0.8 else -1)
    zc = float(rng.uniform(50, 150))
    params = {"xc": float(rng.uniform(-2, 2)), "yc": float(rng.uniform(-2, 2)), "zc": zc - R,
              "R": R, "n_image": float(rng.choice([1.0, 1.33])), "opd_ref": float(rng.uniform(90, 110)),
              "ux": float(rng.uniform(-0.05, 0.05)) if seed % 2 else 0.0,
              "uy": float(rng.uniform(-0.05, 0.05)) if seed % 2 else 0.0,
              "half_epd": float(rng.uniform(3, 12)), "wavelength_um": float(rng.uniform(0.4, 1.6))}
    if seed % 4 == 3:
        nv = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), 1.0])
        nv /= np.linalg.norm(nv)
        params.update(zc=zc - abs(R), nx=float(nv[0]), ny=float(nv[1]), nz=float(nv[2]))
    L, M = rng.uniform(-0.15, 0.15, n), rng.uniform(-0.15, 0.15, n)
    rays7 = [rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), np.full(n, zc),
             L, M, np.sqrt(1 - L * L - M * M), rng.uniform(95, 105, n)]
    r, th = np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    px, py = r * np.cos(th), r * np.sin(th)
    want, want_pupil = oracle.wavefront_opd(params, rays7, px, py)
    sysm = hm.HostMathSystem(load_system("cooke_generic"))
    got, pupil = sysm.wavefront_opd(params, [np.ascontiguousarray(a, dtype=dtype) for a in rays7],
                                    np.ascontiguousarray(px, dtype=dtype),
                                    np.ascontiguousarray(py, dtype=dtype))
    sysm.close()
    assert np.isfinite(want).all()
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(want).max()))
    np.testing.assert_allclose(pupil, want_pupil, rtol=0, atol=1e-10 * abs(R))


def test_pupil_fill_is_the_definition():
    """`ol_pupil_fill` (psf/fft.py:101-137), host-run: A exp(-i 2 pi (OPD - plane)) of the
    compacted samples at their cells of the zero-padded grid, everything else untouched."""
    from optiland_amd import load_system
    rng = np.random.default_rng(5)
    n_side, grid = 37, 128
    yy, xx = np.mgrid[0:n_side, 0:n_side]
    X = (xx - (n_side - 1) / 2) / ((n_side - 1) / 2)
    Y = (yy - (n_side - 1) / 2) / ((n_side - 1) / 2)
    inside = (X * X + Y * Y) <= 1.0
    cell = np.flatnonzero(inside.ravel()).astype(np.int32)
    n = cell.size
    opd = rng.normal(0, 0.3, n)
    inten = rng.uniform(0.2, 1.0, n)
    pxy = (X.ravel()[cell].copy(), Y.ravel()[cell].copy())
    plane = (0.05, -0.2, 0.11)
    sysm = hm.HostMathSystem(load_system("cooke_generic"))
    for use_plane in (False, True):
        g = sysm.pupil_fill(opd, inten, cell, n_side, grid, pupil_xy=pxy if use_plane else None,
                            plane=plane if use_plane else None)
        o = opd - (plane[0] + plane[1] * pxy[0] + plane[2] * pxy[1]) if use_plane else opd
        want = np.zeros((grid, grid), dtype=np.complex128)
        pad = (grid - n_side) // 2
        sub = np.zeros(n_side * n_side, dtype=np.complex128)
        sub[cell] = np.sqrt(inten) * np.exp(-2j * np.pi * o)
        want[pad:pad + n_side, pad:pad + n_side] = sub.reshape(n_side, n_side)
        np.testing.assert_allclose(g, want, rtol=0, atol=1e-14)
    sysm.close()


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
@pytest.mark.parametrize("seed", range(28))
def test_random_fused_spot(seed, dtype):
    """`ol_trace_spot` (generate -> trace -> reduce), host-run, on the random Newton-Raphson
    lenses of tests/test_gpu_fuzz.py::test_random_fused_spot with the same generator scalars,
    against the oracle's generate + trace + numpy moments: per-ray hits and the seven
    moments; per-ray field / vignetting planes or one launch-uniform field."""
    from oracle import oracle
    from tests.test_gpu_fuzz import _random_raygen
    from tests.test_gpu_spot import _check_moments, _scale
    table, _ = random_nr_system(seed)
    rng = np.random.default_rng(50_000 + seed)
    table.raygen = _random_raygen(rng, table)
    n = 20_011
    r, th = np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    c = lambda a: np.ascontiguousarray(a, dtype=dtype)  # noqa: E731
    px, py = c(r * np.cos(th)), c(r * np.sin(th))
    planes = bool(seed % 2)
    if planes:
        hx, hy = c(rng.uniform(-1, 1, n)), c(rng.uniform(-1, 1, n))
        vx, vy = c(rng.uniform(0.8, 1.0, n)), c(rng.uniform(0.8, 1.0, n))
    else:
        f, v = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))), (0.95, 0.9)
        hx, hy, vx, vy = c(np.full(n, f[0])), c(np.full(n, f[1])), c(np.full(n, v[0])), c(np.full(n, v[1]))
    center = (float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2)))
    d = lambda a: a.astype(np.float64)  # noqa: E731
    g = oracle.generate_rays(table.raygen, d(hx), d(hy), d(px), d(py), d(vx), d(vy))
    g["opd"] = np.zeros(n)
    o = oracle.trace(table, g, 0, record=False)
    wx, wy, wi = o["x"], o["y"], o["i"]
    m = wi > 0
    dx, dy = wx[m] - center[0], wy[m] - center[1]
    r2 = (dx * dx + dy * dy)
    r2 = r2[~np.isnan(r2)]
    want = np.array([m.sum(), dx.sum(), dy.sum(), (dx * dx).sum(), (dy * dy).sum(), wi[m].sum(),
                     r2.max() if r2.size else 0.0])
    sysm = hm.HostMathSystem(table)
    if planes:
        got, hits, status = sysm.trace_spot(px, py, 0, hx=hx, hy=hy, vx=vx, vy=vy, center=center,
                                            want_hits=True)
    else:
        got, hits, status = sysm.trace_spot(px, py, 0, hx=f[0], hy=f[1], vx=v[0], vy=v[1],
                                            center=center, want_hits=True)
    sysm.close()
    assert (status & ~0x20) == 0  # (0x20: OL_STATUS_NAN_DIRECTION, informational)
    gx, gy, gi = (h.astype(np.float64) for h in hits)
    scale = _scale(table, wx, wy)
    assert want[0] > 0.05 * n, "bundle lost: the fuzz case tests nothing"
    if dtype == np.float64:
        assert np.array_equal(gi > 0, wi > 0)
        _check_moments(got, want, scale, 1e-7)
        tol = 1e-7
    else:
        assert ((gi > 0) == (wi > 0)).mean() > 0.998
        tol = 1e-4
    both = (gi > 0) & (wi > 0)
    np.testing.assert_allclose(gx[both], wx[both], rtol=0, atol=tol * scale)
    np.testing.assert_allclose(gy[both], wy[both], rtol=0, atol=tol * scale)
    np.testing.assert_allclose(gi[both], wi[both], rtol=0, atol=max(tol, 1e-9) * 10)


@pytest.mark.parametrize("case", ["double_gauss", "cooke_generic", "rc_asphere", "zernike_nopol",
                                  "aspheric_singlet"])
def test_fused_opd_equals_the_unfused_chain(case):
    """`ol_trace_opd` (generate -> trace -> OPD + its twelve moments in one kernel), host-run,
    against the chain it fuses -- `ol_generate_rays`, record-last `ol_trace`,
    `ol_wavefront_opd` -- run through the same harness: the maps must agree to rounding (same
    per-ray code, the fused kernel only skips the planes), the moments with numpy sums."""
    from optiland_amd import load_system
    from tests._util import load_case
    try:
        table = load_system(case)
    except KeyError:
        table, _ = load_case(case)
    if not table.raygen:
        pytest.skip("no ray-generation scalars")
    rng = np.random.default_rng(3)
    n = 4001
    r, th = np.sqrt(rng.random(n)) * 0.95, 2 * np.pi * rng.random(n)
    px, py = r * np.cos(th), r * np.sin(th)
    field = (0.0, 0.6)
    sysm = hm.HostMathSystem(table)
    # a plausible reference sphere: centred on the chief ray's image point, radius to z = 0
    chief, _ = sysm.generate_rays(field[0], field[1], np.zeros(1), np.zeros(1))
    sysm.trace(chief, 0, record=False)
    zc = float(chief[2][0])
    params = {"xc": float(chief[0][0]), "yc": float(chief[1][0]), "zc": zc, "R": abs(zc) * 0.8 + 5.0,
              "n_image": 1.0, "opd_ref": float(chief[7][0]), "ux": 0.01, "uy": -0.02,
              "half_epd": float(table.raygen["EPD"]) / 2, "wavelength_um": 0.55}
    opd, inten, pupil, mom, status = sysm.trace_opd(params, px, py, 0, field=field)
    assert (status & ~0x20) == 0  # (0x20: OL_STATUS_NAN_DIRECTION, informational)
    rays, _ = sysm.generate_rays(field[0], field[1], px, py)
    sysm.trace(rays, 0, record=False)
    want_opd, want_pupil = sysm.wavefront_opd(params, rays[:6] + [rays[7]], px, py)
    sysm.close()
    np.testing.assert_array_equal(np.isnan(opd), np.isnan(want_opd))
    np.testing.assert_allclose(np.nan_to_num(opd), np.nan_to_num(want_opd), rtol=0, atol=1e-9)
    np.testing.assert_allclose(np.nan_to_num(inten), np.nan_to_num(rays[6]), rtol=0, atol=0)
    np.testing.assert_allclose(np.nan_to_num(pupil), np.nan_to_num(want_pupil), rtol=0, atol=1e-11)
    w, o, X, Y = inten, opd, pupil[0], pupil[1]
    ok = ~np.isnan(o) & ~np.isnan(w)
    alive = ok & (w > 0)
    want_m = [w[ok].sum(), (w * X)[ok].sum(), (w * Y)[ok].sum(), (w * X * X)[ok].sum(),
              (w * X * Y)[ok].sum(), (w * Y * Y)[ok].sum(), (w * o)[ok].sum(), (w * o * X)[ok].sum(),
              (w * o * Y)[ok].sum(), alive.sum(), o[alive].sum(), (o * o)[alive].sum()]
    if ok.all():
        np.testing.assert_allclose(mom, want_m, rtol=1e-10, atol=1e-9)


@pytest.mark.parametrize("dtype", DTYPES, ids=IDS)
def test_reduction_entry_points_equal_their_definitions(dtype):
    """`ol_spot_moments`, `ol_spot_max_r2`, `ol_radial_energy`, `ol_irradiance` through the
    product's engine class on the host build: the bin searches (`edge_bin`,
    `radial_step_index` in epilogue_device.h) against numpy on data with hits exactly on
    edges, outside the detector, NaN coordinates, NaN and zero and negative weights."""
    import torch
    from optiland_amd import load_system
    eng = hm.make_engine_class()(load_system("cooke_generic"))
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    rng = np.random.default_rng(9)
    n = 20_000
    x, y = rng.normal(0, 1.0, n), rng.normal(0.2, 0.7, n)
    w = rng.uniform(0, 1, n)
    xe, ye = np.linspace(-2, 2, 33), np.linspace(-1.5, 1.5, 17)
    # exact edges (first, interior, last), outside, NaN
    x[:6] = [xe[0], xe[5], xe[-1], 2.5, np.nan, 0.1]
    y[:6] = [ye[0], ye[-1], ye[3], 0.0, 0.0, np.nan]
    w[6:12] = [0.0, -1.0, np.nan, 1.0, 0.5, 0.0]
    tx, ty, tw = (torch.tensor(np.asarray(v, dtype=dtype)) for v in (x, y, w))
    xs, ys, ws = tx.double().numpy(), ty.double().numpy(), tw.double().numpy()
    # irradiance == numpy.histogram2d over power > 0
    got = eng.irradiance(tx, ty, tw, torch.tensor(xe), torch.tensor(ye)).numpy()
    ok = ws > 0
    with np.errstate(invalid="ignore"):
        want, _, _ = np.histogram2d(xs[ok], ys[ok], bins=[xe, ye], weights=ws[ok])
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    # radial energy: per-step energy, cumulative sum == sum of energies with r <= step
    cx, cy = 0.05, 0.15
    r_step = np.linspace(0.0, 2.5, 41)
    r_step[7] = float(np.hypot(xs[20] - cx, ys[20] - cy))   # a hit exactly on a step
    r_step.sort()
    bins = eng.radial_energy(tx, ty, tw, cx, cy, torch.tensor(r_step)).numpy()
    r = np.hypot(xs - cx, ys - cy)
    with np.errstate(invalid="ignore"):
        want_cum = np.array([np.nansum(np.where(np.isnan(r) | (r > v), 0.0, ws)) for v in r_step])
    np.testing.assert_allclose(np.cumsum(bins), want_cum, rtol=1e-12, atol=1e-12)
    # spot moments / max radius over i > 0
    m = ws > 0
    got_m = eng.spot_moments(tx, ty, tw).numpy()
    with np.errstate(invalid="ignore"):
        want_m = [m.sum(), xs[m].sum(), ys[m].sum(), (xs[m] ** 2).sum(), (ys[m] ** 2).sum(), m.sum()]
    fin = np.isfinite(want_m)
    np.testing.assert_allclose(got_m[fin], np.asarray(want_m)[fin], rtol=1e-12)
    assert np.array_equal(np.isnan(got_m), ~fin)
    got_r = float(eng.spot_max_r2(tx, ty, tw, cx, cy)[0])
    r2 = ((xs - cx) ** 2 + (ys - cy) ** 2)[m]
    np.testing.assert_allclose(got_r, np.nanmax(r2), rtol=1e-14)  # (fused multiply-add)
    eng.close()
