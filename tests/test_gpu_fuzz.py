"""Randomised systems: HIP against the oracle on prescriptions nobody hand-picked.

Forty seeded random systems of 2-6 surfaces -- planes, conics of either sign and any
conic constant, even aspheres, mirrors, decentred and tilted frames, radial / rectangular /
elliptical apertures, index steps up and down, absorbing media -- traced with ray bundles
wide enough to produce misses, clipping and total internal reflection (conic-only systems;
aspheres get bundles inside the region where their Newton iteration has a unique root).  fp64 must agree
with the oracle to 1e-9 of the position scale with IDENTICAL NaN and clip masks (the
Newton surfaces to 1e-7: gradient reuse, see test_gpu_parity).  fp32 is checked on the
rays the oracle itself finds well away from every branch point (fp32 inputs alone move a
grazing ray across a miss / TIR / rim threshold), to the 1e-4 contract.
"""

import numpy as np
import pytest
import torch

from optiland_amd import system as S
from optiland_amd.system import SystemTable
from tests._util import PLANES, assert_close_planes

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rot(rng, max_deg):
    a, b, c = np.radians(rng.uniform(-max_deg, max_deg, 3))
    rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
    ry = np.array([[np.cos(b), 0, np.sin(b)], [0, 1, 0], [-np.sin(b), 0, np.cos(b)]])
    rz = np.array([[np.cos(c), -np.sin(c), 0], [np.sin(c), np.cos(c), 0], [0, 0, 1]])
    return rx @ ry @ rz


def random_system(seed):
    rng = np.random.default_rng(seed)
    ns = int(rng.integers(2, 7))  # traced surfaces
    surf = np.zeros(ns + 1, dtype=S.SURFACE_DESC_DTYPE)
    optics = np.zeros((ns + 1, 1), dtype=S.SURFACE_OPTICS_DTYPE)
    coeffs = []
    surf["rot"] = np.eye(3).reshape(-1)
    surf["norm_radius"] = 1.0
    surf[0]["geom_kind"] = S.GEOM_PLANE
    surf[0]["interaction"] = S.INTERACT_RECORD_ONLY
    surf[0]["origin"] = (0.0, 0.0, -20.0)
    n_prev = 1.0
    optics[0, 0] = (1.0, 1.0, 0.0)
    z = 0.0
    direction = 1.0
    has_nr = False
    for i in range(1, ns + 1):
        row = surf[i]
        kind = rng.choice([S.GEOM_PLANE, S.GEOM_STANDARD, S.GEOM_STANDARD, S.GEOM_STANDARD,
                           S.GEOM_EVEN_ASPHERE])
        row["geom_kind"] = kind
        if kind != S.GEOM_PLANE:
            row["radius"] = rng.choice([-1, 1]) * rng.uniform(25.0, 400.0) if rng.random() > 0.1 \
                else np.inf
            row["conic"] = rng.choice([0.0, 0.0, -1.0, rng.uniform(-2.5, 1.5)])
        else:
            row["radius"] = np.inf
        if kind == S.GEOM_EVEN_ASPHERE:
            has_nr = True
            c = [rng.uniform(-2e-4, 2e-4), rng.uniform(-2e-7, 2e-7), rng.uniform(-1e-10, 1e-10)]
            row["coeff_offset"], row["n_coeff"] = len(coeffs), len(c)
            coeffs.extend(c)
            row["max_iter"], row["tol"] = 100, 1e-12
        mirror = rng.random() < 0.2
        row["interaction"] = S.INTERACT_REFLECT if mirror else S.INTERACT_REFRACT
        n_next = n_prev if mirror else float(rng.choice([1.0, rng.uniform(1.3, 1.9)]))
        absorb = float(rng.choice([0.0, 0.0, rng.uniform(1e-3, 5e-2)]))
        optics[i, 0] = (n_prev, n_next, absorb)
        n_prev = n_next
        z += direction * rng.uniform(3.0, 25.0)
        if mirror:
            direction = -direction
        row["origin"] = (rng.uniform(-0.8, 0.8) if rng.random() < 0.4 else 0.0,
                         rng.uniform(-0.8, 0.8) if rng.random() < 0.4 else 0.0, z)
        if rng.random() < 0.35:
            row["rot"] = _rot(rng, 6.0).reshape(-1)
            row["flags"] = S.SURF_ROTATED
        ak = rng.choice([S.AP_NONE, S.AP_NONE, S.AP_RADIAL, S.AP_RECTANGULAR, S.AP_ELLIPTICAL])
        row["aperture_kind"] = ak
        if ak == S.AP_RADIAL:
            row["aperture"] = (rng.choice([0.0, 1.0]), rng.uniform(5.0, 9.0), 0, 0)
        elif ak == S.AP_RECTANGULAR:
            row["aperture"] = (-rng.uniform(4, 8), rng.uniform(4, 8), -rng.uniform(4, 8), rng.uniform(4, 8))
        elif ak == S.AP_ELLIPTICAL:
            row["aperture"] = (rng.uniform(5, 9), rng.uniform(4, 8), rng.uniform(-1, 1), rng.uniform(-1, 1))
    table = SystemTable(surfaces=surf, coeffs=np.array(coeffs, dtype=np.float64), optics=optics,
                        wavelengths=np.array([0.55]), name=f"fuzz{seed}")
    n = 4000
    # half the bundle is tame, half is wild (steep, far off axis): misses and TIR.  Systems
    # with an even asphere keep the whole bundle tame: 40 mm off axis the random r^4 / r^6
    # terms fold the surface over, the Newton iteration turns chaotic (several roots, or
    # none) and where it ends after the reference's batch-wide 100 iterations is decided by
    # rounding noise -- not something any second implementation can reproduce ray by ray.
    half = n if has_nr else n // 2
    xy = np.concatenate([rng.uniform(-7, 7, (2, half)), rng.uniform(-40, 40, (2, n - half))], 1)
    lm = np.concatenate([rng.uniform(-0.12, 0.12, (2, half)), rng.uniform(-0.55, 0.55, (2, n - half))], 1)
    rays = {"x": xy[0], "y": xy[1], "z": np.full(n, -20.0), "L": lm[0], "M": lm[1]}
    rays.update(N=np.sqrt(1 - lm[0] ** 2 - lm[1] ** 2), i=np.ones(n))
    return table, rays, has_nr


@pytest.mark.parametrize("seed", range(40))
def test_random_system_fp64(seed):
    from oracle import oracle
    from optiland_amd.engine import HipSystem
    table, rays, has_nr = random_system(seed)
    want = oracle.trace(table, rays, 0, record=True)["record"]
    hip = HipSystem(table, DEV)
    try:
        planes = [torch.tensor(rays[k], dtype=torch.float64, device=DEV) for k in PLANES[:7]]
        planes.append(torch.zeros_like(planes[0]))
        got = hip.trace(planes, 0, record=True).record[:, :, :planes[0].numel()].cpu().numpy()
    finally:
        hip.close()
    tol = 1e-7 if has_nr else 1e-9
    assert_close_planes(got, want, tol, tol, f"fuzz{seed}")
    assert np.array_equal(got[:, 6, :] == 0, want[:, 6, :] == 0)


@pytest.mark.parametrize("seed", range(40))
def test_random_system_fp32_on_well_conditioned_rays(seed):
    from oracle import oracle
    from optiland_amd.engine import HipSystem
    table, rays, _ = random_system(seed)
    n = rays["x"].size
    r32 = {k: v.astype(np.float32).astype(np.float64) for k, v in rays.items()}
    want = oracle.trace(table, r32, 0, record=True)["record"]
    # well-conditioned = the oracle's verdict (hit / miss / clipped pattern) survives a
    # 1e-3 mm / 1e-4 rad perturbation of the launch state in every direction tried
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
        # (every plane: total internal reflection leaves the hit finite and kills only
        # the direction)
        stable &= np.all(np.isnan(alt) == np.isnan(want), axis=(0, 1))
        stable &= np.all((alt[:, 6, :] == 0) == (want[:, 6, :] == 0), axis=0)
        # near total internal reflection / grazing: directions move a lot under the nudge
        with np.errstate(invalid="ignore"):
            stable &= np.all(np.nan_to_num(np.abs(alt[:, 3:6, :] - want[:, 3:6, :])) < 1e-2,
                             axis=(0, 1))
    assert stable.sum() > 0.3 * n
    hip = HipSystem(table, DEV)
    try:
        planes = [torch.tensor(r32[k], dtype=torch.float32, device=DEV) for k in PLANES[:7]]
        planes.append(torch.zeros_like(planes[0]))
        got = hip.trace(planes, 0, record=True).record[:, :, :n].double().cpu().numpy()
    finally:
        hip.close()
    assert_close_planes(got[:, :, stable], want[:, :, stable], 1e-4, 1e-4, f"fuzz{seed}:f32")


def random_polarised_system(seed):
    """Conic / plane surfaces with random Simple / Fresnel / Polarizer / Retarder coatings,
    tilts and mirrors; tame bundle (the PRT of a lost ray is NaN in both)."""
    rng = np.random.default_rng(5000 + seed)
    ns = int(rng.integers(2, 6))
    surf = np.zeros(ns + 1, dtype=S.SURFACE_DESC_DTYPE)
    optics = np.zeros((ns + 1, 1), dtype=S.SURFACE_OPTICS_DTYPE)
    coeffs = []
    surf["rot"] = np.eye(3).reshape(-1)
    surf["norm_radius"] = 1.0
    surf[0]["geom_kind"], surf[0]["interaction"] = S.GEOM_PLANE, S.INTERACT_RECORD_ONLY
    surf[0]["origin"] = (0.0, 0.0, -20.0)
    optics[0, 0] = (1.0, 1.0, 0.0)
    n_prev, z, direction = 1.0, 0.0, 1.0
    for i in range(1, ns + 1):
        row = surf[i]
        row["geom_kind"] = rng.choice([S.GEOM_PLANE, S.GEOM_STANDARD, S.GEOM_STANDARD])
        row["radius"] = (rng.choice([-1, 1]) * rng.uniform(30.0, 300.0)
                         if row["geom_kind"] == S.GEOM_STANDARD else np.inf)
        row["conic"] = (rng.choice([0.0, -1.0, rng.uniform(-2.0, 1.0)])
                        if row["geom_kind"] == S.GEOM_STANDARD else 0.0)
        mirror = rng.random() < 0.2
        row["interaction"] = S.INTERACT_REFLECT if mirror else S.INTERACT_REFRACT
        n_next = n_prev if mirror else float(rng.choice([1.0, rng.uniform(1.3, 1.9)]))
        if not mirror and abs(n_next - n_prev) < 0.05:
            # an undeviated ray at a CURVED surface leaves the reference's s = k0 x k1 as
            # pure rounding noise (no longer orthogonal to k0): its PRT there is garbage
            # that no implementation can reproduce -- always step the index
            n_next = n_prev + 0.3 if n_prev < 1.5 else 1.0
        if i == ns:  # image plane: same medium, like every real system
            row["geom_kind"], row["radius"], row["interaction"] = S.GEOM_PLANE, np.inf, S.INTERACT_REFRACT
            n_next = n_prev
        optics[i, 0] = (n_prev, n_next, 0.0)
        n_prev = n_next
        z += direction * rng.uniform(3.0, 20.0)
        if mirror and i != ns:
            direction = -direction
        row["origin"] = (rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), z)
        if rng.random() < 0.3:
            row["rot"] = _rot(rng, 5.0).reshape(-1)
            row["flags"] = S.SURF_ROTATED
        ck = rng.choice([S.COAT_NONE, S.COAT_SIMPLE, S.COAT_FRESNEL, S.COAT_FRESNEL,
                         S.COAT_POLARIZER, S.COAT_RETARDER]) if i != ns else S.COAT_NONE
        row["coating_kind"] = ck
        if ck == S.COAT_SIMPLE:
            row["coat"] = (rng.uniform(0.5, 1.0), rng.uniform(0.0, 0.5))
        elif ck in (S.COAT_POLARIZER, S.COAT_RETARDER):
            ax = rng.normal(size=3)
            ax[2] *= 0.2
            ax /= np.linalg.norm(ax)
            row["coat"] = (float(len(coeffs)), 0.0)
            coeffs.extend(ax.tolist())
            if ck == S.COAT_RETARDER:
                coeffs.append(float(rng.uniform(0.2, 3.0)))
    table = SystemTable(surfaces=surf, coeffs=np.array(coeffs, dtype=np.float64), optics=optics,
                        wavelengths=np.array([0.55]), name=f"polfuzz{seed}")
    table.polarization = {"is_polarized": True, "Ex": 1.0, "Ey": 0.5, "phase_x": 0.0, "phase_y": 0.7}
    n = 1500
    rays = {"x": rng.uniform(-5, 5, n), "y": rng.uniform(-5, 5, n), "z": np.full(n, -20.0)}
    L, M = rng.uniform(-0.1, 0.1, n), rng.uniform(-0.1, 0.1, n)
    rays.update(L=L, M=M, N=np.sqrt(1 - L * L - M * M), i=np.ones(n))
    return table, rays


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(25))
def test_random_polarised_system_with_directions_that_are_not_unit_vectors(seed, dtype):
    """`OL_TRACE_NONUNIT_K` (ABI 11) on the device: the bundles of the reference's iterative /
    robust aimers (|k|^2 - 1 ~ 1e-3) against the oracle, which restates
    polarized_rays.py:136-202 with k as it comes.  The image plane (equal indices: the
    reference's s is rounding noise there) is left out of the range, as integration.py leaves
    it to the reference.  Host twin: tests/test_hostmath_fuzz.py."""
    from oracle import oracle
    from optiland_amd.engine import HipSystem
    from optiland_amd.rays import prt_to_complex
    table, rays = random_polarised_system(seed)
    g = np.random.default_rng(99 + seed)
    scale = 1.0 + 1e-3 * g.uniform(-1.0, 1.0, rays["x"].size)
    rays = dict(rays)
    for k in ("L", "M", "N"):
        rays[k] = rays[k] * scale
    if dtype == torch.float32:
        rays = {k: v.astype(np.float32).astype(np.float64) for k, v in rays.items()}
    n = rays["x"].size
    last = table.num_surfaces - 2
    out = oracle.trace(table, rays, 0, record=True, polarized=True, last=last)
    hip = HipSystem(table, DEV)
    try:
        planes = [torch.tensor(rays[k], dtype=dtype, device=DEV) for k in PLANES[:7]]
        planes.append(torch.zeros_like(planes[0]))
        prt = torch.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype, device=DEV)
        res = hip.trace(planes, 0, record=True, prt=prt, prt_identity=True, last=last,
                        nonunit_directions=True)
        got = res.record[:, :, :n].double().cpu().numpy()
        p = prt_to_complex(prt).cpu().numpy().astype(np.complex128)
    finally:
        hip.close()
    tol = 1e-9 if dtype == torch.float64 else 1e-4
    assert_close_planes(got, out["record"], tol, tol, f"polfuzz{seed}")
    assert np.array_equal(np.isnan(p.real), np.isnan(out["prt"].real))
    np.testing.assert_allclose(np.nan_to_num(p), np.nan_to_num(out["prt"]), rtol=0,
                               atol=tol * 10)


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(25))
def test_random_polarised_system(seed, dtype):
    """Rays, PRT matrices (real or complex planes) and the update_intensity epilogue of
    random coated systems against the oracle: fp64 1e-9, fp32 1e-4."""
    from oracle import oracle
    from optiland_amd.engine import HipSystem
    from optiland_amd.rays import prt_to_complex
    table, rays = random_polarised_system(seed)
    if dtype == torch.float32:
        rays = {k: v.astype(np.float32).astype(np.float64) for k, v in rays.items()}
    n = rays["x"].size
    out = oracle.trace(table, rays, 0, record=True, polarized=True)
    hip = HipSystem(table, DEV)
    try:
        planes = [torch.tensor(rays[k], dtype=dtype, device=DEV) for k in PLANES[:7]]
        planes.append(torch.zeros_like(planes[0]))
        k0 = [planes[3].clone(), planes[4].clone(), planes[5].clone()]
        i0 = planes[6].clone()
        prt = torch.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype, device=DEV)
        res = hip.trace(planes, 0, record=True, prt=prt, prt_identity=True)
        got = res.record[:, :, :n].double().cpu().numpy()
        p = prt_to_complex(prt).cpu().numpy().astype(np.complex128)
        iu = hip.polarized_intensity(prt, k0, i0, table.polarization).double().cpu().numpy()
    finally:
        hip.close()
    tol = 1e-9 if dtype == torch.float64 else 1e-4
    assert_close_planes(got, out["record"], tol, tol, f"polfuzz{seed}")
    assert np.array_equal(np.isnan(p.real), np.isnan(out["prt"].real))
    np.testing.assert_allclose(np.nan_to_num(p), np.nan_to_num(out["prt"]), rtol=0,
                               atol=tol * 10)
    r = rays
    want_i, status = oracle.polarized_intensity(out["prt"], r["L"], r["M"], r["N"], r["i"],
                                                table.polarization)
    assert (status & ~0x20) == 0  # (0x20: OL_STATUS_NAN_DIRECTION, informational)
    np.testing.assert_allclose(np.nan_to_num(iu), np.nan_to_num(want_i), rtol=0, atol=tol * 10)


def random_nr_system(seed):
    """Every Newton-Raphson geometry kind with random (mild) coefficients: odd / even
    asphere, XY polynomial, Chebyshev, biconic, toroidal, Zernike; decentres, tilts,
    apertures; bundle inside the region where the iteration has a unique root."""
    rng = np.random.default_rng(9000 + seed)
    ns = int(rng.integers(2, 5))
    surf = np.zeros(ns + 1, dtype=S.SURFACE_DESC_DTYPE)
    optics = np.zeros((ns + 1, 1), dtype=S.SURFACE_OPTICS_DTYPE)
    coeffs = []
    surf["rot"] = np.eye(3).reshape(-1)
    surf["norm_radius"] = 1.0
    surf[0]["geom_kind"], surf[0]["interaction"] = S.GEOM_PLANE, S.INTERACT_RECORD_ONLY
    surf[0]["origin"] = (0.0, 0.0, -20.0)
    optics[0, 0] = (1.0, 1.0, 0.0)
    kinds = [S.GEOM_EVEN_ASPHERE, S.GEOM_ODD_ASPHERE, S.GEOM_POLYNOMIAL, S.GEOM_CHEBYSHEV,
             S.GEOM_BICONIC, S.GEOM_TOROIDAL, S.GEOM_ZERNIKE]
    n_prev, z = 1.0, 0.0
    for i in range(1, ns + 1):
        row = surf[i]
        kind = int(kinds[(seed + i) % len(kinds)]) if i < ns else S.GEOM_PLANE
        row["geom_kind"] = kind
        row["radius"] = rng.choice([-1, 1]) * rng.uniform(40.0, 300.0) if kind != S.GEOM_PLANE else np.inf
        row["conic"] = rng.choice([0.0, -1.0, rng.uniform(-1.5, 0.8)]) if kind != S.GEOM_PLANE else 0.0
        if kind != S.GEOM_PLANE:
            row["max_iter"], row["tol"] = 100, 1e-12
            row["coeff_offset"] = len(coeffs)
        if kind == S.GEOM_EVEN_ASPHERE:
            c = [rng.uniform(-1e-4, 1e-4), rng.uniform(-1e-6, 1e-6), rng.uniform(-1e-8, 1e-8)]
        elif kind == S.GEOM_ODD_ASPHERE:
            c = [rng.uniform(-1e-3, 1e-3), rng.uniform(-1e-4, 1e-4), rng.uniform(-1e-5, 1e-5),
                 rng.uniform(-1e-6, 1e-6)]
        elif kind == S.GEOM_POLYNOMIAL:
            g = (rng.uniform(-1, 1, (3, 4)) * np.array([[1e-3, 1e-3, 1e-4, 1e-5]])
                 * np.array([[1.0], [1.0], [0.1]]))
            g[0, 0] = 0.0
            row["poly_cols"] = 4
            c = g.reshape(-1).tolist()
        elif kind == S.GEOM_CHEBYSHEV:
            g = rng.uniform(-1, 1, (3, 3)) * 2e-3
            g[0, 0] = 0.0
            row["poly_cols"] = 3
            coeffs.extend([12.0, 14.0])  # norm_x, norm_y (bundle stays inside)
            c = g.reshape(-1).tolist()
        elif kind == S.GEOM_BICONIC:
            c = [float(rng.choice([-1, 1]) * rng.uniform(40.0, 300.0)), float(rng.uniform(-1.2, 0.5))]
        elif kind == S.GEOM_TOROIDAL:
            c = [float(rng.choice([-1, 1]) * rng.uniform(40.0, 300.0)), float(rng.uniform(-1.0, 0.5)),
                 rng.uniform(-1e-4, 1e-4), rng.uniform(-1e-6, 1e-6)]
            row["conic"] = 0.0
        elif kind == S.GEOM_ZERNIKE:
            row["norm_radius"] = 12.0
            terms = [(1, 1), (1, -1), (2, 0), (2, 2), (2, -2), (3, 1), (3, -1), (4, 0), (3, 3)]
            c = []
            for (nn, mm) in terms:
                c.extend([float(rng.uniform(-5e-4, 5e-4)), float(nn), float(mm),
                          float(rng.choice([1.0, np.sqrt(2.0 * (nn + 1))]))])
        else:
            c = []
        if kind == S.GEOM_ZERNIKE:
            row["n_coeff"] = len(c) // 4
        elif kind == S.GEOM_CHEBYSHEV:
            row["n_coeff"] = len(c)
        else:
            row["n_coeff"] = len(c)
        coeffs.extend(c)
        row["interaction"] = S.INTERACT_REFRACT
        n_next = float(rng.uniform(1.3, 1.9)) if n_prev == 1.0 else 1.0
        if i == ns:
            n_next = n_prev
        optics[i, 0] = (n_prev, n_next, 0.0)
        n_prev = n_next
        z += rng.uniform(4.0, 15.0)
        row["origin"] = (rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), z)
        if rng.random() < 0.3:
            row["rot"] = _rot(rng, 3.0).reshape(-1)
            row["flags"] = S.SURF_ROTATED
        if rng.random() < 0.4:
            row["aperture_kind"] = S.AP_RADIAL
            row["aperture"] = (0.0, rng.uniform(4.0, 7.0), 0, 0)
    table = SystemTable(surfaces=surf, coeffs=np.array(coeffs, dtype=np.float64), optics=optics,
                        wavelengths=np.array([0.55]), name=f"nrfuzz{seed}")
    n = 2000
    rays = {"x": rng.uniform(-5, 5, n), "y": rng.uniform(-5, 5, n), "z": np.full(n, -20.0)}
    L, M = rng.uniform(-0.08, 0.08, n), rng.uniform(-0.08, 0.08, n)
    rays.update(L=L, M=M, N=np.sqrt(1 - L * L - M * M), i=np.ones(n))
    return table, rays


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(28))
def test_random_newton_raphson_system(seed, dtype):
    from oracle import oracle
    from optiland_amd.engine import HipSystem
    table, rays = random_nr_system(seed)
    if dtype == torch.float32:
        rays = {k: v.astype(np.float32).astype(np.float64) for k, v in rays.items()}
    n = rays["x"].size
    want = oracle.trace(table, rays, 0, record=True)
    assert want["status"] == 0
    hip = HipSystem(table, DEV)
    try:
        planes = [torch.tensor(rays[k], dtype=dtype, device=DEV) for k in PLANES[:7]]
        planes.append(torch.zeros_like(planes[0]))
        got = hip.trace(planes, 0, record=True).record[:, :, :n].double().cpu().numpy()
    finally:
        hip.close()
    assert not np.isnan(want["record"]).all()
    tol = 1e-7 if dtype == torch.float64 else 1e-4
    assert_close_planes(got, want["record"], tol, tol, f"nrfuzz{seed}")


def _random_aperture_tokens(rng, coeffs, depth=0):
    """Random boolean aperture tree -> reverse-Polish token list (packer._flatten_aperture)."""
    if depth >= 3 or rng.random() < 0.45:
        kind = rng.choice([S.AP_RADIAL, S.AP_OFFSET_RADIAL, S.AP_RECTANGULAR, S.AP_ELLIPTICAL,
                           S.AP_POLYGON])
        if kind == S.AP_RADIAL:
            return [[kind, rng.choice([0.0, 1.5]), rng.uniform(3.0, 8.0), 0.0, 0.0]]
        if kind == S.AP_OFFSET_RADIAL:
            return [[kind, 0.0, rng.uniform(2.0, 6.0), rng.uniform(-2, 2), rng.uniform(-2, 2)]]
        if kind == S.AP_RECTANGULAR:
            return [[kind, -rng.uniform(1, 7), rng.uniform(1, 7), -rng.uniform(1, 7), rng.uniform(1, 7)]]
        if kind == S.AP_ELLIPTICAL:
            return [[kind, rng.uniform(2, 8), rng.uniform(2, 8), rng.uniform(-1, 1), rng.uniform(-1, 1)]]
        nv = int(rng.integers(3, 9))
        th = np.sort(rng.uniform(0, 2 * np.pi, nv))
        rr = rng.uniform(2.0, 8.0, nv)
        off = len(coeffs)
        for a, b in zip(rr * np.cos(th), rr * np.sin(th)):
            coeffs.extend([float(a), float(b)])
        return [[kind, float(off), float(nv), 0.0, 0.0]]
    op = rng.choice([S.AP_OP_UNION, S.AP_OP_INTERSECTION, S.AP_OP_DIFFERENCE])
    return (_random_aperture_tokens(rng, coeffs, depth + 1)
            + _random_aperture_tokens(rng, coeffs, depth + 1) + [[op, 0.0, 0.0, 0.0, 0.0]])


@pytest.mark.parametrize("seed", range(30))
def test_random_aperture_trees_clip_exactly(seed):
    """Random boolean trees over every leaf kind (incl. polygons) on a two-surface lens:
    the clipped mask (i == 0) must equal the oracle's ray for ray, fp64; hit coordinates
    to 1e-9."""
    from oracle import oracle
    from optiland_amd.engine import HipSystem
    rng = np.random.default_rng(20_000 + seed)
    surf = np.zeros(4, dtype=S.SURFACE_DESC_DTYPE)
    optics = np.zeros((4, 1), dtype=S.SURFACE_OPTICS_DTYPE)
    surf["rot"] = np.eye(3).reshape(-1)
    surf["norm_radius"] = 1.0
    coeffs = []
    surf[0]["interaction"] = S.INTERACT_RECORD_ONLY
    surf[0]["origin"] = (0, 0, -10.0)
    optics[0, 0] = (1.0, 1.0, 0.0)
    for i, (R, n1, n2, z) in enumerate(((60.0, 1.0, 1.6, 0.0), (-70.0, 1.6, 1.0, 5.0)), start=1):
        surf[i]["geom_kind"], surf[i]["radius"] = S.GEOM_STANDARD, R
        surf[i]["interaction"] = S.INTERACT_REFRACT
        surf[i]["origin"] = (0, 0, z)
        optics[i, 0] = (n1, n2, 0.0)
        toks = _random_aperture_tokens(rng, coeffs)
        if len(toks) == 1:
            surf[i]["aperture_kind"] = int(toks[0][0])
            surf[i]["aperture"] = toks[0][1:]
        else:
            surf[i]["aperture_kind"] = S.AP_COMPOSITE
            surf[i]["aperture"] = (float(len(coeffs)), float(len(toks)), 0.0, 0.0)
            for t in toks:
                coeffs.extend(float(v) for v in t)
    surf[3]["geom_kind"], surf[3]["interaction"] = S.GEOM_PLANE, S.INTERACT_REFRACT
    surf[3]["origin"] = (0, 0, 40.0)
    optics[3, 0] = (1.0, 1.0, 0.0)
    table = SystemTable(surfaces=surf, coeffs=np.array(coeffs, dtype=np.float64), optics=optics,
                        wavelengths=np.array([0.55]), name=f"apfuzz{seed}")
    n = 6000
    rays = {"x": rng.uniform(-9, 9, n), "y": rng.uniform(-9, 9, n), "z": np.full(n, -10.0),
            "L": np.zeros(n), "M": np.zeros(n), "N": np.ones(n), "i": np.ones(n)}
    want = oracle.trace(table, rays, 0, record=True)["record"]
    hip = HipSystem(table, DEV)
    try:
        planes = [torch.tensor(rays[k], dtype=torch.float64, device=DEV) for k in PLANES[:7]]
        planes.append(torch.zeros_like(planes[0]))
        got = hip.trace(planes, 0, record=True).record[:, :, :n].cpu().numpy()
        # the fused spot kernel shares the aperture code through a different instantiation
    finally:
        hip.close()
    assert np.array_equal(got[:, 6, :] == 0, want[:, 6, :] == 0)
    assert_close_planes(got, want, 1e-9, 1e-9, f"apfuzz{seed}")


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(30))
def test_random_ray_generation(seed, dtype):
    """ol_generate_rays against the oracle over random generator scalars: the three field
    kinds, object at infinity or finite, telecentric, every apodization, per-ray planes
    and launch-uniform scalars, vignetting factors and trace_generic's pupil pre-scale."""
    from oracle import oracle
    from optiland_amd import _capi
    from optiland_amd.engine import HipSystem
    from optiland_amd import load_system
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
        rg["apod_b"] = float(rng.uniform(2.0, 6.0))   # super-Gaussian order n >= 2
    if int(rg["apod_kind"]) == 4:
        rg["apod_b"] = float(rng.uniform(0.5, 3.0))   # polynomial power
    table.raygen = rg
    hip = HipSystem(table, DEV)
    try:
        n = 3001
        r, th = np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
        px, py = r * np.cos(th), r * np.sin(th)
        hx, hy = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        vx, vy = rng.uniform(0.7, 1.0, n), rng.uniform(0.7, 1.0, n)
        dev = lambda a: torch.as_tensor(a, dtype=dtype, device=DEV)  # noqa: E731
        seen = lambda a: dev(a).double().cpu().numpy()  # noqa: E731
        prescale = bool(rng.random() < 0.5)
        flags = _capi.RAYGEN_PRESCALE_PUPIL if prescale else 0
        got = hip.generate_rays(dev(hx), dev(hy), dev(px), dev(py), dev(vx), dev(vy), flags=flags)
        pxs, pys = (seen(px) * seen(vx), seen(py) * seen(vy)) if prescale else (seen(px), seen(py))
        want = oracle.generate_rays(rg, seen(hx), seen(hy), pxs, pys, seen(vx), seen(vy))
        scale = max(1.0, abs(rg["z_first"]), rg["EPD"], rg["offset"] + rg["EPL"])
        tol = 1e-12 if dtype == torch.float64 else 2e-6
        for k, g in zip(("x", "y", "z", "L", "M", "N", "i"), got):
            gv = g.double().cpu().numpy()
            s_ = scale if k in "xyz" else 1.0
            if dtype == torch.float32 and k == "i":
                # a pupil point within fp32 rounding of an apodization edge may fall on
                # either side of it: compare away from the edges
                ok = np.abs(gv - want[k]) < 1e-3
                assert ok.mean() > 0.995
                continue
            np.testing.assert_allclose(gv, want[k], rtol=0, atol=tol * s_ * 10, err_msg=k)
        # launch-uniform scalars == constant planes
        c = np.ones(n)
        a = hip.generate_rays(0.3, -0.4, dev(px), dev(py), 0.9, 0.8)
        b = hip.generate_rays(dev(0.3 * c), dev(-0.4 * c), dev(px), dev(py), dev(0.9 * c), dev(0.8 * c))
        for u, v in zip(a, b):
            assert float((u - v).abs().max()) <= (1e-5 if dtype == torch.float32 else 1e-12) * scale
    finally:
        hip.close()


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(28))
def test_random_newton_raphson_system_polarised(seed, dtype):
    """The Newton-Raphson family with Fresnel / Simple coatings and a polarised state
    (the POLK x NR kernel instantiations of configuration C5): rays, PRT and the
    update_intensity epilogue against the oracle."""
    from oracle import oracle
    from optiland_amd.engine import HipSystem
    from optiland_amd.rays import prt_to_complex
    table, rays = random_nr_system(seed)
    rng = np.random.default_rng(40_000 + seed)
    for i in range(1, table.num_surfaces - 1):
        ck = rng.choice([S.COAT_FRESNEL, S.COAT_FRESNEL, S.COAT_SIMPLE, S.COAT_NONE])
        table.surfaces[i]["coating_kind"] = ck
        if ck == S.COAT_SIMPLE:
            table.surfaces[i]["coat"] = (rng.uniform(0.6, 1.0), rng.uniform(0.0, 0.4))
    table.polarization = {"is_polarized": bool(seed % 2), "Ex": 0.8, "Ey": 0.6, "phase_x": 0.3,
                          "phase_y": -0.4}
    if dtype == torch.float32:
        rays = {k: v.astype(np.float32).astype(np.float64) for k, v in rays.items()}
    n = rays["x"].size
    out = oracle.trace(table, rays, 0, record=True, polarized=True)
    assert (out["status"] & ~0x20) == 0
    hip = HipSystem(table, DEV)
    try:
        planes = [torch.tensor(rays[k], dtype=dtype, device=DEV) for k in PLANES[:7]]
        planes.append(torch.zeros_like(planes[0]))
        k0 = [planes[3].clone(), planes[4].clone(), planes[5].clone()]
        i0 = planes[6].clone()
        prt = torch.empty((9, n), dtype=dtype, device=DEV)
        res = hip.trace(planes, 0, record=True, prt=prt, prt_identity=True)
        got = res.record[:, :, :n].double().cpu().numpy()
        p = prt_to_complex(prt).cpu().numpy().astype(np.complex128)
        iu = hip.polarized_intensity(prt, k0, i0, table.polarization).double().cpu().numpy()
    finally:
        hip.close()
    tol = 1e-7 if dtype == torch.float64 else 1e-4
    assert_close_planes(got, out["record"], tol, tol, f"nrpol{seed}")
    assert np.array_equal(np.isnan(p.real), np.isnan(out["prt"].real))
    np.testing.assert_allclose(np.nan_to_num(p), np.nan_to_num(out["prt"]), rtol=0, atol=tol * 10)
    want_i, status = oracle.polarized_intensity(out["prt"], rays["L"], rays["M"], rays["N"],
                                                rays["i"], table.polarization)
    assert (status & ~0x20) == 0  # (0x20: OL_STATUS_NAN_DIRECTION, informational)
    np.testing.assert_allclose(np.nan_to_num(iu), np.nan_to_num(want_i), rtol=0, atol=tol * 10)


def _random_raygen(rng, table):
    """Generator scalars that put a tame bundle onto a `random_nr_system` lens."""
    z1 = float(table.surfaces[1]["origin"][2])
    infinite = bool(rng.random() < 0.5)
    kind = int(rng.integers(0, 2)) if not infinite else int(rng.choice([0, 2]))
    rg = {"object_infinite": 1.0 if infinite else 0.0, "field_kind": float(kind),
          "EPL": z1 + float(rng.uniform(0.0, 6.0)), "EPD": float(rng.uniform(3.0, 7.0)),
          "offset": float(rng.uniform(5, 30)) if infinite else 0.0,
          "z_first": z1 if infinite else z1 - float(rng.uniform(40, 120)),
          "tele_dz": 0.0, "apod_kind": float(rng.integers(0, 7)),
          "apod_a": float(rng.uniform(0.6, 1.2)), "apod_b": float(rng.uniform(0.3, 0.9))}
    rg["max_field"] = float(rng.uniform(0.5, 3.0))          # degrees or mm
    if kind == 2:
        rg["field_scale"] = float(rng.uniform(0.005, 0.04))  # slope per unit H
    if kind == 1 and rng.random() < 0.5:
        rg["tele_dz"] = float(rng.uniform(20, 60))           # object-space telecentric
    if int(rg["apod_kind"]) == 5:
        rg["apod_b"] = float(rng.uniform(2.0, 6.0))
    if int(rg["apod_kind"]) == 4:
        rg["apod_b"] = float(rng.uniform(0.5, 3.0))
    return rg


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("seed", range(28))
def test_random_fused_spot(seed, dtype):
    """generate -> trace -> reduce in ONE kernel (`ol_trace_spot`: its own instantiations
    per Newton-Raphson class, field planes / uniform field, apodization on / off) on random
    Newton-Raphson-family lenses with random generator scalars, against the oracle's
    generate + trace + numpy moments; per-ray hits as well as the seven moments."""
    from tests.test_gpu_spot import _check_moments, _oracle_spot, _scale
    from optiland_amd.engine import HipSystem
    table, _ = random_nr_system(seed)
    rng = np.random.default_rng(50_000 + seed)
    table.raygen = _random_raygen(rng, table)
    n = 20_011
    r, th = np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    dev = lambda a: torch.as_tensor(a, dtype=dtype, device=DEV)  # noqa: E731
    px, py = dev(r * np.cos(th)), dev(r * np.sin(th))
    planes = bool(seed % 2)              # per-ray field planes or one launch-uniform field
    if planes:
        hx, hy = dev(rng.uniform(-1, 1, n)), dev(rng.uniform(-1, 1, n))
        vx, vy = dev(rng.uniform(0.8, 1.0, n)), dev(rng.uniform(0.8, 1.0, n))
    else:
        f, v = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))), (0.95, 0.9)
        one = torch.ones(n, dtype=dtype, device=DEV)
        hx, hy, vx, vy = one * f[0], one * f[1], one * v[0], one * v[1]
    center = (float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.2, 0.2)))
    want, (wx, wy, wi) = _oracle_spot(table, 0, hx, hy, px, py, vx, vy, center)
    hits = [torch.empty(n, dtype=dtype, device=DEV) for _ in range(3)]
    hip = HipSystem(table, DEV)
    try:
        if planes:
            got = hip.trace_spot(px, py, 0, hx=hx, hy=hy, vx=vx, vy=vy, center=center, hits=hits)
        else:
            got = hip.trace_spot(px, py, 0, field=f, vig=v, center=center, hits=hits)
        got = got.cpu().numpy()
    finally:
        hip.close()
    gx, gy, gi = (h.double().cpu().numpy() for h in hits)
    scale = _scale(table, wx, wy)
    assert want[0] > 0.05 * n, "bundle lost: the fuzz case tests nothing"
    if dtype == torch.float64:
        assert np.array_equal(gi > 0, wi > 0)
        _check_moments(got, want, scale, 1e-7)
        tol = 1e-7
    else:
        assert ((gi > 0) == (wi > 0)).mean() > 0.998   # rim rays may round either way
        tol = 1e-4
    both = (gi > 0) & (wi > 0)
    np.testing.assert_allclose(gx[both], wx[both], rtol=0, atol=tol * scale)
    np.testing.assert_allclose(gy[both], wy[both], rtol=0, atol=tol * scale)
    np.testing.assert_allclose(gi[both], wi[both], rtol=0, atol=max(tol, 1e-9) * 10)


@pytest.mark.parametrize("seed", range(20))
def test_random_wavefront_opd(seed):
    """`ol_wavefront_opd` (reference-sphere path length + tilt removal, wavefront/
    strategy.py:83-139,163-215) against the oracle on random image-plane bundles and
    random reference spheres, fp64: OPD to 1e-9 waves (relative to its magnitude),
    pupil coordinates to 1e-10 of the sphere radius."""
    from oracle import oracle
    from optiland_amd.engine import HipSystem
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
    if seed % 4 == 3:   # ABI 4: planar reference through (xc, yc, zc) with this normal
        nv = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), 1.0])
        nv /= np.linalg.norm(nv)
        params.update(zc=zc - abs(R), nx=float(nv[0]), ny=float(nv[1]), nz=float(nv[2]))
    L, M = rng.uniform(-0.15, 0.15, n), rng.uniform(-0.15, 0.15, n)
    rays7 = [rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), np.full(n, zc),
             L, M, np.sqrt(1 - L * L - M * M), rng.uniform(95, 105, n)]
    r, th = np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    px, py = r * np.cos(th), r * np.sin(th)
    want, want_pupil = oracle.wavefront_opd(params, rays7, px, py)
    hip = HipSystem(load_system("cooke_generic"), DEV)
    try:
        dev = lambda a: torch.as_tensor(a, dtype=torch.float64, device=DEV).contiguous()  # noqa: E731
        got, pupil = hip.wavefront_opd(params, [dev(a) for a in rays7], dev(px), dev(py))
        got, pupil = got.cpu().numpy(), pupil.cpu().numpy()
    finally:
        hip.close()
    assert np.isfinite(want).all()
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-9 * max(1.0, np.abs(want).max()))
    np.testing.assert_allclose(pupil, want_pupil, rtol=0, atol=1e-10 * abs(R))


@pytest.mark.parametrize("seed", range(36))
def test_polarised_zernike_pair_equals_the_one_ray_form(seed):
    """Configuration C5's generating launch on PAIRS of fp32 rays (OL_POLZ_PAIR: two rays per
    lane as one f32x2, csrc/surface_math.h "pair forms") against the one-ray-per-lane form of the
    same launch ON THE DEVICE: record, PRT planes, updated intensity and status bit for bit, on
    the shaken C5 systems of tests/test_polz_pair.py (the host twin)."""
    from optiland_amd.engine import HipSystem
    from tests import test_polz_pair as tp
    table = tp.random_c5_table(seed)
    rng = np.random.default_rng(seed)
    hip = HipSystem(table, DEV)
    try:
        n = int(rng.choice([2, 254, 1000, 4098, 100_000]))
        px, py = tp._pupil(n, rng, reach=1.0 if seed % 3 else 1.08)
        px, py = px.to(DEV), py.to(DEV)
        state = tp.POLARISED if seed % 2 else tp.STATE
        kw = dict(field=(float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))),
                  update_intensity=state if seed % 3 != 1 else None)
        one = tp._launch(hip, px, py, 0, 1, **kw)
        pair = tp._launch(hip, px, py, 0, 3, **kw)
        tp.assert_same_bits(one, pair, f"seed {seed}")
    finally:
        hip.close()


def test_polarised_zernike_pair_with_field_planes_on_the_device():
    """... and with per-ray field planes, vignetting planes and an apodized pupil: each ray of a
    lane generated from ITS field point (tests/test_polz_pair.py: field_plane_cases)."""
    from optiland_amd.engine import HipSystem
    from tests import test_polz_pair as tp
    tp.field_plane_cases(lambda table: HipSystem(table, DEV), DEV)
