"""`OL_SURF_REFERENCE_ROOT` (ABI 10, opt-in: `integration.enable(reference_root=True)`): the conic
intersection in the reference's OWN form, `(-b +- sqrt(d)) / 2a` with R-scaled coefficients
(geometries/standard.py:112-146), instead of the kernel's cancellation-free one.

Where the reference's formula is ill conditioned it carries a systematic error and its goldens
encode it (tests/test_operand.py::test_opd_diff_on_axis on the Hubble: 0.00132951 waves with
the reference's form, 0.00132994 with the stable one, tolerance 1.1e-7).  These tests hold the
option to what it promises on two ill-conditioned surfaces: WITH the flag the kernel source
follows the oracle (a restatement of the reference's formula) far more closely than without,
WITHOUT it the kernel is the one that is closer to a long-double evaluation.  `host` = the host
build of the kernel source (runs without a GPU), `cuda` = the device.
"""

import math

import numpy as np
import pytest
import torch

from optiland_amd import system as S
from optiland_amd.system import SystemTable
from tests import _hostmath as hm
from tests._util import PLANES

WHERE = [pytest.param("host", id="host"), pytest.param("cuda", marks=pytest.mark.gpu, id="cuda")]


def _table(radius, conic, flag, interaction=S.INTERACT_REFLECT):
    desc = np.zeros(2, dtype=S.SURFACE_DESC_DTYPE)
    optics = np.zeros((2, 1), dtype=S.SURFACE_OPTICS_DTYPE)
    desc["rot"] = np.eye(3).reshape(-1)
    desc["norm_radius"] = 1.0
    desc[0]["geom_kind"] = S.GEOM_PLANE
    desc[0]["interaction"] = S.INTERACT_RECORD_ONLY
    desc[0]["origin"] = (0, 0, -math.inf)
    optics[0, 0] = (1.0, 1.0, 0.0)
    d = desc[1]
    d["geom_kind"] = S.GEOM_STANDARD
    d["interaction"] = interaction
    d["radius"], d["conic"] = radius, conic
    d["flags"] = S.SURF_REFERENCE_ROOT if flag else 0
    optics[1, 0] = (1.0, 1.0 if interaction == S.INTERACT_REFLECT else 1.5, 0.0)
    return SystemTable(surfaces=desc, coeffs=np.zeros(0), optics=optics,
                       wavelengths=np.array([0.55]))


def _engine(table, where):
    if where == "host":
        if not hm.available():
            pytest.skip("hipcc (used as host C++ compiler) missing")
        return hm.make_engine_class()(table), "cpu"
    from optiland_amd.engine import HipSystem
    return HipSystem(table, "cuda:0"), "cuda:0"


def _bundle(n, seed, half_width):
    g = np.random.default_rng(seed)
    x, y = g.uniform(-half_width, half_width, (2, n))
    L, M = g.normal(0, 2e-4, (2, n))       # nearly axial: where |a| = |1 + k| bites
    N = np.sqrt(1 - L * L - M * M)
    return dict(x=x, y=y, z=np.full(n, -100.0), L=L, M=M, N=N, i=np.ones(n))


def _t_long_double(rays, R, k):
    ld = np.longdouble
    x, y, z, L, M, N = (rays[q].astype(ld) for q in ("x", "y", "z", "L", "M", "N"))
    R, k = ld(R), ld(k)
    a = k * N * N + L * L + M * M + N * N
    b = 2 * k * N * z + 2 * L * x + 2 * M * y - 2 * N * R + 2 * N * z
    c = k * z * z - 2 * R * z + x * x + y * y + z * z
    sq = np.sqrt(b * b - 4 * a * c)
    q = -(b + np.copysign(sq, b)) / 2          # cancellation-free pair of roots
    t1, t2 = q / a, c / q
    return np.where(np.abs(z + t1 * N) <= np.abs(z + t2 * N), t1, t2)


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("radius,conic,half_width", [(-11040.0, -1.0023, 1200.0),   # Hubble primary
                                                     (1e7, 0.0, 3.0)])              # near-flat sphere
def test_reference_root_follows_the_reference_formula(radius, conic, half_width, where):
    from oracle import oracle
    n = 4096
    rays = _bundle(n, 3, half_width)
    want = oracle.trace(_table(radius, conic, False), rays, 0, record=True)["record"][1, 2]
    exact = np.asarray(rays["z"] + _t_long_double(rays, radius, conic) * rays["N"], dtype=np.float64)
    got = {}
    for flag in (False, True):
        eng, dev = _engine(_table(radius, conic, flag), where)
        try:
            planes = [torch.tensor(np.asarray(rays[q], dtype=np.float64), dtype=torch.float64,
                                   device=dev) for q in PLANES[:7]]
            planes.append(torch.zeros(n, dtype=torch.float64, device=dev))
            res = eng.trace(planes, 0, record=True)
            got[flag] = res.record[1, 2, :n].double().cpu().numpy()
        finally:
            eng.close()
    err_ref_formula = np.max(np.abs(want - exact))          # what the reference is off by
    assert err_ref_formula > 1e-10, err_ref_formula          # ... a real, ill-conditioned case
    # default: the kernel is the accurate side
    assert np.max(np.abs(got[False] - exact)) < err_ref_formula / 20
    # with the flag: the kernel carries the REFERENCE's error (differs from the oracle by
    # rounding only -- another sqrt / quotient sequence -- not by the conditioning)
    assert np.max(np.abs(got[True] - want)) < err_ref_formula / 20
    assert np.max(np.abs(got[True] - exact)) > err_ref_formula / 2


def test_flag_is_ignored_where_it_does_not_apply():
    """Planes, infinite radii and Newton geometries keep their own intersection: the flag on
    them changes nothing (the packer sets it on every row when the option is on)."""
    if not hm.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    from tests._util import load_case
    table, data = load_case("aspheric_singlet")
    r = data["rays_in"]
    n = r.shape[1]
    out = []
    for flag in (0, S.SURF_REFERENCE_ROOT):
        import copy
        tb = copy.deepcopy(table)
        keep = tb.surfaces["geom_kind"] != S.GEOM_STANDARD
        tb.surfaces["flags"][keep] |= flag
        eng = hm.make_engine_class()(tb)
        planes = [torch.tensor(r[q], dtype=torch.float64) for q in range(7)]
        planes.append(torch.zeros(n, dtype=torch.float64))
        out.append(eng.trace(planes, 0, record=True).record[:, :, :n].clone())
        eng.close()
    assert torch.equal(out[0].nan_to_num(), out[1].nan_to_num())
