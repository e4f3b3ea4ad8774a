"""Round 5: the Newton stop rule on a ray that STARTS badly.

A random lens of the differential fuzz (tools/make_fuzz_tables.py seed 7016, built through the
reference's public API): its first surface is an oblate biconic mirror (Rx = -33.8, kx = 0.40)
lit at up to 32 degrees; rays that hit within a millimetre of the rim, where the surface is
nearly vertical, see the residuals 4.0, 2.3, 0.97, 0.13, 2e-3, 7e-7, 1e-13 mm.  The reference
(geometries/newton_raphson.py:119-168) iterates on and converges; the kernel's per-ray rule
"leave when the residual no longer halves" -- meant for the rounding floor of fp32 -- used to
cut such a ray off after its second step, 1 mm off the surface (28 of 60 000 random systems of
tools/host_long_fuzz.py were this).  Since round 5 the rule applies at the rounding floor only.
"""
import os

import numpy as np
import pytest
import torch

from oracle import oracle
from optiland_amd.system import SystemTable
from tests._util import GOLDEN

WHERE = [pytest.param("cuda", marks=pytest.mark.gpu), "host"]


def _engine(table, where):
    if where == "cuda":
        from optiland_amd.engine import HipSystem
        return HipSystem(table, "cuda:0")
    from tests import _hostmath as hm
    if not hm.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    return hm.make_engine_class()(table, "cpu")


def _bundle(table, n=3000, seed=7016):
    rng = np.random.default_rng(90_000 + seed)
    r, th = np.sqrt(rng.random(n)) * 0.9, 2 * np.pi * rng.random(n)
    px, py = r * np.cos(th), r * np.sin(th)
    hx, hy = np.full(n, rng.uniform(-0.6, 0.6)), np.full(n, rng.uniform(-1, 1))
    g = oracle.generate_rays(table.raygen, hx, hy, px, py)
    g["opd"] = np.zeros(n)
    return g


@pytest.mark.parametrize("where", WHERE)
def test_rim_rays_of_a_biconic_mirror_converge_like_the_reference(where):
    table = SystemTable.load(os.path.join(GOLDEN, "fuzz_r05_biconic_rim.json"))
    g = _bundle(table)
    want = oracle.trace(table, g, 0, record=True)["record"]
    # the known-answer ray: the reference's loop restated in NumPy on this ray gives t = 57.159891
    # and a residual of 4e-14 at the recorded point
    j = 2529
    np.testing.assert_allclose(want[1, :3, j], [29.055302, -7.714972, -20.655066], atol=2e-6)
    assert abs(want[1, 7, j] - 57.159891) < 1e-6 and want[1, 6, j] > 0.9   # alive at the mirror
    hip = _engine(table, where)
    dev = hip.device
    try:
        rays = [torch.as_tensor(g[k], dtype=torch.float64, device=dev).contiguous()
                for k in ("x", "y", "z", "L", "M", "N", "i", "opd")]
        got = hip.trace(rays, 0, record=True).record[:, :, :rays[0].numel()].cpu().numpy()
    finally:
        hip.close()
    # (the reference leaves its batch at max |f| < 1e-6 mm; at the rim that much of position is
    # 5e-8 of direction)
    np.testing.assert_allclose(got[1, :, j], want[1, :, j], rtol=0, atol=1e-6)
    # every ray that is alive AT the mirror lands where the reference's does; the rim rays
    # (sqrt term of the x profile below 0.02) are among them
    sf = table.surfaces[1]
    cx, kx = 1.0 / float(sf["radius"]), float(sf["conic"])
    R, o = np.array(sf["rot"]).reshape(3, 3), np.array(sf["origin"])
    loc = R @ (want[1, :3] - o[:, None])
    rim = (1.0 - (1.0 + kx) * cx * cx * loc[0] ** 2 < 0.02) & np.isfinite(want[1, 0]) \
        & (want[1, 6] > 0)
    assert rim.sum() >= 3 and rim[j]
    alive = np.isfinite(want[1, 0]) & (want[1, 6] > 0)
    np.testing.assert_allclose(got[1][:, alive], want[1][:, alive], rtol=0, atol=2e-6)


@pytest.mark.parametrize("where", WHERE)
def test_total_reflection_at_the_last_surface_is_reported(where):
    """Round 5: OL_STATUS_NAN_DIRECTION -- informational, raises nothing: a ray left the last
    traced surface with a position and no direction (total internal reflection there; a conic,
    polarised lens of the seam fuzz kept as tests/golden/fuzz_r05_tir_at_last_surface.json).
    The drop-in needs the bit to end `Optic.trace` the way the reference does, with `x += 0 L`
    (tests/test_reference_integration.py::test_a_ray_without_a_direction…)."""
    from optiland_amd import system as S
    table = SystemTable.load(os.path.join(GOLDEN, "fuzz_r05_tir_at_last_surface.json"))
    hip = _engine(table, where)
    dev = hip.device
    try:
        d = torch.linspace(-0.9, 0.9, 41, dtype=torch.float64, device=dev)
        px, py = (t.reshape(-1).contiguous() for t in torch.meshgrid(d, d, indexing="ij"))
        n = px.numel()
        prt = torch.empty((18 if table.needs_complex_prt else 9, n), dtype=torch.float64, device=dev)
        for field, scale, expect in (((0.3, -0.5), 1.0, True), ((0.0, 0.0), 0.05, False)):
            res = hip.trace_generate((px * scale).contiguous(), (py * scale).contiguous(), 0,
                                     field=field, prt=prt, defer_status=True)
            bits = int(hip._status.item())
            hip.raise_for_status(bits)                      # ... and nothing is raised
            last = res.record[-1, :, :n]
            lost = torch.isnan(last[3]) & torch.isfinite(last[0])
            assert bool(lost.any()) == expect, field
            assert bool(bits & S.STATUS_NAN_DIRECTION) == expect, field
    finally:
        hip.close()
