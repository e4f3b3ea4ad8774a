"""The DEVICE against the host run of its own source (needs an MI355X).

tests/hostmath executes `optiland_amd/csrc/surface_math.h` on the host; here the kernel and
that host run trace the same rays of every golden system and are compared ray by ray.  The
two share every formula and branch and differ only in what a GPU adds: the launch glue
(plane addressing, record rows, the write-only PRT) and the hardware's 1-ulp
`v_rcp / v_sqrt / v_rsq / v_exp_f32`.  So this holds the device far tighter than any
comparison with an independent implementation can:

    fp64   1e-13 of the group scale (IEEE divide / sqrt on both sides; measured on an
           MI355X: <= 8e-16 on every golden system, Newton systems included)
    fp32   8 x the margin the kernel was measured to have against the goldens
           (tests/golden/fp32_margins.json), i.e. the rounding noise of the path itself

and it checks the glue independently of the oracle.  The measured deviations are written to
gpurun_out/hostmath_vs_device.json.
"""

import json
import os

import numpy as np
import pytest
import torch

from tests import _hostmath as hm
from tests._util import assert_close_planes, fp32_group_tolerances, golden_cases, load_case

pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not hm.available(), reason="hipcc (host C++ compiler) missing")]

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SEEN = {}


@pytest.fixture(scope="module")
def both():
    from optiland_amd.engine import HipSystem
    cache = {}

    def get(case):
        if case not in cache:
            table, data = load_case(case)
            cache[case] = (HipSystem(table, "cuda:0"), hm.HostMathSystem(table), table, data)
        return cache[case]

    yield get
    for dev, host, _, _ in cache.values():
        dev.close()
        host.close()
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "hostmath_vs_device.json"), "w") as f:
        json.dump(_SEEN, f, indent=1, sort_keys=True)


def _deviation(dev, host):
    """max |dev - host| / group scale, per group (the tolerance model of assert_close_planes)."""
    out = {}
    groups = {"pos": (0, 1, 2), "dir": (3, 4, 5), "i": (6,), "opd": (7,)}
    for g, planes in groups.items():
        h = host[:, list(planes), :]
        d = dev[:, list(planes), :]
        fin = np.isfinite(h)
        if not fin.any():
            continue
        scale = np.max(np.abs(h[fin]))
        out[g] = float(np.max(np.abs(d[fin] - h[fin])) / scale) if scale > 0 else 0.0
    return out


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("case", golden_cases())
def test_device_equals_host_run_of_the_same_source(both, case, dtype):
    from optiland_amd.rays import new_prt
    dev, host, table, data = both(case)
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    r = data["rays_in"]
    n = r.shape[1]
    hrays = [np.array(r[k], dtype=dtype, order="C", copy=True) for k in range(7)]
    hrays.append(np.zeros(n, dtype=dtype))
    drays = [torch.tensor(hrays[k], device="cuda:0") for k in range(8)]
    polarized = "prt" in data
    hprt = dprt = None
    if polarized:
        hprt = hm.new_prt(n, dtype, table.needs_complex_prt)
        dprt = new_prt(n, tdt, "cuda:0", table.needs_complex_prt)
    hrec, hstatus = host.trace(hrays, 0, record=True, prt=hprt)
    res = dev.trace(drays, 0, record=True, prt=dprt)
    drec = res.record[:, :, :n].cpu().numpy()
    assert (hstatus & ~0x20) == 0 and (res.status & ~0x20) == 0
    h64, d64 = hrec.astype(np.float64), drec.astype(np.float64)
    _SEEN[f"{case}:{np.dtype(dtype).name}"] = _deviation(d64, h64)
    if dtype == np.float64:
        assert_close_planes(d64, h64, 1e-13, 1e-13, f"{case}:f64 device vs host run")
    else:
        gt = fp32_group_tolerances(case, factor=8.0)
        assert gt is not None
        assert_close_planes(d64, h64, 1e-4, 1e-4, f"{case}:f32 device vs host run", group_tol=gt)
    assert np.array_equal(d64[:, 6, :] == 0, h64[:, 6, :] == 0)
    if polarized:
        dp, hp = dprt.cpu().numpy().astype(np.float64), hprt.astype(np.float64)
        assert np.array_equal(np.isnan(dp), np.isnan(hp))
        tol = 1e-13 if dtype == np.float64 else 2e-5
        np.testing.assert_allclose(np.nan_to_num(dp), np.nan_to_num(hp), rtol=0, atol=tol)


@pytest.mark.parametrize("dtype", [np.float64, np.float32], ids=["f64", "f32"])
@pytest.mark.parametrize("degree", [1, 3, 4, 6, 8, 10])
def test_zernike_degrees_device_equals_host_run(degree, dtype):
    """The one-polynomial Zernike evaluator (unrolled degrees, loop form) and the level form
    above the cap: device == host run of the same source, and == the oracle."""
    from oracle import oracle
    from optiland_amd.engine import HipSystem
    from tests.test_hostmath import _zernike_singlet
    from tests._util import PLANES
    table, rays = _zernike_singlet(0, degree, True)
    if dtype == np.float32:
        rays = {k: v.astype(np.float32).astype(np.float64) for k, v in rays.items()}
    want = oracle.trace(table, rays, 0, record=True)["record"]
    hrays = [np.array(rays[k], dtype=dtype, order="C", copy=True) for k in PLANES[:7]]
    hrays.append(np.zeros(hrays[0].size, dtype=dtype))
    n = hrays[0].size
    host = hm.HostMathSystem(table)
    hrec, _ = host.trace(hrays, 0, record=True)
    host.close()
    dev = HipSystem(table, "cuda:0")
    try:
        drays = [torch.tensor(h, device="cuda:0") for h in hrays]
        drec = dev.trace(drays, 0, record=True).record[:, :, :n].double().cpu().numpy()
    finally:
        dev.close()
    tol = 1e-7 if dtype == np.float64 else 1e-4
    assert_close_planes(drec, want, tol, tol, f"zern{degree}: device vs oracle")
    h64 = hrec.astype(np.float64)
    _SEEN[f"zernike_degree_{degree}:{np.dtype(dtype).name}"] = _deviation(drec, h64)
    tol = 1e-13 if dtype == np.float64 else 2e-5
    assert_close_planes(drec, h64, tol, tol, f"zern{degree}: device vs host run")
