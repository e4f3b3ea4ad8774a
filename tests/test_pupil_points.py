"""`ol_pupil_points` (ABI 8): the reference's deterministic pupil samplers evaluated on the
device from the point index (optiland/distribution.py:161-220) -- against this repo's host
samplers (`optiland_amd/distribution.py`, themselves bit-identical to the reference's,
tests/test_host_tracer.py).  Same number of points, same ORDER; the uniform grid bit for bit
in both precisions; hexapolar within 1 ulp of fp64 (cos / sin come from another libm) and,
after rounding, equal in fp32 but for isolated last-bit cases.
`-m gpu`: the HIP library; otherwise the host build of the same source."""

import numpy as np
import pytest
import torch

from optiland_amd import load_system, tracer as tr
from optiland_amd.distribution import create_distribution, uniform_rows

WHERE = [pytest.param("cuda", marks=pytest.mark.gpu), "host"]
_MEASURED = {}


def _dump():
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "pupil_points.json"), "w") as fh:
        json.dump(_MEASURED, fh, indent=1, sort_keys=True)


def _engine(where):
    table = load_system("double_gauss")
    if where == "cuda":
        from optiland_amd.engine import HipSystem
        return HipSystem(table, "cuda:0"), table
    from tests import _hostmath as hm
    if not hm.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    return hm.make_engine_class()(table), table


def _host(name, num):
    d = create_distribution(name)
    d.generate_points(num)
    return np.asarray(d.x, dtype=np.float64), np.asarray(d.y, dtype=np.float64)


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("rings", [0, 1, 6, 64, 1825])
def test_hexapolar_points_from_the_index(rings, where):
    if rings == 1825 and where != "cuda":
        pytest.skip("1e7 points: on the device only")
    eng, _ = _engine(where)
    try:
        hx, hy = _host("hexapolar", rings)
        for dtype in (torch.float64, torch.float32):
            x, y = eng.pupil_points("hexapolar", rings, dtype)
            assert x.numel() == hx.size == 1 + 3 * rings * (rings + 1)
            gx, gy = x.cpu().numpy(), y.cpu().numpy()
            if dtype == torch.float64:
                # the radius of a point is exact; its cos / sin come from another libm than
                # NumPy's -- each within an ulp of the true value, so the two agree to a
                # couple of ulps of the RADIUS (a coordinate near a zero crossing carries the
                # absolute, not the relative, error of its cosine)
                ulp_r = np.spacing(np.hypot(hx, hy).clip(1e-300))
                ex, ey = np.abs(gx - hx) / ulp_r, np.abs(gy - hy) / ulp_r
                _MEASURED[f"{where} hexapolar {rings} f64 max ulp(r)"] = float(max(ex.max(),
                                                                               ey.max()))
                assert float(max(ex.max(), ey.max())) <= 3.0, (float(ex.max()), float(ey.max()))
                np.testing.assert_allclose(np.hypot(gx, gy), np.hypot(hx, hy), rtol=6e-16,
                                           atol=0)
            else:
                wx, wy = hx.astype(np.float32), hy.astype(np.float32)
                bad = int((gx != wx).sum() + (gy != wy).sum())
                _MEASURED[f"{where} hexapolar {rings} f32 values that differ"] = bad
                _dump()
                assert bad <= max(2, gx.size // 1_000_000), bad   # isolated last-bit cases
                np.testing.assert_allclose(gx, wx, rtol=1.2e-7, atol=1e-38)
                np.testing.assert_allclose(gy, wy, rtol=1.2e-7, atol=1e-38)
    finally:
        eng.close()


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("side", [2, 3, 4, 5, 10, 101, 512, 3568])
def test_uniform_points_from_the_index_are_bit_identical(side, where):
    if side == 3568 and where != "cuda":
        pytest.skip("1e7 points: on the device only")
    eng, _ = _engine(where)
    try:
        hx, hy = _host("uniform", side)
        first, offset = uniform_rows(side)
        assert int(offset[-1]) == hx.size and first.size == side
        for dtype in (torch.float64, torch.float32):
            x, y = eng.pupil_points("uniform", side, dtype)
            npd = np.float64 if dtype == torch.float64 else np.float32
            np.testing.assert_array_equal(x.cpu().numpy(), hx.astype(npd))
            np.testing.assert_array_equal(y.cpu().numpy(), hy.astype(npd))
    finally:
        eng.close()


@pytest.mark.parametrize("where", WHERE)
def test_tracer_samples_named_distributions_on_the_device(where, monkeypatch):
    """`HipRayTracer.trace(..., "hexapolar" | "uniform")` takes its pupil planes from
    `ol_pupil_points` (no host sampling, no upload) and traces the same rays as with the host
    sampler (`OPTILAND_HIP_DEVICE_PUPIL=0`)."""
    eng, table = _engine(where)
    try:
        dev = "cuda:0" if where == "cuda" else "cpu"
        w = float(table.wavelengths[0])
        for name, num in (("hexapolar", 9), ("uniform", 21)):
            tr._PUPIL_PLANES.clear()
            calls = []
            orig = type(eng).pupil_points

            def spy(self, *a, _o=orig, **k):
                calls.append(a)
                return _o(self, *a, **k)

            monkeypatch.setattr(type(eng), "pupil_points", spy)
            a = tr.HipRayTracer(table, dev, dtype=torch.float64, engine=eng)
            ra = a.trace(0.0, 0.7, w, num, name)
            assert len(calls) == 1
            monkeypatch.setattr(type(eng), "pupil_points", orig)
            tr._PUPIL_PLANES.clear()
            monkeypatch.setenv("OPTILAND_HIP_DEVICE_PUPIL", "0")
            b = tr.HipRayTracer(table, dev, dtype=torch.float64, engine=eng)
            rb = b.trace(0.0, 0.7, w, num, name)
            monkeypatch.delenv("OPTILAND_HIP_DEVICE_PUPIL")
            for k in ("x", "y", "z", "L", "M", "N", "i", "opd"):
                np.testing.assert_allclose(getattr(ra, k).cpu().numpy(),
                                           getattr(rb, k).cpu().numpy(), rtol=1e-12, atol=1e-12,
                                           err_msg=f"{name} {k}")
        tr._PUPIL_PLANES.clear()
    finally:
        eng.close()
