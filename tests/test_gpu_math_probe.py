"""The kernels' arithmetic primitives on the device (`ol_math_probe`, ABI 8): the fp64
quotient / reciprocal / square root / reciprocal square root built from the hardware seeds
(`v_rcp_f64`, `v_rsq_f64`) + two refinement steps (surface_math.h: OL_FAST_F64) and the fp32
1-ulp instructions, held to their stated error bounds over millions of operands and to IEEE's
special values (`rcp_f64(-inf)` used to fall into the refinement and return NaN)."""

import ctypes as C

import numpy as np
import pytest
import torch

from optiland_amd import _capi

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_MEASURED = {}


def _dump():
    import json
    import os
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "math_probe.json"), "w") as fh:
        json.dump(_MEASURED, fh, indent=1, sort_keys=True)
OPS = {"rcp": 0, "div": 1, "sqrt": 2, "rsqrt": 3}


def _probe(op, a, b=None):
    lib = _capi.load()
    out = torch.empty_like(a)
    rc = lib.ol_math_probe(OPS[op], _capi.F32 if a.dtype == torch.float32 else _capi.F64,
                           a.numel(), C.c_void_p(a.data_ptr()),
                           C.c_void_p(b.data_ptr()) if b is not None else None,
                           C.c_void_p(out.data_ptr()), None)
    assert rc == 0, lib.ol_last_error()
    torch.cuda.synchronize()
    return out


def _ulps(got, want):
    """|got - want| in units of the last place of `want` (float64 arrays)."""
    return np.abs(got - want) / np.spacing(np.abs(want))


@pytest.mark.parametrize("op", ["rcp", "div", "sqrt", "rsqrt"])
def test_fp64_seed_plus_refinement_is_within_one_ulp(op):
    g = np.random.default_rng(7)
    n = 4_000_000
    # magnitudes a trace forms (1e-12 .. 1e12) and a band across the whole normal range
    for lo, hi in ((-12, 12), (-290, 290)):
        a = g.uniform(1, 2, n) * 10.0 ** g.uniform(lo, hi, n) * g.choice([-1.0, 1.0], n)
        b = g.uniform(1, 2, n) * 10.0 ** g.uniform(lo / 2, hi / 2, n) * g.choice([-1.0, 1.0], n)
        if op in ("sqrt", "rsqrt"):
            a = np.abs(a)
        ta = torch.tensor(a, device=DEV)
        tb = torch.tensor(b, device=DEV)
        got = _probe(op, ta, tb if op == "div" else None).cpu().numpy()
        al, bl = a.astype(np.longdouble), b.astype(np.longdouble)
        want = {"rcp": 1 / al, "div": al / bl, "sqrt": np.sqrt(al), "rsqrt": 1 / np.sqrt(al)}[op]
        ok = np.isfinite(want.astype(np.float64)) & (np.abs(want) > 1e-300)
        err = np.abs(got[ok].astype(np.longdouble) - want[ok]) / np.spacing(
            np.abs(want[ok]).astype(np.float64))
        worst, exact = float(err.max()), float((err <= 0.5 + 1e-9).mean())
        _MEASURED[f"f64 {op} 1e{lo}..1e{hi}"] = {"max_ulp": worst, "correctly_rounded": exact}
        _dump()
        # quotient / reciprocal / square root: within 1 ulp and almost always THE correctly
        # rounded value; the reciprocal square root (direction normalisation only) ends on a
        # product of two rounded factors: within 2 ulp
        assert worst <= (2.0 if op == "rsqrt" else 1.0), (op, lo, hi, worst)
        assert exact > (0.5 if op == "rsqrt" else 0.9), (op, exact)


def test_fp64_special_values_follow_ieee():
    inf, nan = np.inf, np.nan
    a = torch.tensor([0.0, -0.0, inf, -inf, nan, 1.0, -4.0, 4.0], device=DEV)
    r = _probe("rcp", a).cpu().numpy()
    assert np.array_equal(r[:4], [inf, -inf, 0.0, -0.0]) and np.signbit(r[3]) and np.isnan(r[4])
    assert r[5] == 1.0 and r[6] == -0.25
    s = _probe("sqrt", a).cpu().numpy()
    assert s[0] == 0 and s[1] == 0 and np.signbit(s[1]) and s[2] == inf
    assert np.isnan(s[3]) and np.isnan(s[4]) and np.isnan(s[6]) and s[7] == 2.0
    q = _probe("rsqrt", a).cpu().numpy()
    assert q[0] == inf and q[2] == 0.0 and np.isnan(q[3]) and np.isnan(q[6]) and q[7] == 0.5
    num = torch.tensor([1.0, 1.0, 1.0, 1.0, 0.0, -3.0, 3.0, 0.0], device=DEV)
    den = torch.tensor([0.0, -0.0, inf, -inf, 0.0, inf, -inf, 5.0], device=DEV)
    d = _probe("div", num, den).cpu().numpy()
    assert d[0] == inf and d[1] == -inf and d[2] == 0.0 and not np.signbit(d[2])
    assert d[3] == 0.0 and np.signbit(d[3])           # 1 / -inf = -0 (was NaN: mask 0x260)
    assert np.isnan(d[4]) and d[5] == 0.0 and np.signbit(d[5]) and d[6] == 0.0 and d[7] == 0.0


@pytest.mark.parametrize("op", ["rcp", "div", "sqrt", "rsqrt"])
def test_fp32_hardware_instructions_are_within_their_ulp_budget(op):
    """v_rcp_f32 / v_sqrt_f32 / v_rsq_f32: 1 ulp each; a quotient is a reciprocal times the
    numerator (2 roundings).  Well inside the 1e-4 fp32 parity budget of the trace."""
    g = np.random.default_rng(11)
    n = 2_000_000
    a = (g.uniform(1, 2, n) * 10.0 ** g.uniform(-12, 12, n)).astype(np.float32)
    b = (g.uniform(1, 2, n) * 10.0 ** g.uniform(-6, 6, n)).astype(np.float32)
    got = _probe(op, torch.tensor(a, device=DEV),
                 torch.tensor(b, device=DEV) if op == "div" else None).cpu().numpy()
    a64, b64 = a.astype(np.float64), b.astype(np.float64)
    want = {"rcp": 1 / a64, "div": a64 / b64, "sqrt": np.sqrt(a64), "rsqrt": 1 / np.sqrt(a64)}[op]
    err = np.abs(got.astype(np.float64) - want) / np.spacing(np.abs(want).astype(np.float32))
    assert float(err.max()) <= (2.5 if op == "div" else 1.5), (op, float(err.max()))
