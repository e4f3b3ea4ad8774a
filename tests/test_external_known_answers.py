"""Zemax ray-trace data as an external known answer.

The reference's own tests compare a single toroidal surface against Zemax OpticStudio
output (reference tests/test_geometries.py:1483-1634 positive R_x, :1636-1840 negative
R_x; tolerance rtol 1e-5 / atol 1e-6).  The same numbers are replayed here against the
CPU oracle and -- through the C ABI -- against the HIP kernel, in both precisions.  The
system tables were packed from the reference's lens definitions
(tools/make_golden.py:zemax_toroid_tables); the Zemax values are copied from the cited
assertions, none was produced by this repository.
"""

import numpy as np
import pytest
import torch

from tests._util import load_case_table

RT, AT = 1e-5, 1e-6

# (table, fan axis, input coordinates, z_image, transverse positions, transverse cosines, N)
CASES = {
    "posRx_yfan": ("zemax_toroid_posRx", "y", np.linspace(-5.0, 5.0, 5), 15.0,
                   [-8.123193233401276e-001, -4.676255499616224e-001, 0.0,
                    4.676255499616224e-001, 8.123193233401276e-001],
                   [3.251509839270260e-001, 1.537950377308984e-001, 0.0,
                    -1.537950377308984e-001, -3.251509839270260e-001],
                   [9.456621160072382e-001, 9.881027711576116e-001, 1.0,
                    9.881027711576116e-001, 9.456621160072382e-001]),
    "posRx_xfan": ("zemax_toroid_posRx", "x", np.linspace(-5.0, 5.0, 5), 15.0,
                   [-4.668385225648558e000, -2.333547899735358e000, 0.0,
                    2.333547899735358e000, 4.668385225648558e000],
                   [2.502086086422164e-002, 1.250260502601134e-002, 0.0,
                    -1.250260502601134e-002, -2.502086086422164e-002],
                   [9.996869292541608e-001, 9.999218393792406e-001, 1.0,
                    9.999218393792406e-001, 9.996869292541608e-001]),
    "negRx_yfan": ("zemax_toroid_negRx", "y", np.linspace(-10.0, 10.0, 5), 77.0,
                   [4.842002236238105e-001, -4.633747816929823e-002, 0.0,
                    4.633747816929823e-002, -4.842002236238105e-001],
                   [1.407949486740093e-001, 6.643870254059459e-002, 0.0,
                    -6.643870254059459e-002, -1.407949486740093e-001],
                   [9.900387782445106e-001, 9.977905084759640e-001, 1.0,
                    9.977905084759640e-001, 9.900387782445106e-001]),
}
# reference tests/test_geometries.py:1464-1481 and :1664-1680: Zemax sag of the two toroids
SAG_XY = (np.array([0.0, 2.5, 0.0, -2.5, 5.0, -5.0, 2.5, -2.5]),
          np.array([0.0, 0.0, 2.5, 0.0, 2.5, -2.5, -2.5, 2.5]))
SAG_NEG = [0.0, -6.253911140455271e-002, 7.867099677109624e-002, -6.253911140455271e-002,
           -1.715614452665938e-001, -1.715614452665938e-001, 1.623025381703616e-002,
           1.623025381703616e-002]


def _rays(case):
    _, axis, c, *_ = CASES[case]
    n = len(c)
    z = np.zeros(n)
    x, y = (c, z) if axis == "x" else (z, c)
    return {"x": x.copy(), "y": y.copy(), "z": z.copy(), "L": z.copy(), "M": z.copy(),
            "N": np.ones(n), "i": np.ones(n)}


def _check(case, out, fp32=False):
    """fp64: the reference's own tolerance against Zemax (rtol 1e-5, atol 1e-6).  fp32:
    positions to 1e-6 of the propagation length (ulp(77 mm) alone is 7.6e-6 mm -- the
    reference's absolute 1e-6 mm is below fp32 resolution there), cosines to 1e-6."""
    _, axis, _, z_img, pos, cos, N = CASES[case]
    a, b = ("x", "y") if axis == "x" else ("y", "x")
    pt = dict(rtol=RT, atol=AT * (z_img if fp32 else 1.0))
    np.testing.assert_allclose(out[a], pos, **pt)
    np.testing.assert_allclose(out[b], 0.0, **pt)
    np.testing.assert_allclose(out["z"], z_img, **pt)
    np.testing.assert_allclose(out["L" if axis == "x" else "M"], cos, RT, AT)
    np.testing.assert_allclose(out["M" if axis == "x" else "L"], 0.0, RT, AT)
    np.testing.assert_allclose(out["N"], N, RT, AT)


@pytest.mark.parametrize("case", sorted(CASES))
def test_oracle_reproduces_zemax_ray_data(case):
    from oracle import oracle
    table = load_case_table(CASES[case][0])
    out = oracle.trace(table, _rays(case), 0, record=False)
    _check(case, out)


def test_oracle_reproduces_zemax_sag():
    from oracle import oracle
    table = load_case_table("zemax_toroid_negRx")
    got = [oracle.sag(table, 1, float(x), float(y)) for x, y in zip(*SAG_XY)]
    np.testing.assert_allclose(got, SAG_NEG, RT, AT)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_hip_reproduces_zemax_ray_data(case, dtype):
    from optiland_amd.engine import HipSystem
    table = load_case_table(CASES[case][0])
    hip = HipSystem(table, "cuda:0")
    try:
        r = _rays(case)
        planes = [torch.tensor(r[k], dtype=dtype, device="cuda:0")
                  for k in ("x", "y", "z", "L", "M", "N", "i")]
        planes.append(torch.zeros_like(planes[0]))
        hip.trace(planes, 0, record=False)
        out = {k: planes[j].double().cpu().numpy()
               for j, k in enumerate(("x", "y", "z", "L", "M", "N"))}
        _check(case, out, fp32=dtype == torch.float32)
    finally:
        hip.close()


# --- ray generator known answers (reference tests/test_rays.py:686-735) -----------------
RAYGEN_CASES = {
    # TessarLens, angle field at infinity: Hx = Hy = 0.5, Px = Py = [0.1, 0.2]
    "tessar": ("sample_TessarLens", 0.5, 0.5, [0.1, 0.2], [0.1, 0.2],
               dict(x=[-0.23535066, -0.1909309], y=[-0.23535066, -0.1909309],
                    z=[-0.88839505, -0.88839505], L=[0.17519154, 0.17519154],
                    M=[0.17519154, 0.17519154], N=[0.96882189, 0.96882189], i=[1.0, 1.0])),
    # UVProjectionLens, object-height field, object-space telecentric: Hy = 1, Px = 0.8
    "uv_telecentric": ("sample_UVProjectionLens", 0.0, 1.0, [0.8], [0.0],
                       dict(x=[0.0], y=[48.0], z=[-110.85883544], L=[0.10674041], M=[0.0],
                            N=[0.99428692], i=[1.0])),
}


@pytest.mark.parametrize("case", sorted(RAYGEN_CASES))
def test_oracle_reproduces_raygen_known_answers(case):
    from oracle import oracle
    name, hx, hy, px, py, want = RAYGEN_CASES[case]
    table = load_case_table(name)
    n = len(px)
    got = oracle.generate_rays(table.raygen, np.full(n, hx), np.full(n, hy), np.array(px),
                               np.array(py))
    for k, v in want.items():
        np.testing.assert_allclose(got[k], v, rtol=0, atol=1e-8, err_msg=k)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("case", sorted(RAYGEN_CASES))
def test_hip_reproduces_raygen_known_answers(case, dtype):
    from optiland_amd.engine import HipSystem
    name, hx, hy, px, py, want = RAYGEN_CASES[case]
    table = load_case_table(name)
    hip = HipSystem(table, "cuda:0")
    try:
        tx = torch.tensor(px, dtype=dtype, device="cuda:0")
        ty = torch.tensor(py, dtype=dtype, device="cuda:0")
        planes = hip.generate_rays(hx, hy, tx, ty)  # launch-uniform field scalars
        atol = 1e-8 if dtype == torch.float64 else 2e-5  # fp32: ulp(110 mm) = 7.6e-6
        for k, p in zip(("x", "y", "z", "L", "M", "N", "i"), planes):
            np.testing.assert_allclose(p.double().cpu().numpy(), want[k], rtol=0, atol=atol,
                                       err_msg=k)
    finally:
        hip.close()
