"""The LIVE reference against the kernel SOURCE, with nothing in between (build container).

tests/test_reference_fuzz.py holds packer + oracle to the reference on random lenses, and
the GPU suite holds the kernel to the oracle.  With the kernel arithmetic compiled for the
host (tests/hostmath) the chain can be closed directly, on the CPU: 150 random lenses built
through the reference's public API, traced by its NumPy backend, and by `pack_optic` ->
`ol_system_create` -> `ol_generate_rays` -> `ol_trace` of the host harness -- the source of
`raygen_device.h` and `surface_math.h` -- on the same field and pupil points, in fp64 and in
fp32.  Every surface's recorded x, y, z, L, M, N, intensity, opd; PRT matrices and
`update_intensity` on the polarised ones.

CPU only, skipped where /root/reference does not exist.
"""

import numpy as np
import pytest

from tests import _hostmath as hm
from tests.test_reference_fuzz import REF, build_random_lens, ref  # noqa: F401 (fixture)

import os

pytestmark = [
    pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "optiland")),
                       reason="reference package not present"),
    pytest.mark.skipif(not hm.available(), reason="hipcc (used as host C++ compiler) missing"),
]
PLANES = ("x", "y", "z", "L", "M", "N", "intensity", "opd")


@pytest.mark.parametrize("seed", range(150))
def test_random_reference_lens_equals_the_kernel_source(ref, seed):  # noqa: F811
    be = ref
    from optiland_amd.packer import pack_optic
    from optiland_amd.rays import _state_dict
    lens, rng = build_random_lens(seed, be)
    w = float(lens.primary_wavelength)
    table = pack_optic(lens, wavelengths=[w])
    n = 400
    r, th = np.sqrt(rng.random(n)) * 0.9, 2 * np.pi * rng.random(n)
    px, py = r * np.cos(th), r * np.sin(th)
    hx, hy = float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-1, 1))
    sysm = hm.HostMathSystem(table)
    polarised = table.polarization is not None
    vxf, vyf = lens.fields.get_vig_factor(hx, hy)
    vx, vy = 1.0 - float(np.asarray(vxf)), 1.0 - float(np.asarray(vyf))

    def host(dtype):
        rays, st = sysm.generate_rays(hx, hy, np.ascontiguousarray(px * vx, dtype=dtype),
                                      np.ascontiguousarray(py * vy, dtype=dtype), vx, vy)
        assert st == 0
        k0 = [rays[3].copy(), rays[4].copy(), rays[5].copy()]
        i0 = rays[6].copy()
        prt = np.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype) if polarised else None
        rec, status = sysm.trace(rays, 0, record=True, prt=prt, prt_identity=polarised)
        return rec.astype(np.float64), status, prt, k0, i0

    with np.errstate(all="ignore"):
        try:
            out = lens.trace_generic(hx, hy, px, py, w)
        except ValueError as e:  # Zernike / Chebyshev range errors: the status word must say so
            _, status, _, _, _ = host(np.float64)
            assert status != 0, f"reference raised {e!r}, kernel source reports status 0"
            sysm.close()
            return
    want = {k: np.asarray(getattr(lens.surfaces, k), dtype=np.float64) for k in PLANES}
    scale = max(1.0, float(np.nanmax(np.abs(want["z"][1:][np.isfinite(want["z"][1:])]))))
    for dtype, tol0 in ((np.float64, 1e-7), (np.float32, 1e-4)):
        rec, status, prt, k0, i0 = host(dtype)
        assert status == 0
        for j, k in enumerate(PLANES):
            a, b = rec[:, j, :], want[k]
            assert a.shape == b.shape, k
            if k in "xyz" and not np.isfinite(b[0]).all():  # object at infinity: row 0 is +-inf
                a, b = a[1:], b[1:]
            if dtype == np.float64:
                assert np.array_equal(np.isnan(a), np.isnan(b)), f"{k}: NaN masks differ"
            else:  # fp32: a ray within rounding of a miss / TIR edge may fall on either side
                same = np.isnan(a) == np.isnan(b)
                assert same.mean() > 0.995, k
                a, b = np.where(same, a, 0.0), np.where(same, b, 0.0)
            tol = tol0 * (scale if k in ("x", "y", "z", "opd") else 1.0)
            np.testing.assert_allclose(np.nan_to_num(a, posinf=0, neginf=0),
                                       np.nan_to_num(b, posinf=0, neginf=0), rtol=0, atol=tol,
                                       err_msg=f"seed {seed} {np.dtype(dtype).name} plane {k}")
        if polarised and dtype == np.float64:
            np.testing.assert_allclose(np.nan_to_num(hm.prt_to_complex(prt)),
                                       np.nan_to_num(np.asarray(out.p)), rtol=0, atol=1e-7)
            # update_intensity of the same bundle (rays/polarized_rays.py:122-133)
            iu, st = sysm.polarized_intensity(prt, k0, i0, _state_dict(lens.polarization_state))
            assert st == 0
            ref_rays = out
            ref_rays.update_intensity(lens.polarization_state)
            np.testing.assert_allclose(np.nan_to_num(iu), np.nan_to_num(np.asarray(ref_rays.i)),
                                       rtol=0, atol=1e-7)
    sysm.close()
