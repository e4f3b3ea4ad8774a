"""`ol_trace_generate` (ABI 6): ray generation fused into the recording trace kernel, and
`record_first_surface`.

The fused launch must reproduce the two-launch chain `ol_generate_rays` -> `ol_trace` BIT FOR
BIT (same device functions, same order of operations): every recorded row, the PRT planes of
polarised systems, the status bits -- on every golden system that carries generator scalars,
in fp32 and fp64, for `trace()`-style launches and for `trace_generic`-style ones (pupil
range check, vignetting pre-scaling).  `-m gpu`: the HIP library on the MI355X;
`-m "not gpu"`: the product's engine class on the host build of the kernel source.
"""

import numpy as np
import pytest
import torch

from optiland_amd import _capi, load_system, tracer as tr
from tests._util import golden_cases, load_case

WHERE = [pytest.param("cuda", marks=pytest.mark.gpu), "host"]
CASES = [c for c in golden_cases() if not c.startswith("fuzz_")] + ["fuzz_00", "fuzz_03"]


def _engine(table, where):
    if where == "cuda":
        from optiland_amd.engine import HipSystem
        return HipSystem(table, "cuda:0"), "cuda:0"
    from tests import _hostmath as hm
    if not hm.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    return hm.make_engine_class()(table), "cpu"


def _pupil(n, dtype, dev, seed):
    g = np.random.default_rng(seed)
    r, th = np.sqrt(g.random(n)) * 0.98, 2 * np.pi * g.random(n)
    return (torch.tensor(r * np.cos(th), dtype=dtype, device=dev),
            torch.tensor(r * np.sin(th), dtype=dtype, device=dev))


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("case", CASES)
def test_fused_generate_equals_generate_then_trace(case, where):
    table, _ = load_case(case)
    if not table.raygen:
        pytest.skip("no ray-generation scalars in this table")
    eng, dev = _engine(table, where)
    try:
        if not eng.can_trace_generate():
            pytest.skip("apodized pupil: the two-launch path is the only one")
        pol = table.polarization is not None
        if table.uses_polarization and not pol:
            pytest.skip("coatings need a polarisation state")
        for dtype in (torch.float64, torch.float32):
            for n in (1, 257, 1000):
                px, py = _pupil(n, dtype, dev, n)
                for field, vig, flags in (((0.0, 0.7), (1.0, 1.0), 0),
                                          ((0.3, -0.5), (0.9, 0.8),
                                           _capi.RAYGEN_CHECK_PUPIL | _capi.RAYGEN_PRESCALE_PUPIL)):
                    # two launches
                    rec_a = eng.alloc_record(n, dtype)
                    rays = eng.row0_planes(rec_a, n)
                    eng.generate_rays(field[0], field[1], px, py, vig[0], vig[1], out=rays,
                                      flags=flags)
                    prt_a = torch.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype,
                                        device=dev) if pol else None
                    a = eng.trace(rays, 0, record=rec_a, prt=prt_a, prt_identity=pol,
                                  zero_status=False)
                    # one launch
                    prt_b = torch.empty_like(prt_a) if pol else None
                    b = eng.trace_generate(px, py, 0, field=field, vig=vig, flags=flags,
                                           prt=prt_b)
                    ra, rb = a.record[:, :, :n].cpu().numpy(), b.record[:, :, :n].cpu().numpy()
                    assert ra.shape == rb.shape
                    np.testing.assert_array_equal(ra, rb, err_msg=f"{case} {dtype} n={n}")
                    if pol:
                        np.testing.assert_array_equal(prt_a.cpu().numpy(), prt_b.cpu().numpy())
                    # the last two rows alone (lazy-record launches) and a final-state copy
                    S = eng.num_surfaces
                    out = [torch.empty(n, dtype=dtype, device=dev) for _ in range(8)]
                    c = eng.trace_generate(px, py, 0, field=field, vig=vig, flags=flags,
                                           prt=torch.empty_like(prt_a) if pol else None,
                                           record_first=max(S - 2, 0), rays_out=out)
                    rc = c.record[: min(2, S), :, :n].cpu().numpy()
                    np.testing.assert_array_equal(rc, ra[max(S - 2, 0):])
                    np.testing.assert_array_equal(torch.stack(out).cpu().numpy(), ra[-1])
    finally:
        eng.close()


@pytest.mark.parametrize("where", WHERE)
def test_record_first_surface_on_plain_traces(where):
    table, data = load_case("double_gauss")
    eng, dev = _engine(table, where)
    try:
        S = eng.num_surfaces
        for dtype in (torch.float64, torch.float32):
            src = [torch.tensor(data["rays_in"][k], dtype=dtype, device=dev) for k in range(7)]
            src.append(torch.zeros_like(src[0]))
            n = src[0].numel()
            full = eng.trace([t.clone() for t in src], 0, record=True)
            for first in (3, S - 2, S - 1):
                part = eng.trace([t.clone() for t in src], 0, record=True, record_first=first)
                assert part.record.shape[0] == S - first and part.first == first
                np.testing.assert_array_equal(part.record[:, :, :n].cpu().numpy(),
                                              full.record[first:, :, :n].cpu().numpy())
                np.testing.assert_array_equal(part.row(S - 1, "x").cpu().numpy(),
                                              full.row(S - 1, "x").cpu().numpy())
        with pytest.raises(ValueError):
            eng.trace([t.clone() for t in src], 0, record=True, record_first=S)
    finally:
        eng.close()


@pytest.mark.parametrize("where", WHERE)
def test_fused_generate_status_bits_and_refusals(where):
    table = load_system("double_gauss")
    eng, dev = _engine(table, where)
    try:
        px = torch.tensor([0.0, 0.5, 1.5], dtype=torch.float64, device=dev)
        py = torch.zeros_like(px)
        with pytest.raises(ValueError, match="pupil coordinates must be within"):
            eng.trace_generate(px, py, 0, field=(0.0, 0.0), flags=_capi.RAYGEN_CHECK_PUPIL)
        with pytest.raises(ValueError, match="field coordinates must be within"):
            eng.trace_generate(px * 0.1, py, 0, field=(0.0, 1.5), flags=_capi.RAYGEN_CHECK_FIELD)
        ok = eng.trace_generate(px * 0.1, py, 0, field=(0.0, 1.0), flags=_capi.RAYGEN_CHECK_PUPIL)
        assert bool(torch.isfinite(ok.record[-1, 0, :3]).all())
    finally:
        eng.close()
    # ABI 8: an apodized pupil is served by the generating launch too; ABI 10: also when the
    # trace is polarised
    apod, _ = load_case("apodized_gaussian_trace")
    eng, dev = _engine(apod, where)
    try:
        assert eng.can_trace_generate() and eng.can_trace_generate(field_planes=True)
    finally:
        eng.close()
    pol, _ = load_case("zernike_fresnel_fringe")
    eng, dev = _engine(pol, where)
    try:
        assert eng.can_trace_generate() and eng.can_trace_generate(field_planes=True)
    finally:
        eng.close()


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("name", ["double_gauss", "rc_asphere", "zernike_fresnel_fringe"])
def test_tracer_fused_and_two_launch_paths_agree(name, where):
    """`HipRayTracer.trace / trace_generic` through `ol_trace_generate` (default) and through
    the two launches (`fuse_generate = False`): identical rays, records, L0 / M0 / N0, PRT and
    polarised intensities; record-last mode returns the same final state."""
    table = load_system(name)
    eng, dev = _engine(table, where)
    try:
        for dtype in (torch.float64, torch.float32):
            a = tr.HipRayTracer(table, dev, dtype=dtype, engine=eng)
            b = tr.HipRayTracer(table, dev, dtype=dtype, engine=eng)
            b.fuse_generate = False
            w = float(table.wavelengths[0])
            px, py = _pupil(500, dtype, dev, 7)
            for call in (lambda t: t.trace(0.0, 0.7, w, 5, "hexapolar"),
                         lambda t: t.trace_generic(0.0, 1.0, px, py, w)):
                ra, rb = call(a), call(b)
                assert a.last_fused_launch is not None and b.last_fused_launch is None
                for k in ("x", "y", "z", "L", "M", "N", "i", "opd", "L0", "M0", "N0"):
                    np.testing.assert_array_equal(getattr(ra, k).cpu().numpy(),
                                                  getattr(rb, k).cpu().numpy(), k)
                for k in ("x", "L", "intensity", "opd"):
                    np.testing.assert_array_equal(getattr(a.surfaces, k).cpu().numpy(),
                                                  getattr(b.surfaces, k).cpu().numpy())
                if table.polarization is not None:
                    np.testing.assert_array_equal(ra._prt.cpu().numpy(), rb._prt.cpu().numpy())
                    np.testing.assert_array_equal(ra._i0.cpu().numpy(), rb._i0.cpu().numpy())
                # record-last
                a.record_all = b.record_all = False
                la, lb = call(a), call(b)
                a.record_all = b.record_all = True
                for k in ("x", "y", "z", "L", "M", "N", "i", "opd"):
                    np.testing.assert_array_equal(getattr(la, k).cpu().numpy(),
                                                  getattr(ra, k).cpu().numpy(), k)
                    np.testing.assert_array_equal(getattr(lb, k).cpu().numpy(),
                                                  getattr(ra, k).cpu().numpy(), k)
                assert a.surfaces.x.numel() == 0
    finally:
        eng.close()


STATES = [
    {"is_polarized": False},
    {"is_polarized": True, "Ex": 1.0, "Ey": 0.0, "phase_x": 0.0, "phase_y": 0.0},
    {"is_polarized": True, "Ex": 0.6, "Ey": 0.8, "phase_x": 0.3, "phase_y": -1.1},
]


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("case", ["zernike_fresnel_fringe", "zernike_fresnel_polarized",
                                  "polarizer_retarder", "polarizer_only",
                                  "coated_mirror_polarised", "fuzz_03", "fuzz_05"])
def test_update_intensity_epilogue_of_the_generating_launch(case, where):
    """ABI 7: `ol_trace_extras.updated_intensity` -- `PolarizedRays.update_intensity`
    (rays/polarized_rays.py:68-133) inside `ol_trace_generate`, from the matrix in registers
    and the regenerated launch direction -- equals `ol_polarized_intensity` on what the same
    launch wrote (PRT planes, row 0 of the record): same device function, same inputs.  Real
    and complex (retarder) matrices, unpolarised / linear / elliptical states, both dtypes;
    the PRT planes and the record are those of a launch without the epilogue."""
    table, _ = load_case(case)
    if not table.raygen or table.polarization is None:
        pytest.skip("not a polarised case with generator scalars")
    eng, dev = _engine(table, where)
    try:
        if not eng.can_trace_generate() or not eng.can_fuse_update_intensity():
            pytest.skip("library without the fused epilogue")
        for dtype in (torch.float64, torch.float32):
            n = 777
            px, py = _pupil(n, dtype, dev, 21)
            cplx = 18 if table.needs_complex_prt else 9
            for st in STATES:
                prt0 = torch.empty((cplx, n), dtype=dtype, device=dev)
                plain = eng.trace_generate(px, py, 0, field=(0.0, 0.6), prt=prt0)
                assert plain.updated_intensity is None
                prt1 = torch.empty((cplx, n), dtype=dtype, device=dev)
                fused = eng.trace_generate(px, py, 0, field=(0.0, 0.6), prt=prt1,
                                           update_intensity=st)
                assert fused.updated_intensity is not None
                np.testing.assert_array_equal(prt1.cpu().numpy(), prt0.cpu().numpy())
                np.testing.assert_array_equal(fused.record[:, :, :n].cpu().numpy(),
                                              plain.record[:, :, :n].cpu().numpy())
                r0 = plain.rows(0)
                want = eng.polarized_intensity(prt0, (r0[3], r0[4], r0[5]), r0[6], st)
                got = fused.updated_intensity
                a, b = got.cpu().numpy(), want.cpu().numpy()
                assert np.array_equal(np.isnan(a), np.isnan(b))
                tol = 1e-13 if dtype == torch.float64 else 2e-6
                np.testing.assert_allclose(a, b, rtol=tol, atol=tol)
    finally:
        eng.close()


def test_update_intensity_epilogue_is_refused_without_a_prt():
    """C ABI, host build: `ol_trace_extras.updated_intensity` on an unpolarised launch is an
    argument error (OL_EINVAL with a message), not something silently ignored."""
    import ctypes as C
    from tests import _hostmath as hm
    if not hm.available():
        pytest.skip("hipcc (used as host C++ compiler) missing")
    eng = hm.make_engine_class()(load_system("double_gauss"))
    try:
        n = 16
        px, py = _pupil(n, torch.float64, "cpu", 3)
        rec = eng.alloc_record(n, torch.float64)
        out = torch.empty(n, dtype=torch.float64)
        st = _capi.PolarizationStateC(0, 0, 0.0, 0.0, 0.0, 0.0)
        ex = _capi.TraceExtras(None, 0.0, 0.0, 0, 0, C.addressof(st), out.data_ptr())
        p = eng._raygen_params()
        inp, keep = eng._raygen_inputs(0.0, 0.5, px, py, 1.0, 1.0, 0)
        rc = eng.lib.ol_trace_generate(eng._handle, 1, n, C.byref(p), C.byref(inp), 0,
                                       rec.data_ptr(), int(rec.shape[2]), None, None, 0,
                                       eng._status.data_ptr(), C.byref(ex), None)
        assert rc == -1  # OL_EINVAL
        assert b"polarised launch" in eng.lib.ol_last_error()
    finally:
        eng.close()


# ------------------------------------------------------------------------------------------
# ABI 8: per-ray field planes, apodized pupils and the spot epilogue in the generating launch
# ------------------------------------------------------------------------------------------
APODIZATIONS = [  # (apod_kind, a, b): optiland/apodization/*.py as packed by packer._pack_apodization
    (1, 0.7, 0.0),    # gaussian
    (2, 0.9, 0.0),    # cosine squared
    (3, 1.6, 0.0),    # hann
    (4, 1.0, 2.0),    # polynomial
    (5, 0.8, 4.0),    # super-gaussian
    (6, 0.95, 0.4),   # tukey
]


def _table(name):
    try:
        return load_system(name)
    except Exception:  # not a shipped table: a golden case
        return load_case(name)[0]


def _fields(n, dtype, dev, seed, discrete):
    g = np.random.default_rng(seed)
    if discrete:   # C1: three field points x the pupil, expanded (real_ray_tracer.py:95-98)
        pts = np.array([[0.0, 0.0], [0.0, 0.7], [0.0, 1.0]])
        idx = np.arange(n) * 3 // max(n, 1)
        hx, hy = pts[idx, 0], pts[idx, 1]
    else:
        hx, hy = g.uniform(-1, 1, n), g.uniform(-1, 1, n)
    return (torch.tensor(hx, dtype=dtype, device=dev), torch.tensor(hy, dtype=dtype, device=dev))


def _two_launch_then_fused(eng, dev, n, dtype, px, py, field, vig, flags):
    rec_a = eng.alloc_record(n, dtype)
    rays = eng.row0_planes(rec_a, n)
    vx, vy = vig if vig is not None else (None, None)
    eng.generate_rays(field[0], field[1], px, py, vx, vy, out=rays, flags=flags)
    a = eng.trace(rays, 0, record=rec_a, zero_status=False)
    b = eng.trace_generate(px, py, 0, field=field, vig=vig, flags=flags)
    return a.record[:, :, :n].cpu().numpy(), b.record[:, :, :n].cpu().numpy()


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("case", CASES)
def test_generating_launch_with_field_planes_equals_two_launches(case, where):
    """`trace_generic(Hx[], Hy[], Px[], Py[])` and the fields x pupil expansion of a
    multi-field trace as ONE launch (`trace_kernel<..., GEN | kGenFieldPlanes>`): every
    recorded row bit for bit what `ol_generate_rays` + `ol_trace` give, with and without
    per-ray vignetting planes, fp32 and fp64, ragged sizes."""
    table, _ = load_case(case)
    if not table.raygen:
        pytest.skip("no ray-generation scalars in this table")
    eng, dev = _engine(table, where)
    try:
        if not eng.can_trace_generate(field_planes=True):
            pytest.skip("no generating launch for this table")
        if table.polarization is not None or table.uses_polarization:
            pytest.skip("polarised: test_polarised_generating_launch_with_fields_and_apodization")
        for dtype in (torch.float64, torch.float32):
            for n, discrete in ((1, False), (259, True), (1000, False)):
                px, py = _pupil(n, dtype, dev, n + 1)
                hx, hy = _fields(n, dtype, dev, n + 2, discrete)
                g = np.random.default_rng(n)
                vplanes = (torch.tensor(g.uniform(0.8, 1.0, n), dtype=dtype, device=dev),
                           torch.tensor(g.uniform(0.8, 1.0, n), dtype=dtype, device=dev))
                for vig, flags in ((None, _capi.RAYGEN_CHECK_FIELD),
                                   (vplanes, _capi.RAYGEN_CHECK_FIELD | _capi.RAYGEN_CHECK_PUPIL
                                    | _capi.RAYGEN_PRESCALE_PUPIL),
                                   ((0.9, 0.85), _capi.RAYGEN_PRESCALE_PUPIL)):
                    ra, rb = _two_launch_then_fused(eng, dev, n, dtype, px, py, (hx, hy), vig,
                                                    flags)
                    np.testing.assert_array_equal(ra, rb, err_msg=f"{case} {dtype} n={n}")
        # range check of the field planes inside the generating launch
        bad = torch.tensor([0.0, 1.5], dtype=torch.float64, device=dev)
        ok = torch.tensor([0.0, 0.1], dtype=torch.float64, device=dev)
        with pytest.raises(ValueError, match="field coordinates must be within"):
            eng.trace_generate(ok, ok, 0, field=(ok, bad), vig=None,
                               flags=_capi.RAYGEN_CHECK_FIELD)
    finally:
        eng.close()


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("kind,a,b", APODIZATIONS)
@pytest.mark.parametrize("name", ["double_gauss", "rc_asphere", "zernike_nopol"])
def test_generating_launch_with_apodized_pupil_equals_two_launches(name, kind, a, b, where):
    """Apodized pupils (ray_generator.py:81-85: the initial intensity) in the generating
    launch, alone and together with per-ray field planes: every apodization the packer
    knows, on a conic-only, an even-asphere and a Zernike system (the three Newton families
    of the generating kernels), bit for bit the two-launch result."""
    import copy
    table = copy.deepcopy(_table(name))
    if table.polarization is not None or table.uses_polarization:
        pytest.skip("polarised table")
    table.raygen = dict(table.raygen, apod_kind=kind, apod_a=a, apod_b=b)
    eng, dev = _engine(table, where)
    try:
        assert eng.can_trace_generate()
        for dtype in (torch.float64, torch.float32):
            n = 517
            px, py = _pupil(n, dtype, dev, 40 + kind)
            hx, hy = _fields(n, dtype, dev, 50 + kind, False)
            for field, vig, flags in (((0.0, 0.7), (1.0, 1.0), 0),
                                      ((hx, hy), None, _capi.RAYGEN_CHECK_FIELD)):
                ra, rb = _two_launch_then_fused(eng, dev, n, dtype, px, py, field, vig, flags)
                np.testing.assert_array_equal(ra, rb, err_msg=f"{name} apod {kind} {dtype}")
                assert not np.all(ra[0, 6] == 1.0)  # the apodization really is in row 0
    finally:
        eng.close()


@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("name", ["double_gauss", "rc_asphere", "zernike_nopol"])
def test_spot_epilogue_of_the_generating_launch(name, where):
    """ABI 8: `ol_trace_extras.spot_slots` on `ol_trace_generate` -- generate, trace, record
    and reduce in one launch (the per-step form of the sharded trace).  The record is the one
    a launch without the epilogue writes; the moments are the masked sums of its last row."""
    table = _table(name)
    eng, dev = _engine(table, where)
    try:
        for dtype in (torch.float64, torch.float32):
            for n in (1, 255, 30_011 if where == "cuda" else 2_003):
                px, py = _pupil(n, dtype, dev, n + 3)
                slots = eng.alloc_spot_slots()
                cx, cy = 0.01, 0.25
                plain = eng.trace_generate(px, py, 0, field=(0.0, 0.6))
                fused = eng.trace_generate(px, py, 0, field=(0.0, 0.6), spot=(slots, cx, cy))
                assert torch.equal(fused.record[:, :, :n].nan_to_num(),
                                   plain.record[:, :, :n].nan_to_num())
                got = eng.reduce_spot_slots(slots).cpu().numpy()
                x, y, i = (plain.row(plain.last, k) for k in (0, 1, 6))
                m = i > 0
                dx, dy = x[m].double() - cx, y[m].double() - cy
                ok = ~(torch.isnan(dx) | torch.isnan(dy))
                want = np.array([float(m.sum()), float(dx.sum()), float(dy.sum()),
                                 float((dx * dx).sum()), float((dy * dy).sum()),
                                 float(i[m].double().sum()),
                                 float((dx * dx + dy * dy)[ok].max()) if bool(ok.any()) else 0.0])
                assert got[0] == want[0]
                np.testing.assert_allclose(got[1:6], want[1:6], rtol=1e-10, atol=1e-9)
                assert got[6] == want[6]
                if where == "cuda":
                    # ... and bit-equal to the library's own plane reduction of that row
                    # (`ol_spot_moments`: the count and the max; the sums agree to rounding
                    # -- different summation trees)
                    mom = eng.spot_moments(x, y, i) if hasattr(eng, "spot_moments") else None
                    if mom is not None:
                        assert float(mom[0]) == got[0] or float(mom[5]) == got[0]
        # refused combinations: per-ray fields or an apodized pupil with the epilogue
        n = 16
        px, py = _pupil(n, torch.float64, dev, 1)
        hx = torch.zeros(n, dtype=torch.float64, device=dev)
        with pytest.raises(RuntimeError, match="one field point"):
            eng.trace_generate(px, py, 0, field=(hx, hx), vig=None,
                               spot=(eng.alloc_spot_slots(), 0.0, 0.0))
    finally:
        eng.close()


@pytest.mark.parametrize("where", WHERE)
def test_tracer_takes_the_generating_launch_for_field_arrays(where):
    """`HipRayTracer.trace_generic(Hx[], Hy[], Px[], Py[])` and `trace()` over several field
    points: ONE launch (`last_fused_launch` set), results identical to the two launches."""
    table = load_system("cooke_generic")
    eng, dev = _engine(table, where)
    try:
        for dtype in (torch.float64, torch.float32):
            a = tr.HipRayTracer(table, dev, dtype=dtype, engine=eng)
            b = tr.HipRayTracer(table, dev, dtype=dtype, engine=eng)
            b.fuse_generate = False
            w = float(table.wavelengths[0])
            n = 300
            px, py = _pupil(n, dtype, dev, 9)
            hx, hy = _fields(n, dtype, dev, 10, True)
            for call in (lambda t: t.trace_generic(hx, hy, px, py, w),
                         lambda t: t.trace(torch.tensor([0.0, 0.0, 0.0]),
                                           torch.tensor([0.0, 0.7, 1.0]), w, 6, "hexapolar")):
                ra, rb = call(a), call(b)
                assert a.last_fused_launch is not None and b.last_fused_launch is None
                for k in ("x", "y", "z", "L", "M", "N", "i", "opd", "L0", "M0", "N0"):
                    np.testing.assert_array_equal(getattr(ra, k).cpu().numpy(),
                                                  getattr(rb, k).cpu().numpy(), k)
                for k in ("x", "L", "intensity", "opd"):
                    np.testing.assert_array_equal(getattr(a.surfaces, k).cpu().numpy(),
                                                  getattr(b.surfaces, k).cpu().numpy())
    finally:
        eng.close()


@pytest.mark.parametrize("where", WHERE)
def test_packed_pair_form_of_the_generating_launch(where):
    """Round 4: the lean fp32 form of `ol_trace_generate` runs one packed PAIR of rays per lane
    (`trace_kernel<float, 2, true, 0, 0, SPOT, kGenUniform>`).  Same bits as one ray per lane
    (`OL_TUNE_RAYS_PER_THREAD = 1`) for even and odd counts, with the spot epilogue, and when
    the planes are NOT 8-byte aligned (the C ABI then keeps one ray per lane by itself)."""
    table = load_system("double_gauss")
    eng, dev = _engine(table, where)
    dtype = torch.float32
    try:
        for n in (2, 3, 511, 512, 513, 100_001):
            px, py = _pupil(n + 1, dtype, dev, n)
            got = {}
            for rpt in (0, 1):
                assert eng.lib.ol_set_tuning(_capi.TUNE_RAYS_PER_THREAD, rpt) == 0
                pairs = getattr(eng.lib, "ol_hostmath_pair_launches", None)  # host build only
                before = pairs() if pairs is not None else None
                out = [torch.empty(n, dtype=dtype, device=dev) for _ in range(8)]
                slots = eng.alloc_spot_slots()
                r = eng.trace_generate(px[:n], py[:n], 0, field=(0.0, 0.7), rays_out=out,
                                       spot=(slots, 0.0, 0.0))
                got[rpt] = (r.record[:, :, :n].cpu().numpy(), torch.stack(out).cpu().numpy(),
                            eng.reduce_spot_slots(slots).cpu().numpy())
                if pairs is not None:  # the host mirror took the form the device would take
                    assert pairs() - before == (1 if rpt == 0 else 0)
            assert eng.lib.ol_set_tuning(_capi.TUNE_RAYS_PER_THREAD, 0) == 0
            np.testing.assert_array_equal(got[0][0], got[1][0])
            np.testing.assert_array_equal(got[0][1], got[1][1])
            # count and max: exact; the sums are accumulated in another order
            assert got[0][2][0] == got[1][2][0] and got[0][2][6] == got[1][2][6]
            np.testing.assert_allclose(got[0][2], got[1][2], rtol=1e-12)
            # misaligned planes (a view one element into its storage): still the same rays
            out = [torch.empty(n + 1, dtype=dtype, device=dev)[1:] for _ in range(8)]
            m = eng.trace_generate(px[1:], py[1:], 0, field=(0.0, 0.7), rays_out=out)
            a = eng.trace_generate(px[1:].clone(), py[1:].clone(), 0, field=(0.0, 0.7))
            np.testing.assert_array_equal(m.record[:, :, :n].cpu().numpy(),
                                          a.record[:, :, :n].cpu().numpy())
            np.testing.assert_array_equal(torch.stack(out).cpu().numpy(),
                                          a.record[-1, :, :n].cpu().numpy())
    finally:
        eng.lib.ol_set_tuning(_capi.TUNE_RAYS_PER_THREAD, 0)
        eng.close()


# ------------------------------------------------------------------------------------------
# ABI 10: POLARISED generating launches with per-ray field planes / an apodized pupil
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("where", WHERE)
@pytest.mark.parametrize("case", ["zernike_fresnel_fringe", "zernike_fresnel_polarized",
                                  "polarizer_retarder", "coated_mirror_polarised", "fuzz_03"])
def test_polarised_generating_launch_with_fields_and_apodization(case, where):
    """`PolarizedRays` bundles take per-ray fields (`trace_generic(Hx[], Hy[], ...)`, the
    fields x pupil expansion of a multi-field trace: real_ray_tracer.py:88-98, 120-154) and
    apodized pupils (ray_generator.py:81-85) like any other -- since ABI 10 in ONE launch too:
    every recorded row AND the nine / eighteen PRT planes bit for bit what `ol_generate_rays`
    + `ol_trace` give, and the `update_intensity` epilogue (whose `_i0` is then the
    apodization, polarized_rays.py:51) equal to `ol_polarized_intensity` on what the launch
    wrote.  Real and complex matrices, fp32 and fp64, with and without vignetting planes."""
    import copy
    table, _ = load_case(case)
    if not table.raygen or table.polarization is None:
        pytest.skip("not a polarised case with generator scalars")
    for apod in (None, (1, 0.7, 0.0), (4, 1.0, 2.0)):
        tb = copy.deepcopy(table)
        if apod is not None:
            tb.raygen = dict(tb.raygen, apod_kind=apod[0], apod_a=apod[1], apod_b=apod[2])
        eng, dev = _engine(tb, where)
        try:
            assert eng.can_trace_generate(field_planes=True)
            cplx = 18 if tb.needs_complex_prt else 9
            for dtype in (torch.float64, torch.float32):
                n = 389
                px, py = _pupil(n, dtype, dev, 31)
                hx, hy = _fields(n, dtype, dev, 32, False)
                g = np.random.default_rng(7)
                vpl = (torch.tensor(g.uniform(0.8, 1.0, n), dtype=dtype, device=dev),
                       torch.tensor(g.uniform(0.8, 1.0, n), dtype=dtype, device=dev))
                for field, vig, flags in (((hx, hy), None, _capi.RAYGEN_CHECK_FIELD),
                                          ((hx, hy), vpl, _capi.RAYGEN_CHECK_FIELD
                                           | _capi.RAYGEN_PRESCALE_PUPIL),
                                          ((0.0, 0.6), (1.0, 1.0), 0)):
                    if apod is None and not isinstance(field[0], torch.Tensor):
                        continue   # (the launch-uniform plain form: the ABI 7 tests)
                    # two launches
                    rec_a = eng.alloc_record(n, dtype)
                    rays = eng.row0_planes(rec_a, n)
                    vx, vy = vig if vig is not None else (None, None)
                    eng.generate_rays(field[0], field[1], px, py, vx, vy, out=rays, flags=flags)
                    prt_a = torch.empty((cplx, n), dtype=dtype, device=dev)
                    a = eng.trace(rays, 0, record=rec_a, prt=prt_a, prt_identity=True,
                                  zero_status=False)
                    # one
                    prt_b = torch.empty((cplx, n), dtype=dtype, device=dev)
                    st = STATES[2]
                    b = eng.trace_generate(px, py, 0, field=field, vig=vig, flags=flags, prt=prt_b,
                                           update_intensity=st)
                    np.testing.assert_array_equal(a.record[:, :, :n].cpu().numpy(),
                                                  b.record[:, :, :n].cpu().numpy(),
                                                  err_msg=f"{case} apod={apod} {dtype}")
                    np.testing.assert_array_equal(prt_a.cpu().numpy(), prt_b.cpu().numpy())
                    if apod is not None:
                        assert not np.all(b.record[0, 6, :n].cpu().numpy() == 1.0)
                    r0 = b.rows(0)
                    want = eng.polarized_intensity(prt_b, (r0[3], r0[4], r0[5]), r0[6], st)
                    got = b.updated_intensity
                    assert got is not None
                    x, y = got.cpu().numpy(), want.cpu().numpy()
                    assert np.array_equal(np.isnan(x), np.isnan(y))
                    tol = 1e-13 if dtype == torch.float64 else 2e-6
                    np.testing.assert_allclose(x, y, rtol=tol, atol=tol)
        finally:
            eng.close()
