"""The reference's OBJECT PROTOCOL under the drop-in (build container only).

A stock reference `Optic` pickles, copies, deep-copies, round-trips through
`to_dict` / `from_dict` and prints `info()` (surfaces/standard_surface.py:90-99 keeps the
listeners out of the state for exactly that; optic/optic.py:121 hangs the tracer on the
optic).  The drop-in must not take any of it away: after a trace under `install()`,
`enable()` or `enable(lazy_records=True)` every one of these still works, the copy traces
bit-identically to the original, and nothing of this process's device state (engine handles,
id()-based change-detector tokens, the record block) travels with it.

The fused trace is stood in for by the oracle-backed engine / the host build of the kernel
source (same fixtures as tests/test_reference_integration.py).
"""

import copy
import importlib
import inspect
import io
import multiprocessing
import os
import pickle
import pkgutil
import sys

import numpy as np
import pytest

REF = os.environ.get("OPTILAND_REFERENCE", "/root/reference")
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "optiland")),
                                reason="reference package not present")

from tests.test_reference_integration import _np, hip_on_cpu, ref  # noqa: E402,F401


def _sample_classes():
    """(module, class) of every lens in `optiland.samples`, collected without importing the
    reference at collection time (the fixture owns sys.path)."""
    root = os.path.join(REF, "optiland", "samples")
    out = []
    if not os.path.isdir(root):
        return out
    for fn in sorted(os.listdir(root)):
        if not fn.endswith(".py") or fn.startswith("_"):
            continue
        with open(os.path.join(root, fn)) as fh:
            for line in fh:
                if line.startswith("class ") and "(" in line:
                    out.append((fn[:-3], line[6:line.index("(")].strip()))
    return out


SAMPLES = _sample_classes()


def _make(mod, name):
    m = importlib.import_module(f"optiland.samples.{mod}")
    return getattr(m, name)()


def _activate(mode, lens):
    from optiland_amd import integration as ig
    if mode == "install":
        ig.install(lens, force=True)
    else:
        ig.enable(force=True, lazy_records=(mode == "enable-lazy"))
    return ig


def _trace(lens):
    w = lens.primary_wavelength
    return lens.trace(0.0, 0.7 if lens.fields.max_field != 0 else 0.0, w, 4, "hexapolar")


def _same_rays(be, a, b):
    for k in ("x", "y", "z", "L", "M", "N", "i", "opd"):
        np.testing.assert_array_equal(_np(be, getattr(a, k)), _np(be, getattr(b, k)), err_msg=k)


@pytest.mark.parametrize("mode", ["install", "enable", "enable-lazy"])
def test_traced_optic_pickles_and_the_copy_traces_bit_identically(hip_on_cpu, mode):
    be = hip_on_cpu
    from optiland.samples.objectives import CookeTriplet
    lens = CookeTriplet()
    ig = _activate(mode, lens)
    try:
        r0 = _trace(lens)
        comp = ig.hip_tracer_of(lens)
        assert comp.last_path == "hip"
        blob = pickle.dumps(lens)
        # stock optic after the same trace: the same order of magnitude, not 8 (S + 1) copies
        # of the record block (the per-surface arrays are views of one block)
        assert len(blob) < 400_000, len(blob)
        twin = pickle.loads(blob)
        comp2 = ig.hip_tracer_of(twin)
        assert comp2 is not None and comp2 is not comp
        # nothing of this process's device state travelled
        assert len(comp2._hip_engines) == 0 and len(comp2._hip_memo) == 0
        assert comp2._hip_surface_cache == {} and comp2._hip_engine is None
        # the recorded state travelled (what Surface._record_real leaves behind)
        for k in ("x", "y", "z", "L", "M", "N", "intensity", "opd"):
            np.testing.assert_array_equal(_np(be, getattr(twin.surfaces, k)),
                                          _np(be, getattr(lens.surfaces, k)), err_msg=k)
        r1 = _trace(twin)
        assert comp2.last_path == "hip"
        _same_rays(be, r0, r1)
        # ... and the original is untouched by all this
        _same_rays(be, r0, _trace(lens))
        # the returned rays pickle too
        _same_rays(be, r0, pickle.loads(pickle.dumps(r0)))
    finally:
        ig.disable()


def test_polarized_bundle_pickles_with_its_matrices_unread(hip_on_cpu):
    """`PolarizedRays.p` is produced from the kernel's planes on first read: a pickle taken
    BEFORE that read must still carry the matrices."""
    be = hip_on_cpu
    from optiland.rays import PolarizationState
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import integration as ig
    lens = CookeTriplet()
    lens.set_polarization(PolarizationState(is_polarized=False))
    ig.install(lens, force=True)
    rays = lens.trace(0.0, 1.0, 0.55, 3, "hexapolar")
    assert "p" not in rays.__dict__  # still planes
    twin = pickle.loads(pickle.dumps(rays))
    np.testing.assert_array_equal(_np(be, twin.p.real), _np(be, rays.p.real))
    optic2 = pickle.loads(pickle.dumps(lens))
    r2 = optic2.trace(0.0, 1.0, 0.55, 3, "hexapolar")
    np.testing.assert_array_equal(_np(be, r2.i), _np(be, rays.i))


def _pool_job(blob):
    lens = pickle.loads(blob)
    r = lens.trace(0.0, 1.0, 0.55, 3, "hexapolar")
    return np.asarray(r.y.detach().cpu().numpy(), dtype=np.float64)


def test_multiprocessing_over_traced_optics(hip_on_cpu, monkeypatch):
    """`multiprocessing.Pool.map` over optics that were traced under the drop-in (joblib /
    tolerancing sweeps do this): the workers unpickle them and trace.  (fork start method: the
    workers inherit the patched `_make_engine`; the optics go through pickle all the same.)"""
    be = hip_on_cpu
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import integration as ig
    lenses = []
    for k in range(3):
        lens = CookeTriplet()
        ig.install(lens, force=True)
        lens.surfaces[1].geometry.radius = 22.0 + k
        lens.update_paraxial() if hasattr(lens, "update_paraxial") else None
        lens.trace(0.0, 1.0, 0.55, 3, "hexapolar")
        lenses.append(lens)
    want = [_np(be, L.trace(0.0, 1.0, 0.55, 3, "hexapolar").y) for L in lenses]
    ctx = multiprocessing.get_context("fork")
    with ctx.Pool(2) as pool:
        got = pool.map(_pool_job, [pickle.dumps(L) for L in lenses])
    for a, b in zip(got, want):
        np.testing.assert_array_equal(a, b)
    assert not np.array_equal(want[0], want[1])  # three different prescriptions


@pytest.mark.parametrize("mod,name", SAMPLES, ids=[n for _, n in SAMPLES])
def test_protocol_sweep_over_every_sample(hip_on_cpu, mod, name):
    """pickle, copy, deepcopy, to_dict / from_dict and info() on every `optiland.samples`
    lens after a trace under `install()`, each copy traced again and held against the
    original."""
    be = hip_on_cpu
    from optiland.optic import Optic
    from optiland_amd import integration as ig
    import time
    lens = _make(mod, name)
    tracer = ig.install(lens, force=True)
    t0 = time.perf_counter()
    r0 = _trace(lens)
    slow = time.perf_counter() - t0 > 1.0   # (a lens with iterative ray aiming: seconds per trace)
    path = tracer.last_path
    assert path in ("hip", "reference", "reference-rays")
    # pickle
    twin = pickle.loads(pickle.dumps(lens))
    _same_rays(be, r0, _trace(twin))
    assert ig.hip_tracer_of(twin).last_path == path
    # deepcopy: same contract
    deep = copy.deepcopy(lens)
    _same_rays(be, r0, _trace(deep))
    assert deep.surfaces.trace.__self__ is deep.surfaces  # the seam follows the copy
    assert ig.hip_tracer_of(deep).optic is deep
    if slow:
        return
    # shallow copy shares the surfaces; it must at least exist and trace
    shallow = copy.copy(lens)
    _same_rays(be, r0, _trace(shallow))
    # to_dict / from_dict: a plain reference optic comes back -- the same one a STOCK optic's
    # own round trip gives (the reference's dict form is lossy for some samples, e.g. the
    # UV projection lens; what matters here is that the drop-in adds or loses nothing)
    assert lens.to_dict() == _make(mod, name).to_dict()
    rebuilt = Optic.from_dict(lens.to_dict())
    stock = Optic.from_dict(_make(mod, name).to_dict())
    r3, r4 = _trace(rebuilt), _trace(stock)
    for k in ("x", "y", "i"):
        np.testing.assert_allclose(_np(be, getattr(r3, k)), _np(be, getattr(r4, k)),
                                   rtol=1e-9, atol=1e-9, equal_nan=True, err_msg=k)
    # info() prints the prescription table
    buf, old = io.StringIO(), sys.stdout
    sys.stdout = buf
    try:
        lens.info()
    finally:
        sys.stdout = old
    assert "Surface" in buf.getvalue() or "Type" in buf.getvalue()


def test_disable_restores_the_reference_classes(hip_on_cpu):
    """`disable()` leaves no descriptor of the drop-in on the reference's classes
    (`Surface.x ... .opd`, `PolarizedRays.p`), and a bundle handed out before keeps its `p`."""
    be = hip_on_cpu
    from optiland.rays import PolarizationState, PolarizedRays
    from optiland.samples.objectives import CookeTriplet
    from optiland.surfaces.standard_surface import Surface
    from optiland_amd import integration as ig
    ig.enable(force=True)   # (whatever earlier tests left behind: a clean disable first)
    ig.disable()
    stock_getstate = Surface.__getstate__
    assert stock_getstate.__qualname__ == "Surface.__getstate__"
    assert "p" not in PolarizedRays.__dict__
    ig.enable(force=True)
    lens = CookeTriplet()
    lens.set_polarization(PolarizationState(is_polarized=True, Ex=1.0, Ey=0.0, phase_x=0.0, phase_y=0.0))
    rays = lens.trace(0.0, 1.0, 0.55, 3, "hexapolar")
    assert "p" in PolarizedRays.__dict__ and "p" not in rays.__dict__
    ig.disable()
    assert "p" not in PolarizedRays.__dict__
    for k in ("x", "y", "z", "L", "M", "N", "intensity", "opd"):
        assert k not in Surface.__dict__, k
    assert Surface.__getstate__ is stock_getstate
    assert tuple(rays.p.shape) == (rays.x.shape[0], 3, 3)  # materialised by disable()
    assert lens.surfaces.x.shape[0] == len(lens.surfaces.surfaces)
    # and the drop-in can be switched on again afterwards
    ig.enable(force=True)
    try:
        again = lens.trace(0.0, 1.0, 0.55, 3, "hexapolar")
        np.testing.assert_array_equal(_np(be, again.p.real), _np(be, rays.p.real))
    finally:
        ig.disable()


def test_invalidate_rereads_a_data_write(hip_on_cpu):
    """The change detector cannot see `tensor.data.fill_()` (no version bump); the documented
    override `tracer.invalidate()` must then force a REAL re-read of the prescription -- the
    packer's device-scalar cache and the per-surface row cache included."""
    be = hip_on_cpu
    import torch
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import integration as ig
    lens = CookeTriplet()
    tracer = ig.install(lens, force=True)
    radius = torch.tensor(22.01359, dtype=torch.float64)
    lens.surfaces[1].geometry.radius = radius
    y0 = _np(be, lens.trace(0.0, 1.0, 0.55, 3, "hexapolar").y)
    lens.trace(0.0, 1.0, 0.55, 3, "hexapolar")   # steady state: memo + caches warm
    radius.data.fill_(33.0)
    tracer.invalidate()
    y1 = _np(be, lens.trace(0.0, 1.0, 0.55, 3, "hexapolar").y)
    assert not np.allclose(y0, y1)
    ref_lens = CookeTriplet()
    ref_lens.surfaces[1].geometry.radius = torch.tensor(33.0, dtype=torch.float64)
    np.testing.assert_allclose(y1, _np(be, ref_lens.trace(0.0, 1.0, 0.55, 3, "hexapolar").y),
                               rtol=1e-9, atol=1e-10)
    # a forced full pack (no tokens) never consults the scalar cache
    from optiland_amd.packer import pack_optic
    radius.data.fill_(25.0)
    assert abs(pack_optic(lens).surfaces[1]["radius"] - 25.0) < 1e-12


def test_subclass_reset_is_called_by_the_surface_group_seam(hip_on_cpu):
    """`SurfaceGroup.trace` resets every surface first (surface_group.py:373-380); the seam
    writes the reset state directly for stock surfaces, but a subclass with its OWN `reset()`
    must still see the call."""
    from optiland.rays import RealRays
    from optiland.samples.objectives import CookeTriplet
    from optiland_amd import integration as ig
    lens = CookeTriplet()
    ig.install(lens, force=True)
    calls = []
    surf = lens.surfaces.surfaces[2]
    cls = type(surf)

    class Counting(cls):
        def reset(self):
            calls.append(1)
            super().reset()

    surf.__class__ = Counting
    rays = RealRays([0.0, 0.1], [0.0, 0.2], [-5.0, -5.0], [0.0, 0.0], [0.0, 0.0], [1.0, 1.0],
                    [1.0, 1.0], [0.55, 0.55])
    n = len(calls)
    lens.surfaces.trace(rays)
    assert len(calls) == n + 1
