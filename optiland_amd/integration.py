"""Plug the fused HIP trace in behind a live reference installation.

Two seams of the reference are used:

1. the backend registry -- `optiland.backend._backends` (backend/__init__.py:
   100-112); `register_backend()` adds a `"hip"` entry, a `TorchBackend` pinned
   to the HIP device so every un-accelerated `be.*` call keeps working on the
   same device tensors (the reference's own test does the same with a "dummy"
   backend, tests/test_backend.py:79-85);
2. the tracer attribute -- `Optic.ray_tracer` (optic/optic.py:121,740,763);
   `install(optic)` replaces it with `OptilandHipRayTracer`, a subclass of the
   reference's `RealRayTracer` whose `trace` / `trace_generic` run the HIP path
   and then populate exactly what the reference leaves behind: the returned
   `RealRays` / `PolarizedRays` object and every `Surface`'s recorded
   x, y, z, L, M, N, intensity, opd (surfaces/standard_surface.py:260-274).

`enable()` also patches the third seam of SURVEY.md section 8b, the one that covers
direct callers: `SurfaceGroup.trace(rays, skip)` (surfaces/surface_group.py:245-257),
reached with caller-built rays by `IncoherentIrradiance` / `RadiantIntensity` with
`user_initial_rays`, `ExtendedSourceOptic.trace` and `RealImageHeightField` -- the
rays are traced IN PLACE through surfaces[skip:] by one `ol_trace` launch.

Anything the fused path does not implement (autograd, BSDF, GRIN, NURBS, thin
films, non-paraxial aiming ...) raises `UnsupportedSystem` inside the packer
and the call is forwarded, untouched, to the reference implementation.

This module imports the reference lazily; it is inert where `optiland` is not
installed (e.g. the GPU test box).
"""

from __future__ import annotations

import collections
import contextlib
import os

import numpy as np
import torch

from . import fingerprint as _fp
from . import system as _S
from . import tracer as _tracer
from .packer import UnsupportedSystem, pack_optic, pack_surfaces
from .rays import prt_to_complex

BACKEND_NAME = "hip"


def _host_or_device(v):
    """Coordinates as the device front end takes them: torch tensors untouched (device
    tensors stay on the device), python / numpy scalars as floats, anything else as a
    float64 numpy array."""
    if isinstance(v, torch.Tensor):
        return v.detach()
    if isinstance(v, (int, float)):
        return float(v)
    a = np.asarray(v, dtype=np.float64)
    return float(a) if a.ndim == 0 else a


class _PupilPoints:
    """A caller's distribution object (reference `BaseDistribution`: backend arrays in
    `.x` / `.y`) as `HipRayTracer.trace` reads it."""

    def __init__(self, dist):
        self.x, self.y = _host_or_device(dist.x), _host_or_device(dist.y)


def register_backend(name: str = BACKEND_NAME):
    """Insert the HIP backend into the reference's registry and return it."""
    import optiland.backend as be
    from optiland.backend.torch_backend import TorchBackend

    if name in be._backends:
        return be._backends[name]

    class HipBackend(TorchBackend):
        """TorchBackend on the ROCm device (+ the fused trace via install())."""

        @property
        def name(self) -> str:  # noqa: D401
            return name

        def __init__(self):
            super().__init__()
            if torch.cuda.is_available():
                self._config.set_device("cuda")

    be._backends[name] = HipBackend()
    return be._backends[name]


def _table_key(table):
    return (table.surfaces.tobytes(), table.coeffs.tobytes(), table.optics.tobytes(),
            table.wavelengths.tobytes(), repr(sorted(table.raygen.items())),
            repr(table.fields), repr(table.polarization), float(table.last_thickness))


_TRACER_CLASS = None
_ORIGINALS = {}
_MAX_ENGINES = 8  # device tables kept per tracer (a few kB each)
_MAX_MEMO = 32    # (wavelength -> change-detector token) entries kept per tracer
_EMPTY = {}       # (dtype, device) -> shared empty tensor for Surface.u / Surface.aoi


def _empty(dtype, device):
    key = (dtype, str(device))
    t = _EMPTY.get(key)
    if t is None:
        t = _EMPTY[key] = torch.empty(0, dtype=dtype, device=device)
    return t


def _surface_views(res):
    """The per-surface plane views of a record block (one tuple of eight per recorded
    surface): tensor bookkeeping only, nothing of the reference is touched -- so it can run
    while the kernels that fill the block are still in flight."""
    n = res.n
    rows = res.record[: res.last - res.first + 1, :, :n].unbind(0)
    return [row.unbind(0) for row in rows]


def _bind_surfaces(surfaces, res, views=None, owner=None):
    """Every reference `Surface` gets its recorded vectors (standard_surface.py:260-274) as
    zero-copy views of the record block: three tensor ops per surface instead of the
    reference's `reset()` (eleven fresh device tensors) + eight assignments.  `owner`: bind
    only the surfaces whose pending entry is still this object (a surface written since --
    `Surface.reset()`, a reference-side trace -- keeps what was written)."""
    if views is None:
        views = _surface_views(res)
    empty = _empty(res.record.dtype, res.record.device)
    for surf, row in zip(surfaces[res.first: res.last + 1], views):
        if owner is not None:
            if _PENDING.get(surf) is not owner:
                continue
            del _PENDING[surf]
        surf.x, surf.y, surf.z, surf.L, surf.M, surf.N, surf.intensity, surf.opd = row
        surf.u = surf.aoi = empty  # Surface.reset(): only paraxial traces fill these


# --------------------------------------------------------------------------------------
# lazy per-surface records (SURVEY.md 7 step 4, 8b output contract)
# --------------------------------------------------------------------------------------
# A trace that nobody has asked the interior surfaces of runs record-LAST (the image plane and
# the row before it: what the returned rays need); what it was launched with is remembered,
# and the first read of ANY `Surface.x / .y / ... / .opd` afterwards re-runs it record-all and
# binds every surface -- so `SurfaceGroup.x` (surface_group.py:108-153) and friends still see
# exactly what `Surface._record_real` (standard_surface.py:260-274) would have stored.  The
# fused analysis kernels (analysis_seams.py) register their traces the same way.
# Implementation: data descriptors for the eight recorded attributes on the reference's
# `Surface` class (values stay in the instance `__dict__`), and a weak side table
# surface -> pending trace.  A write to any of the eight (Surface.reset(), a reference-side
# trace) drops that surface's pending entry.
_PENDING: "weakref.WeakKeyDictionary" = None  # created with the descriptors
# eager traces too make their per-surface views on first read (`_PendingViews`);
# OPTILAND_HIP_DEFER_VIEWS=0: bind every surface inside the call, as rounds 1-2 did
_DEFER_VIEWS = os.environ.get("OPTILAND_HIP_DEFER_VIEWS", "1") != "0"
_RECORDED = ("x", "y", "z", "L", "M", "N", "intensity", "opd")


class _RecordedPlane:
    """Data descriptor of one recorded attribute of the reference's `Surface`."""

    __slots__ = ("name",)

    def __init__(self, name):
        self.name = name

    def __get__(self, obj, cls=None):
        if obj is None:
            return self
        if _PENDING:
            pend = _PENDING.get(obj)
            if pend is not None:
                pend.materialize()
        try:
            return obj.__dict__[self.name]
        except KeyError:
            raise AttributeError(self.name) from None

    def __set__(self, obj, value):
        if _PENDING:
            _written(obj)
        obj.__dict__[self.name] = value

    def __delete__(self, obj):
        if _PENDING:
            _written(obj)
        try:
            del obj.__dict__[self.name]
        except KeyError:
            raise AttributeError(self.name) from None


def _written(surf):
    """A recorded attribute of `surf` is about to be written.  Views that only wait to be made
    (`_PendingViews`: an eager trace) are made first -- the other attributes of this and of
    every other surface then hold exactly what an eager bind would have left; a trace that has
    not been RUN (`_PendingRecord`, opt-in lazy records) is not run for a write: the surface
    leaves it and keeps its reset state."""
    pend = _PENDING.get(surf)
    if pend is None:
        return
    if isinstance(pend, _PendingViews):
        pend.materialize()
    else:
        del _PENDING[surf]


def _install_lazy_descriptors():
    global _PENDING
    import weakref

    from optiland.surfaces.standard_surface import Surface

    if _PENDING is None:
        _PENDING = weakref.WeakKeyDictionary()
    if not isinstance(Surface.__dict__.get("x"), _RecordedPlane):
        for name in _RECORDED:
            setattr(Surface, name, _RecordedPlane(name))
        # copies / pickles of a surface take its `__dict__` (standard_surface.py:90-94): views
        # that only wait to be made are made first, so that the copy carries what an eager
        # bind would have left on the original
        orig = Surface.__getstate__
        _ORIGINALS["surface_getstate"] = orig

        def __getstate__(self):
            pend = _PENDING.get(self) if _PENDING else None
            if pend is not None:
                # views that only wait to be made are made; a lazy record that has not been
                # run is run: the copy carries what `Surface._record_real` would have left
                pend.materialize()
            state = orig(self)
            # the recorded vectors are VIEWS of one record block: torch pickles / deep-copies
            # the whole storage behind every view (8 x (S+1) copies of the block).  The copy
            # gets what the reference's own `_record_real` would have left: one array each
            for name in _RECORDED + ("u", "aoi"):
                t = state.get(name)
                if isinstance(t, torch.Tensor) and t.numel() * t.element_size() \
                        < t.untyped_storage().nbytes():
                    state[name] = t.clone()
            return state

        Surface.__getstate__ = __getstate__


def _remove_lazy_descriptors():
    from optiland.surfaces.standard_surface import Surface

    if _PENDING:
        for pend in {id(p): p for p in list(_PENDING.values())}.values():
            pend.materialize()  # nothing may stay unbound once the descriptors are gone
    for name in _RECORDED:
        if isinstance(Surface.__dict__.get(name), _RecordedPlane):
            delattr(Surface, name)
    orig = _ORIGINALS.pop("surface_getstate", None)
    if orig is not None:
        Surface.__getstate__ = orig


class _LazyPrt:
    """Data descriptor for `PolarizedRays.p` (rays/polarized_rays.py:50-54: the (N, 3, 3)
    complex polarisation ray-tracing matrices).  The kernels keep the matrices as nine (or,
    with a retarder, eighteen) real planes; the reference's layout is a transposing copy of
    360 MB into 720 MB at 1e7 rays plus a NaN scan that synchronises -- 0.5 of the 0.9 ms a
    polarised `Optic.trace_generic` took.  A traced bundle carries the planes
    (`__dict__["_hip_prt"]`, so that copies of the object carry them too) and gets its `p`
    when somebody reads it; a write behaves like a plain attribute.  Installed on the
    reference's class while the drop-in is in use (`_install_lazy_prt`); `disable()` gives
    every live bundle that still waits its `p` and takes the descriptor off the class again
    (`_remove_lazy_prt`)."""

    def __get__(self, obj, cls=None):
        if obj is None:
            return self
        d = obj.__dict__
        try:
            return d["p"]
        except KeyError:
            planes = d.pop("_hip_prt", None)
            if planes is None:
                raise AttributeError("p") from None
            out = d["p"] = prt_to_complex(planes)
            return out

    def __set__(self, obj, value):
        d = obj.__dict__
        d.pop("_hip_prt", None)
        d["p"] = value

    def __delete__(self, obj):
        d = obj.__dict__
        had = d.pop("_hip_prt", None) is not None
        if d.pop("p", None) is None and not had:
            raise AttributeError("p")


_PRT_WAITING = None  # weak set of ray bundles that carry `_hip_prt` planes instead of `p`


def _install_lazy_prt():
    global _PRT_WAITING
    import weakref

    from optiland.rays import PolarizedRays as RefPolarizedRays

    if _PRT_WAITING is None:
        _PRT_WAITING = weakref.WeakSet()
    if not isinstance(RefPolarizedRays.__dict__.get("p"), _LazyPrt):
        RefPolarizedRays.p = _LazyPrt()


def _carry_prt_planes(rays, planes):
    """`rays.p` = the matrices these planes hold, produced on first read."""
    d = rays.__dict__
    d.pop("p", None)
    d["_hip_prt"] = planes
    _PRT_WAITING.add(rays)


def _remove_lazy_prt():
    """`disable()`: the reference's `PolarizedRays` is a plain class again.  Bundles the
    drop-in handed out and nobody has read the `p` of get it now (a deep copy of such a
    bundle, taken before the read, is not known here: read `p` before `disable()`)."""
    from optiland.rays import PolarizedRays as RefPolarizedRays

    if _PRT_WAITING is not None:
        for rays in list(_PRT_WAITING):
            planes = rays.__dict__.pop("_hip_prt", None)
            if planes is not None and "p" not in rays.__dict__:
                rays.__dict__["p"] = prt_to_complex(planes)
        _PRT_WAITING.clear()
    if isinstance(RefPolarizedRays.__dict__.get("p"), _LazyPrt):
        del RefPolarizedRays.p


class _PendingRecord:
    """One record-all trace that has not been run: the table it was launched on, the inputs
    of the fused launch, the optic whose surfaces it belongs to."""

    def __init__(self, optic, table, engine, dtype, launch):
        import weakref

        self.optic = weakref.ref(optic)
        self.table, self.engine, self.dtype, self.launch = table, engine, dtype, launch
        self.done = False

    def materialize(self):
        if self.done:
            return
        self.done = True
        optic = self.optic()
        if optic is None:
            return
        surfaces = optic.surfaces.surfaces
        if len(surfaces) != self.table.num_surfaces:
            for s in surfaces:
                if _PENDING.get(s) is self:
                    del _PENDING[s]
            return  # the optic was rebuilt in between: nothing sensible to bind
        eng = self.engine
        if getattr(eng, "_handle", True) is None or getattr(eng, "table", self.table) \
                is not self.table:  # evicted and closed, or patched to another prescription
            eng = _tracer._make_engine(self.table, getattr(eng, "device", None))
        front = _tracer.HipRayTracer(self.table, dtype=self.dtype, engine=eng)
        hx, hy, px, py, vig, wavelength, flags = self.launch
        front._run(hx, hy, px, py, vig, wavelength, False, flags)  # record-all, eager
        _bind_surfaces(surfaces, front.surfaces._res, owner=self)
        front.surfaces._bind(None)
        self.launch = None


class _PendingViews:
    """An EAGER trace whose record block exists, but whose per-surface plane views have not
    been made: 13 surfaces x (an unbind + ten attribute writes) are a quarter of the host time
    of a small trace, and most callers read the returned rays only.  The views are made when
    somebody reads a recorded attribute of any surface of the optic."""

    def __init__(self, group, res):
        import weakref

        self.group = weakref.ref(group)  # the SurfaceGroup whose surfaces[first..last] it binds
        self.res = res
        self.done = False

    def materialize(self):
        if self.done:
            return
        self.done = True
        group, res = self.group(), self.res
        self.res = None
        if group is None:
            return
        surfaces = group.surfaces
        if len(surfaces) <= res.last:
            for s in surfaces:
                if _PENDING.get(s) is self:
                    del _PENDING[s]
            return
        _bind_surfaces(surfaces, res, owner=self)


def _mark_pending(surfaces, pend, dtype):
    empty = None
    for surf in surfaces:
        d = surf.__dict__
        if empty is None:
            old = d.get("x")
            empty = _empty(dtype, old.device if isinstance(old, torch.Tensor) else "cpu")
        d["x"] = d["y"] = d["z"] = d["L"] = d["M"] = d["N"] = d["intensity"] = d["opd"] = empty
        d["u"] = d["aoi"] = empty
        _PENDING[surf] = pend
    return pend


def register_pending_views(group, res):
    """Eager trace of `group.surfaces[res.first .. res.last]`, views deferred
    (`_PendingViews`); same bookkeeping as a pending record."""
    _install_lazy_descriptors()
    return _mark_pending(group.surfaces[res.first: res.last + 1], _PendingViews(group, res),
                         res.record.dtype)


def register_pending_record(optic, table, engine, dtype, launch):
    """Mark every surface of `optic` as "recorded state = this trace, not yet run".  The
    instance attributes are emptied first (as `Surface.reset()` leaves them): nothing can
    see stale arrays.  A copy / pickle of a surface (`Surface.__getstate__`) runs the trace."""
    _install_lazy_descriptors()
    return _mark_pending(optic.surfaces.surfaces,
                         _PendingRecord(optic, table, engine, dtype, launch), dtype)


def forget_pending_record(optic):
    """Drop a trace that was registered for `optic`'s surfaces and has not been RUN (a launch
    whose status turned out bad after it was registered): nothing will try to produce it."""
    if not _PENDING:
        return
    for surf in optic.surfaces.surfaces:
        if isinstance(_PENDING.get(surf), _PendingRecord):
            del _PENDING[surf]


class _Volatile:
    """Holder of drop-in state that hangs on a REFERENCE object (`SurfaceGroup.__dict__`) and
    must not travel with it: change-detector tokens embed `id()`s, engines own device handles.
    `copy.copy` / `copy.deepcopy` / `pickle` of the reference object carry an EMPTY holder --
    the copy re-packs on its first trace, exactly like a freshly built optic."""

    __slots__ = ("value",)

    def __init__(self, value=None):
        self.value = value

    def __reduce__(self):
        return (_Volatile, ())

    def __copy__(self):
        return _Volatile()

    def __deepcopy__(self, memo):
        return _Volatile()


class _VolatileCache(collections.OrderedDict):
    """LRU of (engine, table) entries on a reference `SurfaceGroup`; copies and pickles of the
    group start with an empty one (see `_Volatile`)."""

    def __reduce__(self):
        return (_VolatileCache, ())

    def __copy__(self):
        return _VolatileCache()

    def __deepcopy__(self, memo):
        return _VolatileCache()


class _BoundSgTrace:
    """`install(optic)`'s per-instance replacement of `SurfaceGroup.trace`: what
    `types.MethodType(_sg_trace, group)` would be, but one that survives `pickle` (a bound
    method of a module-level function pickles as `getattr(group, "_sg_trace")`, which does
    not exist) and `copy.deepcopy` (re-bound to the COPY of the group)."""

    __slots__ = ("group",)

    def __init__(self, group):
        self.group = group

    def __call__(self, rays, skip=0):
        return _sg_trace(self.group, rays, skip)

    @property
    def __self__(self):
        return self.group

    def __reduce__(self):
        return (_BoundSgTrace, (self.group,))

    def __deepcopy__(self, memo):
        import copy
        return _BoundSgTrace(copy.deepcopy(self.group, memo))


# attributes of an `OptilandHipRayTracer` that are caches of THIS process / device (handles,
# id()-based tokens, the last table): a copy or a pickle of the tracer starts without them
_TRACER_VOLATILE = {
    "_hip_engines": collections.OrderedDict, "_hip_memo": collections.OrderedDict,
    "_hip_surface_cache": dict, "_hip_engine": lambda: None, "_hip_table": lambda: None,
    "_hip_trusted": lambda: None, "_hip_trust_depth": lambda: 0,
}


def _restore_tracer(state):
    """Unpickle an `OptilandHipRayTracer` (its `__reduce__`): the class is created on first
    use of the drop-in (it subclasses the reference's `RealRayTracer`, imported lazily), so a
    pickle cannot name it -- it names this function instead."""
    cls = _make_tracer_class()
    new = cls.__new__(cls)
    new.__dict__.update(state)
    for k, make in _TRACER_VOLATILE.items():
        new.__dict__[k] = make()
    return new


def _make_tracer_class():
    global _TRACER_CLASS
    if _TRACER_CLASS is not None:
        return _TRACER_CLASS
    import optiland.backend as be
    from optiland.raytrace.real_ray_tracer import RealRayTracer
    from optiland.rays import PolarizedRays as RefPolarizedRays
    from optiland.rays import RealRays as RefRealRays

    class OptilandHipRayTracer(RealRayTracer):
        """`RealRayTracer` whose ray generation + surface loop run in HIP kernels.

        Device-resident: coordinates that arrive as device tensors are used as they are
        and range-checked inside the ray-generation kernel (`OL_RAYGEN_CHECK_*`), the rays
        are generated straight into row 0 of the record block, the returned rays and
        every `Surface`'s recorded arrays are views of that block.  The host does a
        change check of the optic (`fingerprint.optic_token`, no device access), two
        launches and ONE status read-back per call."""

        def __init__(self, optic, device=None, force=False):
            super().__init__(optic)
            self._hip_device = device
            self._hip_force = force  # tests: intercept regardless of backend/device
            # True: traces of ONE field point run record-last and the per-surface arrays are
            # materialised on first access (see the lazy-record block above)
            self._hip_lazy = False
            # packed-table fingerprint -> (engine, table, {dtype: HipRayTracer}): analyses
            # alternate between the optic's wavelengths call by call, so the last few
            # device tables stay alive instead of being re-created for every trace
            self._hip_engines = collections.OrderedDict()
            # wavelength -> (token, kept objects, table key | UnsupportedSystem)
            self._hip_memo = collections.OrderedDict()
            self._hip_surface_cache = {}  # id(surface) -> (token, packed row): incremental pack
            self._hip_engine = None  # most recently used (introspection)
            self._hip_table = None
            self.pack_count = 0      # packs really performed (introspection for tests)
            self.engine_updates = 0  # ... of which patched the device table in place
            self.speculative_hits = 0    # launches queued before the change check, kept
            self.speculative_misses = 0  # ... dropped because the optic had changed
            self._hip_spec_ok = True     # False right after a miss: validate before launching
            # inside `unchanged(optic)`: wavelengths whose table was validated in the scope --
            # their change check is not repeated until the scope ends (None: no scope)
            self._hip_trusted = None
            self._hip_trust_depth = 0
            self.last_path = None  # "hip" | "reference" (introspection for tests)

        def _portable_state(self):
            """`__dict__` minus this process's device caches (handles, id()-based tokens, the
            last record block -- gigabytes at 1e7 rays)."""
            return {k: v for k, v in self.__dict__.items() if k not in _TRACER_VOLATILE}

        def __deepcopy__(self, memo):
            # the reference deep-copies optics (tolerancing, optimisation): the copy starts
            # with empty device caches instead of clones of handles and tokens
            import copy
            new = type(self).__new__(type(self))
            memo[id(self)] = new
            for k, v in self._portable_state().items():
                new.__dict__[k] = copy.deepcopy(v, memo)
            for k, make in _TRACER_VOLATILE.items():
                new.__dict__[k] = make()
            return new

        def __reduce__(self):
            # pickle (multiprocessing / joblib over optics, optic/optic.py:121: the tracer is
            # an attribute of the Optic): same contract as a deep copy.  The two-argument
            # form lets pickle restore the state AFTER the object exists, so the cycle
            # tracer -> optic -> tracer is handled by its memo.
            return (_restore_tracer, ({},), self._portable_state())

        def __setstate__(self, state):
            self.__dict__.update(state)

        # ---------------------------------------------------------- eligibility
        def _eligible(self) -> bool:
            if be.get_backend() not in (BACKEND_NAME, "torch"):
                return False  # NumPy-backend traces are never intercepted
            inst = be._backends[be.get_backend()]
            if inst._config.grad_mode.requires_grad:  # autograd stays on torch ops
                return False
            if self._hip_force:  # tests: skip the device check only
                return True
            return inst._config.get_device() == "cuda"

        def _dtype(self):
            if be.get_backend() in (BACKEND_NAME, "torch"):
                return be._backends[be.get_backend()]._config.get_precision()
            return torch.float64

        def stats(self) -> dict:
            """What the drop-in holds and did for this optic: `last_path` ("hip",
            "reference-rays", "reference"), `packs` (tables really packed), `engines` (device
            tables alive for it), and the process-wide record pools (`placed_bytes`: device
            memory of the library's own arenas, `pools`: one entry per pooled block shape;
            `engine.record_pool_stats()`)."""
            from . import engine as _E

            pools = _E.record_pool_stats()
            return {"last_path": getattr(self, "last_path", None), "packs": self.pack_count,
                    "engines": len(getattr(self, "_hip_engines", ()) or ()),
                    "placed_bytes": pools["placed_bytes"], "pools": pools["pools"],
                    "pool_idle_s": pools["idle_s"]}

        def invalidate(self):
            """Forget everything the change detector relies on: the token memo, the
            per-surface packed rows and the device scalars read back so far.  The next trace
            re-reads the WHOLE prescription from the live objects (the manual override for
            edits the detector cannot see, e.g. a `tensor.data` write)."""
            from . import packer as _packer

            self._hip_memo.clear()
            self._hip_surface_cache.clear()
            self._hip_spec_ok = False  # validate before launching, not after
            _packer.clear_tensor_values()

        def _entry_for(self, wavelength, tok=None, keep=None):
            """(engine, table, fronts) for the optic AS IT IS NOW at `wavelength`.  `tok`,
            `keep`: its change-detector token (and the objects the token's ids belong to) if
            the caller has just taken it."""
            w = float(wavelength.item()) if hasattr(wavelength, "item") else float(wavelength)
            _keep = keep if tok is not None else None
            trusted = self._hip_trusted
            if trusted is not None and tok is None and w in trusted:
                memo = self._hip_memo.get(w)  # validated earlier in this `unchanged` scope
                hit = self._hip_engines.get(memo[2]) if memo is not None \
                    and not isinstance(memo[2], UnsupportedSystem) else None
                if hit is not None:
                    self._hip_engine, self._hip_table = hit[0], hit[1]
                    return hit
            if trusted is not None:
                trusted.add(w)
            if _fp.ENABLED:
                if tok is None:
                    seen = getattr(trusted, "token", None)
                    if seen is not None:   # (same optic, same scope: only the wavelength differs)
                        tok, _keep = (w,) + seen[0][1:], seen[1]
                    else:
                        tok, _keep = _fp.optic_token(self.optic, w)
                        if trusted is not None:
                            trusted.token = (tok, _keep)
                memo = self._hip_memo.get(w)
                if memo is not None and memo[0] == tok:
                    key = memo[2]
                    if isinstance(key, UnsupportedSystem):
                        raise key
                    hit = self._hip_engines.get(key)
                    if hit is not None:
                        self._hip_engines.move_to_end(key)
                        self._hip_engine, self._hip_table = hit[0], hit[1]
                        return hit
            self.pack_count += 1
            try:
                # incremental: surfaces whose change-detector token is the one they were
                # last packed under are taken from the per-surface cache, not read again
                table = pack_optic(self.optic, wavelengths=[w],
                                   tokens=None if tok is None else tok[1],
                                   cache=self._hip_surface_cache, keep=_keep)
            except UnsupportedSystem as exc:
                self._remember(w, exc)
                raise
            key = _table_key(table)
            hit = self._hip_engines.get(key)
            if hit is None:
                # an optic edited between traces: the engine this wavelength was traced on
                # last time is patched in place (ol_system_update: four small async copies)
                # instead of being replaced by a new one (allocations, blocking copies)
                old_memo = self._hip_memo.get(w)
                old_key = old_memo[2] if old_memo is not None else None
                old = self._hip_engines.get(old_key) if isinstance(old_key, tuple) else None
                upd = getattr(old[0], "update", None) if old is not None else None
                try:
                    patched = upd is not None and upd(table)
                except Exception:
                    # a failed in-place update (ol_system_update) may have left the engine's
                    # tables behind the prescription: it must not stay cached under its key
                    self._hip_engines.pop(old_key, None)
                    if hasattr(old[0], "close"):
                        old[0].close()
                    patched = False
                if patched:
                    del self._hip_engines[old_key]
                    for front in old[2].values():  # same engine, new prescription
                        front.rebind(table)
                    hit = (old[0], table, old[2])
                    self.engine_updates += 1
                else:
                    hit = (_tracer._make_engine(table, self._hip_device), table, {})
                self._hip_engines[key] = hit
                while len(self._hip_engines) > _MAX_ENGINES:  # evict least recently used
                    _, (old, _t, _f) = self._hip_engines.popitem(last=False)
                    if hasattr(old, "close"):
                        old.close()
            else:
                self._hip_engines.move_to_end(key)
            self._remember(w, key, None if (tok is None or _keep is None) else (tok, _keep))
            self._hip_engine, self._hip_table = hit[0], hit[1]
            return hit

        def _remember(self, w, key, before=None):
            if not _fp.ENABLED:
                return
            # token taken AFTER the pack: whatever the pack itself touched (lazy caches of
            # the reference objects) is then part of the steady state.  A RE-pack of a
            # wavelength that has been packed before reads (almost) nothing new -- the lazy
            # caches exist, unchanged surfaces come from the per-surface cache -- so the token
            # taken before it (`before`) stands; should the pack have touched something after
            # all, the next trace sees a mismatch and re-packs once more, incrementally.
            if before is not None and w in self._hip_memo:
                tok, keep = before
            else:
                tok, keep = _fp.optic_token(self.optic, w)
            self._hip_memo[w] = (tok, keep, key)
            self._hip_memo.move_to_end(w)
            while len(self._hip_memo) > _MAX_MEMO:
                self._hip_memo.popitem(last=False)

        def _front_for(self, wavelength, tok=None, keep=None, wavefront_fp64=False):
            """The stand-alone device tracer (`tracer.HipRayTracer`) on the current table in
            the backend's precision: it owns the whole device-side call sequence.
            `wavefront_fp64`: for the wavefront seams of a float32 backend -- a front that
            computes in fp64 (an OPD in waves is a difference of path lengths 1e5 waves long:
            the fused wavefront kernels are fp64 only) and says in
            `_hip_out_dtype` what precision its results are handed over in."""
            eng, table, fronts = self._entry_for(wavelength, tok, keep)
            dtype = self._dtype()
            key = dtype
            if wavefront_fp64 and dtype != torch.float64:
                key, dtype = ("wavefront", dtype), torch.float64
            front = fronts.get(key)
            if front is None:
                front = fronts[key] = _tracer.HipRayTracer(table, dtype=dtype, engine=eng)
                front._hip_out_dtype = key[1] if isinstance(key, tuple) else None
            front.ray_aiming_config = self.ray_aiming_config
            front.lazy_records = self._hip_lazy
            return front, table

        # ---------------------------------------------------------------- trace
        def _finish(self, front, table, mine, wavelength, update_intensity,
                    before_commit=None):
            """Hand the device results over in the reference's own classes: the returned
            `RealRays` / `PolarizedRays` and every `Surface`'s recorded arrays.

            Everything that only BUILDS objects (plane views, the rays object, the epilogue
            launches) comes first; `before_commit()` -- the status read-back of a launch whose
            checks were deferred, the one synchronisation of the call, which may raise -- runs
            after it and before the first reference object is modified: the host work
            overlaps the kernels instead of following them, and a range error still leaves
            the optic's surfaces as they were."""
            lazy = front.last_was_lazy
            res = front._last_res if lazy else front.surfaces._res
            views = None if (lazy or _DEFER_VIEWS) else _surface_views(res)
            n, dtype, dev = res.n, res.record.dtype, res.record.device
            polarized = table.polarization is not None
            cls = RefPolarizedRays if polarized else RefRealRays
            out = cls.__new__(cls)  # fill attributes directly: no be.* round trip
            out.x, out.y, out.z, out.L, out.M, out.N, out.i, out.opd = mine.planes()
            out.w = torch.full((n,), float(wavelength), dtype=dtype, device=dev)
            out.is_normalized = True
            out.L0, out.M0, out.N0 = mine.L0, mine.M0, mine.N0
            if polarized:
                # `p` in the reference's (N, 3, 3) complex layout: produced on first read
                _install_lazy_prt()
                _carry_prt_planes(out, mine._prt)
                out._i0, out._L0, out._M0, out._N0 = mine._i0, mine._L0, mine._M0, mine._N0
            # final propagation by the image thickness (0 in every sample; identity then):
            # real_ray_tracer.py:104-110, BEFORE the polarised epilogue (:112-113)
            thick = float(table.last_thickness)
            if thick != 0.0:
                last_surface = self.optic.surfaces[-1]
                last_surface.material_post.propagation_model.propagate(out, thick)
            if polarized and update_intensity:
                # (the state as packed: the change detector covers the live object, and reading
                # it again would be four blocking read-backs in front of the epilogue launch)
                fused = getattr(mine, "_i_updated", None)  # epilogue of the trace launch
                out.i = fused if fused is not None else front.engine.polarized_intensity(
                    mine._prt, (mine._L0, mine._M0, mine._N0), mine._i0, table.polarization)
            if before_commit is not None:
                before_commit()
            if thick == 0.0 and front.last_status & _S.STATUS_NAN_DIRECTION:
                # The reference's trace ends with `x += t L` by the last thickness
                # (real_ray_tracer.py:104-110, homogeneous.py:40-42) even when that is 0 -- and
                # 0 * NaN is NaN: a ray that was totally reflected at the last surface (a
                # position, no direction) comes back WITHOUT a position, while the surface's
                # recorded row keeps it.  The kernel says when there is such a ray
                # (OL_STATUS_NAN_DIRECTION); only then do the returned rays get planes of their
                # own for x, y, z.
                out.x = out.x + 0.0 * out.L
                out.y = out.y + 0.0 * out.M
                out.z = out.z + 0.0 * out.N
            if lazy:
                register_pending_record(self.optic, table, front.engine, front.dtype,
                                        front.last_fused_launch)
            elif _DEFER_VIEWS:
                register_pending_views(self.optic.surfaces, res)
            else:
                _bind_surfaces(self.optic.surfaces.surfaces, res, views)
            # the record block now lives exactly as long as the reference objects that view
            # it (the Surfaces, the returned rays): the cached front must not pin it too --
            # up to _MAX_ENGINES x dtypes fronts would each hold their last 4 GB at 1e7 rays
            front.surfaces._bind(None)
            return out

        def _speculate(self, wavelength):
            """(front, table, token, w) of the table memoised for this wavelength, NOT yet
            validated against the live optic -- or None when there is nothing to gamble on."""
            if not _fp.ENABLED:
                return None
            w = float(wavelength.item()) if hasattr(wavelength, "item") else float(wavelength)
            memo = self._hip_memo.get(w)
            if memo is None or isinstance(memo[2], UnsupportedSystem):
                return None
            hit = self._hip_engines.get(memo[2])
            if hit is None or not hit[1].raygen:
                return None
            self._hip_engines.move_to_end(memo[2])  # speculative use is use: keep LRU honest
            eng, table, fronts = hit
            dtype = self._dtype()
            front = fronts.get(dtype)
            if front is None:
                front = fronts[dtype] = _tracer.HipRayTracer(table, dtype=dtype, engine=eng)
            front.ray_aiming_config = self.ray_aiming_config
            front.lazy_records = self._hip_lazy
            return front, table, memo[0], w

        def _run(self, wavelength, call, update_intensity, original):
            """One intercepted trace.  `call(front)` performs the device call sequence on a
            `HipRayTracer`; `original()` is the reference's own method.

            The launch is SPECULATIVE when a table is memoised for this wavelength: the
            kernels are queued on it at once and the change check of the live optic
            (fingerprint.optic_token, ~0.1 ms of pure host work) runs while the GPU traces;
            only then comes the one status read-back.  If the optic did change, the results
            of that launch are dropped (they live in a fresh block nobody has seen) and the
            call is repeated on a re-packed table."""
            # (an optic that is being edited between traces -- an optimiser, a tolerancing
            # loop -- misses every time: after a miss the next call validates first instead
            # of queueing a launch that is thrown away)
            spec = self._speculate(wavelength) if self._hip_spec_ok else None
            self._hip_spec_ok = True
            tok = _keep = None
            if spec is not None:
                front, table, tok0, w = spec
                mine = err = None
                front.defer_checks = True
                try:
                    mine = call(front)
                except Exception as exc:  # noqa: BLE001 - judged after the change check
                    err = exc
                finally:
                    front.defer_checks = False
                trusted = self._hip_trusted
                if trusted is not None and w in trusted:
                    tok = tok0          # validated earlier in this `unchanged` scope
                else:
                    tok, _keep = _fp.optic_token(self.optic, w)
                    if trusted is not None and tok == tok0:
                        trusted.add(w)
                if tok == tok0:
                    self.speculative_hits += 1
                    self._hip_engine, self._hip_table = front.engine, table
                    try:
                        if err is not None:
                            raise err
                        out = self._finish(front, table, mine, wavelength, update_intensity,
                                           before_commit=front.check_status)
                    except BaseException:
                        # the cached front must not pin the record block of a failed call
                        front.surfaces._bind(None)
                        front._last_res = None
                        raise
                    self.last_path = "hip"
                    return out
                self.speculative_misses += 1
                self._hip_spec_ok = False
                front.surfaces._bind(None)  # drop the stale launch's block
                front._last_res = None
            packs = self.pack_count
            try:
                front, table = self._front_for(wavelength, tok, _keep)
            except UnsupportedSystem:
                self.last_path = "reference"
                return original()
            if not table.raygen:
                # aiming mode / field type outside the device generator: the reference
                # builds the rays, its surface loop enters the HIP path through the
                # SurfaceGroup.trace seam (when enable() / install() patched it)
                self.last_path = "reference-rays"
                return original()
            # an optic that had to be re-packed for this call is being edited between traces:
            # the next call validates first as well; speculation resumes after a quiet call
            self._hip_spec_ok = self.pack_count == packs
            mine = call(front)
            self.last_path = "hip"
            return self._finish(front, table, mine, wavelength, update_intensity)

        def trace(self, Hx, Hy, wavelength, num_rays=100, distribution="hexapolar"):
            def original():
                return _ORIGINALS["trace"](self, Hx, Hy, wavelength, num_rays, distribution)
            if not self._eligible():
                self.last_path = "reference"
                return original()
            dist = distribution if isinstance(distribution, str) else _PupilPoints(distribution)
            hx, hy = _host_or_device(Hx), _host_or_device(Hy)
            return self._run(wavelength,
                             lambda front: front.trace(hx, hy, wavelength, num_rays, dist,
                                                       update_intensity="defer"),
                             True, original)

        def trace_generic(self, Hx, Hy, Px, Py, wavelength):
            def original():
                return _ORIGINALS["trace_generic"](self, Hx, Hy, Px, Py, wavelength)
            if not self._eligible() or _tracer.coordinates_unlike_the_reference(Hx, Hy, Px, Py):
                # (lists, tuples, 2-D arrays: the reference's own method raises what the
                # reference raises -- real_ray_tracer.py:120-194)
                self.last_path = "reference"
                return original()
            args = [_host_or_device(v) for v in (Hx, Hy, Px, Py)]
            return self._run(wavelength, lambda front: front.trace_generic(*args, wavelength),
                             False, original)

    # the reference's own implementations, captured before enable() can patch them
    _ORIGINALS["trace"] = RealRayTracer.trace
    _ORIGINALS["trace_generic"] = RealRayTracer.trace_generic
    from optiland.surfaces.surface_group import SurfaceGroup

    _ORIGINALS["sg_trace"] = SurfaceGroup.trace
    _TRACER_CLASS = OptilandHipRayTracer
    return OptilandHipRayTracer


# --------------------------------------------------------------------------------------
# SurfaceGroup.trace(rays, skip) -- the seam for callers that bring their own rays
# --------------------------------------------------------------------------------------
_SG = {"device": None, "force": False, "count": 0, "fallbacks": 0, "foreign": 0}
# settings of the class-wide patch (enable())
_ENABLE = {"device": None, "force": False, "lazy": False}
_PLANE_ATTRS = ("x", "y", "z", "L", "M", "N", "i", "opd")


def _sg_force(group) -> bool:
    return bool(_SG["force"] or group.__dict__.get("_hip_force", False))


def _sg_backend_ok(be, force: bool) -> bool:
    if be.get_backend() not in (BACKEND_NAME, "torch"):
        return False
    cfg = be._backends[be.get_backend()]._config
    if cfg.grad_mode.requires_grad:
        return False
    return True if force else cfg.get_device() == "cuda"


def _sg_engine(group, table, dev):
    """(engine, table) for this SurfaceGroup's packed table on the rays' device,
    LRU-cached on the group against the packed bytes (a changed surface re-packs and
    misses)."""
    cache = group.__dict__.setdefault("_hip_engines", _VolatileCache())
    dev = dev if dev.type == "cuda" else (group.__dict__.get("_hip_device") or _SG["device"])
    key = (_table_key(table), str(dev))
    hit = cache.get(key)
    if hit is None:
        hit = cache[key] = (_tracer._make_engine(table, dev), table)
        while len(cache) > _MAX_ENGINES:
            _, (old, _t) = cache.popitem(last=False)
            if hasattr(old, "close"):
                old.close()
    else:
        cache.move_to_end(key)
    return hit


def _sg_table(group, wavelength):
    """`pack_surfaces(group.surfaces, tolerate=True)`, memoised on the group against the
    change-detector token of its surfaces (fingerprint.py): a caller that pushes bundle
    after bundle through an unchanged SurfaceGroup packs it once.  None = the fused path
    refuses the whole group."""
    tok = None
    if _fp.ENABLED:
        tok, _keep = _fp.surfaces_token(group.surfaces, wavelength)
        memo = group.__dict__.get("_hip_sg_memo")
        memo = memo.value if memo is not None else None
        if memo is not None and memo[0] == tok:
            return memo[2]
    try:
        table = pack_surfaces(group.surfaces, [wavelength], name="SurfaceGroup", tolerate=True)
    except UnsupportedSystem:
        table = None
    if tok is not None:
        tok, keep = _fp.surfaces_token(group.surfaces, wavelength)  # after the pack
        group.__dict__["_hip_sg_memo"] = _Volatile((tok, keep, table))
    return table


def _sg_planes(rays, force: bool):
    """The 8 state planes of a reference ray bundle as they must be for a launch, or
    None (not device tensors of one floating dtype / size, autograd attached)."""
    planes = [getattr(rays, k, None) for k in _PLANE_ATTRS]
    if any(not isinstance(t, torch.Tensor) for t in planes):
        return None
    dtype, dev, n = planes[0].dtype, planes[0].device, planes[0].numel()
    # (one-dimensional planes of ONE length: a bundle whose planes disagree in shape -- what
    # `trace_generic` builds from 2-D coordinate arrays, (3,3) intensities next to (9,)
    # positions -- fails to broadcast in the reference's loop, and that loop is who says so)
    if dtype not in (torch.float32, torch.float64) or n == 0 \
            or any(t.dtype != dtype or t.device != dev or t.ndim != 1 or t.numel() != n
                   or t.requires_grad for t in planes):
        return None
    if not force and dev.type != "cuda":
        return None
    return planes


def _sg_run(group, eng, rays, planes, first, last, polarized, nonunit=False):
    """One fused launch over surfaces[first..last]; rays and surfaces updated in place.
    nonunit: a polarised bundle whose direction cosines are not unit vectors
    (`OL_TRACE_NONUNIT_K`)."""
    n, dtype = planes[0].numel(), planes[0].dtype
    planes = [t.detach().reshape(-1).contiguous() for t in planes]
    prt = None
    if polarized:
        p = rays.p.reshape(n, 9)
        # 18 planes (real + imaginary) only when they can be non-zero: behind a retarder in
        # THIS table, or in a bundle that arrives with an imaginary part already (traced
        # through one elsewhere); otherwise the 9-plane real form and the kernels that go
        # with it (92-99 VGPRs for the complex-PRT instances against 60-70)
        need18 = bool(eng.table.needs_complex_prt)
        if p.is_complex() and not need18:
            need18 = bool(torch.count_nonzero(p.imag))
        if need18:
            if p.is_complex():
                prt = torch.cat([p.real.t(), p.imag.t()]).to(dtype).contiguous()  # (18, n)
            else:  # a fresh torch-backend PolarizedRays holds a REAL identity
                prt = torch.cat([p.t().to(dtype), torch.zeros((9, n), dtype=dtype,
                                                              device=p.device)]).contiguous()
        else:
            prt = (p.real if p.is_complex() else p).t().to(dtype).contiguous()  # (9, n)
    res = eng.trace(planes, 0, record=True, prt=prt, first=first, last=last,
                    **({"nonunit_directions": True} if nonunit else {}))
    if _DEFER_VIEWS:
        register_pending_views(group, res)   # the traced surfaces' views: on first read
    else:
        for s in range(first, last + 1):
            surf = group.surfaces[s]
            surf.x, surf.y, surf.z, surf.L, surf.M, surf.N, surf.intensity, surf.opd = \
                res.rows(s)
    pre = res.rows(last - 1)[3:6] if last > first else planes[3:6]
    if not (first == 0 and last == 0):  # an object surface alone interacts with nothing
        # the reference stores L0.. inside refract() / reflect(), i.e. AFTER localize():
        # the pre-interaction cosines in the LAST surface's frame (real_rays.py:170-172)
        srow = eng.table.surfaces[last]
        if srow["flags"] & 1:  # SURF_ROTATED
            R = torch.as_tensor(np.asarray(srow["rot"]).reshape(3, 3), dtype=dtype,
                                device=planes[0].device)
            pre = tuple(R @ torch.stack(list(pre)))
        rays.L0, rays.M0, rays.N0 = pre
    rays.x, rays.y, rays.z, rays.L, rays.M, rays.N, rays.i, rays.opd = res.rows(last)
    if polarized:
        _install_lazy_prt()
        _carry_prt_planes(rays, prt)        # `p` is produced from the planes on first read


def _unit_directions(rays) -> bool:
    """All finite direction cosines of the bundle are unit vectors -- to 1e-9 in float64, to
    2e-6 in float32 (a correctly normalised float32 vector is off by ~1e-7; the aimers' bundles
    by 1e-3).  One small reduction and one read-back; polarised caller-made bundles only."""
    L, M, N = (t.detach() for t in (rays.L, rays.M, rays.N))
    off = (L * L + M * M + N * N - 1.0).abs()
    if not off.numel():
        return True
    tol = 1e-9 if L.dtype == torch.float64 else 2e-6
    return float(torch.nan_to_num(off, nan=0.0, posinf=0.0).max()) <= tol


def _hip_surface_group_trace(group, rays, skip):
    """SurfaceGroup.trace on the HIP path, or None when the call is not eligible (the
    caller then runs the reference's own loop).  Mirrors surface_group.py:245-257:
    `reset()`, trace surfaces[skip:] in order, every traced Surface records
    x,y,z,L,M,N,intensity,opd (standard_surface.py:260-274), `rays` is mutated in place
    and returned; L0/M0/N0 are the direction cosines before the last interaction
    (interactions/refractive_reflective_model.py:41) and PolarizedRays.p is advanced
    from its CURRENT value (rays/polarized_rays.py:180-202).

    Surfaces the fused path does not implement (thin lens, grating, phase, Forbes / NURBS
    / grid-sag geometry, GRIN media, BSDF, thin-film coating ...) do not disqualify the
    whole system: they are traced by their own `Surface.trace(rays)` on the same device
    tensors, and the runs of supported surfaces between them are one launch each
    (`ol_trace`'s [first, last] range; SURVEY.md section 8b "split the system into supported
    segments").  A bundle a thin lens left un-normalised (`is_normalized == False`) takes
    one more reference surface, whose propagation renormalises it
    (propagation/homogeneous.py:55-56)."""
    import optiland.backend as be
    from optiland.rays import PolarizedRays as RefPolarizedRays
    from optiland.rays import RealRays as RefRealRays

    force = _sg_force(group)
    if type(rays) not in (RefRealRays, RefPolarizedRays) or not _sg_backend_ok(be, force):
        return None
    n_s = len(group.surfaces)
    skip = int(skip)
    if not (0 <= skip < n_s) or _sg_planes(rays, force) is None:
        return None
    w = getattr(rays, "w", None)
    if not isinstance(w, torch.Tensor) or w.numel() == 0:
        return None
    lo, hi = (float(v) for v in torch.stack(torch.aminmax(w.detach())).tolist())  # one sync
    if not (lo == hi and lo > 0.0):
        return None  # per-ray wavelengths: per-ray n(w), not a launch constant
    polarized = type(rays) is RefPolarizedRays
    table = _sg_table(group, lo)
    if table is None:
        return None
    foreign = set(table.unsupported)
    if len(foreign) >= n_s - skip - (1 if skip == 0 else 0):
        return None  # nothing but the object row would run fused
    if table.uses_polarization and not polarized:
        return None  # RealRays.update() ignores Jones matrices; keep that on the reference
    nonunit = polarized and not _unit_directions(rays)
    if nonunit:
        # The reference's iterative / robust aimers hand out direction cosines that are not unit
        # vectors (|k|^2 - 1 ~ 1e-3) and nothing renormalises them.  Its PRT algebra takes k as
        # it comes -- O_in = (s, k0 x s, k0), O_out = (s, k1 x s, k1) stop being orthonormal
        # and the matrix picks up factors |k0| |k1| (polarized_rays.py:136-202).  Round 5 left
        # such bundles to the reference's surface loop (the rank-2 kernel form is that matrix
        # only for |k| = 1: 3e-3 of the PRT); round 6: the kernels have the exact form
        # (OL_TRACE_NONUNIT_K, surface_math.h: interact).  One kind of surface stays with the
        # reference: an uncoated refracting surface between EQUAL indices (the image surface of
        # every lens) -- there k1 = k0, the reference's s is the rounding noise of k0 x k1 (or
        # its fallback axes when that noise is exactly zero) and the product
        # s s^T + |k0|^2 (I - s s^T) depends on it at the 1e-3 level.
        rows, opt = table.surfaces, table.optics[:, 0]
        same = (opt["n1"] == opt["n2"]) & (rows["coating_kind"] == _S.COAT_NONE) \
            & (rows["interaction"] == _S.INTERACT_REFRACT)
        foreign = foreign | {int(i) for i in np.nonzero(same)[0] if i >= max(skip, 1)}
        if len(foreign) >= n_s - skip - (1 if skip == 0 else 0):
            return None
    eng, table = _sg_engine(group, table, rays.x.device)

    # SurfaceGroup.reset() (surface_group.py:373-380): every recorded attribute of every surface
    # an empty array.  The reference's own reset makes 11 fresh tensors per surface (143 tiny
    # device allocations for a double Gauss: 0.4 of the 0.5 ms this seam took on a small
    # bundle); one shared empty tensor says the same.
    empty = _empty(rays.x.dtype, rays.x.device)
    base_reset = _ORIGINALS.get("surface_reset")
    if base_reset is None:
        from optiland.surfaces.standard_surface import Surface
        base_reset = _ORIGINALS["surface_reset"] = Surface.reset
    for surf in group.surfaces:
        if _PENDING:   # what an earlier trace left pending is overwritten
            _PENDING.pop(surf, None)
        if type(surf).reset is not base_reset:
            surf.reset()   # a subclass with its own reset(): it is the one that knows
            continue
        d = surf.__dict__
        d["x"] = d["y"] = d["z"] = d["L"] = d["M"] = d["N"] = d["intensity"] = d["opd"] = empty
        d["u"] = d["aoi"] = empty
    s = skip
    while s < n_s:
        planes = _sg_planes(rays, force) if getattr(rays, "is_normalized", True) else None
        if s in foreign or planes is None:
            group.surfaces[s].trace(rays)  # the reference's own surface, same tensors
            _SG["foreign"] += 1
            s += 1
            continue
        e = s
        while e + 1 < n_s and (e + 1) not in foreign:
            e += 1
        _sg_run(group, eng, rays, planes, s, e, polarized, nonunit)
        s = e + 1
    return rays


def _sg_trace(self, rays, skip=0):
    """Replacement for SurfaceGroup.trace (class-wide under enable(), on one group under
    install())."""
    out = _hip_surface_group_trace(self, rays, skip)
    if out is None:
        _SG["fallbacks"] += 1
        return _ORIGINALS["sg_trace"](self, rays, skip)
    _SG["count"] += 1
    return out


def _companion(rt):
    """The `OptilandHipRayTracer` behind a reference tracer object: the object itself when
    `install()` put one there, else (under `enable()`) its lazily created companion."""
    cls = _make_tracer_class()
    if isinstance(rt, cls):
        return rt
    comp = rt.__dict__.get("_hip_companion")
    if comp is None:
        comp = cls(rt.optic, device=_ENABLE["device"], force=_ENABLE["force"])
        comp.ray_generator = rt.ray_generator
        rt.__dict__["_hip_companion"] = comp
    comp._hip_device, comp._hip_force = _ENABLE["device"], _ENABLE["force"]
    comp._hip_lazy = _ENABLE["lazy"]
    comp.ray_aiming_config = rt.ray_aiming_config
    return comp


class _TrustedScope(set):
    """The wavelengths validated inside an `unchanged` scope, and the token of the optic taken
    for the first of them: tokens of ONE optic differ between wavelengths in their first
    element only (`fingerprint.optic_token`), so the second and third wavelength of a spot
    diagram re-use the walk over the surfaces (3 token walks become 1)."""

    token = None   # (tok, keep) of the first wavelength validated in this scope


@contextlib.contextmanager
def unchanged(optic):
    """Scope in which `optic` is known not to be edited -- a loop of the reference that only
    traces (`SpotDiagram._generate_data`: fields x wavelengths).  The change detector validates
    each wavelength's table once inside the scope and its verdict stands until the scope ends
    (9 token walks of a 3 x 3 spot diagram become 3).  No effect on optics the drop-in does not
    serve."""
    comp = hip_tracer_of(optic)
    if comp is None:
        yield
        return
    if comp._hip_trust_depth == 0:
        comp._hip_trusted = _TrustedScope()
    comp._hip_trust_depth += 1
    try:
        yield
    finally:
        comp._hip_trust_depth -= 1
        if comp._hip_trust_depth == 0:
            comp._hip_trusted = None


def hip_tracer_of(optic):
    """The drop-in tracer serving `optic`, or None when neither `install(optic)` nor
    `enable()` is in effect for it."""
    from optiland.raytrace.real_ray_tracer import RealRayTracer

    rt = getattr(optic, "ray_tracer", None)
    if rt is None:
        return None
    cls = _make_tracer_class()
    if isinstance(rt, cls):
        return rt
    if getattr(RealRayTracer, "_hip_enabled", False) and isinstance(rt, RealRayTracer):
        return _companion(rt)
    return None


def _set_record_pool(placed_records):
    """`placed_records` of enable() / install(): None = leave as is (the default policy is
    "auto", seeded by the environment variable OPTILAND_HIP_PLACED_RECORDS = auto | 0 | n);
    "auto"; False / 0 = off; True = 2 blocks per shape from the first request on; n = n."""
    from .engine import HipSystem, _default_slots

    if placed_records is None:
        if os.environ.get("OPTILAND_HIP_PLACED_RECORDS") is None:
            return
        placed_records = _default_slots()
    if placed_records == "auto":
        HipSystem.enable_record_pool("auto")
    else:
        HipSystem.enable_record_pool(2 if placed_records is True else int(placed_records))


def _set_reference_root(reference_root):
    """`reference_root` of enable() / install(): None = leave as is (OPTILAND_HIP_REFERENCE_ROOT=1
    seeds it).  The option is part of every change-detector token and of the packer's cache
    keys: tables packed before a change are re-packed at the next trace."""
    if reference_root is not None:
        from . import system as _S

        _S.OPTIONS["reference_root"] = bool(reference_root)


def _set_reference_newton(reference_newton):
    """`reference_newton` of enable() / install(): None = leave as is
    (OPTILAND_HIP_REFERENCE_NEWTON=1 seeds it); part of the tokens and cache keys like
    `reference_root`."""
    if reference_newton is not None:
        from . import system as _S

        _S.OPTIONS["reference_newton"] = bool(reference_newton)


def enable(device=None, force=False, analyses=True, lazy_records=False, placed_records=None,
           reference_root=None, reference_newton=None):
    """Route EVERY `Optic` (existing and future) through the HIP path.

    Patches `RealRayTracer.trace / trace_generic` (raytrace/real_ray_tracer.py:58-154)
    in place; each tracer instance lazily gets an `OptilandHipRayTracer` companion.
    Ineligible calls (numpy backend, autograd, unsupported systems) run the original
    methods.  `disable()` restores them.  This is the activation to prefer with the
    stock `"torch"` backend: about thirty sites of the reference branch on
    `be.get_backend() == "torch"` (Forbes, NURBS, optimisers ...), which a backend
    registered under another name does not satisfy.

    `analyses` (default on): also put the FUSED kernels behind the reference's own analysis
    classes (`analysis_seams.enable()`): `SpotDiagram` / `EncircledEnergy` get their
    image-plane hits from `ol_trace_spot`, the chief-ray wavefront strategy its OPD map from
    `ol_trace_opd`, `ScalarFFTPSF` its pupil function from `ol_pupil_fill` -- no record
    block, no ray planes.  The one observable difference: after such an analysis the
    `Surface` objects hold that trace lazily (below).

    `lazy_records` (default off): `Optic.trace` / `trace_generic` calls for ONE field point run
    record-last -- the returned rays and nothing else touch HBM -- and the per-surface arrays
    (`Surface.x ... .opd`, `SurfaceGroup.x ...`) are produced by re-running the trace
    record-all on their first read.  A consumer that only uses the returned rays pays for
    24 planes instead of 8 (S + 2); one that does read the surfaces pays one extra
    record-last launch.  Results are identical either way.

    `placed_records` (default "auto"; True = 2, a count, or 0 = off): record blocks of 256 MB
    and more come from a pool of PLACED windows (`engine.RecordPool`: where this part writes the
    record-all pattern 7.1 instead of 5.8 TB/s; a 1e7-ray double-Gauss trace 0.60 instead of
    0.73 ms) and go back to it when the caller's last view of the block dies -- as many blocks
    per shape may be alive at a time, further ones are ordinary allocations.  Costs the arenas
    behind the windows (~40 GiB per shape on the boxes measured, the two most recently used
    shapes kept) and ~0.3 s of probing.  "auto" builds the pool at the SECOND trace of a shape
    (a loop, not a one-off) and only while at least half of the device memory is free.

    `reference_root` (default off): conic surfaces are intersected in the reference's OWN form,
    `(-b +- sqrt(d)) / 2a` with R-scaled coefficients (geometries/standard.py:112-146), instead
    of the cancellation-free form -- for users who need the reference's NUMBERS where its
    formula is ill conditioned (a nearly parabolic mirror, `|1 + k| << 1`): there the reference
    is off by up to `eps |b| / |2a|` in the intersection distance, systematically, and its own
    golden `tests/test_operand.py::test_opd_diff_on_axis` (Hubble, on axis) encodes that: it
    passes through the drop-in with this option and misses by 4e-7 waves without it.

    `reference_newton` (default off): Newton-Raphson surfaces (aspheres, polynomials, Zernike,
    Chebyshev, biconic, toroidal) stop by the reference's OWN rule -- the whole batch of a trace
    call iterates in lockstep and stops when `max_j |f_j| < tol`, `max_iter` updates when any ray
    of the batch is NaN, and the normal is taken at the end point
    (geometries/newton_raphson.py:137-166) -- instead of per ray.  The default converges
    FURTHER than the reference (its rays are within 1e-7 of the reference's with the factory
    tolerance); this option gives the reference's numbers, to rounding, also for a user-set
    loose `tol` / small `max_iter` and in what OPD / PSF consumers make of 1e-7.  Costs one
    counting launch per Newton surface and one verifying launch per trace call, and the fused
    analysis kernels stand back for such optics (the reference's own analysis code then runs on
    top of the drop-in's `Optic.trace`).
    """
    _set_record_pool(placed_records)
    _set_reference_root(reference_root)
    _set_reference_newton(reference_newton)
    cls = _make_tracer_class()
    from optiland.raytrace.real_ray_tracer import RealRayTracer

    from . import analysis_seams

    if getattr(RealRayTracer, "_hip_enabled", False):
        # already patched: a second call only updates the settings (device / force) that
        # future companions and the SurfaceGroup seam read
        _ENABLE.update(device=device, force=force, lazy=bool(lazy_records))
        _SG.update(device=device, force=force)
        (analysis_seams.enable if analyses else analysis_seams.disable)()
        return
    _ENABLE.update(device=device, force=force, lazy=bool(lazy_records))
    if analyses:
        analysis_seams.enable()

    def trace(self, Hx, Hy, wavelength, num_rays=100, distribution="hexapolar"):
        if isinstance(self, cls):
            return cls.trace(self, Hx, Hy, wavelength, num_rays, distribution)
        return _companion(self).trace(Hx, Hy, wavelength, num_rays, distribution)

    def trace_generic(self, Hx, Hy, Px, Py, wavelength):
        if isinstance(self, cls):
            return cls.trace_generic(self, Hx, Hy, Px, Py, wavelength)
        return _companion(self).trace_generic(Hx, Hy, Px, Py, wavelength)

    RealRayTracer.trace = trace
    RealRayTracer.trace_generic = trace_generic
    RealRayTracer._hip_enabled = True

    from optiland.surfaces.surface_group import SurfaceGroup

    _ORIGINALS.setdefault("sg_trace", SurfaceGroup.trace)
    _SG.update(device=device, force=force)

    SurfaceGroup.trace = _sg_trace


def disable():
    from optiland.raytrace.real_ray_tracer import RealRayTracer

    if getattr(RealRayTracer, "_hip_enabled", False):
        RealRayTracer.trace = _ORIGINALS["trace"]
        RealRayTracer.trace_generic = _ORIGINALS["trace_generic"]
        RealRayTracer._hip_enabled = False
        from optiland.surfaces.surface_group import SurfaceGroup

        SurfaceGroup.trace = _ORIGINALS["sg_trace"]
        _SG.update(device=None, force=False)
        from . import analysis_seams

        analysis_seams.disable()
        _ENABLE.update(lazy=False)
        _remove_lazy_descriptors()
        _remove_lazy_prt()
        from .engine import HipSystem

        HipSystem.reset_record_pool()


def install(optic, device=None, force=False, analyses=True, lazy_records=False,
            placed_records=None, reference_root=None, reference_newton=None):
    """Replace `optic.ray_tracer` with the HIP tracer (keeps the aiming config).  `analyses`,
    `lazy_records`: see `enable()` -- the class-wide analysis seams only act on optics the
    drop-in serves."""
    _set_record_pool(placed_records)
    _set_reference_root(reference_root)
    _set_reference_newton(reference_newton)
    cls = _make_tracer_class()
    if analyses:
        from . import analysis_seams

        analysis_seams.enable()
    old = optic.ray_tracer
    new = cls(optic, device=device, force=force)
    new._hip_lazy = bool(lazy_records)
    new.ray_aiming_config = dict(getattr(old, "ray_aiming_config", new.ray_aiming_config))
    optic.ray_tracer = new
    # this optic's SurfaceGroup too: caller-built rays, and the surface loop of traces whose
    # ray generation stays on the reference (aiming modes, unsupported surfaces bridged)
    from optiland.surfaces.surface_group import SurfaceGroup

    _ORIGINALS.setdefault("sg_trace", SurfaceGroup.trace)
    group = optic.surfaces
    group.__dict__["_hip_force"], group.__dict__["_hip_device"] = bool(force), device
    group.__dict__["trace"] = _BoundSgTrace(group)
    return new


def uninstall(optic):
    from optiland.raytrace.real_ray_tracer import RealRayTracer

    cfg = dict(optic.ray_tracer.ray_aiming_config)
    optic.ray_tracer = RealRayTracer(optic)
    optic.ray_tracer.ray_aiming_config = cfg
    for k in ("trace", "_hip_force", "_hip_device", "_hip_sg_memo", "_hip_engines"):
        optic.surfaces.__dict__.pop(k, None)
