"""Host-side mirror of the reference's real-ray tracer for the fused HIP path.

`HipRayTracer` exposes the interface of `RealRayTracer`
(optiland/raytrace/real_ray_tracer.py:36-173): `trace(Hx, Hy, wavelength,
num_rays, distribution)`, `trace_generic(Hx, Hy, Px, Py, wavelength)`,
`set_aiming(...)`, `ray_aiming_config` -- same argument meaning, same
`ValueError` texts -- but runs ray generation, the whole surface sequence and
the recording in HIP kernels behind the C ABI.  The per-surface recorded state
(`SurfaceGroup.x/.y/.../.intensity/.opd`, surfaces/surface_group.py:108-153) is
exposed without copies as views of one (S+1, 8, N) device block.

It works from a `SystemTable` alone (the GPU box has no reference package) or
from a live reference `Optic` through `optiland_amd.integration`.
"""

from __future__ import annotations

import math
import os
from collections import OrderedDict

import numpy as np
import torch

from . import _capi
from .distribution import create_distribution
from .rays import PolarizedRays, RealRays, _state_dict
from .system import SystemTable


def _make_engine(table: SystemTable, device):
    """Engine factory (tests substitute an oracle-backed stand-in here; the
    product path has no CPU fallback)."""
    from .engine import HipSystem

    return HipSystem(table, device)


class RecordedSurfaces:
    """Stacked per-surface state of the last trace: `(S+1, N)` views (no copy).

    Mirrors the read side of `SurfaceGroup` (surfaces/surface_group.py:108-153).
    """

    _PLANES = {"x": 0, "y": 1, "z": 2, "L": 3, "M": 4, "N": 5, "intensity": 6, "opd": 7}

    def __init__(self):
        self._res = None

    def _bind(self, res):
        self._res = res

    def __getattr__(self, name):
        planes = object.__getattribute__(self, "_PLANES")
        if name in planes:
            res = object.__getattribute__(self, "_res")
            if res is None or res.record is None:
                return torch.empty((0, 0))
            return res.stack(planes[name])
        raise AttributeError(name)

    @property
    def num_surfaces(self):
        return 0 if self._res is None else self._res.last - self._res.first + 1


# Device planes of the deterministic pupil samplers, shared by every tracer of the process:
# a spot diagram over three wavelengths runs on three tracers (one device table each), and
# 400 hexapolar rings are 26 ms of host sampling + 8 MB of upload per tracer otherwise.  The
# planes are kernel INPUTS only (never written).  Least recently used out beyond 256 MB.
_PUPIL_PLANES: "OrderedDict" = OrderedDict()
_PUPIL_PLANES_CAP = 256 << 20


def _shared_pupil_planes(key, dtype, device, to_device, engine=None):
    name = key[0]
    if name in ("random", "sobol"):
        d = create_distribution(name)
        d.generate_points(key[1])
        return to_device(d.x), to_device(d.y)
    full = (key, dtype, str(device))
    hit = _PUPIL_PLANES.get(full)
    if hit is None:
        can = getattr(engine, "can_pupil_points", None)
        if name in ("hexapolar", "uniform") and can is not None and can() \
                and (name == "hexapolar" or int(key[1]) >= 2):
            # sampled ON the device from the point index (`ol_pupil_points`): no host pass
            # over the points, no upload
            hit = engine.pupil_points(name, key[1], dtype)
        else:
            d = create_distribution(name)
            d.generate_points(key[1])
            hit = (to_device(d.x), to_device(d.y))
        _PUPIL_PLANES[full] = hit
        total = sum(2 * h[0].numel() * h[0].element_size() for h in _PUPIL_PLANES.values())
        while total > _PUPIL_PLANES_CAP and len(_PUPIL_PLANES) > 1:
            _, old = _PUPIL_PLANES.popitem(last=False)
            total -= 2 * old[0].numel() * old[0].element_size()
    else:
        _PUPIL_PLANES.move_to_end(full)
    return hit


def _broadcast_message(a: int, b: int) -> str:
    """NumPy's text for the mismatch the reference runs into when Hx, Hy, Px, Py are arrays of
    different lengths (real_ray_tracer.py:175-194 pads scalars only; the first array operation
    of the generator then fails to broadcast)."""
    return f"operands could not be broadcast together with shapes ({int(a)},) ({int(b)},) "


def _size_mismatch_message(nhx, nhy, npx, npy):
    """The text of the ValueError the reference's NumPy backend ends in when the coordinate
    arrays of `trace_generic` differ in length, or None when they agree (sizes; None = a scalar
    or a one-element array, which broadcasts).  Which operation fails first decides the text:
    `FieldGroup.get_vig_factor` broadcasts Hx against Hy (fields/field_group.py:93-122);
    `Px * (1 - vx)` / `Py * (1 - vy)` meet the field length next (real_ray_tracer.py:134-137);
    without field arrays the scalars were padded to the LONGER pupil array
    (real_ray_tracer.py:175-194) and the shorter one fails against it in
    `AngleField.get_ray_origins` (fields/field_types/angle.py:40-47)."""
    if nhx is not None and nhy is not None and nhx != nhy:
        return ("shape mismatch: objects cannot be broadcast to a single shape.  Mismatch is "
                f"between arg 0 with shape ({int(nhx)},) and arg 1 with shape ({int(nhy)},).")
    f = nhx if nhx is not None else nhy
    if f is not None:
        for p in (npx, npy):
            if p is not None and p != f:
                return _broadcast_message(p, f)
        return None
    if npx is not None and npy is not None and npx != npy:
        return _broadcast_message(min(npx, npy), max(npx, npy))
    return None


def coordinates_unlike_the_reference(*coords) -> bool:
    """True when `trace_generic` of the reference does NOT accept these coordinate arguments
    (real_ray_tracer.py:120-194): a Python list / tuple dies in `x >= -1` of
    `_validate_normalized_coordinates` (:168-169, TypeError); an array of two or more dimensions
    passes `_validate_array_size` untouched (:186-194) and fails to broadcast against the
    flattened planes of the ray generator (ValueError) -- unless it is a single row."""
    for v in coords:
        if isinstance(v, (list, tuple)):
            return True
        shape = getattr(v, "shape", None)
        if shape is not None and len(shape) >= 2:
            return True
    return False


def _reject_unlike_the_reference(*coords):
    """The standalone tracer's form of the above: the exceptions of the reference's NumPy
    backend, same type and text (the live drop-in hands such calls to the reference's own
    method instead, integration.py)."""
    for v in coords:
        if isinstance(v, (list, tuple)):
            raise TypeError(f"'>=' not supported between instances of '{type(v).__name__}' "
                            "and 'int'")
    for v in coords:
        shape = tuple(getattr(v, "shape", ()) or ())
        if len(shape) >= 2 and not (len(shape) == 2 and shape[0] == 1):
            dims = ",".join(str(int(k)) for k in shape)
            raise ValueError(f"operands could not be broadcast together with shapes ({dims}) "
                             f"({int(np.prod(shape))},) ")


def _can_field_planes(can) -> bool:
    try:
        return bool(can(field_planes=True))
    except TypeError:  # an engine stand-in with the ABI-6 signature
        return False


class HipRayTracer:
    """Drop-in for `RealRayTracer` on a packed system."""

    def __init__(self, table: SystemTable, device=None, dtype=torch.float32, engine=None):
        self.dtype = dtype
        self.engine = engine if engine is not None else _make_engine(table, device)
        self.device = self.engine.device
        self.surfaces = RecordedSurfaces()
        self.ray_aiming_config = {"mode": "paraxial", "max_iter": 10, "tol": 1e-6}
        self.record_all = True  # drop-in semantics; False = image plane only
        # one field point per call: ray generation fused into the trace launch
        # (`ol_trace_generate`) when the engine offers it; False = always two launches
        self.fuse_generate = os.environ.get("OPTILAND_HIP_FUSE_GENERATE", "1") != "0"
        # lazy records: a fused single-field launch records its last two surfaces only and
        # remembers its inputs (`last_fused_launch`) so that the caller can re-run it
        # record-all when somebody asks for the interior surfaces (integration.py)
        self.lazy_records = False
        self.last_fused_launch = None
        self.last_was_lazy = False
        self._last_res = None
        # True: launches return without the status read-back (the one sync of a call);
        # the caller does `check_status()` itself -- integration.py overlaps its change
        # check of the live optic with the kernels this way
        self.defer_checks = False
        self.last_status = 0   # the status word of the last `_finish_checks`
        self._pupil_cache = {}  # (distribution name, num_rays) -> device planes
        self.rebind(table)

    def rebind(self, table: SystemTable):
        """Point this tracer at another packed table of the same engine (the engine's device
        tables were just patched in place, `HipSystem.update`): everything that does not depend
        on the prescription -- cached pupil planes, configuration -- stays."""
        self.table = table
        self._uses_polarization = bool(table.uses_polarization)
        self._complex_prt = bool(table.needs_complex_prt)
        if getattr(self, "_fields_of", None) == table.fields:
            return  # (an optimiser's re-pack: the field list is what it was)
        self._fields_of = list(table.fields)
        f = np.asarray(table.fields, dtype=np.float64).reshape(-1, 4)
        self._fields = f
        # facts of the (immutable, packed) table that every launch asks for: taken once --
        # as numpy reductions per call they were ~35 us of a ~230 us small trace
        self._has_vignetting = bool(f.shape[0] != 0 and np.any(f[:, 2:]))

    # ------------------------------------------------------------ configuration
    def set_aiming(self, mode: str, max_iter: int = 10, tol: float = 1e-6, **kwargs):
        """real_ray_tracer.py:42-56.  Only paraxial aiming runs on device."""
        if mode != "paraxial":
            raise NotImplementedError(
                f"ray aiming mode {mode!r} is a host-side Newton loop in the reference "
                "(rays/ray_aiming/iterative.py) and is outside the fused path")
        self.ray_aiming_config = {"mode": mode, "max_iter": max_iter, "tol": tol, **kwargs}

    # ------------------------------------------------------------------ helpers
    def _dev(self, v):
        if isinstance(v, torch.Tensor):
            return v.to(device=self.device, dtype=self.dtype).reshape(-1)
        return torch.as_tensor(np.atleast_1d(np.asarray(v, dtype=np.float64)),
                               dtype=self.dtype, device=self.device).reshape(-1)

    @staticmethod
    def _as_scalar(v):
        """float(v) when `v` is a host scalar / one-element host array, else None."""
        if isinstance(v, torch.Tensor):
            return None if (v.is_cuda or v.numel() != 1) else float(v)
        if isinstance(v, (int, float, np.floating, np.integer)):
            return float(v)
        a = np.asarray(v)
        return float(a.reshape(-1)[0]) if a.size == 1 else None

    def _validate_normalized_coordinates(self, x, y, coord_type="field"):
        """real_ray_tracer.py:156-173 (same message) for values that live on the HOST.
        Device planes are validated inside the ray-generation kernel, which reads them
        anyway (`OL_RAYGEN_CHECK_*` -> status bits, surfaced by `_finish_checks`)."""
        for v in (x, y):
            if isinstance(v, torch.Tensor) and v.is_cuda:
                continue
            s = self._as_scalar(v)
            if s is not None:
                ok = -1.0 <= s <= 1.0
            else:
                a = v.detach().numpy() if isinstance(v, torch.Tensor) else \
                    np.asarray(v, dtype=np.float64)
                ok = bool(np.all((a >= -1) & (a <= 1)))
            if not ok:
                raise ValueError(f"Normalized {coord_type} coordinates must be within (-1, 1)")

    def _finish_checks(self, eng):
        """One read-back of the device status word (range checks of the ray generator +
        the trace kernel's own bits)."""
        status_t = getattr(eng, "_status", None)
        if status_t is None:  # engines that raise eagerly (tests' oracle stand-in)
            self.last_status = 0
            return
        # (kept: the informational bits -- STATUS_NAN_DIRECTION -- are the caller's to use)
        self.last_status = int(status_t.item())
        eng.raise_for_status(self.last_status)

    def _vig_factor(self, hx, hy):
        """FieldGroup.get_vig_factor (fields/field_group.py:93-122): nearest field
        point in normalised field coordinates -> (vx, vy), per ray, on device."""
        f = self._fields
        if f.shape[0] == 0 or not np.any(f[:, 2:]):
            return None, None
        max_field = self.table.raygen.get("max_field", 0.0)
        pts = f[:, :2] / max_field if max_field != 0 else f[:, :2]
        P = torch.as_tensor(pts, dtype=self.dtype, device=self.device)
        d2 = (hx[:, None] - P[None, :, 0]) ** 2 + (hy[:, None] - P[None, :, 1]) ** 2
        idx = torch.argmin(d2, dim=1)
        V = torch.as_tensor(f[:, 2:], dtype=self.dtype, device=self.device)
        return V[idx, 0], V[idx, 1]

    def _vig_scalar(self, hx: float, hy: float):
        """`_vig_factor` for one field point, on the host: (1 - vx, 1 - vy)."""
        f = self._fields
        if not self._has_vignetting:
            return 1.0, 1.0
        max_field = self.table.raygen.get("max_field", 0.0)
        pts = f[:, :2] / max_field if max_field != 0 else f[:, :2]
        idx = int(np.argmin((hx - pts[:, 0]) ** 2 + (hy - pts[:, 1]) ** 2))
        return 1.0 - float(f[idx, 2]), 1.0 - float(f[idx, 3])

    def _wavelength_index(self, wavelength):
        w = float(wavelength.item()) if hasattr(wavelength, "item") else float(wavelength)
        return self.table.wavelength_index(w), w

    def _pupil_planes(self, distribution, num_rays):
        """Device planes of a pupil distribution; named distributions are cached per
        (name, num_rays) so that repeated traces do not re-sample / re-upload them."""
        if isinstance(distribution, str):
            if distribution == "random" and self.device.type == "cuda":
                # distribution.py:132-158 with the draws made on the device, as the reference's
                # torch backend makes them (a fresh sample per call; 1e6 points took 31 ms
                # as host arrays + upload, the trace they feed 0.1 ms)
                u = torch.rand((2, int(num_rays)), dtype=self.dtype, device=self.device)
                r, th = u[0].sqrt_(), u[1].mul_(2.0 * math.pi)
                return r * th.cos(), r * th.sin()
            key = (distribution, int(num_rays) if num_rays is not None else None)
            hit = self._pupil_cache.get(key)
            if hit is None:
                hit = _shared_pupil_planes(key, self.dtype, self.device, self._dev, self.engine)
                if distribution not in ("random", "sobol"):  # those: a fresh sample per call
                    self._pupil_cache[key] = hit
            return hit
        return self._dev(distribution.x), self._dev(distribution.y)

    # -------------------------------------------------------------------- trace
    def _alloc_state(self, n):
        """(record, ray planes): in record-all mode the rays live in row 0 of the record
        block -- the object surface only records its input, so the trace need not copy
        that row (zero-copy object row)."""
        eng = self.engine
        if self.record_all:
            record = eng.alloc_record(n, self.dtype)
            return record, eng.row0_planes(record, n)
        buf = torch.empty((8, max(n, 1)), dtype=self.dtype, device=self.device)
        return False, [buf[k, :n] for k in range(8)]

    def _run(self, hx, hy, px, py, vig, wavelength, update_intensity, flags):
        """hx, hy: floats (launch-uniform field) or device planes; px, py: device planes;
        vig: (1 - vx, 1 - vy) as floats or planes; flags: OL_RAYGEN_*."""
        n = int(px.numel())
        eng = self.engine
        self.last_fused_launch = None
        self.last_was_lazy = False
        uniform = isinstance(hx, float) and isinstance(hy, float) \
            and isinstance(vig[0], float) and isinstance(vig[1], float)
        # per-ray field planes (trace_generic with arrays, the fields x pupil expansion of a
        # multi-field trace): one launch as well when the engine serves them (ABI 8)
        per_ray = isinstance(hx, torch.Tensor) and isinstance(hy, torch.Tensor) \
            and (vig[0] is None or isinstance(vig[0], torch.Tensor))
        can = getattr(eng, "can_trace_generate", None)
        if self.fuse_generate and n > 0 and can is not None \
                and ((uniform and can()) or (per_ray and _can_field_planes(can))):
            return self._run_fused(hx, hy, px, py, vig, wavelength, update_intensity, flags)
        record, rays = self._alloc_state(n)
        eng.generate_rays(hx, hy, px, py, vig[0], vig[1], out=rays, flags=flags)
        # the generator zeroes the status word only when it also writes range bits into it
        checked = bool(flags & (_capi.RAYGEN_CHECK_FIELD | _capi.RAYGEN_CHECK_PUPIL))
        return self._launch(rays, record, wavelength, update_intensity, zero_status=not checked)

    def _run_fused(self, hx, hy, px, py, vig, wavelength, update_intensity, flags):
        """One field point: generate + trace + record in ONE launch (`ol_trace_generate`).
        `record_all`: every surface, row 0 = the generated rays (which are also the initial
        direction cosines / intensity a polarised bundle keeps -- no copies).  Otherwise the
        last TWO surfaces only: the image plane (the returned rays) and the row before it
        (the pre-interaction cosines L0, M0, N0)."""
        wl, w = self._wavelength_index(wavelength)
        eng = self.engine
        n = int(px.numel())
        self.last_fused_launch = (hx, hy, px, py, vig, wavelength, flags)
        polarized = self.table.polarization is not None
        # (a polarised bundle records from row 0 anyway -- its initial cosines and intensity
        # live there -- so there is nothing to be lazy about)
        lazy = self.record_all and self.lazy_records and not polarized
        full = self.record_all and not lazy
        self.last_was_lazy = lazy
        if not polarized and self._uses_polarization:
            # rays/ray_generator.py:89-94
            raise ValueError("Polarization must be set when surfaces have "
                             "polarization-dependent coatings.")
        last = eng.num_surfaces - 1
        # a polarised bundle keeps its initial cosines and intensity (polarized_rays.py:50-54):
        # they are row 0, so row 0 is recorded whenever the trace is polarised
        first_row = 0 if (full or polarized) else max(last - 1, 0)
        prt = None
        if polarized:
            prt = torch.empty((18 if self._complex_prt else 9, n), dtype=self.dtype,
                              device=self.device)  # written by the kernel (starts from I)
        # polarised Optic.trace: update_intensity as an epilogue of the same launch
        fuse = None
        if polarized and update_intensity and getattr(eng, "can_fuse_update_intensity",
                                                      lambda: False)():
            fuse = _state_dict(self.table.polarization)
        kw = {} if fuse is None else {"update_intensity": fuse}
        res = eng.trace_generate(px, py, wl, field=(hx, hy), vig=vig, record=True,
                                 record_first=first_row, prt=prt, flags=flags,
                                 defer_status=True, **kw)
        if not self.defer_checks:
            self._finish_checks(eng)
        k_init = i0 = None
        if polarized:
            r0 = res.rows(0)
            k_init, i0 = (r0[3], r0[4], r0[5]), r0[6]
        return self._wrap(res, res.rows(res.last), w, prt, k_init, i0, update_intensity,
                          bind=full)

    def trace_rays(self, planes, wavelength, update_intensity=False):
        """Trace rays the CALLER generated (x, y, z, L, M, N, i [, opd] arrays of one
        length; the reference's own RayGenerator for aiming modes / field types the
        device generator does not cover)."""
        src = [self._dev(p) for p in planes]
        n = int(src[0].numel())
        if self.table.polarization is not None and n:
            # the PRT update of the kernels is the reference's only for unit direction cosines
            # (surface_math.h: prt_apply_diag): fail loudly instead of being 1e-3 off
            off = (src[3] * src[3] + src[4] * src[4] + src[5] * src[5] - 1.0).abs()
            if float(torch.nan_to_num(off, nan=0.0, posinf=0.0).max()) > 1e-6:
                raise ValueError("trace_rays: a polarised trace needs unit direction cosines "
                                 "(|L^2 + M^2 + N^2 - 1| <= 1e-6)")
        record, rays = self._alloc_state(n)
        for dst, s_ in zip(rays, src):
            dst.copy_(s_)
        if len(src) < 8:
            rays[7].zero_()
        return self._launch(rays, record, wavelength, update_intensity, zero_status=True)

    def _launch(self, rays, record, wavelength, update_intensity, zero_status):
        wl, w = self._wavelength_index(wavelength)
        eng = self.engine
        n = int(rays[0].numel())
        polarized = self.table.polarization is not None
        if not polarized and self._uses_polarization:
            # rays/ray_generator.py:89-94
            raise ValueError("Polarization must be set when surfaces have "
                             "polarization-dependent coatings.")
        prt = None
        k_init = i0 = None
        if polarized:
            prt = torch.empty((18 if self._complex_prt else 9, n), dtype=self.dtype,
                              device=self.device)  # written by the kernel (starts from I)
            if isinstance(record, torch.Tensor):
                # record-all: the rays ARE row 0 of the record block, which the trace never
                # rewrites -- the initial cosines / intensity need no copies
                k_init, i0 = (rays[3], rays[4], rays[5]), rays[6]
            else:  # the trace writes the final state back into these planes
                k_init = (rays[3].clone(), rays[4].clone(), rays[5].clone())
                i0 = rays[6].clone()
        deferred = hasattr(eng, "_status")
        kw = {"defer_status": True, "zero_status": zero_status} if deferred else {}
        res = eng.trace(rays, wl, record=record, prt=prt, prt_identity=prt is not None, **kw)
        if not self.defer_checks:
            self._finish_checks(eng)
        fin = res.rows(res.last) if res.record is not None else rays
        return self._wrap(res, fin, w, prt, k_init, i0, update_intensity)

    def _wrap(self, res, fin, w, prt, k_init, i0, update_intensity, bind=None):
        """The result objects of one launch: bound recorded surfaces + the returned rays."""
        eng = self.engine
        # record-last / lazy launches bind no surfaces; the rows they did record serve
        # L0 / M0 / N0 below
        if bind is None:
            bind = self.record_all
        self.surfaces._bind(res if bind else None)
        self._last_res = res
        if prt is not None:
            out = PolarizedRays(*fin[:7], w, fin[7], engine=eng, prt=prt, i0=i0, k_init=k_init)
            # real_ray_tracer.py:112-113 -- trace() only.  `update_intensity == "defer"`: the
            # caller places the epilogue itself (integration._finish, after the final
            # propagation) and takes `_i_updated` when the launch already produced it
            fused = getattr(res, "updated_intensity", None)
            out._i_updated = fused
            if update_intensity is True:
                if fused is not None:
                    out.i = fused
                else:
                    out.update_intensity(_state_dict(self.table.polarization))
        else:
            out = RealRays(*fin[:7], w, fin[7])
        # pre-interaction cosines at the last surface = directions recorded on the
        # previous one, expressed in the last surface's frame (real_rays.py:170-172)
        if res.record is not None and res.last > res.first:
            L0, M0, N0 = res.rows(res.last - 1)[3:6]
            s = self.table.surfaces[res.last]
            if s["flags"] & 1:
                R = torch.as_tensor(np.asarray(s["rot"]).reshape(3, 3), dtype=self.dtype,
                                    device=self.device)
                k = torch.stack([L0, M0, N0])
                L0, M0, N0 = R @ k
            out.L0, out.M0, out.N0 = L0, M0, N0
        return out

    def trace(self, Hx, Hy, wavelength, num_rays=100, distribution="hexapolar",
              update_intensity: bool = True):
        """real_ray_tracer.py:58-118: every field point x every pupil point.
        (`update_intensity=False`: the caller applies the polarised epilogue itself.)"""
        self._validate_normalized_coordinates(Hx, Hy, "field")
        Px, Py = self._pupil_planes(distribution, num_rays)
        sx, sy = self._as_scalar(Hx), self._as_scalar(Hy)
        if sx is not None and sy is not None:  # one field point: launch-uniform scalars
            return self._run(sx, sy, Px, Py, self._vig_scalar(sx, sy), wavelength,
                             update_intensity=update_intensity, flags=0)
        Hx, Hy = self._dev(Hx), self._dev(Hy)
        nf, npup = Hx.numel(), Px.numel()
        hx, hy = Hx.repeat_interleave(npup), Hy.repeat_interleave(npup)
        px, py = Px.repeat(nf), Py.repeat(nf)
        vxf, vyf = self._vig_factor(hx, hy)
        vig = (None, None) if vxf is None else (1 - vxf, 1 - vyf)
        return self._run(hx, hy, px, py, vig, wavelength, update_intensity=update_intensity,
                         flags=_capi.RAYGEN_CHECK_FIELD)

    def reset_status(self):
        """Clear the device status word before queueing launches with
        check_status=False (they OR their bits in and never clear it)."""
        st = getattr(self.engine, "_status", None)
        if st is not None:
            st.zero_()

    def check_status(self):
        """Read the device status word back and raise what it holds (for callers that
        queued launches with check_status=False)."""
        self._finish_checks(self.engine)

    def trace_spot(self, Hx: float, Hy: float, wavelength, num_rays=100,
                   distribution="hexapolar", center=(0.0, 0.0), hits: bool = False,
                   check_status: bool = True, recorded_row: bool = False, local: bool = False):
        """`trace(Hx, Hy, ...)` for ONE field point fused with the image-plane
        reduction (`ol_trace_spot`): rays are generated, traced and folded into masked
        moments about `center` in one kernel and never exist in HBM.  Returns
        (moments7, hits) with moments7 = {count, sum dx, sum dy, sum dx^2, sum dy^2,
        sum i, max r^2} (float64 device tensor) and hits = (x, y, intensity) at the last
        surface or None.  Unpolarised systems (the polarised `update_intensity` epilogue needs
        the PRT planes) -- unless `recorded_row`: what a spot diagram reads of a POLARISED trace
        is the recorded last row, positions and the geometric intensity, which no PRT matrix
        enters (`OL_SPOT_POLARIZED_OK`).  `local`: hits and moments in the last surface's own
        frame (`OL_SPOT_HITS_LOCAL`).  The masks are those of
        analysis/spot_diagram/core.py:470-476."""
        Hx, Hy = float(Hx), float(Hy)
        self._validate_normalized_coordinates(Hx, Hy, "field")
        has_state = self.table.polarization is not None
        if has_state and not recorded_row:
            raise ValueError("trace_spot: fused spot reduction needs an unpolarised system")
        # (coatings that need a polarization state on an optic WITHOUT one: the library refuses
        # with the reference's own error, rays/ray_generator.py:89-94)
        flags = (_capi.SPOT_POLARIZED_OK if (recorded_row and has_state) else 0) \
            | (_capi.SPOT_HITS_LOCAL if local else 0)
        px, py = self._pupil_planes(distribution, num_rays)
        self.last_spot_pupil = (px, py)
        wl, _ = self._wavelength_index(wavelength)
        out3 = None
        if hits:
            n = int(px.numel())
            buf = torch.empty((3, max(n, 1)), dtype=self.dtype, device=self.device)
            out3 = [buf[k, :n] for k in range(3)]
        mom = self.engine.trace_spot(px, py, wl, field=(Hx, Hy), vig=self._vig_scalar(Hx, Hy),
                                     center=center, hits=out3, check_status=check_status,
                                     flags=flags)
        return mom, out3

    def trace_generic(self, Hx, Hy, Px, Py, wavelength):
        """real_ray_tracer.py:120-154: caller-supplied per-ray coordinates; the
        pupil is pre-scaled by (1 - v) (:134-137, `OL_RAYGEN_PRESCALE_PUPIL`) and the
        polarised update_intensity epilogue is NOT applied."""
        _reject_unlike_the_reference(Hx, Hy, Px, Py)
        self._validate_normalized_coordinates(Hx, Hy, "field")
        self._validate_normalized_coordinates(Px, Py, "pupil")
        sx, sy = self._as_scalar(Hx), self._as_scalar(Hy)
        px, py = self._dev(Px), self._dev(Py)
        flags = _capi.RAYGEN_CHECK_PUPIL
        sizes = [None if s_ is not None else int(np.size(v) if not isinstance(v, torch.Tensor)
                                                  else v.numel())
                 for s_, v in ((sx, Hx), (sy, Hy))] + [int(px.numel()), int(py.numel())]
        bad = _size_mismatch_message(*[None if k == 1 else k for k in sizes])
        if bad is not None:
            raise ValueError(bad)
        if sx is not None and sy is not None:
            n = max(px.numel(), py.numel())
            hx, hy = sx, sy
            vig = self._vig_scalar(sx, sy)
            vignetted = vig != (1.0, 1.0)
        else:
            hx, hy = self._dev(Hx), self._dev(Hy)
            n = max(a.numel() for a in (hx, hy, px, py))
            hx, hy = (a.expand(n).contiguous() if a.numel() == 1 else a for a in (hx, hy))
            if hx.numel() != n or hy.numel() != n:
                raise ValueError(_broadcast_message(n, min(hx.numel(), hy.numel())))
            vxf, vyf = self._vig_factor(hx, hy)
            vignetted = vxf is not None
            vig = (1 - vxf, 1 - vyf) if vignetted else (None, None)
            flags |= _capi.RAYGEN_CHECK_FIELD
        px, py = (a.expand(n).contiguous() if a.numel() == 1 else a for a in (px, py))
        if px.numel() != n or py.numel() != n:
            raise ValueError(_broadcast_message(n, min(px.numel(), py.numel())))
        if vignetted:
            flags |= _capi.RAYGEN_PRESCALE_PUPIL
        return self._run(hx, hy, px, py, vig, wavelength, update_intensity=False, flags=flags)
