"""The fused kernels behind the reference's OWN analysis classes.

`integration.enable()` / `install()` make `Optic.trace` a launch of the fused trace; the
reference's analyses on top of it would still pay for a record-all trace (8 planes for each
of the S + 1 surfaces) and reduce it with a chain of backend array operations, although
all they read is the image plane.  This module patches the methods that sit between "an
optic" and "the arrays the analysis keeps" so that they call the kernels that were built
for exactly that:

* `SpotDiagram._generate_field_data` (analysis/spot_diagram/core.py:440-481) and
  `EncircledEnergy._generate_field_data` (analysis/encircled_energy.py:170-197)
  -> `ol_trace_spot` with hit planes: generate -> trace -> image-plane x, y, intensity
  (+ the seven masked moments) in ONE kernel, three planes written instead of 8 (S + 1);
* `ChiefRayStrategy.compute_wavefront_data` (wavefront/strategy.py:163-215)
  -> `ol_trace_opd`: pupil points -> OPD in waves, intensity, pupil intersection points;
  its constructor (strategy.py:156-160) takes the exit-pupil position from the packed table
  instead of `optic.paraxial.XPL()` + `surfaces.positions` (6 of the 8 ms of an `OPD(...)`);
* `ScalarFFTPSF._generate_pupils` / `_pad_pupils` (psf/fft.py:123-161, 203-230)
  -> `ol_pupil_fill`: the pupil function scattered straight into the zero-padded FFT grid;
* `CentroidStrategy.compute_wavefront_data` (strategy.py:307-364; `BestFitStrategy` inherits
  it) -> one generating launch recording the image surface + `ol_wavefront_fit` +
  `ol_wavefront_opd_fitted`: the reference's ~100 array operations and half a dozen host
  decisions between them as a chain of device passes with one read-back at the end;
* `HexagonalDistribution.generate_points` (distribution.py:201-220) -> `ol_pupil_points`: the
  pupil grid of an analysis in one launch instead of a Python loop over the rings (round 4:
  10 of the 11 ms of an OPD at 256 rings).

Every patched method first asks whether the call is one the fused path covers -- drop-in
active for this optic, torch backend on the HIP device without autograd, a system the
packer accepts with device-side ray generation, unpolarised, (for the wavefront) fp64 --
and otherwise runs the reference's own method untouched, which then still traces through
the drop-in's `Optic.trace`.  Results are the reference's own dataclasses.

What `Optic.trace()` would have left on the `Surface` objects -- the recorded arrays of the
analysis' last trace -- is registered as a PENDING record (integration.py: lazy records) and
produced by a record-all re-run of that trace if anybody reads it.
"""

from __future__ import annotations

import math

import threading

import numpy as np
import torch

from .packer import UnsupportedSystem

_ORIG: dict = {}
STATS = {"spot": 0, "spot_fallback": 0, "ee": 0, "ee_fallback": 0, "opd": 0, "opd_fallback": 0,
         "pupil": 0, "pupil_fallback": 0, "opd_init": 0, "opd_init_fallback": 0,
         "dist": 0, "dist_fallback": 0, "opd_fit": 0, "opd_fit_fallback": 0, "spot_grid": 0,
         "spot_radius": 0}


def _why(seam, reason):
    """OPTILAND_HIP_SEAM_LOG=<file>: one line per declined seam call (which, why, and -- under
    pytest -- in which test): how `tools/ref_consumers_seams.sh` attributes its fall-backs."""
    import os

    path = os.environ.get("OPTILAND_HIP_SEAM_LOG")
    if path:
        with open(path, "a") as f:
            f.write(f"{seam}: {reason} [{os.environ.get('PYTEST_CURRENT_TEST', '')}]\n")


def _front(optic, wavelength, need_fp64=False, final_propagation=False, recorded_row=False,
           polarised_ok=False):
    """(front, table) -- the stand-alone device tracer on the CURRENT packed table of
    `optic` -- when the fused analysis kernels apply to it, else None."""
    from . import integration as ig

    comp = ig.hip_tracer_of(optic)
    if comp is None or not comp._eligible():
        _why("front", "optic not served by the drop-in (backend / device / autograd)")
        return None
    try:
        front, table = comp._front_for(wavelength)
    except UnsupportedSystem as exc:
        _why("front", f"unsupported system: {exc}")
        return None
    polarised = table.polarization is not None or table.uses_polarization
    if polarised and recorded_row and table.polarization is not None \
            and hasattr(getattr(front.engine, "lib", None), "ol_trace_spot_batch"):
        # the caller reads the RECORDED last row only (spot diagram, encircled energy): positions
        # and the geometric intensity, which no PRT matrix enters -- OL_SPOT_POLARIZED_OK, ABI 10
        polarised = False
    if not table.raygen or (polarised and not polarised_ok):
        _why("front", "no device ray generation" if not table.raygen else "polarised")
        return None  # reference-side ray generation / polarised epilogue: not fused
    if table.reference_newton_surfaces():
        # opt-in reference stop rule: the Newton iteration count is a property of each trace
        # call's batch (engine._newton_counts) -- the reference's own analysis code runs on top
        # of the drop-in's `Optic.trace`, which finds it
        _why("front", "reference-rule Newton surfaces")
        return None
    if float(table.last_thickness) != 0.0 and not final_propagation:
        _why("front", "last surface has a thickness")
        # `Optic.trace` propagates the rays on by the LAST surface's thickness
        # (real_ray_tracer.py:104-110; 0 in every sample: the last surface is the image
        # plane).  `final_propagation`: the caller either does not look at the returned rays
        # (the spot / encircled-energy data are the RECORDED last row) or hands the
        # propagation to the kernel (`_final_propagation`: the OPD seams, ABI 10).  Anybody
        # else keeps the reference's own analysis code on top of the drop-in's trace.
        return None
    if need_fp64 and front.dtype != torch.float64:
        # a float32 backend (round 6): the wavefront kernels compute in fp64 and the seam hands
        # the maps over in the backend's precision (`_out`).  (Until round 6 such an optic kept
        # the reference's own fp32 chain: an OPD whose last two digits are rounding noise of
        # path lengths 2e5 waves long.)
        try:
            front, table = comp._front_for(wavelength, wavefront_fp64=True)
        except UnsupportedSystem as exc:
            _why("front", f"unsupported system: {exc}")
            return None
    if not hasattr(front.engine, "trace_spot"):
        return None
    comp.last_path = "hip"  # (introspection, as after an intercepted Optic.trace)
    return front, table


def _final_propagation(optic, table, wavelength) -> dict:
    """`last_thickness` / `last_absorb` of `ol_wavefront_params`: what ends `Optic.trace`
    (real_ray_tracer.py:104-110 with propagation/homogeneous.py:44-53) when the last surface
    has a thickness -- {} (0 / 0) otherwise."""
    t = float(table.last_thickness)
    if t == 0.0:
        return {}
    k = _f(optic.surfaces[-1].material_post.k(wavelength))
    return {"last_thickness": t,
            "last_absorb": (4.0 * math.pi * k / float(wavelength)) * 1e3 if k > 0 else 0.0}


def _out(front, t):
    """A result of a fused wavefront launch in the precision the backend works in (a float32
    backend is served by the fp64 kernels: `_front(need_fp64=True)`)."""
    dt = getattr(front, "_hip_out_dtype", None)
    if dt is None or not isinstance(t, torch.Tensor) or not t.is_floating_point():
        return t
    return t.to(dt)


def _register(optic, front, table, launch):
    from . import integration as ig

    dt = getattr(front, "_hip_out_dtype", None)
    if dt is not None:
        # (the surfaces' recorded arrays, if anybody reads them, are the backend's precision)
        hx, hy, px, py, vig, w, flag = launch
        launch = (hx, hy, px.to(dt), py.to(dt), vig, w, flag)
    ig.register_pending_record(optic, table, front.engine, dt or front.dtype, launch)


def _scalar(v):
    if isinstance(v, torch.Tensor):
        return float(v.detach().cpu().reshape(-1)[0]) if v.numel() == 1 else None
    try:
        a = np.asarray(v, dtype=np.float64)
    except (TypeError, ValueError):
        return None
    return float(a.reshape(-1)[0]) if a.size == 1 else None


def _dist_arg(distribution):
    from . import integration as ig

    return distribution if isinstance(distribution, str) else ig._PupilPoints(distribution)


# Engines whose status word is read ONCE, when the reference's fields x wavelengths loop is over
# (`_spot_generate_data`), instead of after every cell's launch -- per thread; None: read per call.
_LOOP = threading.local()


def _image_hits(optic, field, wavelength, num_rays, distribution, local=False, moments=True):
    """(front, table, moments7 (host; None unless `moments`), (x, y, i)) of one fused spot
    launch, or None."""
    hx, hy = _scalar(field[0]), _scalar(field[1])
    if hx is None or hy is None:
        return None
    got = _front(optic, wavelength, final_propagation=True, recorded_row=True)
    if got is None:     # (the data are the recorded row: no final propagation, no PRT matrix)
        return None
    front, table = got
    dist = _dist_arg(distribution)
    # A caller that needs nothing of the launch on the HOST (the encircled-energy cells: device
    # arrays only) inside the grid loop: no read-back per cell -- the status word accumulates in
    # the engine and `_spot_generate_data` reads it once (2 x 9 synchronisations of a 3 x 3
    # `EncircledEnergy` were 0.23 of its 0.49 ms, profiles/r06_seam_profile.txt)
    eng = front.engine
    pending = getattr(_LOOP, "engines", None)
    defer = pending is not None and not moments and getattr(eng, "_status", None) is not None
    if defer and all(e is not eng for e in pending):
        eng._status.zero_()
        pending.append(eng)
    mom, hits = front.trace_spot(hx, hy, wavelength, num_rays, dist, hits=True,
                                 recorded_row=True, local=local, check_status=not defer)
    # what Optic.trace() would have left on the Surface objects, produced on first read
    px, py = front.last_spot_pupil
    _register(optic, front, table, (hx, hy, px, py, front._vig_scalar(hx, hy), wavelength, 0))
    return front, table, (mom.cpu().numpy() if moments else None), hits


def _spot_generate_data(self):
    """analysis/spot_diagram/core.py:420-438: the fields x wavelengths loop -- nothing inside it
    edits the optic, so its tables are validated once per wavelength (`integration.unchanged`).
    Round 5: the WHOLE grid is one launch (`ol_trace_spot_batch`: the cell is the launch grid's
    second dimension) and one read-back when every cell is one the fused kernel serves; the
    reference's own loop -- with the per-cell seam inside it -- otherwise."""
    from . import integration as ig

    with ig.unchanged(self.optic):
        out = None
        try:
            out = _spot_grid(self)
        except UnsupportedSystem:
            out = None
        if out is not None:
            return out
        _LOOP.engines = []
        try:
            out = _ORIG["spot_data"](self)
            try:
                for eng in _LOOP.engines:    # (cells that deferred their status: one read each)
                    eng.raise_for_status(int(eng._status.item()))
            except ValueError:
                # (the record registered for the last cell is that of a trace that raises)
                ig.forget_pending_record(self.optic)
                raise
        finally:
            _LOOP.engines = None
        return out


def _spot_grid(self):
    """[[SpotData per wavelength] per field] from ONE `ol_trace_spot_batch` launch per device
    table (normally one), or None when the grid is not one the batch serves: a per-cell method
    that is not one of this module's seams (a user's subclass), non-scalar fields, a sampler
    that draws a fresh sample per call ("random": every cell of the reference's loop gets its
    own draw), a tilted image surface in local coordinates, an optic the fused kernels decline."""
    per_cell = getattr(type(self), "_generate_field_data", None)
    if per_cell is _spot_generate_field_data:
        masked, coordinates = True, self.coordinates
    elif per_cell is _ee_generate_field_data:
        masked, coordinates = False, "global"   # encircled_energy.py:193-197: unmasked, global
    else:
        return None
    if "spot_data" not in _ORIG or isinstance(self.distribution, str) and \
            self.distribution in ("random", "sobol"):
        return None
    fields = [(_scalar(fp.coord[0]), _scalar(fp.coord[1])) for fp in self.fields]
    if not fields or any(hx is None or hy is None for hx, hy in fields):
        return None
    wls = [wp.value for wp in self.wavelengths]
    if not wls:
        return None
    from . import integration as ig
    if len(wls) > ig._MAX_ENGINES:
        # one engine per wavelength is collected below and the tracer keeps _MAX_ENGINES of them:
        # the ninth would evict -- and close -- the first before the launch
        return None
    main = None      # the front whose geometry the launch reads; the others lend index rows
    cells, last = [], None
    for wi, w in enumerate(wls):
        got = _front(self.optic, w, final_propagation=True, recorded_row=True)
        if got is None:
            return None
        front, table = got
        if not hasattr(front.engine, "trace_spot_batch"):
            return None
        s = table.surfaces[-1]
        # tilted image surface in local coordinates: the kernel leaves the hits in the last
        # surface's own frame (OL_SPOT_HITS_LOCAL) -- visualization/system/utils.py:17-47
        tilted_local = bool(s["flags"] & 1) and coordinates == "local"
        polarised = table.polarization is not None
        if main is None:
            main = (front, table)
        elif front.dtype != main[0].dtype or table.num_surfaces != main[1].num_surfaces:
            return None
        wl, wv = front._wavelength_index(w)
        for fi, (hx, hy) in enumerate(fields):
            if wi == 0 and not (-1.0 <= hx <= 1.0 and -1.0 <= hy <= 1.0):
                # real_ray_tracer.py:156-173 (host scalars: decided here, once per field)
                raise ValueError("Normalized field coordinates must be within (-1, 1)")
            vx, vy = front._vig_scalar(hx, hy)
            cells.append((fi, wi, (hx, hy, vx, vy, 0.0, 0.0, wl, front.engine), wv, front, table))
    from optiland.analysis.spot_diagram.core import SpotData

    data = [[None] * len(wls) for _ in fields]
    front, table = main
    px, py = front._pupil_planes(_dist_arg(self.distribution), self.num_rings)
    n = int(px.numel())
    from . import _capi

    flags = (_capi.SPOT_HITS_LOCAL if tilted_local else 0) \
        | (_capi.SPOT_POLARIZED_OK if polarised else 0)
    if any(getattr(c[2][7], "_handle", True) is None for c in cells):
        return None   # an engine of the grid was closed in the meantime: the per-cell loop
    eng = front.engine
    status_t = getattr(eng, "_status", None)
    if status_t is not None:
        # the ONE read-back of the grid: the per-cell counts and the launch's status word in the
        # same transfer (round 6: they were two synchronisations)
        status_t.zero_()
        mom, hits = eng.trace_spot_batch(px, py, [c[2] for c in cells], hits=True, flags=flags,
                                         check_status=False)
        both = torch.cat([mom[:, 0], status_t.to(torch.float64)]).cpu().numpy()
        front.last_status = int(both[-1])
        eng.raise_for_status(front.last_status)
        counts = both[:-1].astype(np.int64)
    else:   # (engines that raise eagerly: the tests' oracle stand-in)
        mom, hits = eng.trace_spot_batch(px, py, [c[2] for c in cells], hits=True, flags=flags)
        counts = mom[:, 0].cpu().numpy().astype(np.int64)
    xs, ys, ins = hits[:, 0, :n], hits[:, 1, :n], hits[:, 2, :n]
    clipped = masked and bool((counts != n).any())
    if clipped:
        # core.py:470-473 (ignore rays with zero intensity) for ALL cells at once: one
        # `nonzero` over the (cells, n) mask -- cell-major, so every cell's survivors are one
        # contiguous run of `counts[k]` entries -- and three gathers
        sel = (ins > 0).reshape(-1).nonzero().reshape(-1)
        gx, gy, gi = (t.reshape(-1)[sel] for t in (xs, ys, ins))
        ends = np.cumsum(counts)
    s = table.surfaces[-1]
    ox, oy = float(s["origin"][0]), float(s["origin"][1])
    for k, (fi, wi, cell, wv, fr, tb) in enumerate(cells):
        if clipped:
            lo, hi = int(ends[k] - counts[k]), int(ends[k])
            x, y, inten = gx[lo:hi], gy[lo:hi], gi[lo:hi]
        else:
            x, y, inten = xs[k], ys[k], ins[k]
        if coordinates == "local" and not tilted_local:
            # visualization/system/utils.py:17-47 with an untilted image surface:
            # localize = translate by the (folded) origin
            if ox != 0.0:
                x = x - ox
            if oy != 0.0:
                y = y - oy
        data[fi][wi] = SpotData(x=x, y=y, intensity=inten)
        STATS["spot" if masked else "ee"] += 1
        if fi == len(fields) - 1 and wi == len(wls) - 1:
            last = (fr, tb, (cell[0], cell[1], px, py, (cell[2], cell[3]), wv, 0))
    STATS["spot_grid"] += 1
    # what `rms_spot_radius` / `geometric_spot_radius` need to work on the WHOLE grid at once
    # (`_spot_radius`): the (cells, n) blocks the SpotData objects are views of -- only when
    # every cell is such a view (nothing clipped, no local shift applied)
    shifted = coordinates == "local" and not tilted_local and (ox != 0.0 or oy != 0.0)
    self.__dict__["_hip_grid"] = None if (clipped or shifted or not masked) else {
        "xs": xs, "ys": ys, "fields": len(fields), "wls": len(wls),
        "objs": [[data[fi][wi] for wi in range(len(wls))] for fi in range(len(fields))]}
    if last is not None:
        # what the LAST Optic.trace() of the reference's loop would have left on the Surface
        # objects, produced on first read
        _register(self.optic, last[0], last[1], last[2])
    return data


def _spot_radius(self, kind):
    """core.py:342-370 (`geometric_spot_radius`, `rms_spot_radius`) for ALL cells of the grid in
    one pass over the (cells, n) blocks `ol_trace_spot_batch` wrote -- when `self.data` still IS
    what `_spot_grid` handed out.  The centres are the reference's own
    (`_get_reference_centers`: chief rays through the drop-in, or centroids).  The reference's
    form: a deep copy of every cell (`_center_spots`), then five elementwise launches per cell
    -- ~100 launches for a 3 x 3 grid, 1.3 of the 2.9 ms of `SpotDiagram(400 rings)` +
    `rms_spot_radius()` on the MI355X (profiles/r05_spotdiag.json)."""
    g = self.__dict__.get("_hip_grid")
    if g is None:
        return None
    data = self.data
    F, W = g["fields"], g["wls"]
    if len(data) != F or any(len(row) != W for row in data):
        return None
    for fi in range(F):
        for wi in range(W):
            sd, mine = data[fi][wi], g["objs"][fi][wi]
            if sd is not mine or sd.x.data_ptr() != g["xs"][wi * F + fi].data_ptr() \
                    or sd.y.data_ptr() != g["ys"][wi * F + fi].data_ptr() \
                    or sd.x.numel() != g["xs"].shape[1]:
                return None     # somebody replaced a cell or its arrays: the reference's code
    got = _chief_centers(self, F)
    if got is not None:
        cx, cy = got
    else:
        centers = self._get_reference_centers(data)
        cx = torch.stack([c[0].reshape(()) for c in centers]).to(g["xs"].dtype)
        cy = torch.stack([c[1].reshape(()) for c in centers]).to(g["xs"].dtype)
    n = g["xs"].shape[1]
    dx = g["xs"].view(W, F, n) - cx.view(1, F, 1)     # (cells are wavelength-major)
    dy = g["ys"].view(W, F, n) - cy.view(1, F, 1)
    r2 = dx * dx + dy * dy
    # (sqrt is monotone: the largest radius is the root of the largest squared radius)
    out = torch.sqrt(r2.mean(dim=2) if kind == "rms" else r2.amax(dim=2))
    rows = out.t().unbind(0)                          # per field: (W,) views
    STATS["spot_radius"] += 1
    return [list(row.unbind(0)) for row in rows]


def _chief_centers(self, F):
    """The stock `ChiefRayReference` (spot_diagram/reference.py:84-110, global coordinates): the
    image-plane (x, y) of the chief ray of every field at the reference wavelength -- F calls of
    `Optic.trace_generic(Px=0, Py=0)` there, ONE one-ray-per-cell `ol_trace_spot_batch` launch
    here; what the LAST of those calls would have left on the optic's surfaces is registered as
    a pending record, like every seam does.  None: any other strategy / coordinates / an optic
    whose returned rays differ from the recorded row (a trailing thickness)."""
    strat = getattr(self, "_reference_strategy", None)
    try:
        from optiland.analysis.spot_diagram.reference import ChiefRayReference
    except ImportError:
        return None
    if type(strat) is not ChiefRayReference or self.coordinates not in ("global", "local"):
        return None
    fields = [(_scalar(fp.coord[0]), _scalar(fp.coord[1])) for fp in self.fields]
    if len(fields) != F or any(hx is None or hy is None for hx, hy in fields):
        return None
    w = self.wavelengths[self._analysis_ref_wavelength_index].value
    got = _front(self.optic, w, recorded_row=True)
    if got is None:
        return None
    front, table = got
    eng = front.engine
    if not hasattr(eng, "trace_spot_batch") or float(table.last_thickness) != 0.0 \
            or getattr(eng, "_status", None) is None:
        return None
    if self.coordinates == "local":
        # reference.py:103-107 localises the hit to the image surface (`transform`:
        # visualization/system/utils.py:17-47): for an untilted, undecentred image surface that
        # is x - 0.0, y - 0.0 -- the global hit bit for bit.  Anything else: the reference's code.
        img = table.surfaces[-1]
        if bool(img["flags"] & 1) or float(img["origin"][0]) != 0.0 \
                or float(img["origin"][1]) != 0.0:
            return None
    from . import _capi

    wl, wv = front._wavelength_index(w)
    zero = torch.zeros(1, dtype=front.dtype, device=eng.device)
    cells = []
    for hx, hy in fields:
        front._validate_normalized_coordinates(hx, hy, "field")
        vx, vy = front._vig_scalar(hx, hy)
        cells.append((hx, hy, vx, vy, 0.0, 0.0, wl, eng))
    flags = _capi.SPOT_POLARIZED_OK if table.polarization is not None else 0
    _mom, hits = eng.trace_spot_batch(zero, zero, cells, hits=True, flags=flags,
                                      check_status=False)
    hx, hy = fields[-1]
    _register(self.optic, front, table, (hx, hy, zero, zero, front._vig_scalar(hx, hy), wv, 0))
    return hits[:, 0, 0], hits[:, 1, 0]


def _spot_rms_spot_radius(self):
    out = None
    if isinstance(getattr(self, "data", None), list):
        try:
            out = _spot_radius(self, "rms")
        except (AttributeError, RuntimeError, TypeError):
            out = None
    return out if out is not None else _ORIG["spot_rms"](self)


def _spot_geometric_spot_radius(self):
    out = None
    if isinstance(getattr(self, "data", None), list):
        try:
            out = _spot_radius(self, "geometric")
        except (AttributeError, RuntimeError, TypeError):
            out = None
    return out if out is not None else _ORIG["spot_geo"](self)


def _wavefront_generate_data(self):
    """wavefront/wavefront.py:161-176: the same kind of loop (chief-ray traces + OPD launches)."""
    from . import integration as ig

    with ig.unchanged(self.optic):
        return _ORIG["wavefront_data"](self)


# ------------------------------------------------------------------------------- spot
def _spot_generate_field_data(self, field, wavelength, num_rays, distribution, coordinates):
    out = None
    got = None
    try:
        # a tilted image surface in local coordinates: the kernel leaves the hits in the last
        # surface's own frame (decided after the table is known: two tries at most)
        got = _image_hits(self.optic, field, wavelength, num_rays, distribution)
        tilted_local = False
        if got is not None and bool(got[1].surfaces[-1]["flags"] & 1) and coordinates == "local":
            if hasattr(getattr(got[0].engine, "lib", None), "ol_trace_spot_batch"):  # ABI 10
                got = _image_hits(self.optic, field, wavelength, num_rays, distribution,
                                  local=True)
                tilted_local = True
            else:
                got = None   # an engine without OL_SPOT_HITS_LOCAL: the reference's transform
    except UnsupportedSystem:
        got = None
    if got is not None:
        front, table, mom, (x, y, inten) = got
        s = table.surfaces[-1]
        if int(mom[0]) != x.numel():  # core.py:470-473: ignore rays with zero intensity
            mask = inten > 0
            x, y, inten = x[mask], y[mask], inten[mask]
        if coordinates == "local" and not tilted_local:
            # visualization/system/utils.py:17-47 with an untilted image surface:
            # localize = translate by the (folded) origin
            ox, oy = float(s["origin"][0]), float(s["origin"][1])
            if ox != 0.0:
                x = x - ox
            if oy != 0.0:
                y = y - oy
        from optiland.analysis.spot_diagram.core import SpotData

        out = SpotData(x=x, y=y, intensity=inten)
    if out is None:
        STATS["spot_fallback"] += 1
        return _ORIG["spot"](self, field, wavelength, num_rays, distribution, coordinates)
    STATS["spot"] += 1
    return out


def _ee_generate_field_data(self, field, wavelength, num_rays=100, distribution="hexapolar",
                            coordinates="local"):
    try:
        got = _image_hits(self.optic, field, wavelength, num_rays, distribution, moments=False)
    except UnsupportedSystem:
        got = None
    if got is None:
        STATS["ee_fallback"] += 1
        return _ORIG["ee"](self, field, wavelength, num_rays, distribution, coordinates)
    from optiland.analysis.spot_diagram.core import SpotData

    _front_, _table, _mom, (x, y, inten) = got
    STATS["ee"] += 1
    return SpotData(x=x, y=y, intensity=inten)  # encircled_energy.py:193-197: unmasked, global


# -------------------------------------------------------------------------- wavefront
def _f(v) -> float:
    if isinstance(v, torch.Tensor):
        return float(v.detach().cpu().reshape(-1)[0])
    return float(np.asarray(v, dtype=np.float64).reshape(-1)[0])


def _chief_compute_wavefront_data(self, field, wavelength):
    out = None
    try:
        out = _fused_wavefront(self, field, wavelength)
    except UnsupportedSystem:
        out = None
    if out is None:
        STATS["opd_fallback"] += 1
        return _ORIG["opd"](self, field, wavelength)
    STATS["opd"] += 1
    return out


def _chief_init(self, optic, distribution, **kwargs):
    """wavefront/strategy.py:156-160.  `optic.paraxial.XPL() + optic.surfaces.positions[-1]`
    is three walks over the surfaces in backend array operations (6.2 of the 8.3 ms an
    `OPD(lens, ...)` takes on the MI355X box, profiles/r03_analyses_profile.txt); the packed
    table of the drop-in already holds that number (packer._compute_raygen "pupil_z": the
    host first-order model, else the reference's own value)."""
    import optiland.backend as be
    from optiland.wavefront.strategy import ReferenceStrategy

    pupil_z = None
    try:
        got = _front(optic, _f(optic.primary_wavelength), final_propagation=True)
        if got is not None:
            pupil_z = got[1].raygen.get("pupil_z")
    except Exception:  # noqa: BLE001 - anything unexpected: the reference's own constructor
        pupil_z = None
    if pupil_z is None or not math.isfinite(pupil_z):
        STATS["opd_init_fallback"] += 1
        return _ORIG["chief_init"](self, optic, distribution, **kwargs)
    ReferenceStrategy.__init__(self, optic, distribution, **kwargs)  # optic, n_image, ...
    self.pupil_z = be.array([pupil_z])
    self._hip_pupil_z = float(pupil_z)  # (host copy: spares `_fused_wavefront` a read-back)
    self._hip_pupil_z_of = self.pupil_z
    self._chief_ray = None
    STATS["opd_init"] += 1


def _launch_plane_tilt(rg, hx, hy):
    """strategy.py:83-139 _correct_tilt: the direction cosines (ux, uy) of the launch plane's
    tilt -- AngleField with the object at infinity only, else (0, 0)."""
    if rg.get("object_infinite") and int(rg.get("field_kind", 0)) == 0:
        tx = math.tan(math.radians(hx * rg["max_field"]))
        ty = math.tan(math.radians(hy * rg["max_field"]))
        uz = 1.0 / math.sqrt(1.0 + tx * tx + ty * ty)
        return tx * uz, ty * uz
    return 0.0, 0.0


def _fused_wavefront(self, field, wavelength):
    hx, hy = _scalar(field[0]), _scalar(field[1])
    w = _scalar(wavelength)
    if hx is None or hy is None or w is None:
        return None
    if self.reference_type not in ("sphere", "plane"):
        return None
    got = _front(self.optic, w, need_fp64=True, final_propagation=True, polarised_ok=True)
    if got is None:
        return None
    front, table = got
    rg = table.raygen
    if table.polarization is not None or table.uses_polarization:
        return _fused_wavefront_polarised(self, front, table, hx, hy, w)
    if not hasattr(front.engine, "trace_opd"):
        return None
    if float(table.last_thickness) != 0.0 and not hasattr(
            getattr(front.engine, "lib", None), "ol_trace_spot_batch"):
        _why("opd", "last surface has a thickness and the engine does not propagate (ABI < 10)")
        return None
    dist = self.distribution
    dx, dy = getattr(dist, "x", None), getattr(dist, "y", None)
    if dx is None or dy is None:
        return None
    can_dev = getattr(front.engine, "can_wavefront_reference", None)
    if can_dev is not None and can_dev():
        return _fused_wavefront_device(self, front, table, hx, hy, w, dx, dy)
    if getattr(front, "_hip_out_dtype", None) is not None:
        # (an engine without the device-resident reference -- libraries before ABI 6, the
        # tests' stand-in -- would take the chief ray from a float32 trace: not worth a sphere)
        _why("opd", "fp32 wavefront on an engine without ol_wavefront_reference")
        return None
    # 1. chief ray alone (strategy.py:176-179) -- through Optic.trace_generic, i.e. the
    # drop-in's own one-ray launch; kept on the strategy like the reference does
    self._chief_ray = chief = self.optic.trace_generic(hx, hy, Px=0.0, Py=0.0, wavelength=w)
    c = torch.stack([t.reshape(-1)[0] for t in (chief.x, chief.y, chief.z, chief.L, chief.M,
                                                 chief.N, chief.opd)]).double().cpu().tolist()
    xc, yc, zc, Lc, Mc, Nc, opd_c = c
    n_image = rg.get("n_image")  # packed at the primary wavelength, like strategy.py:57
    if n_image is None:
        n_image = _f(self.n_image)
    ux, uy = _launch_plane_tilt(rg, hx, hy)
    params = dict(xc=xc, yc=yc, zc=zc, n_image=n_image, opd_ref=0.0, ux=ux, uy=uy,
                  half_epd=rg["EPD"] / 2.0, wavelength_um=w,
                  **_final_propagation(self.optic, table, w))
    if self.reference_type == "plane":  # strategy.py:260-284
        R = math.inf
        params.update(R=0.0, nx=Lc, ny=Mc, nz=Nc)
        t_back = 0.0
    else:                               # strategy.py:228-243
        pz = self.__dict__.get("_hip_pupil_z")
        if pz is None or self.pupil_z is not self.__dict__.get("_hip_pupil_z_of"):
            pz = _f(self.pupil_z)  # somebody replaced the attribute: read it
        R = math.sqrt(xc * xc + yc * yc + (zc - pz) ** 2)
        params.update(R=R)
        a_ = Lc * Lc + Mc * Mc + Nc * Nc
        sq = math.sqrt(max(4.0 * a_ * R * R, 0.0))
        t1, t2 = -sq / (2.0 * a_), sq / (2.0 * a_)
        t_back = t2 if t1 < 0.0 else t1
    params["opd_ref"] = opd_c - n_image * t_back  # pupil point (0, 0): no tilt term
    # 2. the full pupil: one launch, no ray planes (strategy.py:190-205)
    px, py = front._dev(_as_input(dx)), front._dev(_as_input(dy))
    wl, _ = front._wavelength_index(w)
    opd, inten, pupil, mom = front.engine.trace_opd(
        params, px, py, wl, field=(hx, hy), vig=front._vig_scalar(hx, hy), want_pupil=True)
    _register(self.optic, front, table, (hx, hy, px, py, front._vig_scalar(hx, hy), w, 0))
    from optiland.wavefront.wavefront_data import WavefrontData

    data = WavefrontData(pupil_x=_out(front, pupil[0]), pupil_y=_out(front, pupil[1]),
                         pupil_z=_out(front, pupil[2]), opd=_out(front, opd),
                         intensity=_out(front, inten), radius=R)
    data._hip_fused = True  # lets the FFT-PSF seam recognise device data it can scatter
    _keep_fp64(front, data, opd, inten)
    return data


def _fused_wavefront_polarised(self, front, table, hx, hy, w):
    """strategy.py:163-215 for a POLARISED optic (round 6: the last f4 decline).  The fused OPD
    kernels carry no PRT matrix, so the two traces stay the drop-in's own polarised launches --
    the chief ray through `Optic.trace_generic`, the bundle through `Optic.trace` (generate,
    trace, PRT, `update_intensity` in ONE launch) -- and what follows them, the reference's chain
    of ~25 elementwise operations over the bundle (path length to the reference sphere, tilt,
    normalisation, pupil coordinates), is ONE `ol_wavefront_opd` launch on the returned planes.
    `prt_matrix` / `E_exits` are the returned `PolarizedRays`' own, as in the reference."""
    if getattr(front, "_hip_out_dtype", None) is not None or not hasattr(front.engine, "wavefront_opd"):
        _why("opd", "polarised wavefront of a float32 backend")
        return None
    dist = self.distribution
    dx, dy = getattr(dist, "x", None), getattr(dist, "y", None)
    if dx is None or dy is None:
        return None
    rg = table.raygen
    optic = self.optic
    self._chief_ray = chief = optic.trace_generic(hx, hy, Px=0.0, Py=0.0, wavelength=w)
    c = torch.stack([t.reshape(-1)[0] for t in (chief.x, chief.y, chief.z, chief.L, chief.M,
                                                 chief.N, chief.opd)]).double().cpu().tolist()
    xc, yc, zc, Lc, Mc, Nc, opd_c = c
    n_image = rg.get("n_image")
    if n_image is None:
        n_image = _f(self.n_image)
    ux, uy = _launch_plane_tilt(rg, hx, hy)
    # (the rays `Optic.trace` returns HAVE crossed the last thickness: no last_thickness here)
    params = dict(xc=xc, yc=yc, zc=zc, n_image=n_image, opd_ref=0.0, ux=ux, uy=uy,
                  half_epd=rg["EPD"] / 2.0, wavelength_um=w)
    if self.reference_type == "plane":
        R = math.inf
        params.update(R=0.0, nx=Lc, ny=Mc, nz=Nc)
        t_back = 0.0
    else:
        pz = self.__dict__.get("_hip_pupil_z")
        if pz is None or self.pupil_z is not self.__dict__.get("_hip_pupil_z_of"):
            pz = _f(self.pupil_z)
        R = math.sqrt(xc * xc + yc * yc + (zc - pz) ** 2)
        params.update(R=R)
        a_ = Lc * Lc + Mc * Mc + Nc * Nc
        sq = math.sqrt(max(4.0 * a_ * R * R, 0.0))
        t1, t2 = -sq / (2.0 * a_), sq / (2.0 * a_)
        t_back = t2 if t1 < 0.0 else t1
    params["opd_ref"] = opd_c - n_image * t_back
    rays = optic.trace(hx, hy, w, None, dist)
    planes = [getattr(rays, k) for k in ("x", "y", "z", "L", "M", "N", "opd")]
    if any(not isinstance(t, torch.Tensor) or t.dtype != torch.float64 for t in planes):
        return None
    r7 = [t.detach().reshape(-1).contiguous() for t in planes]
    px, py = front._dev(_as_input(dx)).contiguous(), front._dev(_as_input(dy)).contiguous()
    if px.numel() != r7[0].numel():
        return None
    opd, pupil = front.engine.wavefront_opd(params, r7, px, py, want_pupil=True)
    intensity = optic.surfaces.intensity[-1, :]          # strategy.py:188: the recorded last row
    kwargs = {}
    prt_matrix = getattr(rays, "p", None)
    exit_fields = getattr(rays, "get_exit_fields", None)
    if prt_matrix is not None and exit_fields:
        kwargs["prt_matrix"] = prt_matrix
        kwargs["E_exits"] = exit_fields(optic.polarization_state)
    from optiland.wavefront.wavefront_data import WavefrontData

    return WavefrontData(pupil_x=pupil[0], pupil_y=pupil[1], pupil_z=pupil[2], opd=opd,
                         intensity=intensity, radius=R, **kwargs)


class _LazyChiefRay:
    """`ChiefRayStrategy._chief_ray` (strategy.py:160, 176): the one-ray `RealRays` of the
    chief ray.  The device-resident path has its eight numbers in device memory and nobody in
    the reference reads the attribute outside `compute_wavefront_data`; it is built (one
    `RealRays` from eight device scalars) only if somebody does."""

    def __init__(self, chief8, wavelength):
        self._chief8, self._w, self._rays = chief8, wavelength, None

    def _make(self):
        if self._rays is None:
            from optiland.rays import RealRays

            c = self._chief8
            self._rays = RealRays(c[0:1], c[1:2], c[2:3], c[3:4], c[4:5], c[5:6], c[6:7], self._w)
            self._rays.opd = c[7:8]
        return self._rays

    def __getattr__(self, name):
        # own slots and dunder probes never build the rays: copy.deepcopy / pickle look up
        # `__deepcopy__`, `__reduce_ex__`, `__setstate__` ... on an instance whose __init__ has
        # not run, and `_make()` reading `self._rays` there would re-enter this method for ever
        if name in ("_chief8", "_w", "_rays") or (name.startswith("__") and name.endswith("__")):
            raise AttributeError(name)
        return getattr(self._make(), name)

    def __reduce__(self):
        """pickle / copy.deepcopy (the reference deep-copies strategies with their optic): the
        copy owns its own eight numbers."""
        c = self._chief8
        return (_LazyChiefRay, (c.clone() if hasattr(c, "clone") else c, self._w))


def _fused_wavefront_device(self, front, table, hx, hy, w, dx, dy):
    """strategy.py:163-215 as TWO launches and no read-back: `ol_wavefront_reference` traces
    the chief ray on the device and leaves the reference sphere / plane there,
    `ol_trace_opd_dev` reads it.  (Round 3: a one-ray `Optic.trace_generic` through the
    drop-in, a read-back of seven scalars and the sphere arithmetic on the host -- 0.4 of the
    0.8 ms of an OPD at 256 rings.)"""
    rg = table.raygen
    n_image = rg.get("n_image")
    if n_image is None:
        n_image = _f(self.n_image)
    ux, uy = _launch_plane_tilt(rg, hx, hy)
    planar = self.reference_type == "plane"
    pz = 0.0
    if not planar:
        pz = self.__dict__.get("_hip_pupil_z")
        if pz is None or self.pupil_z is not self.__dict__.get("_hip_pupil_z_of"):
            pz = _f(self.pupil_z)  # somebody replaced the attribute: read it
    params = dict(n_image=n_image, ux=ux, uy=uy, half_epd=rg["EPD"] / 2.0, wavelength_um=w,
                  **_final_propagation(self.optic, table, w))
    vig = front._vig_scalar(hx, hy)
    wl, _ = front._wavelength_index(w)
    eng = front.engine
    ref, chief = eng.wavefront_reference(params, wl, field=(hx, hy), vig=vig, pupil_z=pz,
                                         planar=planar, want_chief=True)
    self._chief_ray = _LazyChiefRay(chief, w)
    px, py = front._dev(_as_input(dx)), front._dev(_as_input(dy))
    opd, inten, pupil, mom = eng.trace_opd(None, px, py, wl, field=(hx, hy), vig=vig,
                                           want_pupil=True, reference=ref, zero_status=False)
    _register(self.optic, front, table, (hx, hy, px, py, vig, w, 0))
    # `radius`: the reference says float (strategy.py:248 `.item()`); here a 0-d device tensor
    # that becomes that float when somebody reads the attribute (the Huygens PSFs do,
    # psf/huygens_fresnel.py:283-300; OPD maps, Zernike fits and the FFT PSF never do)
    data = _device_wavefront_data()(pupil_x=_out(front, pupil[0]), pupil_y=_out(front, pupil[1]),
                                    pupil_z=_out(front, pupil[2]), opd=_out(front, opd),
                                    intensity=_out(front, inten),
                                    radius=math.inf if planar else ref[3])
    data._hip_fused = True
    _keep_fp64(front, data, opd, inten)
    return data


def _fitted_compute_wavefront_data(self, field, wavelength):
    out = None
    try:
        out = _fused_fitted(self, field, wavelength)
    except UnsupportedSystem:
        out = None
    if out is None:
        STATS["opd_fit_fallback"] += 1
        return _ORIG["opd_fit"](self, field, wavelength)
    STATS["opd_fit"] += 1
    return out


_FITTED_METHODS = ("_create_reference_geometry", "_points_from_rays", "_calculate_weights",
                   "_create_spherical_ref", "_create_planar_ref", "_correct_tilt")


def _fitted_kind(cls):
    """"centroid" / "best_fit" when `cls` computes its reference with the stock methods of
    CentroidStrategy / BestFitStrategy (strategy.py:287-605), None for a subclass that
    overrides any of them (which then keeps the reference's own code path)."""
    from optiland.wavefront.strategy import BestFitStrategy, CentroidStrategy

    for kind, owner in (("best_fit", BestFitStrategy), ("centroid", CentroidStrategy)):
        if issubclass(cls, owner):
            same = all(getattr(owner, m, None) is not None
                       and getattr(cls, m, None) is getattr(owner, m) for m in _FITTED_METHODS)
            return kind if same else None
    return None


def _fused_fitted(self, field, wavelength):
    """strategy.py:307-364 (CentroidStrategy.compute_wavefront_data, inherited by
    BestFitStrategy) as: ONE generating launch that records the image surface only,
    `ol_wavefront_fit` (the reference's chain of array reductions and host decisions as a chain
    of device passes, result left on the device), `ol_wavefront_opd_fitted` -- and one read-back
    at the end (radius, centre, the fit's status) next to the trace's status word."""
    hx, hy = _scalar(field[0]), _scalar(field[1])
    w = _scalar(wavelength)
    if hx is None or hy is None or w is None:
        return None
    if self.reference_type not in ("sphere", "plane"):
        return None
    kind = _fitted_kind(type(self))
    if kind is None:
        return None
    got = _front(self.optic, w, need_fp64=True, final_propagation=True)
    if got is None:
        return None
    front, table = got
    eng = front.engine
    if not getattr(eng, "can_wavefront_fit", lambda: False)() \
            or not getattr(eng, "can_trace_generate", lambda: False)():
        return None
    dist = self.distribution
    dx, dy = getattr(dist, "x", None), getattr(dist, "y", None)
    if dx is None or dy is None:
        return None
    trim = getattr(self, "robust_trim_std", 0.0)
    trim = float(trim) if trim else 0.0
    rg = table.raygen
    n_image = rg.get("n_image")
    if n_image is None:
        n_image = _f(self.n_image)
    ux, uy = _launch_plane_tilt(rg, hx, hy)
    planar = self.reference_type == "plane"
    params = dict(n_image=n_image, ux=ux, uy=uy, half_epd=rg["EPD"] / 2.0, wavelength_um=w)
    px = front._dev(_as_input(dx)).contiguous()
    py = front._dev(_as_input(dy)).contiguous()
    vig = front._vig_scalar(hx, hy)
    wl, _ = front._wavelength_index(w)
    res = eng.trace_generate(px, py, wl, field=(hx, hy), vig=vig, record=True,
                             record_first=eng.num_surfaces - 1, defer_status=True)
    x, y, z, L, M, N, inten, opd_in = res.rows(res.last)
    last = _final_propagation(self.optic, table, w)
    if last:
        # `Optic.trace` ends with a propagation by the last surface's thickness
        # (real_ray_tracer.py:104-110, homogeneous.py:39-53): the fitted strategies read the
        # RETURNED rays (strategy.py:319), i.e. positions moved on by t along the ray and the
        # intensity after t of the last medium -- three elementwise operations on the recorded
        # row (round 6; until then such an optic kept the reference's own chain)
        t = last["last_thickness"]
        x, y, z = x + t * L, y + t * M, z + t * N
        if last["last_absorb"] > 0.0:
            inten = inten * math.exp(-last["last_absorb"] * t)
    r8 = [x, y, z, L, M, N, opd_in, inten]
    ref = eng.wavefront_fit(kind, params, r8, px, py, trim_std=trim, flavour="torch",
                            planar=planar)
    opd, pupil = eng.wavefront_opd_fitted(ref, r8[:7], px, py, want_pupil=True)
    host = ref.cpu()                                    # the one wait for the device
    front._finish_checks(eng)
    from . import _capi
    bits = int(host[-1:].view(torch.int32)[0])
    if bits & _capi.FIT_SINGULAR and not bits & (_capi.FIT_NO_VALID | _capi.FIT_TOO_FEW):
        # wavefront points that do not span space (a collimated beam: one plane): the device
        # fit has no fourth pivot.  The reference's own code -- `lstsq` of its backend on the
        # raw system, a rank-3 minimum-norm sphere -- runs instead, on the drop-in's trace.
        _why("opd_fit", "rank-deficient wavefront points (singular device fit)")
        return None
    eng.raise_for_fit_status(bits)
    if kind == "best_fit" and not planar:
        self.center = tuple(float(v) for v in host[0:3])  # strategy.py:581
    _register(self.optic, front, table, (hx, hy, px, py, vig, w, 0))
    data = _device_wavefront_data()(pupil_x=_out(front, pupil[0]), pupil_y=_out(front, pupil[1]),
                                    pupil_z=_out(front, pupil[2]), opd=_out(front, opd),
                                    intensity=_out(front, inten),
                                    radius=math.inf if planar else float(host[3]))
    data._hip_fused = True
    _keep_fp64(front, data, opd, inten)
    return data


def _keep_fp64(front, data, opd, inten):
    """A float32 backend's map also keeps what the kernel computed (fp64): the FFT-PSF seam
    scatters THAT into the padded grid (`_fused_pupils`) and rounds once, at the end."""
    if getattr(front, "_hip_out_dtype", None) is not None:
        data.__dict__["_hip_fp64"] = (opd, inten)


_DEVICE_WAVEFRONT_DATA = None


def _device_wavefront_data():
    """`WavefrontData` whose `radius` may be handed over as a 0-d device tensor: read back
    (once) when the attribute is read, so that the two launches of
    `_fused_wavefront_device` need no synchronisation and the attribute still IS the float
    the reference's dataclass declares (wavefront_data.py:36).  Pickles as the stock class."""
    global _DEVICE_WAVEFRONT_DATA
    if _DEVICE_WAVEFRONT_DATA is None:
        from optiland.wavefront.wavefront_data import WavefrontData

        class DeviceWavefrontData(WavefrontData):
            @property
            def radius(self):
                r = self.__dict__["_radius"]
                if isinstance(r, torch.Tensor):
                    r = self.__dict__["_radius"] = float(r)
                return r

            @radius.setter
            def radius(self, value):
                self.__dict__["_radius"] = value

            def __reduce__(self):
                return (WavefrontData, (self.pupil_x, self.pupil_y, self.pupil_z, self.opd,
                                        self.intensity, self.radius, self.prt_matrix,
                                        self.E_exits))

        DeviceWavefrontData.__name__ = DeviceWavefrontData.__qualname__ = "WavefrontData"
        _DEVICE_WAVEFRONT_DATA = DeviceWavefrontData
    return _DEVICE_WAVEFRONT_DATA


def _as_input(v):
    if isinstance(v, torch.Tensor):
        return v.detach()
    return np.asarray(v, dtype=np.float64)


# ---------------------------------------------------------------------------- FFT PSF
def _fft_generate_pupils(self):
    """psf/fft.py:123-161 with `ol_pupil_fill`: the samples go straight into the zero-padded
    grid; the list the reference keeps (`self.pupils`, n x n each) are views of it."""
    import optiland.backend as be

    out = None
    try:
        out = _fused_pupils(self, be)
    except UnsupportedSystem:
        out = None
    if out is None:
        STATS["pupil_fallback"] += 1
        self.__dict__.pop("_hip_padded", None)
        return _ORIG["pupils"](self)
    STATS["pupil"] += 1
    return out


def _fused_pupils(self, be):
    n, gsz = int(self.num_rays), int(self.grid_size)
    field = self.fields[0]
    datas = [self.get_data(field, wl) for wl in self.wavelengths]
    if not datas or any(not getattr(d, "_hip_fused", False) for d in datas):
        return None
    got = _front(self.optic, _f(getattr(self.wavelengths[0], "value", self.wavelengths[0])),
                 need_fp64=True)
    if got is None:
        return None
    front, _table = got
    if not hasattr(front.engine, "pupil_fill"):
        return None
    # the disc mask in the reference's own arithmetic (fft.py:140-144)
    x = be.linspace(-1, 1, n)
    x, y = be.meshgrid(x, x)
    cells = torch.nonzero((x.ravel() ** 2 + y.ravel() ** 2) <= 1).reshape(-1)
    dev = datas[0].opd.device
    cell = cells.to(device=dev, dtype=torch.int32)
    before = (gsz - n) // 2
    padded, views = [], []
    out_dt = getattr(front, "_hip_out_dtype", None)
    for d in datas:
        opd, inten = d.__dict__.get("_hip_fp64") or (d.opd, d.intensity)
        if opd.numel() != cell.numel() or opd.dtype != torch.float64:
            return None
        grid = front.engine.pupil_fill(opd.contiguous(), inten.contiguous(), cell, n, gsz)
        if out_dt == torch.float32:
            grid = grid.to(torch.complex64)   # (what `be.exp(1j * ...)` of a float32 backend is)
        padded.append(grid)
        views.append(grid[before:before + n, before:before + n])
    self._hip_padded = (padded, views)
    return views


def _fft_pad_pupils(self):
    """psf/fft.py:203-230: the grids `ol_pupil_fill` wrote ARE the padded pupils -- as long
    as `self.pupils` still holds the views handed out by `_generate_pupils`."""
    kept = self.__dict__.get("_hip_padded")
    if kept is not None and len(kept[1]) == len(self.pupils) \
            and all(a is b for a, b in zip(kept[1], self.pupils)):
        return list(kept[0])
    return _ORIG["pad"](self)


# ------------------------------------------------------------------- pupil distributions
POINTS_HOOK = None  # tests: callable(kind, num, dtype) -> (x, y) standing in for the device


def _device_points(kind, num):
    """(x, y) of a deterministic sampler from `ol_pupil_points` in the backend's precision on
    the HIP device, or None when the call is not one the device sampler serves (another
    backend / device, autograd on, library without the entry point)."""
    import ctypes as C

    import optiland.backend as be

    from . import _capi
    from . import distribution as D
    from . import integration as ig

    if be.get_backend() not in (ig.BACKEND_NAME, "torch"):
        return None
    cfg = be._backends[be.get_backend()]._config
    if cfg.grad_mode.requires_grad:
        return None
    try:
        num = int(num)
    except (TypeError, ValueError):
        return None
    if POINTS_HOOK is not None:  # tests: the host build of the same source
        return POINTS_HOOK(kind, num, cfg.get_precision())
    if cfg.get_device() != "cuda" or not torch.cuda.is_available():
        return None
    try:
        lib = _capi.load()
    except Exception:  # noqa: BLE001 - no library: the reference's own sampler
        return None
    if not hasattr(lib, "ol_pupil_points"):
        return None
    dtype = cfg.get_precision()
    # the backend's configured device (backend/torch/config: `get_device()`), not whatever
    # device happens to be current in this thread
    dev = torch.device(cfg.get_device())
    if dev.type != "cuda":
        return None
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    first = offset = None
    if kind == "hexapolar":
        if num < 0:
            return None
        n, code = D.hexapolar_count(num), 0
    else:  # (no seam uses it; kept for callers of `_device_points("uniform", n)`)
        if num < 2:
            return None
        f, o = D.uniform_rows(num)
        n, code = int(o[-1]), 1
        first, offset = torch.as_tensor(f, device=dev), torch.as_tensor(o, device=dev)
    x = torch.empty(n, dtype=dtype, device=dev)
    y = torch.empty(n, dtype=dtype, device=dev)
    rc = lib.ol_pupil_points(code, num, _capi.F32 if dtype == torch.float32 else _capi.F64, n,
                             C.c_void_p(first.data_ptr()) if first is not None else None,
                             C.c_void_p(offset.data_ptr()) if offset is not None else None,
                             C.c_void_p(x.data_ptr()), C.c_void_p(y.data_ptr()),
                             C.c_void_p(torch.cuda.current_stream().cuda_stream))
    if rc != 0:
        return None
    return x, y


def _hexapolar_generate_points(self, num_rings=6):
    """distribution.py:201-220 is a Python loop over the rings with four backend array
    operations and two concatenations each: 10.2 of the 11 ms an `OPD(lens, ..., num_rays=256)`
    takes on the device (profiles/r04_opd_profile.txt).  Same points, same order, from ONE
    launch of `ol_pupil_points`: NumPy's formula in fp64 rounded to the backend's precision --
    in fp32 that is within 1 ulp of, not bit-identical to, what the reference's torch backend
    computes with its own fp32 `linspace / cos / sin`."""
    got = _device_points("hexapolar", num_rings)
    if got is None:
        STATS["dist_fallback"] += 1
        return _ORIG["dist_hex"](self, num_rings)
    STATS["dist"] += 1
    self.x, self.y = got


# (No sampler of our own behind `UniformDistribution.generate_points`: its consumers build the
# SAME grid once more with the backend's own `linspace` and rely on the two masks agreeing point
# for point (psf/fft.py:140-155) -- the torch backend's `linspace` walks the upper half of the
# interval back from the end point and its disc mask differs from NumPy's in a few rim points
# (25445 vs 25441 at 181 x 181).  The stand-alone tracer's "uniform" keeps NumPy's arithmetic, as
# before.  What the seam below does instead: REMEMBER the backend's own answer.)
_UNIFORM_MEMO: dict = {}


def _uniform_generate_points(self, num_points):
    """distribution.py:176-186 on the torch backend: linspace, meshgrid, two squares, a sum, a
    compare and two boolean selections (each a device-to-host read of the count) -- ~0.15 ms of
    every `FFTPSF(...)` / uniform `OPD(...)` constructor for a grid that depends on `num_points`
    and the backend's arithmetic alone.  The reference's own result is kept per (num_points,
    backend, precision, device) and handed out as COPIES (two launches); autograd on, another
    backend, a subclass: the reference's code."""
    import optiland.backend as be
    from optiland.distribution import UniformDistribution

    original = _ORIG["dist_uniform"]
    try:
        if type(self) is not UniformDistribution or be.get_backend() != "torch":
            return original(self, num_points)
        cfg = be._backends["torch"]._config
        if cfg.grad_mode.requires_grad:
            return original(self, num_points)
        key = (int(num_points), cfg.get_precision(), str(cfg.get_device()),
               torch.cuda.current_device() if str(cfg.get_device()) == "cuda" else -1)
    except Exception:  # noqa: BLE001 - a backend object without that configuration
        return original(self, num_points)
    hit = _UNIFORM_MEMO.get(key)
    if hit is None:
        original(self, num_points)
        x, y = getattr(self, "x", None), getattr(self, "y", None)
        if isinstance(x, torch.Tensor) and isinstance(y, torch.Tensor) and x.numel() <= (1 << 22):
            if len(_UNIFORM_MEMO) >= 16:
                _UNIFORM_MEMO.pop(next(iter(_UNIFORM_MEMO)))
            _UNIFORM_MEMO[key] = (x.detach().clone(), y.detach().clone())
        return None
    STATS["uniform_memo"] = STATS.get("uniform_memo", 0) + 1
    self.x, self.y = hit[0].clone(), hit[1].clone()
    return None


# --------------------------------------------------------------------------- (de)activate
# The seams replace PRIVATE methods of the reference.  Each entry: key in _ORIG -> (module,
# class, method, the parameter names the replacement was written against, replacement).  A
# seam is only installed when the reference's method still has exactly that signature: a
# release that renames a parameter, adds one or drops the method leaves that seam OFF (the
# analysis then runs the reference's own code on top of the drop-in's `Optic.trace`) instead
# of silently losing half of a fast path.  Seams that only work as a pair are installed
# together or not at all (`_GROUPS`).  What was skipped is listed in `SKIPPED`.
_SEAMS = {
    "wavefront_data": ("optiland.wavefront.wavefront", "Wavefront", "_generate_data",
                       ("self",), "_wavefront_generate_data"),
    "spot": ("optiland.analysis.spot_diagram.core", "SpotDiagram", "_generate_field_data",
             ("self", "field", "wavelength", "num_rays", "distribution", "coordinates"),
             "_spot_generate_field_data"),
    "spot_data": ("optiland.analysis.spot_diagram.core", "SpotDiagram", "_generate_data",
                  ("self",), "_spot_generate_data"),
    "spot_rms": ("optiland.analysis.spot_diagram.core", "SpotDiagram", "rms_spot_radius",
                 ("self",), "_spot_rms_spot_radius"),
    "spot_geo": ("optiland.analysis.spot_diagram.core", "SpotDiagram", "geometric_spot_radius",
                 ("self",), "_spot_geometric_spot_radius"),
    "ee": ("optiland.analysis.encircled_energy", "EncircledEnergy", "_generate_field_data",
           ("self", "field", "wavelength", "num_rays", "distribution", "coordinates"),
           "_ee_generate_field_data"),
    "opd": ("optiland.wavefront.strategy", "ChiefRayStrategy", "compute_wavefront_data",
            ("self", "field", "wavelength"), "_chief_compute_wavefront_data"),
    "opd_fit": ("optiland.wavefront.strategy", "CentroidStrategy", "compute_wavefront_data",
                ("self", "field", "wavelength"), "_fitted_compute_wavefront_data"),
    "chief_init": ("optiland.wavefront.strategy", "ChiefRayStrategy", "__init__",
                   ("self", "optic", "distribution", "kwargs"), "_chief_init"),
    "pupils": ("optiland.psf.fft", "ScalarFFTPSF", "_generate_pupils", ("self",),
               "_fft_generate_pupils"),
    "pad": ("optiland.psf.fft", "ScalarFFTPSF", "_pad_pupils", ("self",), "_fft_pad_pupils"),
    "dist_hex": ("optiland.distribution", "HexagonalDistribution", "generate_points",
                 ("self", "num_rings"), "_hexapolar_generate_points"),
    "field_coords": ("optiland.fields.field_group", "FieldGroup", "get_field_coords", ("self",),
                     "_memo_get_field_coords"),
    "dist_uniform": ("optiland.distribution", "UniformDistribution", "generate_points",
                     ("self", "num_points"), "_uniform_generate_points"),
    "wf_init": ("optiland.wavefront.wavefront", "Wavefront", "__init__",
                ("self", "optic", "fields", "wavelengths", "num_rays", "distribution", "strategy",
                 "afocal", "remove_tilt", "kwargs"), "_wavefront_init"),
    "fft_init": ("optiland.psf.fft", "ScalarFFTPSF", "__init__",
                 ("self", "optic", "field", "wavelength", "num_rays", "grid_size", "strategy",
                  "remove_tilt", "kwargs"), "_fft_init"),
}
def _constructor_scope(key):
    """A constructor of the reference that only READS its optic -- `Wavefront.__init__`
    (wavefront/wavefront.py:56-90: fields, strategy, the fields x wavelengths loop) and
    `ScalarFFTPSF.__init__` (psf/fft.py:87-123: the same, then pupils and the transform) -- run
    inside `integration.unchanged(optic)`: the change detector walks the optic once per
    constructor instead of once per seam it passes through (strategy constructor, wavefront
    data, pupil fill: 3-4 walks of ~0.06 ms in a 0.4-0.9 ms call, profiles/r06_seam_profile.txt)."""
    def init(self, *args, **kwargs):
        from . import integration as ig

        optic = kwargs.get("optic", args[0] if args else None)
        with ig.unchanged(optic):
            return _ORIG[key](self, *args, **kwargs)
    return init


_wavefront_init = _constructor_scope("wf_init")
_fft_init = _constructor_scope("fft_init")


def _memo_get_field_coords(self):
    """`FieldGroup.get_field_coords()` (fields/field_group.py:124-139), remembered.  Every analysis
    constructor asks for it (`utils.resolve_fields(optic, "all")`), and on the torch backend each
    call is a dozen tiny device launches and 2 F + 1 device-to-host reads (`float(x / max_field)`
    per coordinate): ~0.15 ms of a 0.5 ms `SpotDiagram(lens)`.  The answer is a pure function of
    the fields' coordinates and of the backend's arithmetic: it is computed ONCE by the
    reference's own code and handed out again while the (x, y) of every field -- plain Python
    numbers -- and the backend / precision / device are what they were.  Tensors as field
    coordinates (an optimisation variable): no memo."""
    import optiland.backend as be

    original = _ORIG["field_coords"]
    vals = []
    for f in self.fields:
        x, y = f.x, f.y
        if type(x) not in (int, float) or type(y) not in (int, float):
            return original(self)
        vals.append((x, y))
    try:
        cfg = be._backends[be.get_backend()]._config
        key = (tuple(vals), be.get_backend(), cfg.get_precision(), str(cfg.get_device()))
    except Exception:  # noqa: BLE001 - a backend without that configuration object: no memo
        return original(self)
    memo = self.__dict__.get("_hip_field_coords")
    if memo is not None and memo[0] == key:
        STATS["field_coords_memo"] = STATS.get("field_coords_memo", 0) + 1
        return list(memo[1])
    out = original(self)
    if all(type(a) in (int, float) and type(b) in (int, float) for a, b in out):
        self.__dict__["_hip_field_coords"] = (key, tuple(out))
    return out


# (dependent seams ..., the seam they need): the FFT-PSF pair scatters the OPD seam's device data
_GROUPS = (("pupils", "pad", "opd"),)
SKIPPED: dict = {}                      # key -> why the seam was not installed


def _resolve(key):
    """(class, current method) of a seam, or a reason (str) why it cannot be installed."""
    import importlib
    import inspect

    mod, cls_name, meth, params, _ = _SEAMS[key]
    try:
        module = importlib.import_module(mod)
    except ImportError as exc:
        # an optional dependency of the reference that is not installed (numba, vtk ...): the
        # seam stays off -- and `enable()` SAYS so: the analyses then run at reference speed
        return f"import of {mod} failed ({exc})"
    cls = getattr(module, cls_name, None)
    if cls is None:
        return f"{mod}.{cls_name} not found"
    fn = cls.__dict__.get(meth)
    if fn is None:
        return f"{cls_name}.{meth} is not defined on the class"
    try:
        have = tuple(inspect.signature(fn).parameters)
    except (TypeError, ValueError) as exc:
        return f"{cls_name}.{meth}: no signature ({exc})"
    if have != params:
        return f"{cls_name}.{meth}{have} is not the {params} this seam was written against"
    return cls, fn


def enable():
    """Patch the reference classes (idempotent).  Safe class-wide: each call falls back to
    the original method unless the optic it concerns is served by the drop-in.  Seams whose
    target no longer has the signature they were written against are left off (`SKIPPED`)."""
    if _ORIG:
        return
    SKIPPED.clear()
    found = {}
    for key in _SEAMS:
        got = _resolve(key)
        if isinstance(got, str):
            SKIPPED[key] = got
        else:
            found[key] = got
    for *dependents, needs in _GROUPS:
        missing = [k for k in (*dependents, needs) if k not in found]
        if missing:   # the dependent seams go together; the one they depend on may stay
            for k in dependents:
                if found.pop(k, None) is not None:
                    SKIPPED[k] = (f"needs {missing[0]}: "
                                  f"{SKIPPED.get(missing[0], 'not installed')}")
    for key, (cls, fn) in found.items():
        _ORIG[key] = fn
        setattr(cls, _SEAMS[key][2], globals()[_SEAMS[key][4]])
    if SKIPPED:
        import warnings

        warnings.warn("optiland_amd: analysis seams left OFF (the reference's own code runs on "
                      "top of the drop-in trace for them): "
                      + "; ".join(f"{k}: {why}" for k, why in SKIPPED.items()),
                      RuntimeWarning, stacklevel=2)
    if not _ORIG:
        _ORIG["_none"] = None   # (enable() stays idempotent even if nothing could be patched)


def disable():
    if not _ORIG:
        return
    import importlib

    for key, fn in _ORIG.items():
        if key == "_none":
            continue
        mod, cls_name, meth, _params, _ = _SEAMS[key]
        setattr(getattr(importlib.import_module(mod), cls_name), meth, fn)
    _ORIG.clear()
