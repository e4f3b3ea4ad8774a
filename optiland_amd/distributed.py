"""Multi-GPU ray sharding: one process per GPU, RCCL over xGMI.

Rays are independent (no inter-ray term anywhere on the path), so
the path shards trivially: rank r traces the contiguous block
[r*N/W, (r+1)*N/W) of the input order -- concatenating the shards in rank order
reproduces the single-device result exactly, field-major order included.  The
only exchange is at the image plane, after the trace:

* `allreduce_spot_moments` -- per-rank masked moments (count, sum x, sum y,
  sum x^2, sum y^2) reduced with ONE all-reduce of six doubles; centroid and
  RMS spot radius follow on every rank (analysis/spot_diagram/core.py:329-372).
  Preferred: 48 bytes on the wire instead of 12-24 B per ray.
* `allgather_hits` -- the literal all-gather of image-plane hits (x, y,
  intensity) for consumers that need every hit (PSF / irradiance binning).
  On the xGMI full mesh each rank's shard crosses one direct link per peer;
  volume = 3*b bytes per ray per peer, so at 1e7 rays/GPU fp32 the gather
  (120 MB per shard) costs more than the 0.6 ms trace -- use the reduction
  where the consumer allows.

`torch.distributed` backend "nccl" is RCCL on ROCm; the CPU tests run the same
code over "gloo".
"""

from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist


def shard_bounds(n: int, world: int, rank: int) -> tuple[int, int]:
    """Contiguous shard [lo, hi) of n rays for `rank`; sizes differ by at most 1."""
    base, rem = divmod(int(n), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _world(group=None):
    if not dist.is_available() or not dist.is_initialized():
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _exchanging(group=None) -> bool:
    """True when a process group exists: the collectives below then RUN, also in a group of
    one rank (a one-rank RCCL group on one MI355X is how the device tests exercise the real
    RCCL calls; a single process without a group launches nothing)."""
    return dist.is_available() and dist.is_initialized()


def allgather_hits(x, y, intensity, n_total: int | None = None, group=None):
    """All-gather image-plane hits in rank order -> (x, y, intensity) of all rays.

    Shards may be ragged (sizes from `shard_bounds`); they are padded to the
    largest shard for the collective and trimmed afterwards.
    """
    world, rank = _world(group)
    if not _exchanging(group):
        return x, y, intensity
    n_local = int(x.numel())
    sizes = torch.tensor([n_local], dtype=torch.int64, device=x.device)
    all_sizes = [torch.zeros_like(sizes) for _ in range(world)]
    dist.all_gather(all_sizes, sizes, group=group)
    all_sizes = [int(s.item()) for s in all_sizes]
    m = max(all_sizes)
    send = torch.zeros((3, m), dtype=x.dtype, device=x.device)
    send[0, :n_local], send[1, :n_local], send[2, :n_local] = x, y, intensity
    recv = torch.empty((world, 3, m), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=group)
    parts = [recv[r, :, : all_sizes[r]] for r in range(world)]
    out = torch.cat(parts, dim=1)
    if n_total is not None:
        assert out.shape[1] == n_total
    return out[0], out[1], out[2]


def allreduce_spot_moments(moments: torch.Tensor, group=None) -> torch.Tensor:
    """Sum the 6-element moment vectors of all ranks (in place) and return it."""
    if _exchanging(group):
        dist.all_reduce(moments, op=dist.ReduceOp.SUM, group=group)
    return moments


def allreduce_max(value: torch.Tensor, group=None) -> torch.Tensor:
    if _exchanging(group):
        dist.all_reduce(value, op=dist.ReduceOp.MAX, group=group)
    return value


def spot_statistics(engine, x, y, intensity, group=None) -> dict:
    """Centroid, RMS and geometric spot radius over ALL ranks' hits with i > 0.

    Two tiny collectives (6 doubles, then 1 double) instead of gathering hits.  A rank
    whose shard is empty (`x is None` or zero length) contributes zero moments without
    launching anything but still takes part in both collectives; when no ray at all
    reaches the image plane the statistics are NaN (as `spot7_statistics` reports them).
    """
    empty = x is None or x.numel() == 0
    if empty:
        mom = torch.zeros(6, dtype=torch.float64, device=engine.device)
    else:
        mom = engine.spot_moments(x, y, intensity)
    mom = allreduce_spot_moments(mom, group)
    cnt = float(mom[0])
    if cnt == 0:
        cx = cy = float("nan")
    else:
        cx, cy = float(mom[1]) / cnt, float(mom[2]) / cnt
    if empty or cnt == 0:  # max r^2 >= 0: zero is the neutral element
        mx = torch.zeros(1, dtype=torch.float64, device=engine.device)
    else:
        mx = engine.spot_max_r2(x, y, intensity, cx, cy)
    mx = allreduce_max(mx, group)
    if cnt == 0:
        return {"count": 0.0, "centroid": (cx, cy), "rms_radius": float("nan"),
                "geometric_radius": float("nan")}
    # mean of (x-cx)^2 + (y-cy)^2 = E[x^2] + E[y^2] - cx^2 - cy^2
    rms2 = float(mom[3]) / cnt + float(mom[4]) / cnt - cx * cx - cy * cy
    return {"count": cnt, "centroid": (cx, cy), "rms_radius": max(rms2, 0.0) ** 0.5,
            "geometric_radius": float(mx[0]) ** 0.5}


def allreduce_spot7(mom7: torch.Tensor, group=None) -> torch.Tensor:
    """Combine the seven doubles of `ol_trace_spot` across ranks (in place): the six
    sums with one SUM all-reduce, the max r^2 with one MAX all-reduce."""
    if _exchanging(group):
        dist.all_reduce(mom7[:6], op=dist.ReduceOp.SUM, group=group)
        dist.all_reduce(mom7[6:], op=dist.ReduceOp.MAX, group=group)
    return mom7


def spot7_statistics(mom7, center) -> dict:
    """Centroid / RMS / geometric radius from moments taken about `center`."""
    m = [float(v) for v in mom7]
    cnt = m[0]
    if cnt == 0:
        return {"count": 0.0, "centroid": (float("nan"),) * 2, "rms_radius": float("nan"),
                "geometric_radius": float("nan"), "center": tuple(center)}
    return {"count": cnt, "centroid": (center[0] + m[1] / cnt, center[1] + m[2] / cnt),
            "rms_radius": max((m[3] + m[4]) / cnt, 0.0) ** 0.5,
            "geometric_radius": m[6] ** 0.5, "center": tuple(center)}


class ShardedTracer:
    """Trace this rank's shard of a global ray list and exchange image-plane hits."""

    def __init__(self, tracer, group=None):
        self.tracer = tracer
        self.group = group
        self.world, self.rank = _world(group)

    @contextlib.contextmanager
    def _whole_batch_newton_counts(self):
        """Reference-rule Newton surfaces (`reference_newton`, newton_raphson.py:137-166): the
        number of updates is a property of the WHOLE batch, so the counts of the shards are
        combined (MAX, 8 S bytes) after every counting launch -- while a SHARDED call runs: a
        trace one rank makes by itself on the same engine must not wait for the others."""
        eng = getattr(self.tracer, "engine", None)
        if eng is None or not _exchanging(self.group) or not hasattr(eng, "newton_count_hook"):
            yield    # (an engine stand-in without reference-rule Newton surfaces)
            return
        was = eng.newton_count_hook
        eng.newton_count_hook = lambda iters: allreduce_max(iters, self.group)
        try:
            yield
        finally:
            eng.newton_count_hook = was

    def _no_reference_newton(self, what: str):
        # the fused entry points generate the rays inside the launch; the counting launches of
        # the reference's stop rule read them (capi: OL_ERR_UNSUPPORTED) -- say so here
        if self.tracer.table.reference_newton_surfaces():
            raise ValueError(f"{what}: not with reference-rule Newton surfaces "
                             "(reference_newton); use trace_generic")

    def trace_spot(self, Hx: float, Hy: float, Px, Py, wavelength, center=(0.0, 0.0)):
        """Fused spot of ONE field point over a GLOBAL pupil list: this rank runs
        `ol_trace_spot` on its shard of (Px, Py); the seven doubles are all-reduced.
        No ray plane is written and 56 bytes cross the wire.  RMS / geometric radius
        are about `center` (e.g. the chief-ray hit); the centroid is absolute."""
        t = self.tracer
        if t.table.polarization is not None or t.table.uses_polarization:
            raise ValueError("trace_spot: fused spot reduction needs an unpolarised system")
        self._no_reference_newton("trace_spot")
        px, py = t._dev(Px), t._dev(Py)
        lo, hi = shard_bounds(px.numel(), self.world, self.rank)
        wl, _ = t._wavelength_index(wavelength)
        hx, hy = float(Hx), float(Hy)
        mom = t.engine.trace_spot(px[lo:hi].contiguous(), py[lo:hi].contiguous(), wl,
                                  field=(hx, hy), vig=t._vig_scalar(hx, hy), center=center)
        return spot7_statistics(allreduce_spot7(mom, self.group).cpu(), center)

    def trace_generic(self, Hx, Hy, Px, Py, wavelength, exchange: str = "reduce"):
        """Global per-ray arrays in; local rays + exchanged image-plane data out."""
        t = self.tracer
        arrs = [t._dev(a) for a in (Hx, Hy, Px, Py)]
        n = max(a.numel() for a in arrs)
        lo, hi = shard_bounds(n, self.world, self.rank)
        out = {"rays": None, "lo": lo, "hi": hi, "n_total": n}
        with self._whole_batch_newton_counts():
            if hi > lo:
                # one-element coordinates are broadcast over the GLOBAL list (n is the same on
                # every rank because it is taken before slicing)
                loc = [a.expand(hi - lo) if a.numel() == 1 else a[lo:hi] for a in arrs]
                rays = out["rays"] = t.trace_generic(*loc, wavelength)
                x, y, i = rays.x, rays.y, rays.i
            else:  # more ranks than rays: nothing to launch, zero contribution below
                x = y = i = torch.empty(0, dtype=t.dtype, device=t.device)
                empty = getattr(t.engine, "newton_counts_of_an_empty_shard", None)
                if empty is not None:
                    empty()
        if exchange == "gather":
            out["hits"] = allgather_hits(x, y, i, n, self.group)
        elif exchange == "reduce":
            out["spot"] = spot_statistics(t.engine, x, y, i, self.group)
        return out

    def alloc_field_record(self, n_total: int):
        """A record block for this rank's shard of `trace_field(...)` over `n_total` pupil
        points, to be REUSED by every step of a loop (`trace_field(..., record=block)`).  On
        the device it is placed where the part writes the record-all store pattern fastest
        (`HipSystem.alloc_record_placed`: up to 21 % over where the allocator would put it)."""
        lo, hi = shard_bounds(int(n_total), self.world, self.rank)
        eng, dtype = self.tracer.engine, self.tracer.dtype
        placed = getattr(eng, "alloc_record_placed", None)
        if placed is not None:
            return placed(hi - lo, dtype)[0]
        return eng.alloc_record(hi - lo, dtype)

    def trace_field(self, Hx: float, Hy: float, Px, Py, wavelength, center=(0.0, 0.0),
                    record=None):
        """ONE field point over a GLOBAL pupil list, record-all, reduce-first exchange -- the
        per-step form of BASELINE config C3 (`bench.py --gpus N`): this rank generates,
        traces, records AND reduces its shard in ONE launch (`ol_trace_generate` with the
        spot epilogue, ABI 8); the 4 KB slot block is the only thing that crosses xGMI.
        `record`: a block from `alloc_field_record` (a step loop reuses one; default: a fresh
        allocation per call).  Returns the record of the local shard and the whole-job
        statistics (RMS / geometric radius about `center`, e.g. the chief-ray hit; the centroid
        is absolute)."""
        t = self.tracer
        if t.table.polarization is not None or t.table.uses_polarization:
            raise ValueError("trace_field: the spot epilogue needs an unpolarised system")
        self._no_reference_newton("trace_field")
        px, py = t._dev(Px), t._dev(Py)
        lo, hi = shard_bounds(px.numel(), self.world, self.rank)
        wl, _ = t._wavelength_index(wavelength)
        hx, hy = float(Hx), float(Hy)
        eng = t.engine
        slots = eng.alloc_spot_slots()
        res = None
        if hi > lo:
            res = eng.trace_generate(px[lo:hi].contiguous(), py[lo:hi].contiguous(), wl,
                                     field=(hx, hy), vig=t._vig_scalar(hx, hy),
                                     record=True if record is None else record,
                                     spot=(slots, float(center[0]), float(center[1])))
        mom = allreduce_spot7(eng.reduce_spot_slots(slots), self.group)
        return {"result": res, "lo": lo, "hi": hi, "n_total": int(px.numel()),
                "spot": spot7_statistics(mom.cpu(), center)}
