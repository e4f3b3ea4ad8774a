"""hipGraph capture of the small-trace inner loop.

Optimisation and tolerancing loops in the reference call `Optic.trace_generic` /
`surface_group.trace` thousands of times with a few hundred rays each
(optimization/operand/ray.py -> optic.trace_generic).  At that size the trace is
launch-bound: the kernels take a few microseconds, the host-side launches and
Python glue ~100.  `GraphedTrace` captures the fixed-shape sequence

    zero status word -> ol_generate_rays (with range checks) -> ol_trace

into one hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on the stream
the C ABI launches on), so a replay is ONE hipGraphLaunch.  Inputs and outputs live
in static device buffers: write the normalised coordinates into `.hx/.hy/.px/.py`,
call `replay()`, read the recorded planes through `.result` (same `TraceResult`
views as an eager trace).  Nothing here is a different numerical path -- the graph
replays exactly the kernels of the eager call, bit for bit.
"""

from __future__ import annotations

import torch

from . import _capi


class GraphedTrace:
    def __init__(self, engine, n: int, dtype=torch.float32, wavelength_index: int = 0,
                 record_all: bool = True, vig=(1.0, 1.0), prescale_pupil: bool = False,
                 check_pupil: bool = True):
        """engine: `HipSystem`; n: rays per replay (fixed); vig: launch-uniform
        (1 - vx, 1 - vy); prescale_pupil / check_pupil: trace_generic semantics."""
        if engine.table.uses_polarization or engine.table.polarization is not None:
            raise ValueError("GraphedTrace covers unpolarised systems")
        self.engine, self.n, self.dtype = engine, int(n), dtype
        dev = engine.device
        self.hx, self.hy, self.px, self.py = (torch.zeros(self.n, dtype=dtype, device=dev)
                                              for _ in range(4))
        self.record_all = record_all
        if record_all:
            # one block, replayed into forever: placed where the part writes it fastest when it
            # is big enough for that to matter (engine.alloc_record_placed; small: plain)
            placed = getattr(engine, "alloc_record_placed", None)
            self.record = placed(self.n, dtype)[0] if placed is not None \
                else engine.alloc_record(self.n, dtype)
            self.rays = engine.row0_planes(self.record, self.n)
        else:
            self.record = False
            buf = torch.empty((8, self.n), dtype=dtype, device=dev)
            self.rays = list(buf.unbind(0))
        self._flags = _capi.RAYGEN_CHECK_FIELD
        if check_pupil:
            self._flags |= _capi.RAYGEN_CHECK_PUPIL
        if prescale_pupil:
            self._flags |= _capi.RAYGEN_PRESCALE_PUPIL
        self._vig, self._wl = (float(vig[0]), float(vig[1])), int(wavelength_index)
        self.result = None
        # warm-up on a side stream (torch's capture protocol), then capture
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.result = self._body()

    def _body(self):
        eng = self.engine
        eng.generate_rays(self.hx, self.hy, self.px, self.py, self._vig[0], self._vig[1],
                          out=self.rays, flags=self._flags)  # zeroes the status word first
        return eng.trace(self.rays, self._wl, record=self.record, defer_status=True,
                         zero_status=False)

    def replay(self, check: bool = True):
        """One hipGraphLaunch.  `check=True` reads the status word back (one sync) and
        raises the reference's exceptions; pass False inside a pipelined loop and call
        `check_status()` when convenient."""
        self.graph.replay()
        if check:
            self.check_status()
        return self.result

    def check_status(self):
        self.engine.raise_for_status(int(self.engine._status.item()))
