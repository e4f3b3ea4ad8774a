#!/usr/bin/env python
"""One arm of an interleaved A/B: HIP-event time of the dominant kernel of a bench
configuration, nothing else (no baselines, no bandwidth probes) -- a few seconds per run.
The library under test is whatever OPTILAND_HIP_LIBRARY names (default: the product).

    python tools/ab_kernel.py --workload zernike --dtype f64 --mode opd [--rays 1e7]
prints:  kernel_ms=<mean of the timed launches> min=<...>
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="double_gauss")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--mode", default="record", choices=("record", "last", "spot", "opd", "gen"))
ap.add_argument("--rays", type=float, default=1e7)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--sustained", action="store_true",
                help="queue every launch back to back (no host sync in between) and report the "
                     "steady state: the part's power management needs ~100 ms of sustained "
                     "load to settle (profiles/r04_clock_transient.txt); use with "
                     "--warmup 150 --steps 60")
a = ap.parse_args()
if a.mode == "opd":
    a.dtype = "f64"
dev = torch.device("cuda", 0)
table, hy, _desc, wavelength = bench.load_workload(a.workload)
wl = table.wavelength_index(wavelength)
hip = HipSystem(table, dev)
dtype = torch.float32 if a.dtype == "f32" else torch.float64
n = int(a.rays)
pol = table.uses_polarization
px, py = bench.make_pupil(n, dtype, 1234, dev)
if a.mode in ("record", "last"):
    record = hip.alloc_record(n, dtype) if a.mode == "record" else None
    rays = bench.make_rays(hip, n, dtype, hy, 1234, dev,
                           out=hip.row0_planes(record, n) if record is not None else None)
    scratch = [torch.empty_like(t) for t in rays] if a.mode == "last" else None
    prt = torch.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype, device=dev) \
        if pol else None

    def step():
        src = rays
        if scratch is not None:
            for d, s_ in zip(scratch, rays):
                d.copy_(s_)
            src = scratch
        e0.record()
        hip.trace(src, wl, record=record if record is not None else False, prt=prt,
                  check_status=False, prt_identity=pol)
        e1.record()
elif a.mode == "gen":  # ol_trace_generate: generation fused into the record-all kernel
    record = hip.alloc_record(n, dtype)
    prt = torch.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype, device=dev) \
        if pol else None

    def step():
        e0.record()
        hip.trace_generate(px, py, wl, field=(0.0, hy), record=record, prt=prt,
                           defer_status=True)
        e1.record()
elif a.mode == "spot":
    mom = torch.zeros(7, dtype=torch.float64, device=dev)

    def step():
        e0.record()
        hip.trace_spot(px, py, wl, field=(0.0, hy), out=mom, check_status=False)
        e1.record()
else:
    from optiland_amd.tracer import HipRayTracer
    from optiland_amd.wavefront import Wavefront
    wf = Wavefront(HipRayTracer(table, dev, dtype=torch.float64, engine=hip), (0.0, hy),
                   wavelength, num_rays=3)
    params = wf.chief_reference()[0]
    mom = torch.zeros(12, dtype=torch.float64, device=dev)

    def step():
        e0.record()
        hip.trace_opd(params, px, py, wl, field=(0.0, hy), want_pupil=True, moments=mom,
                      check_status=False)
        e1.record()

e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
if a.sustained:
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
             for _ in range(a.warmup + a.steps)]
    for e0, e1 in pairs:
        step()
    torch.cuda.synchronize(dev)
    ts = [p0.elapsed_time(p1) for p0, p1 in pairs[a.warmup:]]
    head = [p0.elapsed_time(p1) for p0, p1 in pairs[:a.warmup]]
    print(f"kernel_ms={np.mean(ts):.4f} min={np.min(ts):.4f} median={np.median(ts):.4f} "
          f"max={np.max(ts):.4f} first5={np.mean(head[1:6]) if len(head) > 6 else float('nan'):.4f} "
          f"trough={np.max(head) if head else float('nan'):.4f}")
else:
    for _ in range(a.warmup):
        step()
    ts = []
    for _ in range(a.steps):
        step()
        torch.cuda.synchronize(dev)
        ts.append(e0.elapsed_time(e1))
    print(f"kernel_ms={np.mean(ts):.4f} min={np.min(ts):.4f}")
hip.close()
