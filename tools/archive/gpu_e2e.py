#!/usr/bin/env python
"""End-to-end drop-in throughput: HipRayTracer.trace_generic / trace at 1e7 rays
(host glue + ray generation + trace + recorded views), next to the bare kernel."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optiland_amd import load_system, tracer as tr  # noqa: E402

dev = "cuda:0"
table = load_system("double_gauss")
for dtype in (torch.float32, torch.float64):
    t = tr.HipRayTracer(table, dev, dtype=dtype)
    n = 10_000_000
    g = torch.Generator(device=dev).manual_seed(0)
    r = torch.rand(n, generator=g, device=dev).sqrt()
    th = 2 * np.pi * torch.rand(n, generator=g, device=dev)
    px, py = (r * th.cos()).to(dtype), (r * th.sin()).to(dtype)
    for label, call in (("trace_generic(0, 0.7, Px, Py)", lambda: t.trace_generic(0.0, 0.7, px, py, 0.5876)),):
        for _ in range(3):
            rays = call()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            rays = call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        S = table.num_traced
        print(f"{dtype} {label}: {dt*1e3:.3f} ms/call -> {n*S/dt:.4g} rs/s end to end")
    # small traces: latency
    for m in (100, 10_000):
        pxs, pys = px[:m].contiguous(), py[:m].contiguous()
        for _ in range(5):
            t.trace_generic(0.0, 0.7, pxs, pys, 0.5876)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            t.trace_generic(0.0, 0.7, pxs, pys, 0.5876)
        torch.cuda.synchronize()
        print(f"{dtype} n={m}: {(time.perf_counter()-t0)/50*1e6:.0f} us/call")
    t.engine.close()
