#!/usr/bin/env python
"""GPU box: what the polarised `update_intensity` epilogue costs on the drop-in path.
`Optic.trace` (epilogue applied, rays/polarized_rays.py:122-133) against `Optic.trace_generic`
(not applied) on the same 1e7 pupil points of the C5 system (Zernike + Fresnel, elliptical
state), wall clock per call under integration.enable()."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests import _live  # noqa: E402

be = _live.import_reference()
from optiland_amd import integration  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
be.set_backend("torch")
be.set_device("cuda")
out = {"rays": n}


class Points:
    def __init__(self, x, y):
        self.x, self.y = x, y


def wall(fn, reps=12, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 4)


for precision, dt in (("float32", torch.float32), ("float64", torch.float64)):
    be.set_precision(precision)
    integration.enable()
    g = torch.Generator(device="cuda").manual_seed(5)
    r = torch.rand(n, generator=g, device="cuda", dtype=torch.float32).sqrt()
    th = 2 * np.pi * torch.rand(n, generator=g, device="cuda", dtype=torch.float32)
    px, py = (r * th.cos()).to(dt), (r * th.sin()).to(dt)
    lens, w = _live.build_system("ZernikeFresnelPolarized")
    pts = Points(px, py)
    row = {}
    row["trace_generic_ms"] = wall(lambda: lens.trace_generic(0.0, 0.7, px, py, w))
    row["trace_with_update_intensity_ms"] = wall(lambda: lens.trace(0.0, 0.7, w, n, pts))
    row["epilogue_ms"] = round(row["trace_with_update_intensity_ms"] - row["trace_generic_ms"], 4)
    out[precision] = row
    integration.disable()
be.set_precision("float64")
be.set_device("cpu")
be.set_backend("numpy")
print(json.dumps(out, indent=1))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_pol_e2e.json"), "w"), indent=1)
