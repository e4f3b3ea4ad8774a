#!/usr/bin/env python
"""GPU box: wall time per call of the common entry points through the drop-in, for the five
BASELINE systems, small and large bundles -- a search for host-side costs that are out of
proportion to the kernels (how the eager (N,3,3) PRT layout and the host-side random pupil
were found).  fp32 unless stated."""
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

be.set_backend("torch")
be.set_device("cuda")
be.set_precision("float32")
integration.enable()


def wall(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 3)


out = {}
g = torch.Generator(device="cuda").manual_seed(3)
for name in _live.SYSTEMS:
    lens, w = _live.build_system(name)
    row = {}
    for label, n in (("1e3", 1000), ("1e6", 1_000_000)):
        r = torch.rand(n, generator=g, device="cuda").sqrt() * 0.95
        th = 2 * np.pi * torch.rand(n, generator=g, device="cuda")
        px, py = r * th.cos(), r * th.sin()
        hx = torch.zeros(n, device="cuda")
        hy = torch.full((n,), 0.7, device="cuda")
        row[f"trace_generic_scalar_field_{label}"] = wall(lambda: lens.trace_generic(0.0, 0.7, px, py, w))
        row[f"trace_generic_field_planes_{label}"] = wall(lambda: lens.trace_generic(hx, hy, px, py, w))

        def own_rays():
            rays = lens.ray_tracer.ray_generator.generate_rays(hx, hy, px, py, w) \
                if hasattr(lens.ray_tracer, "ray_generator") else None
            return rays
        try:
            rays0 = own_rays()
            if rays0 is not None:
                import copy
                row[f"surface_group_trace_caller_rays_{label}"] = wall(
                    lambda: lens.surfaces.trace(copy.copy(rays0)), reps=6)
        except Exception as exc:  # noqa: BLE001
            row[f"surface_group_trace_caller_rays_{label}"] = repr(exc)[:60]
    row["trace_hexapolar_6_rings"] = wall(lambda: lens.trace(0.0, 0.7, w, 6, "hexapolar"))
    row["trace_hexapolar_400_rings"] = wall(lambda: lens.trace(0.0, 0.7, w, 400, "hexapolar"))
    row["trace_three_fields_64_rings"] = wall(
        lambda: lens.trace(be.array([0.0, 0.0, 0.0]), be.array([0.0, 0.7, 1.0]), w, 64, "hexapolar"))
    comp = lens.ray_tracer.__dict__.get("_hip_companion")
    row["last_path"] = getattr(comp, "last_path", None)
    out[name] = row
    print(name, json.dumps(row))
integration.disable()
be.set_precision("float64")
be.set_device("cpu")
be.set_backend("numpy")
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_dropin_matrix.json"), "w"), indent=1)
