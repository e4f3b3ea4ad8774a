#!/usr/bin/env python
"""GPU: `Optic.trace_generic` of the reference's DoubleGauss, 1e7 rays fp32, through the drop-in
with ordinary record blocks and with the placed pool (`integration.enable(placed_records=2)`).
Each call keeps ONE result alive while the next is made (the usual loop).
    python tools/gpu_pool_dropin.py > gpurun_out/r04_pool_dropin.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests import _live  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
be = _live.import_reference()
from optiland_amd import engine as E  # noqa: E402
from optiland_amd import integration  # noqa: E402

be.set_backend("torch")
be.set_device("cuda")
be.set_precision("float32")
g = torch.Generator(device="cuda").manual_seed(5)
r = torch.rand(n, generator=g, device="cuda").sqrt()
th = 2 * np.pi * torch.rand(n, generator=g, device="cuda")
px, py = (r * th.cos()).contiguous(), (r * th.sin()).contiguous()
hx, hy = torch.zeros(n, device="cuda"), torch.full((n,), 0.7, device="cuda")


def loop(lens, w, reps):
    ts, keep = [], None
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rays = lens.trace_generic(0.0, 0.7, px, py, w)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
        keep = rays  # the previous result dies here, the current one lives on
    return np.array(ts) * 1e3, keep


print("#", torch.cuda.get_device_name(0), n, "rays fp32, Optic.trace_generic (scalar field), wall ms per call")
ref = None
for label, placed in (("ordinary blocks", 0), ("placed pool, 2 blocks", 2), ("ordinary blocks", 0),
                      ("placed pool, 2 blocks", 2)):
    integration.enable(placed_records=placed)
    lens, w = _live.build_system("DoubleGauss")
    loop(lens, w, 4)
    t0 = time.perf_counter()
    ts, rays = loop(lens, w, 30)
    pools = [p.info for p in E._RECORD_POOLS.values()]
    x = rays.x.clone()
    if ref is None:
        ref = x
    same = bool(torch.equal(x.nan_to_num(), ref.nan_to_num()))
    print(f"{label:24s}: median {np.median(ts):.4f}  min {ts.min():.4f}  p90 {np.percentile(ts, 90):.4f}"
          f"  same rays as the first arm: {same}  pools: "
          f"{[(p['slots'], [round(v) for v in p.get('window_GBps', [])], round(p.get('probe_median_GBps', 0))) for p in pools]}")
    del rays, lens
    integration.disable()
