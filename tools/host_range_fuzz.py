#!/usr/bin/env python
"""CPU (round 5): partial traces -- the `[first, last]` surface range of `ol_trace`, which the
`SurfaceGroup.trace(rays, skip)` seam and the bridging around unsupported surfaces use -- on
the reference-built lenses under fuzz_tables/ (tools/make_fuzz_tables.py): the rays the oracle
records at surface `first - 1` are traced through a random range by the product's engine class
on the HOST build of the kernel source and by the oracle; record rows, final rays (written back
in place, `write_rays`) and `record_first` inside the range.  fp64; lost rays (the oracle's own
hit off a Newton surface) are left out as in tools/gpu_fuzz_tables.py.

    python tools/host_range_fuzz.py
"""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from optiland_amd.system import SystemTable  # noqa: E402
from tests import _hostmath as hm  # noqa: E402

cls = hm.make_engine_class()
checked, worst, over = 0, 0.0, []
for path in sorted(glob.glob(os.path.join(ROOT, "fuzz_tables", "*.json"))):
    table = SystemTable.load(path)
    if table.polarization is not None or table.uses_polarization:
        continue   # (a PRT carried across a range is the bridging tests' subject)
    seed = int(os.path.basename(path)[5:-5])
    rng = np.random.default_rng(123_000 + seed)
    n, S = 500, table.num_surfaces
    if S < 4:
        continue
    r, th = np.sqrt(rng.random(n)) * 0.9, 2 * np.pi * rng.random(n)
    g = oracle.generate_rays(table.raygen, np.full(n, rng.uniform(-0.5, 0.5)), np.full(n, rng.uniform(-1, 1)),
                             r * np.cos(th), r * np.sin(th))
    g["opd"] = np.zeros(n)
    full = oracle.trace(table, g, 0, record=True)["record"]
    first = int(rng.integers(1, S - 1))
    last = int(rng.integers(first, S))
    rec_first = int(rng.integers(first, last + 1))
    start = {k: full[first - 1, j].copy() for j, k in enumerate(("x", "y", "z", "L", "M", "N", "i", "opd"))}
    ok_in = np.isfinite(full[first - 1, :6]).all(axis=0)
    want = oracle.trace(table, start, 0, record=True, first=first, last=last)
    hip = cls(table, "cpu")
    try:
        rays = [torch.as_tensor(start[k].copy()) for k in ("x", "y", "z", "L", "M", "N", "i", "opd")]
        res = hip.trace(rays, 0, record=True, first=first, last=last, write_rays=True,
                        record_first=rec_first, check_status=False)
        got = res.record[:, :, :n].numpy()
        final = np.stack([t.numpy() for t in rays])
    finally:
        hip.close()
    wrec = want["record"][rec_first - first:]
    wfin = np.stack([want[k] for k in ("x", "y", "z", "L", "M", "N", "i", "opd")])
    assert got.shape == wrec.shape, (path, got.shape, wrec.shape)
    # lost rays: the oracle's hit at a Newton surface of the range is off the surface
    lost = ~ok_in
    for s_i in np.nonzero(table.surfaces["max_iter"] > 0)[0]:
        if not first <= s_i <= last:
            continue
        sf = table.surfaces[s_i]
        Rm, o_ = np.array(sf["rot"]).reshape(3, 3), np.array(sf["origin"])
        loc = Rm @ (want["record"][s_i - first, :3] - o_[:, None])
        for j in np.nonzero(~lost)[0]:
            f_ = oracle.sag(table, int(s_i), float(loc[0, j]), float(loc[1, j])) - loc[2, j]
            if not abs(f_) < 1e-3:
                lost[j] = True
    keep = ~lost
    z = wrec[:, 2][np.isfinite(wrec[:, 2])]
    scale = max(1.0, float(np.abs(z).max()) if z.size else 1.0)
    err = 0.0
    for a, b in ((got[:, :, keep], wrec[:, :, keep]), (final[None][:, :, keep], wfin[None][:, :, keep])):
        assert np.array_equal(np.isnan(a), np.isnan(b)), (path, first, last, "NaN masks")
        for k in range(8):
            s_ = scale if k in (0, 1, 2, 7) else 1.0
            d = np.abs(a[:, k] - b[:, k])
            if d.size and np.isfinite(d).any():
                err = max(err, float(np.nanmax(d)) / s_)
    worst = max(worst, err)
    if err > 1e-6:
        over.append((os.path.basename(path), first, last, rec_first, err))
    checked += 1
print(f"checked {checked} lenses, one random [first, last] range and record_first each; worst margin {worst:.3e}")
print("over 1e-6:", len(over))
for o in over[:20]:
    print("   ", o)
