#!/usr/bin/env python
"""GPU: what the passes of `ol_wavefront_fit` and `ol_wavefront_opd_fitted` cost, at an analysis
size (256 hexapolar rings = 197 377 rays) and at 1e7 rays, against the bytes they read
(10 planes x 8 B per ray and pass).  HIP events around the whole chain, 50 repetitions after 10.

    python tools/gpu_fit_timing.py > gpurun_out/r04_fit_timing.txt
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optiland_amd import load_system  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402

dev = torch.device("cuda", 0)
eng = HipSystem(load_system("cooke_generic"), dev)
PARAMS = dict(n_image=1.0, wavelength_um=0.55, ux=0.0, uy=0.2, half_epd=5.0)


def bundle(n):
    g = torch.Generator(device=dev).manual_seed(1)
    px = torch.rand(n, generator=g, device=dev, dtype=torch.float64) * 2 - 1
    py = torch.rand(n, generator=g, device=dev, dtype=torch.float64) * 2 - 1
    hit = torch.randn((3, n), generator=g, device=dev, dtype=torch.float64) * 0.01
    hit[2] = 0
    hit[2] += 50.0
    d = torch.stack([hit[0] - 5 * px, hit[1] - 5 * py, hit[2] + 50.0])
    length = torch.linalg.norm(d, dim=0)
    d = d / length
    opd = length.clone()
    inten = torch.rand(n, generator=g, device=dev, dtype=torch.float64) + 0.1
    return [hit[0].contiguous(), hit[1].contiguous(), hit[2].contiguous(), d[0].contiguous(),
            d[1].contiguous(), d[2].contiguous(), opd, inten], px, py


def timed(fn, reps=50, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3  # us


print("# ol_wavefront_fit + ol_wavefront_opd_fitted on", torch.cuda.get_device_name(0))
from optiland_amd import _capi  # noqa: E402

print("# grid cap of a pass (ol_set_tuning(OL_TUNE_FIT_GRID)), 1e7 rays, centroid with trimming")
r8, px, py = bundle(10_000_000)
for cap in (128, 256, 384, 512, 768, 1024, 1536, 2048):
    assert eng.lib.ol_set_tuning(_capi.TUNE_FIT_GRID, cap) == 0
    us = timed(lambda: eng.wavefront_fit("centroid", PARAMS, r8, px, py), reps=20, warm=5)
    print(f"  cap {cap:5d}: {us:8.1f} us  {6 * 10_000_000 * 80 / 1e9 / (us * 1e-6):6.0f} GB/s")
assert eng.lib.ol_set_tuning(_capi.TUNE_FIT_GRID, 0) == 0
del r8, px, py
for n in (197_377, 10_000_000):
    r8, px, py = bundle(n)
    for kind, trim, passes in (("centroid", 3.0, 6), ("centroid", 0.0, 3), ("best_fit", 0.0, 3)):
        for planar in (False, True):
            npass = passes - (1 if (planar and kind == "centroid") else 0)
            us = timed(lambda: eng.wavefront_fit(kind, PARAMS, r8, px, py, trim_std=trim,
                                                 planar=planar))
            gb = npass * n * 80 / 1e9
            print(f"n={n:>9} fit {kind:9s} trim={trim:3.1f} planar={int(planar)}: {us:9.1f} us "
                  f"({npass} passes, {gb / (us * 1e-6):7.0f} GB/s of the planes read)")
    ref = eng.wavefront_fit("centroid", PARAMS, r8, px, py)
    us = timed(lambda: eng.wavefront_opd_fitted(ref, r8[:7], px, py))
    print(f"n={n:>9} opd_fitted (+3 pupil planes): {us:9.1f} us "
          f"({n * (9 + 4) * 8 / 1e9 / (us * 1e-6):7.0f} GB/s)")
eng.close()
