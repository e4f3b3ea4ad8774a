"""Fused spot kernel vs the un-fused HIP pipeline, same process, CUDA events."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from optiland_amd import load_system
from optiland_amd.engine import HipSystem

dev = torch.device("cuda", 0)
table = load_system("double_gauss")
hip = HipSystem(table, dev)
wl = table.wavelength_index(0.5876)
for dtype in (torch.float32, torch.float64):
    n = 10_000_000
    g = torch.Generator(device=dev).manual_seed(1)
    r = torch.rand(n, generator=g, device=dev).sqrt()
    th = 2 * np.pi * torch.rand(n, generator=g, device=dev)
    px, py = (r * th.cos()).to(dtype), (r * th.sin()).to(dtype)
    hx = torch.zeros(n, dtype=dtype, device=dev)
    hy = torch.full((n,), 0.7, dtype=dtype, device=dev)
    buf = torch.empty((8, n), dtype=dtype, device=dev)
    planes = [buf[k] for k in range(8)]
    mom = torch.zeros(7, dtype=torch.float64, device=dev)

    def unfused():
        hip.generate_rays(hx, hy, px, py, out=planes)
        planes[7].zero_()
        hip.trace(planes, wl, record=False, check_status=False)
        m = hip.spot_moments(planes[0], planes[1], planes[6])
        r2 = hip.spot_max_r2(planes[0], planes[1], planes[6], 0.0, 17.0)
        return m, r2

    def fused():
        mom.zero_()
        return hip.trace_spot(px, py, wl, field=(0.0, 0.7), center=(0.0, 17.0), out=mom,
                              check_status=False)

    for name, fn in (("unfused", unfused), ("fused", fused)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        print(f"{dtype} {name}: {e0.elapsed_time(e1) / 20:.4f} ms per spot of {n} rays")
    a, b = unfused(), fused()
    print("  count", float(a[0][0]), float(b[0]), " max r2", float(a[1][0]), float(b[6]))
