"""GPU box: what `ol_wavefront_fit` says about the wavefront points of a lens without power
(round 5 triage; python tools/gpu_fit_probe.py NAME ...: tables under tests/golden/)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optiland_amd import tracer as tr  # noqa: E402
from optiland_amd.system import SystemTable  # noqa: E402
from optiland_amd.wavefront import OPD  # noqa: E402

for name in sys.argv[1:]:
    table = SystemTable.load(os.path.join(ROOT, "tests", "golden", name + ".json"))
    t = tr.HipRayTracer(table, "cuda:0", dtype=torch.float64)
    w = float(table.wavelengths[0])
    rays = t.trace(0.0, 0.7, w, 5, "hexapolar")
    d = OPD(t, (0.0, 0.7), w, num_rays=5).distribution
    px, py = t._dev(d.x).contiguous(), t._dev(d.y).contiguous()
    rg = table.raygen
    params = dict(n_image=rg["n_image"], wavelength_um=w, ux=0.0, uy=0.0, half_epd=rg["EPD"] / 2.0)
    r8 = [v.contiguous() for v in (rays.x, rays.y, rays.z, rays.L, rays.M, rays.N, rays.opd, rays.i)]
    ref = t.engine.wavefront_fit("best_fit", params, r8, px, py, flavour="numpy")
    host = ref.cpu()
    bits = int(host[-1:].view(torch.int32)[0])
    pts = torch.stack(r8[:3], 1) - (r8[6] / rg["n_image"])[:, None] * torch.stack(r8[3:6], 1)
    ok = torch.isfinite(pts).all(dim=1) & (r8[7] != 0)
    p = pts[ok].cpu().numpy()
    print(name, "n valid", int(ok.sum()), "bits", bits, "centre", host[:3].numpy(), "radius", float(host[3]),
          "| pts mean", p.mean(0), "std", p.std(0))
    try:
        o = OPD(t, (0.0, 0.7), w, num_rays=5, strategy="best_fit_sphere")
        print("   OPD best_fit_sphere: radius", o.data.radius, "max|opd|", float(o.data.opd.abs().max()))
    except Exception as e:  # noqa: BLE001
        print("   OPD best_fit_sphere raised", type(e).__name__, e)
    t.engine.close()
