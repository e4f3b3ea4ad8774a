"""Where does the Zernike + Fresnel kernel spend its time?  Strip features one by one."""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from optiland_amd import load_system, system as S
from optiland_amd.engine import HipSystem

dev = torch.device("cuda", 0)
base = load_system("zernike_fresnel_fringe")
n = 10_000_000
dtype = torch.float32


def variant(name, strip_pol=False, strip_zern=False):
    t = copy.deepcopy(base)
    if strip_pol:
        t.surfaces["coating_kind"] = 0
        t.polarization = None
    if strip_zern:
        for s in t.surfaces:
            if s["geom_kind"] == S.GEOM_ZERNIKE:
                s["geom_kind"] = S.GEOM_STANDARD
                s["max_iter"] = 0
    return name, t


for name, t in (variant("full"), variant("no polarisation", strip_pol=True),
                variant("zernike -> conic", strip_zern=True),
                variant("neither", strip_pol=True, strip_zern=True)):
    hip = HipSystem(t, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    r = torch.rand(n, generator=g, device=dev).sqrt() * 0.9
    th = 2 * np.pi * torch.rand(n, generator=g, device=dev)
    px, py = (r * th.cos()).to(dtype), (r * th.sin()).to(dtype)
    rec = hip.alloc_record(n, dtype)
    rays = hip.row0_planes(rec, n)
    hip.generate_rays(0.0, 1.0, px, py, out=rays)
    pol = t.uses_polarization
    prt = torch.empty((9, n), dtype=dtype, device=dev) if pol else None
    for mode in ("record", "last"):
        src = rays if mode == "record" else [x.clone() for x in rays]
        kw = dict(record=rec if mode == "record" else False, prt=prt, prt_identity=pol,
                  check_status=False)
        for _ in range(3):
            hip.trace(src, 0, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.trace(src, 0, **kw)
        e1.record()
        torch.cuda.synchronize()
        print(f"{name:20s} {mode:7s} {e0.elapsed_time(e1) / 20:.4f} ms")
    hip.close()
