"""C5 without polarisation: vector instructions of the Newton path as a function of the
iteration cap and of the sag functor (run under rocprofv3 --pmc SQ_INSTS_VALU; the wrapper
tools/gpu_nr_iters.sh groups the dispatches of each variant, LAUNCHES per variant in the
order printed here)."""
import copy
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from optiland_amd import load_system, system as S  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402

LAUNCHES = 4
dev = torch.device("cuda", 0)
base = load_system("zernike_fresnel_fringe")
base.surfaces["coating_kind"] = 0
base.polarization = None
n = 10_000_000
dtype = torch.float32


def variants():
    for cap in (0, 1, 2, 3, 100):
        t = copy.deepcopy(base)
        t.surfaces["max_iter"] = np.where(t.surfaces["geom_kind"] == S.GEOM_ZERNIKE, cap, 0)
        yield f"zernike max_iter={cap}", t
    # the same conic as an even asphere without coefficients: the Newton framework alone
    t = copy.deepcopy(base)
    for s in t.surfaces:
        if s["geom_kind"] == S.GEOM_ZERNIKE:
            s["geom_kind"], s["n_coeff"] = S.GEOM_EVEN_ASPHERE, 0
    yield "even asphere, no coefficients, max_iter=100", t
    t = copy.deepcopy(base)
    for s in t.surfaces:
        if s["geom_kind"] == S.GEOM_ZERNIKE:
            s["geom_kind"], s["max_iter"] = S.GEOM_STANDARD, 0
    yield "conic (lean kernel)", t


for name, t in variants():
    hip = HipSystem(t, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    r = torch.rand(n, generator=g, device=dev).sqrt() * 0.9
    th = 2 * np.pi * torch.rand(n, generator=g, device=dev)
    px, py = (r * th.cos()).to(dtype), (r * th.sin()).to(dtype)
    rec = hip.alloc_record(n, dtype)
    rays = hip.row0_planes(rec, n)
    hip.generate_rays(0.0, 1.0, px, py, out=rays)
    for _ in range(LAUNCHES):
        hip.trace(rays, 0, record=rec, check_status=False)
    torch.cuda.synchronize()
    print("VARIANT", name, flush=True)
    hip.close()
