import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
np.set_printoptions(precision=5, suppress=True, linewidth=150)
from oracle import oracle
from optiland_amd.engine import HipSystem
from optiland_amd.rays import prt_to_complex
from tests.test_gpu_fuzz import random_polarised_system
from tests._util import PLANES
for seed in (14, 23):
    table, rays = random_polarised_system(seed)
    n = rays["x"].size
    hip = HipSystem(table, "cuda:0")
    out = oracle.trace(table, rays, 0, record=True, polarized=True, last=1)
    planes = [torch.tensor(rays[k], dtype=torch.float64, device="cuda:0") for k in PLANES[:7]]
    planes.append(torch.zeros_like(planes[0]))
    prt = torch.empty((9, n), dtype=torch.float64, device="cuda:0")
    res = hip.trace(planes, 0, record=True, prt=prt, prt_identity=True, last=1)
    p = prt_to_complex(prt).cpu().numpy()
    err = np.abs(np.nan_to_num(p) - np.nan_to_num(out["prt"])).max(axis=(1, 2))
    j = int(np.argmax(err))
    print("seed", seed, "surface 1:", table.surfaces[1][["geom_kind", "radius", "conic", "origin"]], table.optics[1, 0])
    print(" worst ray", j, "err", err[j], " n bad", int((err > 1e-8).sum()), "of", n)
    print(" ray in:", [rays[k][j] for k in PLANES[:6]])
    print(" rec hip:", res.record[1, :6, j].cpu().numpy(), "\n rec ora:", out["record"][1, :6, j])
    print(" hip P:\n", p[j].real, "\n oracle P:\n", out["prt"][j].real)
    hip.close()
