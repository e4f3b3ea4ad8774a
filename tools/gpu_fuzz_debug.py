import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle
from optiland_amd.engine import HipSystem
from tests.test_gpu_fuzz import random_system
from tests._util import PLANES
seed = 16
table, rays, has_nr = random_system(seed)
n = rays["x"].size
r32 = {k: v.astype(np.float32).astype(np.float64) for k, v in rays.items()}
want = oracle.trace(table, r32, 0, record=True)["record"]
hip = HipSystem(table, "cuda:0")
for dtype in (torch.float32, torch.float64):
    planes = [torch.tensor(r32[k], dtype=dtype, device="cuda:0") for k in PLANES[:7]]
    planes.append(torch.zeros_like(planes[0]))
    got = hip.trace(planes, 0, record=True).record[:, :, :n].double().cpu().numpy()
    d = (np.isnan(got) != np.isnan(want)).any(1)
    js = np.flatnonzero(d.any(0))
    print(dtype, "rays with differing NaN:", js[:10])
    for j in js[:3]:
        s = int(np.flatnonzero(d[:, j])[0])
        print("  ray", j, "first differing surface", s, "kind", int(table.surfaces["geom_kind"][s]),
              "inter", int(table.surfaces["interaction"][s]), "R", table.surfaces["radius"][s], "k", table.surfaces["conic"][s],
              "n1,n2", table.optics[s, 0]["n1"], table.optics[s, 0]["n2"])
        print("   want prev:", want[s - 1, :, j])
        print("   want here:", want[s, :, j])
        print("   got  here:", got[s, :, j])
print(table.surfaces[["geom_kind", "interaction", "radius", "conic", "flags"]])
