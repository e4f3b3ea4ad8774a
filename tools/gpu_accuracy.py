#!/usr/bin/env python
"""Print the measured parity margins of the HIP path on every golden case:
max |got - want| / scale per plane group, fp32 and fp64 (needs a GPU), and the worst
image-plane transverse error relative to the RMS spot radius of the golden bundle.
Also writes gpurun_out/parity_margins.json -- the source of tests/golden/fp32_margins.json
(the per-case fp32 tolerances of tests/test_gpu_parity.py are 4 x these margins)."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optiland_amd.engine import HipSystem  # noqa: E402
from tests._util import golden_cases, load_case  # noqa: E402

GROUPS = {"pos": (0, 1, 2), "dir": (3, 4, 5), "i": (6,), "opd": (7,)}
print(f"{'case':28s} {'dtype':5s} " + " ".join(f"{g:>9s}" for g in GROUPS) + "  img/spot")
DOC = {}
sys.path.insert(0, os.path.join(ROOT, "tests"))
from tests._util import image_plane_error_over_spot  # noqa: E402
for case in golden_cases():
    table, data = load_case(case)
    hip = HipSystem(table, "cuda:0")
    want = data["record"]
    n = want.shape[2]
    for dtype in (torch.float32, torch.float64):
        rays = [torch.tensor(data["rays_in"][k], dtype=dtype, device="cuda:0") for k in range(7)]
        rays.append(torch.zeros(n, dtype=dtype, device="cuda:0"))
        prt = None
        if "prt" in data:
            from optiland_amd.rays import new_prt
            prt = new_prt(n, dtype, "cuda:0", table.needs_complex_prt)
        got = hip.trace(rays, 0, record=True, prt=prt).record[:, :, :n].double().cpu().numpy()
        errs = []
        for g, idx in GROUPS.items():
            w, o = want[:, idx, :], got[:, idx, :]
            fin = np.isfinite(w)
            scale = np.abs(w[fin]).max() if fin.any() else 1.0
            errs.append(np.abs(o[fin] - w[fin]).max() / scale if fin.any() else 0.0)
        tag = 'f32' if dtype == torch.float32 else 'f64'
        img = image_plane_error_over_spot(got, want, data)
        DOC.setdefault(case, {})[tag] = dict(zip(GROUPS, (float(e) for e in errs)), img_over_spot=img)
        print(f"{case:28s} {tag:5s} " + " ".join(f"{e:9.2e}" for e in errs) + f"  {img:9.2e}")
    hip.close()
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "parity_margins.json"), "w") as f:
    json.dump(DOC, f, indent=1, sort_keys=True)
