"""GPU box: the stand-alone analyses on the real engine vs the same host code on the
oracle-backed engine (tests/_fake_engine.py), over every table under fuzz_tables/:
SpotDiagram (fused spot kernel / planes path when polarised), EncircledEnergy, OPD.
fp64; prints the worst relative differences."""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optiland_amd.analysis import EncircledEnergy, SpotDiagram  # noqa: E402
from optiland_amd.system import SystemTable  # noqa: E402
from optiland_amd.tracer import HipRayTracer  # noqa: E402
from optiland_amd.wavefront import OPD  # noqa: E402
from tests._fake_engine import OracleEngine  # noqa: E402

worst = {"spot_rms": 0.0, "spot_geo": 0.0, "spot_centroid": 0.0, "ee": 0.0, "opd": 0.0}
count = {"spot": 0, "ee": 0, "opd": 0, "raised_both": 0}
bad = []


def both(fn, table):
    out = []
    for real in (True, False):
        eng = None if real else OracleEngine(table, "cpu")
        t = HipRayTracer(table, "cuda:0" if real else "cpu", dtype=torch.float64, engine=eng)
        try:
            with np.errstate(all="ignore"):
                out.append(fn(t))
        except (ValueError, NotImplementedError) as e:
            out.append(e)
        finally:
            if real:
                t.engine.close()
    return out


ONLY = {int(a) for a in sys.argv[1:]}   # optional: seeds to look at (triage of flagged lenses)
for path in sorted(glob.glob(os.path.join(ROOT, "fuzz_tables", "*.json"))):
    if ONLY and int(os.path.basename(path)[5:9]) not in ONLY:
        continue
    table = SystemTable.load(path)
    name = os.path.basename(path)

    def spot(t):
        s = SpotDiagram(t, num_rings=5)
        return (np.array(s.rms_spot_radius()), np.array(s.geometric_spot_radius()),
                np.array(s.centroid(), dtype=np.float64))
    a, b = both(spot, table)
    if isinstance(a, Exception) or isinstance(b, Exception):
        assert type(a) is type(b), (name, a, b)
        count["raised_both"] += 1
        continue
    if np.isfinite(b[0]).all() and np.isfinite(b[2]).all():
        scale = max(1.0, float(np.abs(b[2]).max()))
        for k, (u, v) in zip(("spot_rms", "spot_geo", "spot_centroid"), zip(a, b)):
            e = float(np.max(np.abs(u - v)) / scale)
            worst[k] = max(worst[k], e)
            if e > 1e-6:
                bad.append((name, k, e))
        count["spot"] += 1
    if table.polarization is None and not table.uses_polarization:
        def ee(t):
            e = EncircledEnergy(t, num_rays=6, distribution="hexapolar", num_points=32)
            return e.r_step, e.ee
        a, b = both(ee, table)
        if not (isinstance(a, Exception) or isinstance(b, Exception)) and np.isfinite(b[0]).all():
            d = np.abs(a[1] - b[1])[:, 1:]
            # 1e-7: the Newton stop tolerance moves hits (and absorbed energy) by ~1e-8
            frac = float((d > 1e-7 * max(1.0, b[1].max())).mean())
            worst["ee"] = max(worst["ee"], frac)
            if frac > 0.05:
                bad.append((name, "ee", frac))
            count["ee"] += 1

        def opd(t):
            o = OPD(t, (0.0, 0.7), float(table.wavelengths[0]), num_rays=5)
            return o.data.opd.double().cpu().numpy()
        a, b = both(opd, table)
        if not (isinstance(a, Exception) or isinstance(b, Exception)) and np.isfinite(b).all():
            e = float(np.max(np.abs(a - b)) / max(1.0, float(np.abs(b).max())))
            worst["opd"] = max(worst["opd"], e)
            if e > 1e-5:
                bad.append((name, "opd", e))
            count["opd"] += 1
print("compared:", count)
print("worst:", {k: f"{v:.3e}" for k, v in worst.items()})
print("flagged:", len(bad))
for b_ in bad[:20]:
    print("   ", b_)
