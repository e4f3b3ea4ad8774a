"""Round 6: the default-on pair form of the C5-type launch against the one-ray form, bit for bit,
on SEEDS shaken systems (tests/test_polz_pair.py: random_c5_table) on the device; every 10th seed
also against the oracle (fp32 contract 1e-4).  Writes gpurun_out/r06_polz_fuzz.txt."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from optiland_amd.engine import HipSystem  # noqa: E402
from tests import test_polz_pair as tp  # noqa: E402

SEEDS = int(os.environ.get("SEEDS", "1500"))
DEV = "cuda:0"
bad, lost, flagged = [], 0, 0
for seed in range(100, 100 + SEEDS):
    table = tp.random_c5_table(seed)
    rng = np.random.default_rng(seed)
    hip = HipSystem(table, DEV)
    try:
        n = int(rng.choice([2, 64, 254, 1000, 4098, 20000]))
        px, py = (t.to(DEV) for t in tp._pupil(n, rng, reach=1.0 if seed % 3 else 1.08))
        kw = dict(field=(float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1))),
                  update_intensity=(tp.POLARISED if seed % 2 else tp.STATE) if seed % 3 != 1 else None)
        one = tp._launch(hip, px, py, 0, 1, **kw)
        pair = tp._launch(hip, px, py, 0, 3, **kw)
        try:
            tp.assert_same_bits(one, pair, f"seed {seed}")
        except AssertionError as exc:
            bad.append(str(exc))
        lost += int(torch.isnan(one[0][-1, 3]).sum())
        flagged += int(one[3] != 0)
    finally:
        hip.close()
lines = [f"{SEEDS} shaken C5 systems (seeds 100 ...), pair form against one ray per lane on the device: "
         f"{len(bad)} with any differing bit (record, PRT, updated intensity, status); "
         f"{lost} rays lost to total internal reflection, {flagged} launches with a status bit -- same on both sides"]
lines += bad[:20]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
open(os.path.join(ROOT, "gpurun_out", "r06_polz_fuzz.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
