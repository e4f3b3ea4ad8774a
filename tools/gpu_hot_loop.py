"""Round 6: when does the two-workgroups-per-CU cap start to pay in a loop?

The fp32 DoubleGauss record-all launch (bench.py's headline step) into a PLACED block, ROUNDS x
(idle gap, then LAUNCHES launches back to back), HIP events around every launch, for each arm:
  never     no cap (rounds 1-5 default for a placed block)
  always    OL_TUNE_RECORD_WG_CAP = 2 from the first launch on
  after N   the engine's hot-loop policy (engine._HOT_LOOP): OL_TRACE_FEW_WAVES from the N-th
            launch in a row on
Prints the mean kernel time per range of launch numbers: where `always` is slower than `never`
the cap must not be on yet, where it is faster it should be."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from optiland_amd import engine as E, load_system  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402

LAUNCHES = int(os.environ.get("LAUNCHES", "400"))
ROUNDS = int(os.environ.get("ROUNDS", "3"))
dev = torch.device("cuda", 0)
t = load_system("double_gauss")
n, dtype = 10_000_000, torch.float32
g = torch.Generator(device=dev).manual_seed(1)
r = torch.rand(n, generator=g, device=dev).sqrt()
th = 2 * np.pi * torch.rand(n, generator=g, device=dev)
px, py = (r * th.cos()).to(dtype), (r * th.sin()).to(dtype)
hip = HipSystem(t, dev)
rec, info = hip.alloc_record_placed(n, dtype)
print("placed:", info.get("placed"), "GB/s", info.get("probe_best_GBps"), flush=True)
RANGES = [(0, 10), (10, 25), (25, 50), (50, 100), (100, 200), (200, 400)]


def arm(name, after, forced):
    E._HOT_LOOP["after"] = after
    hip.lib.ol_set_tuning(3, 2 if forced else 0)
    rows = []
    for _ in range(ROUNDS):
        torch.cuda.synchronize()
        time.sleep(1.0)                      # the part goes idle
        E._HOT_BLOCKS.clear()
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(LAUNCHES)]
        for a, b in evs:
            a.record()
            hip.trace_generate(px, py, 0, field=(0.0, 0.7), record=rec, defer_status=True)
            b.record()
        torch.cuda.synchronize()
        rows.append([a.elapsed_time(b) for a, b in evs])
    hip.lib.ol_set_tuning(3, 0)
    m = np.mean(np.array(rows), axis=0)
    cells = " ".join(f"[{lo:3d},{hi:3d}) {m[lo:min(hi, LAUNCHES)].mean():.4f}" for lo, hi in RANGES
                     if lo < LAUNCHES)
    print(f"{name:10s} {cells}   all {m.mean():.4f} ms", flush=True)


for rep in range(2):
    arm("never", 0, False)
    arm("always", 0, True)
    for N in (16, 32, 64, 128):
        arm(f"after {N}", N, False)
