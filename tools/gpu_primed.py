"""Round 6: is a block that has JUST been probed under sustained load "hot"?

bench.py's record block comes out of `alloc_record_placed` -- ~0.25 s of back-to-back store
launches -- right in front of its warm-up steps.  Arms, ROUNDS x each, 1 s idle before every
round: the probe, then 5 + 20 launches of the fp32 DoubleGauss step (the driver's window)
  never    no cap in the window (what ships for the first 32 launches into a block)
  primed   OL_TRACE_FEW_WAVES from the first launch on (engine._HOT_BLOCKS primed by the probe)
and, for reference, the same two WITHOUT a probe in front (1 s idle, then the launches)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from optiland_amd import engine as E, load_system  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402

ROUNDS = int(os.environ.get("ROUNDS", "4"))
dev = torch.device("cuda", 0)
t = load_system("double_gauss")
n, dtype = 10_000_000, torch.float32
g = torch.Generator(device=dev).manual_seed(1)
r = torch.rand(n, generator=g, device=dev).sqrt()
th = 2 * np.pi * torch.rand(n, generator=g, device=dev)
px, py = (r * th.cos()).to(dtype), (r * th.sin()).to(dtype)
hip = HipSystem(t, dev)
keep, _ = hip.alloc_record_placed(n, dtype)      # for the arms without a probe


def window(rec, primed):
    E._HOT_BLOCKS.clear()
    E._HOT_LOOP["after"] = 0 if not primed else 1
    if primed:
        key = (rec.device.index or 0, rec.data_ptr())
        E._HOT_BLOCKS[key] = [10**6, time.perf_counter()]
    for _ in range(5):
        hip.trace_generate(px, py, 0, field=(0.0, 0.7), record=rec, defer_status=True)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(20)]
    for a, b in evs:
        a.record()
        if primed:
            E._HOT_BLOCKS[(rec.device.index or 0, rec.data_ptr())][1] = time.perf_counter()
        hip.trace_generate(px, py, 0, field=(0.0, 0.7), record=rec, defer_status=True)
        b.record()
    torch.cuda.synchronize()
    return float(np.mean([a.elapsed_time(b) for a, b in evs]))


for probe in (True, False):
    for primed in (False, True, False, True):
        out = []
        for _ in range(ROUNDS):
            torch.cuda.synchronize()
            time.sleep(1.0)
            if probe:
                rec, info = hip.alloc_record_placed(n, dtype)
                if not info.get("placed"):
                    continue
            else:
                rec = keep
            out.append(window(rec, primed))
            if probe:
                del rec
        print(f"probe {'yes' if probe else 'no '}  {'primed' if primed else 'never '}  "
              f"window mean {np.mean(out):.4f} ms  rounds " + " ".join(f"{v:.4f}" for v in out),
              flush=True)
