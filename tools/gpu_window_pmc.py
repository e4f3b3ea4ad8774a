#!/usr/bin/env python
"""(prepared in round 4, NOT yet run: the round's GPU minutes were spent) WHY does the
record-all store pattern write 7.1 TB/s into a window that straddles a boundary of its arena's
backing and 5.8 TB/s into one that does not (profiles/r04_window_scan_*.txt)?

Run under rocprofv3 with ONE counter (or one small set) per pass -- `tools/gpu_window_pmc.sh`
does the passes.  The process finds the fastest and the median window of a 40 GiB arena with
`HipSystem._probe_windows`, then issues K fills (`ol_stream_fill`, the arithmetic-free store
pattern) into the fast window, K into the median one, K into the fast one again; the analysis
(`--analyse DIR`) groups the stream_fill dispatches of counter_collection.csv by that order and
prints counter totals per window kind.  Candidates: L2 write requests and their stalls
(TCC_EA0_WRREQ*, TCC_EA0_WRREQ_STALL*, TCC_TAG_STALL*, TCC_BUBBLE*), 32 B vs 64 B write splits,
address-translation misses (UTCL2-side counters if the build exposes them).
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
K = 12


def run():
    import numpy as np
    import torch

    from optiland_amd import load_system
    from optiland_amd.engine import HipSystem
    import ctypes as C

    dev = torch.device("cuda", 0)
    hip = HipSystem(load_system("double_gauss"), dev)
    n, b, rows = 10_000_000, 4, hip.num_surfaces
    stride = hip.record_stride(n, b)
    need = rows * 8 * stride * b
    arena = torch.empty(40 << 30, dtype=torch.uint8, device=dev)
    times, offs, pad = hip._probe_windows(arena, arena.numel(), need, b, rows * 8)
    best = min(times, key=times.get)
    med_t = float(np.median([times[o] for o in offs]))
    med = min(offs, key=lambda o: abs(times[o] - med_t))
    print(f"# fast window +{best / 2**30:.2f} GiB {need / times[best] / 1e6:.0f} GB/s, "
          f"median window +{med / 2**30:.2f} GiB {need / times[med] / 1e6:.0f} GB/s", flush=True)
    base = arena.data_ptr() + pad
    stream = hip._stream()
    torch.cuda.synchronize()
    # marker dispatches (a tiny fill) separate the three groups in the trace
    for off in (best, med, best):
        hip.lib.ol_stream_fill(C.c_void_p(base), 1 << 20, b, 1, 0, stream)
        for _ in range(K):
            hip.lib.ol_stream_fill(C.c_void_p(base + off), need, b, rows * 8, 0, stream)
        torch.cuda.synchronize()
    hip.close()


def analyse(d):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        rows = [r for r in csv.DictReader(open(f)) if "stream_fill" in r.get("Kernel_Name", "")]
        rows.sort(key=lambda r: int(r.get("Dispatch_Id", 0)))
        by_counter = {}
        for r in rows:
            by_counter.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for name, vals in by_counter.items():
            # the last 3 x (1 marker + K) dispatches of this counter
            tail = vals[-3 * (K + 1):]
            groups = [tail[i * (K + 1) + 1:(i + 1) * (K + 1)] for i in range(3)]
            mean = [sum(g) / max(len(g), 1) for g in groups]
            print(f"{os.path.basename(os.path.dirname(f)):28s} {name:32s} fast {mean[0]:14.1f}  "
                  f"median {mean[1]:14.1f}  fast {mean[2]:14.1f}  ratio median/fast "
                  f"{mean[1] / max((mean[0] + mean[2]) / 2, 1e-9):.3f}")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
        analyse(sys.argv[2])
    else:
        run()
