#!/usr/bin/env python
"""Round 5: resident-workgroup cap of the record-all launches (OL_TUNE_RECORD_WG_CAP), the real
kernels, one process.  Arms are values of the knob (1 = never capped, 0 = the library's default
policy with the engine's OL_TRACE_FEW_WAVES hint, 2 ... 8 = forced); they alternate launch
sequence by launch sequence on the SAME block -- a plain one and, with --placed, a placed window.
--mode trace: the rays come from eight planes (`ol_trace`) instead of the generating launch.  Regimes as
tools/ab_inproc.py (window: idle gap, 5 warm-up + 20 fenced launches; sustained: 150 queued,
mean of the last 60).

    python tools/ab_wgcap.py --caps 0,2,3,4 --configs dg_f32_gen,dg_f64_gen [--placed]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from optiland_amd import _capi  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402
from ab_inproc import CONFIGS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--caps", default="0,2,3,4")
    ap.add_argument("--configs", default="dg_f32_gen,dg_f64_gen")
    ap.add_argument("--rays", type=float, default=1e7)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--placed", action="store_true", help="also time a placed record block")
    ap.add_argument("--mode", default="gen", choices=("gen", "trace"))
    ap.add_argument("--regimes", default="window,sustained")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    lib = _capi.load()
    caps = [int(c) for c in a.caps.split(",")]
    n = int(a.rays)
    for cfg in a.configs.split(","):
        workload, dt, mode = CONFIGS[cfg]
        assert mode == "gen"
        table, hy, _desc, wavelength = bench.load_workload(workload)
        wl = table.wavelength_index(wavelength)
        dtype = torch.float32 if dt == "f32" else torch.float64
        hip = HipSystem(table, dev)
        px, py = bench.make_pupil(n, dtype, 1234, dev)
        pol = table.uses_polarization
        prt = torch.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype, device=dev) \
            if pol else None
        blocks = {"plain": torch.empty((hip.num_surfaces, 8, hip.record_stride(n, px.element_size())),
                                       dtype=dtype, device=dev)}
        if a.placed:
            rec, info = hip.alloc_record_placed(n, dtype)
            if info.get("placed"):
                blocks["placed"] = rec

        rays = None
        if a.mode == "trace":
            rays = [torch.empty(n, dtype=dtype, device=dev) for _ in range(8)]
            hip.generate_rays(0.0, hy, px, py, out=rays)

        def launch(record):
            if rays is not None:
                hip.trace(rays, wl, record=record, prt=prt, prt_identity=pol, write_rays=False,
                          check_status=False)
            else:
                hip.trace_generate(px, py, wl, field=(0.0, hy), record=record, prt=prt,
                                   zero_status=False, defer_status=True)

        def window(record):
            time.sleep(0.5)
            for _ in range(5):
                launch(record)
            ts = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(20):
                e0.record()
                launch(record)
                e1.record()
                torch.cuda.synchronize(dev)
                ts.append(e0.elapsed_time(e1))
            return float(np.mean(ts))

        def sustained(record):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range(150)]
            for e0, e1 in ev:
                e0.record()
                launch(record)
                e1.record()
            torch.cuda.synchronize(dev)
            return float(np.mean([p.elapsed_time(q) for p, q in ev[90:]]))

        def spaced(record):
            # a loop with host work between its traces: 5 ms of idle before every launch
            ts = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for k in range(70):
                time.sleep(0.005)
                e0.record()
                launch(record)
                e1.record()
                torch.cuda.synchronize(dev)
                if k >= 10:
                    ts.append(e0.elapsed_time(e1))
            return float(np.mean(ts))

        regimes = [(r, {"window": window, "sustained": sustained, "spaced": spaced}[r])
                   for r in a.regimes.split(",")]
        for where, record in blocks.items():
            res = {(r, c): [] for r, _f in regimes for c in caps}
            for rnd in range(a.rounds):
                order = caps if rnd % 2 == 0 else caps[::-1]
                for regime, fn in regimes:
                    for c in order:
                        assert lib.ol_set_tuning(_capi.TUNE_RECORD_WG_CAP, c) == 0
                        res[(regime, c)].append(fn(record))
            lib.ol_set_tuning(_capi.TUNE_RECORD_WG_CAP, 0)
            for regime, _f in regimes:
                base = np.mean(res[(regime, caps[0])])
                for c in caps:
                    v = res[(regime, c)]
                    print(f"{cfg:12s} {where:6s} {regime:9s} cap {c}  mean {np.mean(v):.4f} ms  rounds "
                          + " ".join(f"{m:.4f}" for m in v)
                          + (f"  vs cap {caps[0]} {100.0 * (np.mean(v) / base - 1.0):+.1f} %"
                             if c != caps[0] else ""), flush=True)
        hip.close()
        del blocks, prt, px, py
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
