#!/usr/bin/env python
"""Round 5: A/B arms INSIDE one process.  Every arm is a library (the product or a
`variant_<name>.so` of tools/build_variants.py) loaded side by side; each gets its own
`HipSystem` on the same table and writes the SAME record block, and the arms alternate launch
sequence by launch sequence -- same box, same clocks, same place in memory, seconds apart.
(`tools/ab_kernel.py` is one process per arm: 5 s of start-up each and the box's state of a
minute later.)

Two regimes per (configuration, arm), both asked for by the round-4 verdict:
  window     the driver's shape -- an idle gap (0.5 s: clocks down), then 5 warm-up + 20 timed
             launches, each fenced by a host synchronisation
  sustained  150 launches queued back to back, the mean of the last 60
Mean HIP-event time of the dominant kernel, ms; rounds alternate the arm order.

    python tools/ab_inproc.py --arms product,arith_r04 --configs dg_f64_gen,zf_f32_gen [--rounds 3]
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

CONFIGS = {  # name -> (workload, dtype, mode)
    "dg_f32_gen": ("double_gauss", "f32", "gen"),
    "dg_f64_gen": ("double_gauss", "f64", "gen"),
    "rc_f32_gen": ("rc_asphere", "f32", "gen"),
    "rc_f64_gen": ("rc_asphere", "f64", "gen"),
    "zf_f32_gen": ("zernike_fresnel", "f32", "gen"),
    "zf_f64_gen": ("zernike_fresnel", "f64", "gen"),
    "dg_f32_spot": ("double_gauss", "f32", "spot"),
    "dg_f64_spot": ("double_gauss", "f64", "spot"),
    "rc_f64_spot": ("rc_asphere", "f64", "spot"),
    "z_f64_spot": ("zernike", "f64", "spot"),
    "dg_opd": ("double_gauss", "f64", "opd"),
    "z_opd": ("zernike", "f64", "opd"),
}


def load_arm(name):
    import ctypes as C
    path = _capi.library_path() if name == "product" else \
        os.path.join(ROOT, "optiland_amd", "lib", f"variant_{name}.so")
    lib = C.CDLL(path)
    _capi.bind(lib, path)
    return lib


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arms", default="product")
    ap.add_argument("--configs", default="dg_f32_gen")
    ap.add_argument("--rays", type=float, default=1e7)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--regimes", default="window,sustained")
    ap.add_argument("--placed", action="store_true", help="write into a placed record block")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    arms = a.arms.split(",")
    libs = {nm: load_arm(nm) for nm in arms}
    n = int(a.rays)
    print(f"# arms {arms}; {n} rays; rounds {a.rounds}; block {'placed' if a.placed else 'plain'}")
    for cfg in a.configs.split(","):
        workload, dt, mode = CONFIGS[cfg]
        table, hy, _desc, wavelength = bench.load_workload(workload)
        wl = table.wavelength_index(wavelength)
        dtype = torch.float32 if dt == "f32" else torch.float64
        eng = {}
        for nm in arms:
            _capi._LIB = libs[nm]          # HipSystem binds whatever `_capi.load()` returns
            eng[nm] = HipSystem(table, dev)
        _capi._LIB = libs[arms[0]]
        px, py = bench.make_pupil(n, dtype, 1234, dev)
        pol = table.uses_polarization
        first = eng[arms[0]]
        record = None
        if mode == "gen":
            record = first.alloc_record_placed(n, dtype)[0] if a.placed else \
                torch.empty((first.num_surfaces, 8, first.record_stride(n, px.element_size())),
                            dtype=dtype, device=dev)
        prt = torch.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype, device=dev) \
            if pol else None
        mom = torch.zeros(12 if mode == "opd" else 7, dtype=torch.float64, device=dev)
        opd_params = None
        if mode == "opd":
            from optiland_amd.tracer import HipRayTracer
            from optiland_amd.wavefront import Wavefront
            wf = Wavefront(HipRayTracer(table, dev, dtype=torch.float64, engine=first), (0.0, hy),
                           wavelength, num_rays=3)
            opd_params = wf.chief_reference()[0]

        def launch(hip):
            if mode == "opd":
                hip.trace_opd(opd_params, px, py, wl, field=(0.0, hy), want_pupil=True,
                              moments=mom, check_status=False)
            elif mode == "gen":
                hip.trace_generate(px, py, wl, field=(0.0, hy), record=record, prt=prt,
                                   zero_status=False, defer_status=True)
            else:
                hip.trace_spot(px, py, wl, field=(0.0, hy), out=mom, check_status=False)

        def window(hip):
            time.sleep(0.5)
            for _ in range(5):
                launch(hip)
            ts = []
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(20):
                e0.record()
                launch(hip)
                e1.record()
                torch.cuda.synchronize(dev)
                ts.append(e0.elapsed_time(e1))
            return ts

        def sustained(hip):
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                  for _ in range(150)]
            for e0, e1 in ev:
                e0.record()
                launch(hip)
                e1.record()
            torch.cuda.synchronize(dev)
            return [p.elapsed_time(q) for p, q in ev[90:]]

        res = {(r, nm): [] for r in a.regimes.split(",") for nm in arms}
        for rnd in range(a.rounds):
            order = arms if rnd % 2 == 0 else arms[::-1]
            for regime in a.regimes.split(","):
                for nm in order:
                    ts = (window if regime == "window" else sustained)(eng[nm])
                    res[(regime, nm)].append((float(np.mean(ts)), float(np.min(ts)),
                                              float(np.max(ts))))
        for regime in a.regimes.split(","):
            base = np.mean([m for m, _lo, _hi in res[(regime, arms[0])]])
            for nm in arms:
                v = res[(regime, nm)]
                mean = np.mean([m for m, _lo, _hi in v])
                print(f"{cfg:12s} {regime:9s} {nm:16s} mean {mean:.4f} ms  rounds "
                      + " ".join(f"{m:.4f}" for m, _lo, _hi in v)
                      + f"  min {min(lo for _m, lo, _hi in v):.4f} max {max(hi for _m, _lo, hi in v):.4f}"
                      + (f"  vs {arms[0]} {100.0 * (mean / base - 1.0):+.1f} %" if nm != arms[0] else ""),
                      flush=True)
        for hip in eng.values():
            hip.close()
        del record, prt, px, py
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
