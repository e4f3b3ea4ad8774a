#!/usr/bin/env python
"""In-process A/B of kernel variants (interleaved rounds, HIP-event medians).

    python tools/ab_bench.py --workload double_gauss --dtype f32 --mode record

Variants = rays-per-thread {vector, 1} x compaction {on, off}; every round runs
each variant back to back so clock/DVFS drift hits them equally.
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import WORKLOADS, make_rays  # noqa: E402
from optiland_amd import _capi, load_system  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402
from optiland_amd.system import SystemTable  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="double_gauss")
ap.add_argument("--system-json", default=None)
ap.add_argument("--hy", type=float, default=None)
ap.add_argument("--tol", type=float, default=None, help="override Newton tol on all surfaces")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--mode", default="record")
ap.add_argument("--rays", type=float, default=1e7)
ap.add_argument("--rounds", type=int, default=7)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()

if args.system_json:
    table, hy = SystemTable.load(args.system_json), (args.hy if args.hy is not None else 0.0)
else:
    name, hy, _, _wl = WORKLOADS[args.workload]
    table = load_system(name)
    if args.hy is not None:
        hy = args.hy
if args.tol is not None:
    table.surfaces["tol"] = np.where(table.surfaces["max_iter"] > 0, args.tol, table.surfaces["tol"])
dev = torch.device("cuda", 0)
hip = HipSystem(table, dev)
dtype = torch.float32 if args.dtype == "f32" else torch.float64
n = int(args.rays)
rays = make_rays(hip, n, dtype, hy, 1234, dev)
rec = hip.alloc_record(n, dtype) if args.mode == "record" else None
rec2 = hip.alloc_record(n, dtype) if args.mode == "record" else None
alias_rays = None
if rec2 is not None:
    alias_rays = make_rays(hip, n, dtype, hy, 1234, dev, out=hip.row0_planes(rec2, n))
pol = table.uses_polarization
prt = torch.empty((9, n), dtype=dtype, device=dev) if pol else None
scratch = [torch.empty_like(t) for t in rays]
lib = _capi.load()
variants = [("vec,compact", 2, 1), ("vec,plain", 2, 0), ("rpt1", 1, 0), ("auto", 0, 0)]
if alias_rays is not None:
    variants.append(("auto,row0-alias", 0, 0))
times = {v[0]: [] for v in variants}


def one(rpt, compact, alias=False):
    lib.ol_set_tuning(_capi.TUNE_RAYS_PER_THREAD, rpt)
    lib.ol_set_tuning(_capi.TUNE_COMPACT, compact)
    ms = []
    for _ in range(args.steps):
        src = alias_rays if alias else rays
        if args.mode != "record":
            for d, s_ in zip(scratch, rays):
                d.copy_(s_)
            src = scratch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.trace(src, 0, record=(rec2 if alias else rec) if rec is not None else False, prt=prt,
                  check_status=False, prt_identity=pol)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    return float(np.median(ms))


for name, rpt, comp in variants:
    one(rpt, comp, "alias" in name)  # warm
for _ in range(args.rounds):
    for name, rpt, comp in variants:
        times[name].append(one(rpt, comp, "alias" in name))
S = table.num_traced
print(f"# {args.workload if not args.system_json else args.system_json} {args.dtype} {args.mode} "
      f"n={n:.3g} S={S} tol={args.tol}")
for name in times:
    t = np.array(times[name])
    print(f"{name:12s} median {np.median(t):.4f} ms  min {t.min():.4f}  max {t.max():.4f}  "
          f"-> {n * S / np.median(t) / 1e-3:.4g} rs/s")
