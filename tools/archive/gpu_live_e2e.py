#!/usr/bin/env python
"""GPU box: end-to-end wall time of the LIVE drop-in -- `Optic.trace_generic` /
`Optic.trace` of a reference-built DoubleGauss under `integration.enable()` with device
tensors in, device tensors out -- next to (a) the bare raygen + trace kernels on the same
rays, (b) the reference's stock torch backend on the same GPU, (c) its NumPy backend on
the host.  Writes gpurun_out/live_e2e.json (copied to profiles/ by hand).

    python tools/gpu_live_e2e.py [--rays 1e7] [--precision float32]
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests import _live  # noqa: E402


def wall(fn, reps, sync=True):
    ts = []
    for _ in range(reps):
        if sync:
            torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        if sync:
            torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, float(np.min(ts)) * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=float, default=1e7)
    ap.add_argument("--precision", default="float32")
    ap.add_argument("--system", default="DoubleGauss")
    ap.add_argument("--torch-rays", type=float, default=1e6)
    ap.add_argument("--numpy-rays", type=float, default=1e6)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "live_e2e.json"))
    ap.add_argument("--quick", action="store_true",
                    help="skip the slow comparators (stock torch at full size, NumPy); add the "
                         "small-trace latency table")
    args = ap.parse_args()
    n = int(args.rays)
    be = _live.import_reference()
    from optiland_amd import integration
    dtype = torch.float32 if args.precision == "float32" else torch.float64
    doc = {"system": args.system, "rays": n, "precision": args.precision,
           "device": torch.cuda.get_device_name(0), "host_cores": os.cpu_count()}

    def pupil(m, device, dt):
        g = torch.Generator(device=device).manual_seed(5)
        r = torch.rand(m, generator=g, device=device, dtype=torch.float32).sqrt()
        th = 2 * np.pi * torch.rand(m, generator=g, device=device, dtype=torch.float32)
        return (r * th.cos()).to(dt), (r * th.sin()).to(dt)

    # ---------------------------------------------------------------- the drop-in
    be.set_backend("torch")
    be.set_device("cuda")
    be.set_precision(args.precision)
    integration.enable()
    lens, w = _live.build_system(args.system)
    S = len(lens.surfaces.surfaces) - 1
    px, py = pupil(n, "cuda", dtype)
    hx = torch.zeros(n, device="cuda", dtype=dtype)
    hy = torch.full((n,), 0.7, device="cuda", dtype=dtype)
    for _ in range(3):
        lens.trace_generic(hx, hy, px, py, w)
    assert lens.ray_tracer._hip_companion.last_path == "hip"
    doc["trace_generic_planes_ms"] = wall(lambda: lens.trace_generic(hx, hy, px, py, w), 15)
    for _ in range(2):
        lens.trace_generic(0.0, 0.7, px, py, w)
    doc["trace_generic_scalar_field_ms"] = wall(lambda: lens.trace_generic(0.0, 0.7, px, py, w), 15)
    # back-to-back calls without a sync in between (the status read-back is the only one)
    t0 = time.perf_counter()
    for _ in range(15):
        lens.trace_generic(0.0, 0.7, px, py, w)
    torch.cuda.synchronize()
    doc["trace_generic_scalar_field_back_to_back_ms"] = (time.perf_counter() - t0) / 15 * 1e3

    # bare kernels on the same engine / same rays (what the bench line times)
    comp = lens.ray_tracer._hip_companion
    eng = comp._hip_engine
    rec = eng.alloc_record(n, dtype)
    row0 = eng.row0_planes(rec, n)
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]

    def bare():
        e[0].record()
        eng.generate_rays(0.0, 0.7, px, py, 1.0, 1.0, out=row0)
        e[1].record()
        eng.trace(row0, 0, record=rec, check_status=False)
        e[2].record()
    for _ in range(3):
        bare()
    gen, trc = [], []
    for _ in range(15):
        bare()
        torch.cuda.synchronize()
        gen.append(e[0].elapsed_time(e[1]))
        trc.append(e[1].elapsed_time(e[2]))
    doc["raygen_kernel_ms"], doc["trace_kernel_ms"] = float(np.median(gen)), float(np.median(trc))
    doc["kernels_ms"] = doc["raygen_kernel_ms"] + doc["trace_kernel_ms"]
    doc["ratio_scalar_field_over_kernels"] = doc["trace_generic_scalar_field_ms"][0] / doc["kernels_ms"]
    doc["ratio_planes_over_kernels"] = doc["trace_generic_planes_ms"][0] / doc["kernels_ms"]
    doc["e2e_ray_surfaces_per_s"] = n * S / (doc["trace_generic_scalar_field_ms"][0] * 1e-3)
    del rec, row0

    # host-side profile of one call (where the non-kernel time goes)
    import cProfile
    import io
    import pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    pr.enable()
    for _ in range(5):
        lens.trace_generic(0.0, 0.7, px, py, w)
    torch.cuda.synchronize()
    pr.disable()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25)
    doc["host_profile_5_calls"] = s.getvalue().splitlines()[:60]

    # small traces (optimisation-loop sizes): per-call wall time through the drop-in
    from optiland_amd import fingerprint as _fp
    lat = {}
    for m in (100, 10_000, 1_000_000):
        pxs, pys = pupil(m, "cuda", dtype)
        for _ in range(5):
            lens.trace_generic(0.0, 0.7, pxs, pys, w)
        lat[str(m)] = wall(lambda: lens.trace_generic(0.0, 0.7, pxs, pys, w), 200 if m < 1e6 else 30)[0]
    doc["dropin_small_trace_ms"] = lat
    doc["token_walk"] = "native (csrc/fptoken.c)" if _fp._NATIVE is not None else "python"
    if _fp._NATIVE is not None:
        _fp.use_native(False)
        pxs, pys = pupil(100, "cuda", dtype)
        for _ in range(5):
            lens.trace_generic(0.0, 0.7, pxs, pys, w)
        doc["dropin_100_rays_python_walk_ms"] = wall(
            lambda: lens.trace_generic(0.0, 0.7, pxs, pys, w), 200)[0]
        _fp.use_native(True)

    # Optic.trace, hexapolar rings (host distribution, cached pupil planes?)
    rings = 1000  # 1 + 3*1000*1001 = 3.0e6 rays
    for _ in range(2):
        lens.trace(0.0, 0.7, w, rings, "hexapolar")
    doc["trace_hexapolar_1000_rings_ms"] = wall(lambda: lens.trace(0.0, 0.7, w, rings, "hexapolar"), 5)
    integration.disable()

    # ------------------------------------------------- stock torch backend, same GPU
    if args.quick:
        lens_q, _ = _live.build_system(args.system)
        pxs, pys = pupil(100, "cuda", dtype)
        with torch.no_grad():
            for _ in range(2):
                lens_q.trace_generic(0.0, 0.7, pxs, pys, w)
            doc["stock_torch_100_rays_ms"] = wall(lambda: lens_q.trace_generic(0.0, 0.7, pxs, pys, w), 10)[0]
        be.set_precision("float64")
        be.set_device("cpu")
        be.set_backend("numpy")
        lens_n, _ = _live.build_system(args.system)
        pn = (np.linspace(-0.5, 0.5, 100), np.linspace(0.5, -0.5, 100))
        lens_n.trace_generic(0.0, 0.7, pn[0], pn[1], w)
        t0 = time.perf_counter()
        for _ in range(20):
            lens_n.trace_generic(0.0, 0.7, pn[0], pn[1], w)
        doc["numpy_100_rays_ms"] = (time.perf_counter() - t0) / 20 * 1e3
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(doc, f, indent=1)
        print(json.dumps({k_: v for k_, v in doc.items() if k_ != "host_profile_5_calls"}, indent=1))
        return
    m = int(args.torch_rays)
    lens_t, _ = _live.build_system(args.system)
    pxt, pyt = pupil(m, "cuda", dtype)
    hxt, hyt = torch.zeros(m, device="cuda", dtype=dtype), torch.full((m,), 0.7, device="cuda", dtype=dtype)
    with torch.no_grad():
        for _ in range(2):
            lens_t.trace_generic(hxt, hyt, pxt, pyt, w)
        t_ms = wall(lambda: lens_t.trace_generic(hxt, hyt, pxt, pyt, w), 5)
    doc["stock_torch_backend"] = {"rays": m, "ms": t_ms, "ray_surfaces_per_s": m * S / (t_ms[0] * 1e-3)}
    try:
        m2 = n
        px2, py2 = pupil(m2, "cuda", dtype)
        hx2, hy2 = torch.zeros(m2, device="cuda", dtype=dtype), torch.full((m2,), 0.7, device="cuda", dtype=dtype)
        with torch.no_grad():
            lens_t.trace_generic(hx2, hy2, px2, py2, w)
            t2 = wall(lambda: lens_t.trace_generic(hx2, hy2, px2, py2, w), 3)
        doc["stock_torch_backend_full"] = {"rays": m2, "ms": t2, "ray_surfaces_per_s": m2 * S / (t2[0] * 1e-3)}
    except Exception as exc:  # noqa: BLE001
        doc["stock_torch_backend_full"] = {"error": repr(exc)[:200]}

    # ------------------------------------------------------- NumPy backend, host
    be.set_precision("float64")
    be.set_device("cpu")
    be.set_backend("numpy")
    k = int(args.numpy_rays)
    lens_n, _ = _live.build_system(args.system)
    rng = np.random.default_rng(5)
    r, th = np.sqrt(rng.random(k)), 2 * np.pi * rng.random(k)
    pn = (r * np.cos(th), r * np.sin(th))
    hn = (np.zeros(k), np.full(k, 0.7))
    lens_n.trace_generic(hn[0][:1000], hn[1][:1000], pn[0][:1000], pn[1][:1000], w)
    t0 = time.perf_counter()
    reps = 0
    while time.perf_counter() - t0 < 8 and reps < 5:
        lens_n.trace_generic(hn[0], hn[1], pn[0], pn[1], w)
        reps += 1
    dt = (time.perf_counter() - t0) / reps
    doc["numpy_backend"] = {"rays": k, "s_per_call": dt, "ray_surfaces_per_s": k * S / dt,
                            "threads": "numpy default (single-threaded elementwise)"}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps({k_: v for k_, v in doc.items() if k_ != "host_profile_5_calls"}, indent=1))
    print("\n".join(doc["host_profile_5_calls"]))


if __name__ == "__main__":
    main()
