#!/usr/bin/env python
"""Benchmark of the fused sequential ray-trace hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step = one pass of the hot path over one batch of synthetic rays already
resident in HBM: the whole-system trace of `--rays` rays (default 1e7, fp32)
through the double Gauss (BASELINE.json configs[1]) in record-all mode, i.e. what
`Optic.trace()` observably produces (8 recorded planes for each of the S+1
surfaces).  For N > 1 every rank traces its own `--rays` rays (weak scaling) and
the step ends with the image-plane exchange: per-rank spot moments reduced with
one RCCL all-reduce (default) or the raw all-gather of image-plane hits
(`--exchange gather`).

Rank 0 prints ONE JSON line (metric = ray-surface intersections/s, whole job).
`roofline` refers to the trace kernel in the mode that was run; `cpu_baseline`
times the CPU oracle (oracle/, a C port of the reference's algorithm) on a
bounded sample of the same workload -- the reference itself is pure Python and
does not exist on the GPU box.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MODE_NAMES = {"record": "record-all", "last": "record-last",
              "spot": "fused generate+trace+spot-reduce (no ray planes)"}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (system json, Hy, description, wavelength in um)
    "double_gauss": ("double_gauss", 0.7, "DoubleGauss", 0.5876),
    "cooke": ("cooke_generic", 1.0, "CookeTriplet", 0.55),
    "rc_asphere": ("rc_asphere", 1.0, "RC + even-asphere corrector (Newton-Raphson)", 0.55),
    "zernike_fresnel": ("zernike_fresnel_fringe", 1.0, "Zernike freeform + Fresnel/polarized",
                        0.55),
}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=float, default=1e7, help="rays per GPU per step")
    ap.add_argument("--dtype", choices=("f32", "f64"), default="f32")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="double_gauss")
    ap.add_argument("--mode", choices=("record", "last", "spot"), default="record",
                    help="record: all surfaces (drop-in semantics); last: image plane only; "
                         "spot: fused generate -> trace -> reduce kernel (ol_trace_spot), "
                         "no ray planes at all")
    ap.add_argument("--exchange", choices=("reduce", "gather", "none"), default="reduce",
                    help="image-plane exchange when --gpus > 1")
    ap.add_argument("--object-row", choices=("alias", "copy"), default="alias",
                    help="record mode: generate the rays straight into row 0 of the record "
                         "block (zero-copy object row) or keep separate ray planes")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the image-plane exchange even with one rank (RCCL path check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def init_dist(n_gpus):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
    if n_gpus != world:
        if rank == 0:
            print(f"warning: --gpus {n_gpus} but WORLD_SIZE={world}; using {world}",
                  file=sys.stderr)
    return rank, local, world


def make_pupil(n, dtype, seed, device):
    """Seeded uniform-disc pupil sampling on device."""
    g = torch.Generator(device=device).manual_seed(seed)
    r = torch.rand(n, generator=g, device=device, dtype=torch.float32).sqrt()
    th = 2 * np.pi * torch.rand(n, generator=g, device=device, dtype=torch.float32)
    return (r * th.cos()).to(dtype), (r * th.sin()).to(dtype)


def make_rays(hip, n, dtype, hy, seed, device, out=None):
    """Pupil sampling -> rays via ol_generate_rays.
    `out`: 8 preallocated planes (row 0 of the record block) to generate into."""
    px, py = make_pupil(n, dtype, seed, device)
    hx = torch.zeros(n, dtype=dtype, device=device)
    hyt = torch.full((n,), hy, dtype=dtype, device=device)
    if out is not None:
        hip.generate_rays(hx, hyt, px, py, out=out)
        out[7].zero_()
        return list(out)
    planes = hip.generate_rays(hx, hyt, px, py)
    rays = [p.contiguous().clone() for p in planes]
    rays.append(torch.zeros(n, dtype=dtype, device=device))
    return rays


def cpu_baseline(table, hy, mode, budget_s, wl=0, threads=None):
    """Time the CPU oracle (C port of the reference's algorithm) on a bounded sample,
    same mode: first on ONE thread, then on `threads` host threads (contiguous ray
    chunks, one oracle call per thread; the C code releases the GIL under ctypes).  The
    multi-thread figure is the reported `value` -- the reference's own NumPy path is
    single-threaded, so this is the more demanding CPU baseline."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    if threads is None:
        threads = int(os.environ.get("OL_CPU_THREADS", min(32, os.cpu_count() or 1)))
    n = 1_000_000
    rng = np.random.default_rng(0)
    r, th = np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    rays = oracle.generate_rays(table.raygen, np.zeros(n), np.full(n, hy),
                                r * np.cos(th), r * np.sin(th))
    pol = table.uses_polarization
    S = table.num_traced
    rec = mode == "record"
    oracle.trace(table, {k: v[:1000] for k, v in rays.items()}, wl, record=rec,
                 polarized=pol)  # warm (build + page in)

    def timed(fn, budget):
        reps, total = 0, 0.0
        while total < budget and reps < 64:
            t0 = time.perf_counter()
            fn()
            total += time.perf_counter() - t0
            reps += 1
        return reps, total

    # outputs are preallocated and reused: the harness times the trace, not the page
    # faults of a fresh 830 MB record block per repetition
    rows = table.num_surfaces
    rec_full = np.zeros((rows, 8, n)) if rec else None
    oracle.trace(table, rays, wl, record=rec, polarized=pol, record_out=rec_full)  # first touch
    r1, t1 = timed(lambda: oracle.trace(table, rays, wl, record=rec, polarized=pol,
                                        record_out=rec_full),
                   budget_s / 3 if threads > 1 else budget_s)
    single = n * S * r1 / t1
    if threads <= 1:
        value, reps, total = single, r1, t1
    else:
        bounds = np.linspace(0, n, threads + 1).astype(int)
        del rec_full
        chunks = [{k: v[lo:hi] for k, v in rays.items()} for lo, hi in zip(bounds[:-1], bounds[1:])]
        outs = [np.zeros((rows, 8, hi - lo)) if rec else None
                for lo, hi in zip(bounds[:-1], bounds[1:])]
        pool = ThreadPoolExecutor(max_workers=threads)

        def par():
            list(pool.map(lambda a: oracle.trace(table, a[0], wl, record=rec, polarized=pol,
                                                 record_out=a[1]), zip(chunks, outs)))

        par()  # warm the pool
        reps, total = timed(par, 2 * budget_s / 3)
        pool.shutdown()
        value = n * S * reps / total
    return {
        "value": value,
        "unit": "ray-surfaces/s",
        "cores": threads,
        "kind": "port",
        "single_thread_value": single,
        "sample": f"{reps} x {n} rays x {S} surfaces on {threads} threads ({total:.1f} s) after "
                  f"{r1} x {n} rays on 1 thread ({t1:.1f} s); fp64, oracle/trace_oracle.c "
                  f"(gcc -O2), mode={mode}; {os.cpu_count()} logical host cores available",
    }


def load_traffic(workload, dtype, mode):
    """HBM bytes per launch from the committed rocprofv3 PMC summary, if any."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        return doc.get(f"{workload}:{dtype}:{mode}")
    except (OSError, ValueError):
        return None


def main():
    args = parse_args()
    rank, local, world = init_dist(args.gpus)
    device = torch.device("cuda", local)
    from optiland_amd import load_system
    from optiland_amd.engine import HipSystem

    sys_name, hy, desc, wavelength = WORKLOADS[args.workload]
    table = load_system(sys_name)
    wl = table.wavelength_index(wavelength)
    hip = HipSystem(table, device)
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    b = 4 if args.dtype == "f32" else 8
    n = int(args.rays)
    S = table.num_traced
    pol = table.uses_polarization

    record = hip.alloc_record(n, dtype) if args.mode == "record" else None
    alias = record is not None and args.object_row == "alias"
    spot = args.mode == "spot"
    if spot:
        if pol:
            raise SystemExit("--mode spot needs an unpolarised workload")
        px, py = make_pupil(n, dtype, 1234 + rank, device)
        rays = []
        mom = [torch.zeros(7, dtype=torch.float64, device=device) for _ in range(2)]
    else:
        rays = make_rays(hip, n, dtype, hy, seed=1234 + rank, device=device,
                         out=hip.row0_planes(record, n) if alias else None)
    prt = None
    if pol:  # write-only: every step starts a fresh PRT from the identity in-kernel
        prt = torch.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype, device=device)
    scratch = [torch.empty_like(t) for t in rays] if args.mode == "last" else None

    import torch.distributed as dist
    have_pg = dist.is_available() and dist.is_initialized()
    exchange = args.exchange if (world > 1 or (args.force_exchange and have_pg)) else "none"
    pending = [None, None]  # in-flight all-gathers (double-buffered)
    if exchange == "gather":
        # two buffer pairs: the all-gather of step k runs on RCCL's stream while step
        # k+1 traces; a buffer is only reused after its collective has completed
        gather_buf = [torch.empty((world, 3, n), dtype=dtype, device=device) for _ in range(2)]
        hits = [torch.empty((3, n), dtype=dtype, device=device) for _ in range(2)]
    if exchange == "reduce":
        # spot moments come out of the trace launch itself (epilogue of the same kernel,
        # ol_trace_ex): slotted partial sums -> 7 doubles -> one small all-reduce
        # Per step: zero 4 KB, trace (+ epilogue), ONE async all-gather of the 4 KB slot
        # block; the slots of all ranks are folded (sum / max) only when the statistics
        # are read -- no per-step reduction launches on the critical stream.
        slots = [hip.alloc_spot_slots() for _ in range(2)]
        all_slots = [torch.zeros((world,) + tuple(slots[0].shape), dtype=torch.float64,
                                 device=device) for _ in range(2)]
    step_no = [0]

    def spot_step(ev0=None, ev1=None):
        """Fused pipeline: pupil planes in, seven doubles out (+ the all-reduce of
        those when sharded -- the only exchange this mode ever needs)."""
        k = step_no[0] & 1
        step_no[0] += 1
        if pending[k] is not None:
            for w in pending[k]:
                w.wait()
            pending[k] = None
        mom[k].zero_()
        if ev0 is not None:
            ev0.record()
        hip.trace_spot(px, py, wl, field=(0.0, hy), out=mom[k], check_status=False)
        if ev1 is not None:
            ev1.record()
        if exchange != "none":
            pending[k] = (dist.all_reduce(mom[k][:6], async_op=True),
                          dist.all_reduce(mom[k][6:], op=dist.ReduceOp.MAX, async_op=True))
        return mom[k]

    def step(ev0=None, ev1=None):
        if spot:
            return spot_step(ev0, ev1)
        if args.mode == "record":
            src = rays
        else:  # last-surface mode mutates the rays in place: refresh from the source
            for d, s_ in zip(scratch, rays):
                d.copy_(s_)
            src = scratch
        k = step_no[0] & 1
        step_no[0] += 1
        if exchange != "none" and pending[k] is not None:
            for w in (pending[k] if isinstance(pending[k], tuple) else (pending[k],)):
                w.wait()  # stream-level wait: this buffer pair is free again
            pending[k] = None
        spot_arg = None
        if exchange == "reduce":
            slots[k].zero_()
            if not pol:  # polarised traces reduce after the launch (see below)
                spot_arg = (slots[k], 0.0, 0.0)
        if ev0 is not None:
            ev0.record()
        res = hip.trace(src, wl, record=record if record is not None else False, prt=prt,
                        check_status=False, prt_identity=pol, spot=spot_arg)
        if ev1 is not None:
            ev1.record()
        if exchange != "none":
            if exchange == "reduce":
                if pol:
                    # the epilogue is for unpolarised traces: reduce the image-plane planes
                    # with the stand-alone kernel into slot 0 of the same block
                    xi, yi, ii = ((res.row(res.last, q) for q in (0, 1, 6))
                                  if args.mode == "record" else (src[0], src[1], src[6]))
                    hip.spot_moments(xi, yi, ii, out=slots[k].view(-1)[:6])
                pending[k] = dist.all_gather_into_tensor(all_slots[k].view(-1),
                                                         slots[k].view(-1), async_op=True)
            else:
                if args.mode == "record":
                    x, y, inten = res.row(res.last, 0), res.row(res.last, 1), res.row(res.last, 6)
                else:
                    x, y, inten = src[0], src[1], src[6]
                hits[k][0].copy_(x)
                hits[k][1].copy_(y)
                hits[k][2].copy_(inten)
                pending[k] = dist.all_gather_into_tensor(gather_buf[k].view(-1),
                                                         hits[k].view(-1), async_op=True)
        return res

    for _ in range(args.warmup):
        step()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
           for _ in range(args.steps)]
    torch.cuda.synchronize(device)
    if have_pg:
        dist.barrier()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(*evs[k])
    for w in pending:
        for ww in (w if isinstance(w, tuple) else (w,)):
            if ww is not None:
                ww.wait()
    torch.cuda.synchronize(device)
    if have_pg:
        dist.barrier()
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0
    if exchange == "reduce" and not spot and args.steps:
        # fold the gathered slots of the last step: whole-job spot statistics
        tot = hip.reduce_spot_slots(all_slots[(step_no[0] - 1) & 1].view(-1, 8)).cpu().numpy()
        assert tot[0] > 0, "no ray reached the image plane"
    if have_pg:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    kern_ms = float(np.mean([a.elapsed_time(bb) for a, bb in evs])) if args.steps else float("nan")

    if rank == 0:
        total_rs = float(n) * S * world * args.steps
        value = total_rs / elapsed
        # algorithmic bytes per launch: record-all reads 8 planes and
        # writes 8 planes for each of the S+1 surfaces; record-last reads 8, writes 8.
        if args.mode == "record":
            alg_bytes = 8 * b * (S + 2) * n
        elif spot:
            alg_bytes = 2 * b * n  # the two pupil planes; everything else stays in registers
        else:
            alg_bytes = 16 * b * n
        if pol:
            alg_bytes += 2 * 9 * b * n  # PRT read-modify-write (SURVEY figure)
        # bytes this launch really has to move: with the zero-copy object row, row 0
        # of the record block IS the input, so only S rows are written
        moved_bytes = alg_bytes - (8 * b * n if alias else 0) - (9 * b * n if pol else 0)
        achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
        traffic = load_traffic(args.workload, args.dtype,
                               args.mode + (":alias" if alias else ""))
        out = {
            "metric": "ray-surface intersections/s",
            "value": value,
            "unit": "ray-surfaces/s",
            "rays_per_s": value / S,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"{desc} ({S} traced surfaces incl. image plane), "
                            f"{n:.3g} rays/GPU {args.dtype}, lambda={wavelength} um, "
                            f"uniform-disc pupil, Hy={hy}, "
                            f"mode={MODE_NAMES[args.mode]}",
                "rays_per_gpu": n,
                "surfaces": S,
                "mode": args.mode,
                "object_row": ("zero-copy (rays generated into record row 0)" if alias
                               else "copied") if args.mode == "record" else None,
                "exchange": exchange,
                "parallelism": f"ray-shard x{world}",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "spot_trace_kernel" if spot else "trace_kernel",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel_ms": kern_ms,
                "algorithmic_bytes": alg_bytes,
                "bytes_per_ray_surface": alg_bytes / (float(n) * S),
                "moved_bytes": moved_bytes,
                "moved_GBps": moved_bytes / (kern_ms * 1e-3) / 1e9,
                "note": ("fused spot kernel: only the two pupil planes touch HBM, the kernel is "
                         "vector-ALU bound by construction -- the HBM fraction is reported for "
                         "the contract, not as its limiter" if spot else
                         "achieved = SURVEY 8d algorithmic bytes / kernel time; moved_bytes is "
                         "what this launch has to transfer (zero-copy object row writes S rows "
                         "instead of S+1) and is what the PMC traffic should equal"),
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(table, hy, "last" if spot else args.mode,
                                               args.cpu_seconds, wl)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    hip.close()
    if have_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
