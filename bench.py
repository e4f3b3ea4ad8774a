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
(`--exchange gather`).  `--config c3` is BASELINE.json configs[2]: fp64, 1e8 rays in
total, strong-scaled over the ranks (1.25e7 per GPU at N = 8).

`python bench.py --gpus N` with N > 1 and no launcher environment starts the N ranks
itself (re-exec under `python -m torch.distributed.run`, rendezvous on 127.0.0.1) and
exits non-zero when fewer than N devices are visible; under the driver's torchrun
launch it just joins the group.

Rank 0 prints ONE JSON line (metric = ray-surface intersections/s, whole job).
`roofline` refers to the trace kernel in the mode that was run: `frac` is on the bytes
the launch really moves (== the PMC traffic), `frac_algorithmic` on SURVEY 8d's figure,
`frac_of_achievable` against a device copy measured in the same run.  `cpu_baseline` is
the reference's own NumPy backend (`kind: "reference"`, staged under oracle/_ref) on a
bounded sample, with the C port of the algorithm (oracle/) beside it; `gpu_baseline` is
the reference's stock torch backend on the same GPU; `dropin` is the deliverable itself --
the same reference call (`Optic.trace_generic`, device tensors in / out) under
`integration.enable()` at the full batch size, wall clock, with its ratio to the kernel.
"""

from __future__ import annotations

import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

MODE_NAMES = {"gen": "record-all, rays generated inside the launch (ol_trace_generate: what "
                     "Optic.trace / trace_generic run for one field point)",
              "record": "record-all from eight resident ray planes (ol_trace)",
              "last": "record-last",
              "spot": "fused generate+trace+spot-reduce (no ray planes)",
              "opd": "fused generate+trace+OPD (ol_trace_opd: 2 pupil planes in, OPD + "
                     "intensity + pupil point out, 12 device moments)"}
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (system json, Hy, description, wavelength in um)
    "double_gauss": ("double_gauss", 0.7, "DoubleGauss", 0.5876),
    "cooke": ("cooke_generic", 1.0, "CookeTriplet", 0.55),
    "rc_asphere": ("rc_asphere", 1.0, "RC + even-asphere corrector (Newton-Raphson)", 0.55),
    "zernike_fresnel": ("zernike_fresnel_fringe", 1.0, "Zernike freeform + Fresnel/polarized",
                        0.55),
    # the same singlet with the coatings stripped and polarisation ignored: the Zernike
    # Newton kernels on the unpolarised paths (fused spot / OPD kernels refuse polarised
    # systems)
    "zernike": ("zernike_fresnel_fringe", 1.0, "Zernike freeform, uncoated, unpolarised", 0.55),
}


def load_workload(name):
    from optiland_amd import load_system
    sys_name, hy, desc, wavelength = WORKLOADS[name]
    table = load_system(sys_name)
    if name == "zernike":
        table.surfaces["coating_kind"] = 0
        table.polarization = None
        table.name = "zernike_uncoated"
    return table, hy, desc, wavelength


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rays", type=float, default=1e7, help="rays per GPU per step")
    ap.add_argument("--dtype", choices=("f32", "f64"), default="f32")
    ap.add_argument("--workload", choices=sorted(WORKLOADS), default="double_gauss")
    ap.add_argument("--mode", choices=("gen", "record", "last", "spot", "opd"), default="gen",
                    help="gen (default): all surfaces recorded, the rays generated inside the "
                         "same launch from resident pupil planes (ol_trace_generate, the "
                         "drop-in's Optic.trace path); record: all surfaces, from eight resident "
                         "ray planes (ol_trace); last: image plane only; "
                         "spot: fused generate -> trace -> reduce kernel (ol_trace_spot), "
                         "no ray planes at all; opd: fused generate -> trace -> OPD kernel "
                         "(ol_trace_opd, fp64)")
    ap.add_argument("--exchange", choices=("reduce", "gather", "none"), default="reduce",
                    help="image-plane exchange when --gpus > 1")
    ap.add_argument("--object-row", choices=("alias", "copy"), default="alias",
                    help="record mode: generate the rays straight into row 0 of the record "
                         "block (zero-copy object row) or keep separate ray planes")
    ap.add_argument("--force-exchange", action="store_true",
                    help="run the image-plane exchange even with one rank (RCCL path check)")
    ap.add_argument("--config", choices=("c2", "c3"), default="c2",
                    help="c2 = BASELINE.json configs[1] (double Gauss, 1e7 rays/GPU fp32, weak "
                         "scaling; the default).  c3 = configs[2]: double Gauss, fp64, "
                         "--total-rays (1e8) rays in total, STRONG-scaled over the ranks")
    ap.add_argument("--total-rays", type=float, default=1e8, help="--config c3: whole-job rays")
    ap.add_argument("--plumbing-check", action="store_true",
                    help="TEST ONLY, not a measurement: run the launch / sharding / exchange / "
                         "JSON plumbing on CPU tensors with gloo and the host build of the "
                         "kernel source (tests/hostmath) standing in for the device library -- "
                         "tests/test_bench_cli.py uses it for the N > 1 path; the line says "
                         "`\"data\": \"plumbing-check\"` and carries no roofline")
    ap.add_argument("--placement", choices=("probe", "plain"), default="probe",
                    help="record block placed in the fastest window of an arena (default) or a "
                         "plain allocation")
    ap.add_argument("--settle", type=int, default=120,
                    help="extra launches AFTER the timed region whose last third is reported as "
                         "roofline.steady_state (0 = skip)")
    ap.add_argument("--traffic", choices=("live", "committed"), default="live",
                    help="roofline.traffic: re-measured in THIS run (two short rocprofv3 --pmc "
                         "passes of this command as child processes, when rocprofv3 is on PATH "
                         "and N = 1), or read from profiles/traffic.json")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ref-baselines", action="store_true",
                    help="skip the legs that time the staged reference package (NumPy backend "
                         "on the host, stock torch backend on the GPU)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--ref-seconds", type=float, default=8.0)
    return ap.parse_args()


class _HostEvent:
    """torch.cuda.Event stand-in of --plumbing-check (wall clock)."""

    def __init__(self, enable_timing=True):
        self.t = 0.0

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


PLUMBING = {"on": False}


def sync(device=None):
    if not PLUMBING["on"]:
        torch.cuda.synchronize(device)


def make_event():
    return _HostEvent() if PLUMBING["on"] else torch.cuda.Event(enable_timing=True)


def self_launch(args):
    """`python bench.py --gpus N` outside a launcher: start the N ranks ourselves.

    Re-executes this script under `python -m torch.distributed.run` (one process per
    GPU, rendezvous on 127.0.0.1) and exits with ITS return code; exits 2 with a clear
    message when fewer than N HIP devices are visible.  Under the driver's own torchrun
    launch (RANK / WORLD_SIZE set) this is a no-op."""
    if args.gpus <= 1 or "RANK" in os.environ or "WORLD_SIZE" in os.environ:
        return
    import socket
    import subprocess
    have = args.gpus if args.plumbing_check else torch.cuda.device_count()
    if have < args.gpus:
        print(f"bench.py: error: --gpus {args.gpus} requested but only {have} HIP device(s) "
              f"are visible on this node; refusing to report a smaller job as n_gpus={args.gpus}",
              file=sys.stderr)
        sys.exit(2)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1",
           f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC (RCCL needs it here)
    sys.exit(subprocess.call(cmd, env=env))


def init_dist(n_gpus, plumbing=False):
    if plumbing:
        world = int(os.environ.get("WORLD_SIZE", "1"))
        rank = int(os.environ.get("RANK", "0"))
        if n_gpus != world:
            sys.exit(2)
        if world > 1 or "RANK" in os.environ:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29511")
            dist.init_process_group("gloo", rank=rank, world_size=world)
        return rank, 0, world
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if n_gpus != world:
        if rank == 0:
            print(f"bench.py: error: --gpus {n_gpus} but the launcher started WORLD_SIZE={world} "
                  f"rank(s); start {n_gpus} ranks (or run `python bench.py --gpus {n_gpus}` "
                  f"without a launcher: it starts them itself)", file=sys.stderr)
        sys.exit(2)
    if local >= torch.cuda.device_count():
        print(f"bench.py: error: rank {rank} (LOCAL_RANK {local}) has no HIP device "
              f"({torch.cuda.device_count()} visible)", file=sys.stderr)
        sys.exit(2)
    torch.cuda.set_device(local)
    if world > 1 or "RANK" in os.environ:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local))
        world = dist.get_world_size()  # the ranks RCCL really brought up
    return rank, local, world


def gen_kernel_label(table, dtype, pol):
    """Which instantiation `ol_trace_generate` picks for this workload (trace_kernel.hip:
    launch_gen_nr): conic-only, unpolarised, fp32, no apodization -> one packed pair of rays per
    lane (unless OL_TRACE_RPT=1 asks for one ray per lane)."""
    conic = bool(np.all(np.asarray(table.surfaces["geom_kind"]) <= 1))
    apod = int((table.raygen or {}).get("apod_kind", 0) or 0) != 0
    if dtype == "f32" and conic and not pol and not apod \
            and os.environ.get("OL_TRACE_RPT", "0") in ("0", "3"):
        return "trace_kernel<float, RPT = 2 (packed pair), RECORD, GEN = uniform>"
    return "trace_kernel<..., RPT = 1, GEN = true>"


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
        "cpu_model": cpu_model(),
        "kind": "port",
        "single_thread_value": single,
        "sample": f"{reps} x {n} rays x {S} surfaces on {threads} threads ({total:.1f} s) after "
                  f"{r1} x {n} rays on 1 thread ({t1:.1f} s); fp64, oracle/trace_oracle.c "
                  f"(gcc -O2), mode={mode}; {os.cpu_count()} logical host cores available",
    }


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.lower().startswith("model name"):
                    return ln.split(":", 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def load_traffic(workload, dtype, mode):
    """HBM bytes per launch from the committed rocprofv3 PMC summary, if any."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as f:
            doc = json.load(f)
        return doc.get(f"{workload}:{dtype}:{mode}")
    except (OSError, ValueError):
        return None


def measure_traffic_live(args, kernel_substr, log=None):
    """HBM bytes per launch of the dominant kernel, measured NOW: this command again, short
    (3 steps, no baselines, plain block), under `rocprofv3 --kernel-trace --pmc <counter>` --
    one pass per counter, as MI355X_MICROARCH.md prescribes (FETCH_SIZE / WRITE_SIZE in KiB;
    gfx950 reports half of a wide coalesced read: x 2).  None when rocprofv3 is missing, this
    process is itself such a child, or a pass fails (the committed figure is used then)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile

    if os.environ.get("OPTILAND_BENCH_CHILD") == "1" or shutil.which("rocprofv3") is None:
        return None
    vals = {}
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ol_pmc_", dir="/tmp")
        cmd = ["rocprofv3", "--kernel-trace", "--pmc", ctr, "--output-format", "csv", "-d", d,
               "-o", "t", "--", sys.executable, os.path.abspath(__file__), "--gpus", "1",
               "--steps", "3", "--warmup", "1", "--settle", "0", "--no-cpu-baseline",
               "--no-ref-baselines", "--placement", "plain", "--traffic", "committed",
               "--rays", str(args.rays), "--dtype", args.dtype, "--workload", args.workload,
               "--mode", args.mode, "--object-row", args.object_row]
        try:
            subprocess.run(cmd, cwd="/tmp", timeout=180, capture_output=True,
                           env=dict(os.environ, TMPDIR="/tmp", OPTILAND_BENCH_CHILD="1"))
            rows = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    rows += [float(r["Counter_Value"]) for r in csv.DictReader(fh)
                             if r.get("Counter_Name") == ctr
                             and kernel_substr in r.get("Kernel_Name", "")]
            if not rows:
                return None
            # (the first launch of a process also faults its pages in: the later ones)
            rows = rows[1:] if len(rows) > 1 else rows
            vals[ctr] = sum(rows) / len(rows)
        except (OSError, subprocess.SubprocessError, ValueError, KeyError):
            return None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return {"bytes": 2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024,
            "FETCH_SIZE_KiB": vals["FETCH_SIZE"], "WRITE_SIZE_KiB": vals["WRITE_SIZE"]}


def traffic_meta() -> dict:
    """Where / when the committed PMC passes ran (profiles/traffic.json: "_meta")."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            return json.load(f).get("_meta") or {}
    except (OSError, ValueError):
        return {}


def stream_fill_bandwidth(hip, device, nbytes, store_bytes, planes, reps=6, block=None):
    """GB/s of `ol_stream_fill` over a buffer of `nbytes` in `planes` planes (the write
    footprint and store pattern of the trace launch): a kernel that ONLY writes, every lane
    one element of `store_bytes` into each plane with the trace kernels' own non-temporal
    stores -- timed in THIS run, on this box.  A footprint far beyond the 256 MB Infinity
    Cache is what makes it a yardstick for the record-all kernels (a 1 GiB fill still drains
    its tail into the cache and reads ~25 % high)."""
    import ctypes as C
    unit = int(store_bytes) * int(planes)
    nbytes = int(nbytes) // unit * unit
    if nbytes <= 0 or not hasattr(hip.lib, "ol_stream_fill"):
        return None
    # `block`: the fill goes into THAT memory (the record block of this run's steps, planes at
    # its own stride) -- where a block lies decides 5.8 or 7.1 TB/s, so a ceiling for a kernel
    # is only one when it is measured where the kernel writes
    buf = block if block is not None else torch.empty(nbytes, dtype=torch.uint8, device=device)
    stream = hip._stream()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(2 + reps):
        if k == 2:
            e0.record()
        rc = hip.lib.ol_stream_fill(C.c_void_p(buf.data_ptr()), nbytes, int(store_bytes),
                                    int(planes), 0x3f800000, stream)
        if rc != 0:
            return None
    e1.record()
    torch.cuda.synchronize(device)
    ms = e0.elapsed_time(e1) / reps
    del buf
    return nbytes / (ms * 1e-3) / 1e9


def device_bandwidth(device, gib=1, reps=10):
    """Device-to-device copy and fill bandwidth measured in THIS run (what a pure
    streaming kernel sustains on this box): HIP events around `reps` 1 GiB copies."""
    nel = gib * (1 << 30) // 4
    src = torch.empty(nel, dtype=torch.float32, device=device).normal_()
    dst = torch.empty_like(src)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for _ in range(2):
        dst.copy_(src)
        dst.zero_()
    torch.cuda.synchronize(device)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    for _ in range(reps):
        dst.zero_()
    e2.record()
    torch.cuda.synchronize(device)
    size = nel * 4
    return {"copy_GBps": 2 * size * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9,
            "fill_GBps": size * reps / (e1.elapsed_time(e2) * 1e-3) / 1e9}


def reference_baselines(workload, dtype, wavelength, hy, S, budget_s, device, n=1_000_000,
                        full_rays=10_000_000):
    """The reference ITSELF on this node, same lens, same ray shape, bounded sample:
    `numpy` = Optiland's NumPy backend (`backend/numpy_backend.py`, fp64 always, see
    SURVEY Appendix D) through `Optic.trace_generic` on the host; `torch` = its stock torch
    backend (`backend/torch_backend.py:64-102`) on the same GPU, WITHOUT the drop-in.
    Needs the staged package (oracle/stage_reference.py -> oracle/_ref/); None otherwise."""
    try:
        from tests import _live
        if _live.reference_root() is None:
            return None
        be = _live.import_reference()
    except Exception as exc:  # noqa: BLE001
        print(f"reference baselines skipped: {exc!r}", file=sys.stderr)
        return None
    name = {"double_gauss": "DoubleGauss", "cooke": "CookeTriplet", "rc_asphere": "RCAsphere",
            "zernike_fresnel": "ZernikeFresnelUnpolarized"}.get(workload)
    if name is None:
        return None
    rng = np.random.default_rng(0)
    r, th = np.sqrt(rng.random(n)), 2 * np.pi * rng.random(n)
    px, py = r * np.cos(th), r * np.sin(th)
    out = {}
    try:
        be.set_backend("numpy")
        lens, w = _live.build_system(name)
        hx_, hy_ = np.zeros(n), np.full(n, hy)
        lens.trace_generic(hx_[:1000], hy_[:1000], px[:1000], py[:1000], w)  # warm caches
        reps, t0 = 0, time.perf_counter()
        while reps < 1 or (time.perf_counter() - t0 < budget_s and reps < 8):
            lens.trace_generic(hx_, hy_, px, py, w)
            reps += 1
        dt = time.perf_counter() - t0
        out["numpy"] = {
            "value": n * S * reps / dt, "unit": "ray-surfaces/s", "cores": 1,
            "cpu_model": cpu_model(), "host_logical_cores": os.cpu_count(),
            "kind": "reference",
            "sample": f"{reps} x Optic.trace_generic({n} rays) of the reference's {name}, NumPy "
                      f"backend fp64, {dt:.1f} s; elementwise NumPy kernels run on one core "
                      f"({os.cpu_count()} logical host cores available)",
        }
    except Exception as exc:  # noqa: BLE001
        out["numpy"] = None
        print(f"NumPy-backend baseline failed: {exc!r}", file=sys.stderr)
    try:
        be.set_backend("torch")
        be.set_device("cuda")
        be.set_precision("float32" if dtype == "f32" else "float64")
        tdt = torch.float32 if dtype == "f32" else torch.float64
        lens, w = _live.build_system(name)
        dev = [torch.as_tensor(a, dtype=tdt, device=device)
               for a in (np.zeros(n), np.full(n, hy), px, py)]
        with torch.no_grad():
            lens.trace_generic(*dev, w)  # warm (allocator, material caches)
            torch.cuda.synchronize(device)
            reps, t0 = 0, time.perf_counter()
            while reps < 1 or (time.perf_counter() - t0 < budget_s and reps < 8):
                lens.trace_generic(*dev, w)
                torch.cuda.synchronize(device)
                reps += 1
            dt = time.perf_counter() - t0
        out["torch"] = {
            "value": n * S * reps / dt, "unit": "ray-surfaces/s", "kind": "reference",
            "sample": f"{reps} x Optic.trace_generic({n} rays, device tensors) of the "
                      f"reference's {name} on its stock torch backend, device cuda, {dtype}, "
                      f"no drop-in, {dt:.1f} s",
        }
    except Exception as exc:  # noqa: BLE001
        out["torch"] = None
        print(f"torch-backend baseline failed: {exc!r}", file=sys.stderr)
    try:
        # the deliverable itself: the SAME reference call with the drop-in enabled
        # (optiland_amd.integration.enable(): RealRayTracer.trace_generic -> C ABI -> HIP
        # kernels), device tensors in / out, at the full batch size of this bench line
        from optiland_amd import integration
        be.set_backend("torch")
        be.set_device("cuda")
        be.set_precision("float32" if dtype == "f32" else "float64")
        tdt = torch.float32 if dtype == "f32" else torch.float64
        integration.enable()
        try:
            lens, w = _live.build_system(name)
            m = int(full_rays)
            g = torch.Generator(device=device).manual_seed(5)
            rr = torch.rand(m, generator=g, device=device, dtype=torch.float32).sqrt()
            tt = 2 * np.pi * torch.rand(m, generator=g, device=device, dtype=torch.float32)
            dpx, dpy = (rr * tt.cos()).to(tdt), (rr * tt.sin()).to(tdt)
            for _ in range(3):
                lens.trace_generic(0.0, hy, dpx, dpy, w)
            comp = lens.ray_tracer.__dict__.get("_hip_companion")
            assert comp is not None and comp.last_path == "hip", "drop-in did not intercept"
            times = []
            for _ in range(10):
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                lens.trace_generic(0.0, hy, dpx, dpy, w)
                torch.cuda.synchronize(device)
                times.append(time.perf_counter() - t0)
            ms = float(np.median(times)) * 1e3
            out["dropin"] = {
                "value": m * S / (ms * 1e-3), "unit": "ray-surfaces/s", "ms_per_call": ms,
                "rays": m,
                "sample": f"median of 10 x Optic.trace_generic({m} rays, scalar field, device "
                          f"tensors in / out) of the reference's {name} under "
                          f"integration.enable(), {dtype}: ray generation + record-all trace + "
                          "result objects + status read-back, wall clock",
            }
        finally:
            integration.disable()
        # the same call with lazy per-surface records (opt-in): the launch records the last two
        # surfaces, the interior ones are produced if somebody reads them
        integration.enable(lazy_records=True)
        try:
            lens, w = _live.build_system(name)
            for _ in range(3):
                lens.trace_generic(0.0, hy, dpx, dpy, w)
            times = []
            for _ in range(10):
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                lens.trace_generic(0.0, hy, dpx, dpy, w)
                torch.cuda.synchronize(device)
                times.append(time.perf_counter() - t0)
            out["dropin"]["lazy_records_ms_per_call"] = float(np.median(times)) * 1e3
        finally:
            integration.disable()
    except Exception as exc:  # noqa: BLE001
        if not out.get("dropin"):
            out["dropin"] = None
        print(f"drop-in end-to-end leg failed: {exc!r}", file=sys.stderr)
    finally:
        be.set_precision("float64")
        be.set_device("cpu")
        be.set_backend("numpy")
    return out


# OL_BENCH_EXCH_DEBUG = nozero | nocoll: what the pieces of the reduce-first exchange cost per step
# (diagnostic only: the statistics of such a run are wrong)
_DBG = os.environ.get("OL_BENCH_EXCH_DEBUG", "")


def main():
    args = parse_args()
    self_launch(args)  # N > 1 without a launcher: re-exec under torchrun and exit
    PLUMBING["on"] = bool(args.plumbing_check)
    rank, local, world = init_dist(args.gpus, args.plumbing_check)
    from optiland_amd.distributed import shard_bounds
    if args.plumbing_check:
        # TEST ONLY: CPU tensors + the host build of the kernel source behind the same engine
        # class (tests/_hostmath.py); nothing here is a measurement
        from tests import _hostmath
        device = torch.device("cpu")
        HipSystem = _hostmath.make_engine_class()
        args.no_cpu_baseline = True
    else:
        from optiland_amd.engine import HipSystem
        device = torch.device("cuda", local)

    strong = args.config == "c3"
    if strong:  # BASELINE.json configs[2]: double Gauss, fp64, 1e8 rays over the ranks
        args.workload, args.dtype = "double_gauss", "f64"
    if args.mode == "opd":
        args.dtype = "f64"  # ol_trace_opd is fp64 only (an OPD in waves)
    table, hy, desc, wavelength = load_workload(args.workload)
    wl = table.wavelength_index(wavelength)
    hip = HipSystem(table, device)
    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    b = 4 if args.dtype == "f32" else 8
    if strong:
        lo, hi = shard_bounds(int(args.total_rays), world, rank)
        n, job_rays = hi - lo, int(args.total_rays)
    else:
        n, job_rays = int(args.rays), int(args.rays) * world
    S = table.num_traced
    pol = table.uses_polarization

    gen = args.mode == "gen"
    if gen and not hip.can_trace_generate():
        raise SystemExit("--mode gen needs generator scalars without a pupil apodization")
    # The record block is reused by every step: it is PLACED (engine.alloc_record_placed: the
    # window of an arena in which this launch's own store pattern writes fastest, found with
    # `ol_stream_fill` outside the timed region).  `--placement plain` = one ordinary
    # allocation, as the drop-in makes for every trace it hands to a user.
    placement = None
    record = None

    def make_record():
        if args.placement == "probe" and not args.plumbing_check \
                and hasattr(hip, "alloc_record_placed"):
            # (time-boxed: a rank whose arenas hold no fast window gives up after 2 s and
            # writes into an ordinary block instead of keeping the others at the barrier)
            return hip.alloc_record_placed(n, dtype, time_budget_s=2.0)
        return hip.alloc_record(n, dtype), None

    if args.mode == "record":
        record, placement = make_record()
    # (--mode gen: the block is the LAST thing set up, right in front of the warm-up steps --
    # nothing of the step needs it earlier, and the placement probe's ~0.1 s of sustained
    # writes then leads straight into them instead of being followed by an idle gap)
    alias = record is not None and args.object_row == "alias" and not gen
    opd_mode = args.mode == "opd"
    spot = args.mode == "spot" or opd_mode  # the fused, ray-plane-free pipelines
    if gen:
        px, py = make_pupil(n, dtype, 1234 + rank, device)
        rays = []
    elif spot:
        if pol:
            raise SystemExit(f"--mode {args.mode} needs an unpolarised workload")
        px, py = make_pupil(n, dtype, 1234 + rank, device)
        rays = []
        mom = [torch.zeros(12 if opd_mode else 7, dtype=torch.float64, device=device)
               for _ in range(2)]
        if opd_mode:
            # reference sphere of the chief ray of this field (wavefront/strategy.py:176-184),
            # worked out once outside the timed region like a consumer's first step
            from optiland_amd.tracer import HipRayTracer
            from optiland_amd.wavefront import Wavefront
            wf = Wavefront(HipRayTracer(table, device, dtype=torch.float64, engine=hip),
                           (0.0, hy), wavelength, num_rays=3)
            opd_params = wf.chief_reference()[0]
    else:
        rays = make_rays(hip, n, dtype, hy, seed=1234 + rank, device=device,
                         out=hip.row0_planes(record, n) if alias else None)
    prt = None
    if pol:  # write-only: every step starts a fresh PRT from the identity in-kernel
        prt = torch.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype, device=device)
    scratch = [torch.empty_like(t) for t in rays] if args.mode == "last" else None

    import torch.distributed as dist
    have_pg = dist.is_available() and dist.is_initialized()
    rccl_world = None
    if have_pg:
        # the number of ranks as a REAL collective on the compute device reports it (a sum of
        # ones over the group), not as the launcher's environment claims it
        ones = torch.ones(1, dtype=torch.float64, device=device)
        dist.all_reduce(ones)
        rccl_world = int(round(float(ones.item())))
    exchange = args.exchange if (world > 1 or (args.force_exchange and have_pg)) else "none"
    configured_exchange = exchange
    pending = [None, None]  # in-flight all-gathers (double-buffered)
    # buffers of BOTH exchange kinds whenever one is configured: the kind that is not timed
    # runs afterwards as an extra, untimed-for-`value` leg, so that one run at N > 1 shows
    # the reduce-first exchange and the literal all-gather of hits side by side
    if exchange != "none" and not (args.mode in ("spot", "opd")):
        # two buffer pairs: the all-gather of step k runs on RCCL's stream while step
        # k+1 traces; a buffer is only reused after its collective has completed
        # (strong scaling of a total that does not divide: every rank contributes the
        # largest shard's size, the tail of a smaller shard is padding)
        n_g = -(-job_rays // world) if strong else n
        gather_buf = [torch.empty((world, 3, n_g), dtype=dtype, device=device) for _ in range(2)]
        hits = [torch.zeros((3, n_g), dtype=dtype, device=device) for _ in range(2)]
    if exchange != "none":
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
        if opd_mode:
            hip.trace_opd(opd_params, px, py, wl, field=(0.0, hy), want_pupil=True,
                          moments=mom[k], check_status=False)
        else:
            hip.trace_spot(px, py, wl, field=(0.0, hy), out=mom[k], check_status=False)
        if ev1 is not None:
            ev1.record()
        if exchange != "none":
            if opd_mode:  # the twelve sums add up across ray shards
                pending[k] = (dist.all_reduce(mom[k], async_op=True),)
            else:
                pending[k] = (dist.all_reduce(mom[k][:6], async_op=True),
                              dist.all_reduce(mom[k][6:], op=dist.ReduceOp.MAX, async_op=True))
        return mom[k]

    def step(ev0=None, ev1=None):
        if spot:
            return spot_step(ev0, ev1)
        if args.mode in ("record", "gen"):
            src = rays
        else:  # last-surface mode mutates the rays in place: refresh from the source
            for d, s_ in zip(scratch, rays):
                d.copy_(s_)
            src = scratch
        k = step_no[0] & 1
        step_no[0] += 1
        if exchange != "none" and pending[k] is not None:
            for w in (pending[k] if isinstance(pending[k], tuple) else (pending[k],)):
                # this buffer pair is free again once its collective (two steps back) is done:
                # normally it IS by now, and asking (an event query on the host) spares the
                # compute stream a wait packet between two trace launches
                if _DBG == "alwayswait" or not w.is_completed():
                    w.wait()
            pending[k] = None
        spot_arg = None
        if exchange == "reduce":
            if _DBG != "nozero":
                slots[k].zero_()
            if not pol:  # polarised traces reduce after the launch (see below)
                spot_arg = (slots[k], 0.0, 0.0)
        if ev0 is not None:
            ev0.record()
        if gen:
            # ONE launch per step also at N > 1: the masked image-plane moments are an
            # epilogue of the generating kernel (ABI 8, trace_kernel<..., SPOT, GEN>)
            res = hip.trace_generate(px, py, wl, field=(0.0, hy), record=record, prt=prt,
                                     zero_status=False, defer_status=True, spot=spot_arg)
        else:
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
                                  if record is not None else (src[0], src[1], src[6]))
                    hip.spot_moments(xi, yi, ii, out=slots[k].view(-1)[:6])
                if _DBG not in ("nocoll", "nozero_nocoll"):
                    pending[k] = dist.all_gather_into_tensor(all_slots[k].view(-1),
                                                             slots[k].view(-1), async_op=True)
            else:
                if record is not None:
                    x, y, inten = res.row(res.last, 0), res.row(res.last, 1), res.row(res.last, 6)
                else:
                    x, y, inten = src[0], src[1], src[6]
                hits[k][0, :n].copy_(x)
                hits[k][1, :n].copy_(y)
                hits[k][2, :n].copy_(inten)
                pending[k] = dist.all_gather_into_tensor(gather_buf[k].view(-1),
                                                         hits[k].view(-1), async_op=True)
        return res

    def timed(steps, evs=None):
        """EXACTLY `steps` steps between barrier + synchronize brackets; max over ranks."""
        sync(device)
        if have_pg:
            dist.barrier()
        sync(device)
        t0 = time.perf_counter()
        for k in range(steps):
            step(*(evs[k] if evs else ()))
        for k in range(2):
            w = pending[k]
            for ww in (w if isinstance(w, tuple) else (w,)):
                if ww is not None:
                    ww.wait()
            pending[k] = None
        sync(device)
        if have_pg:
            dist.barrier()
        sync(device)
        dt = time.perf_counter() - t0
        if have_pg:
            tt = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt

    # The interpreter's cyclic garbage collector stays out of the measurements: a generation-2
    # pass over a process with torch.distributed loaded is a 35-53 ms pause of the HOST between
    # two launches, wherever its allocation count happens to trip (profiles/r04_host_gc_stall.txt:
    # no HIP call at all for 39 ms between a step's hipEventRecord and its hipLaunchKernel).
    # Collected once HERE -- in front of the warm-up steps, so that the idle gap it leaves on the
    # device is not the start of a timed region -- and switched off until the GPU legs are done.
    gc.collect()
    gc.disable()
    if have_pg:
        # RCCL's first collective (the barrier of `timed` when the step itself has none) is made
        # HERE, before the placement arena exists and outside every timed region
        dist.barrier()
        sync(device)
    if gen:
        record, placement = make_record()
    if have_pg:
        # the placement probe takes 0.1-0.5 s, rank by rank different (one to three arenas): the
        # ranks meet HERE, so that they start the warm-up together and none of them sits idle at
        # the timed region's barrier with a part that has dropped out of its loaded state
        dist.barrier()
        sync(device)
    for _ in range(args.warmup):
        step()
    evs = [(make_event(), make_event()) for _ in range(args.steps)]
    # (HIP events are created at their first record(): done here, once, so that the timed region
    # creates nothing)
    for e0_, e1_ in evs:
        e0_.record()
        e1_.record()
    sync(device)
    elapsed = timed(args.steps, evs)
    if exchange == "reduce" and not spot and args.steps:
        # fold the gathered slots of the last step: whole-job spot statistics
        tot = hip.reduce_spot_slots(all_slots[(step_no[0] - 1) & 1].view(-1, 8)).cpu().numpy()
        assert tot[0] > 0 or _DBG, "no ray reached the image plane"
    exchange_info = None
    if configured_exchange != "none" and args.steps:
        # the same steps WITHOUT the image-plane exchange (outside the reported region):
        # the difference is what the exchange costs per step, overlap included
        exchange = "none"
        step()  # untimed: the launch without the spot epilogue is a different kernel variant
        bare = timed(args.steps)
        exchange = configured_exchange
        wire = (8 * 64 * 8 if configured_exchange == "reduce" else 3 * b * n) if not spot \
            else (96 if opd_mode else 56)
        kinds = {"reduce": "async all-gather of the 4 KB spot-moment slot block (RCCL)",
                 "gather": "literal RCCL all-gather of image-plane hits (x, y, i)"}
        exchange_info = {
            "kind": kinds[configured_exchange] if not spot else "all-reduce of 7 doubles",
            "ms_per_step_with": elapsed / args.steps * 1e3,
            "ms_per_step_without": bare / args.steps * 1e3,
            "exchange_ms_per_step": (elapsed - bare) / args.steps * 1e3,
            "bytes_per_rank_per_step": wire,
        }
        if not spot:
            # the OTHER exchange kind, same steps, outside the reported region
            other = "gather" if configured_exchange == "reduce" else "reduce"
            exchange = other
            step()
            alt = timed(args.steps)
            exchange = configured_exchange
            exchange_info["other"] = {
                "kind": kinds[other],
                "ms_per_step_with": alt / args.steps * 1e3,
                "exchange_ms_per_step": (alt - bare) / args.steps * 1e3,
                "bytes_per_rank_per_step": 8 * 64 * 8 if other == "reduce" else 3 * b * n,
            }

    kern_each = [a.elapsed_time(bb) for a, bb in evs]
    kern_ms = float(np.mean(kern_each)) if args.steps else float("nan")
    # what every rank did, for rank 0's line (outside the timed region): its shard, where its
    # record block lies and what its kernel took -- a slow rank is visible, not averaged away
    mine = {"rank": rank, "rays": n, "kernel_ms": kern_ms,
            "placement": None if placement is None else {
                k: placement.get(k) for k in ("placed", "arenas_tried", "probes",
                                              "probe_best_GBps", "probe_median_GBps",
                                              "probe_seconds", "gave_up")}}
    per_rank = [mine]
    if have_pg:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    bw = device_bandwidth(device) if (rank == 0 and args.steps and not args.plumbing_check) \
        else None
    # Steady state, OUTSIDE the reported region: the same launch repeated until the part's
    # power management has settled (profiles/r04_clock_transient.txt: after ~1.5 ms of a
    # vector-ALU-heavy kernel the clocks drop by up to 40 % and take ~80 ms of sustained load
    # to come back -- exactly where a "5 warm-up + 20 timed" window sits; HBM-bound kernels
    # barely see it).  Mean of the last third of `--settle` more launches.
    steady = None
    if rank == 0 and args.steps and args.settle > 0 and not args.plumbing_check:
        sev = [(make_event(), make_event()) for _ in range(args.settle)]
        keep_exchange, exchange = exchange, "none"
        for e0_, e1_ in sev:
            step(e0_, e1_)
        for k in range(2):
            w = pending[k]
            for ww in (w if isinstance(w, tuple) else (w,)):
                if ww is not None:
                    ww.wait()
            pending[k] = None
        sync(device)
        exchange = keep_exchange
        each = [a.elapsed_time(bb) for a, bb in sev]
        tail = each[-max(len(each) // 3, 1):]
        steady = {"launches_before": args.warmup + args.steps + len(each) - len(tail),
                  "launches_averaged": len(tail), "kernel_ms": float(np.mean(tail)),
                  "kernel_us_minmax": [min(tail) * 1e3, max(tail) * 1e3]}
        if gen and not pol and args.workload == "double_gauss" and hasattr(hip.lib, "ol_set_tuning"):
            # the same launch with at most two workgroups resident per CU: what a LOOP of
            # hundreds of back-to-back traces can ask for (`ol_set_tuning`): faster once the
            # clocks have settled, slower in the first 25 launches -- hence not the default
            # for a placed block (profiles/r05_ab_wgcap.txt)
            lib_, knob_ = hip.lib, 3  # OL_TUNE_RECORD_WG_CAP
            try:
                prev_cap = int(os.environ.get("OL_RECORD_WG_CAP", "0") or 0)
            except ValueError:
                prev_cap = 0
            if lib_.ol_set_tuning(knob_, 2) == 0:
                try:
                    cev = [(make_event(), make_event()) for _ in range(max(args.settle // 2, 30))]
                    for e0_, e1_ in cev:
                        e0_.record()
                        hip.trace_generate(px, py, wl, field=(0.0, hy), record=record, prt=prt,
                                           zero_status=False, defer_status=True)
                        e1_.record()
                    sync(device)
                finally:
                    lib_.ol_set_tuning(knob_, prev_cap)   # (what the environment had seeded)
                ct = [a.elapsed_time(bb) for a, bb in cev]
                ct = ct[-max(len(ct) // 3, 1):]
                steady["kernel_ms_two_workgroups_per_cu"] = float(np.mean(ct))
            # (round 6: the engine passes that cap by itself once it has seen this many launches
            # in a row into one block -- `steady_state.kernel_ms` above already has it)
            from optiland_amd import engine as _E
            steady["hot_loop_after"] = int(_E._HOT_LOOP["after"])
        if placement is not None and placement.get("placed") and gen:
            # the same launch into an ORDINARY allocation (what a drop-in trace gets), for
            # comparison -- outside the reported region
            plain = hip.alloc_record(n, dtype)
            pev = [(make_event(), make_event()) for _ in range(40)]
            for e0_, e1_ in pev:
                e0_.record()
                hip.trace_generate(px, py, wl, field=(0.0, hy), record=plain, prt=prt,
                                   zero_status=False, defer_status=True)
                e1_.record()
            sync(device)
            pt = [a.elapsed_time(bb) for a, bb in pev][10:]
            placement["kernel_ms_plain_block"] = float(np.mean(pt))
            del plain

    if rank == 0:
        total_rs = float(job_rays) * S * args.steps
        value = total_rs / elapsed
        # algorithmic bytes per launch: record-all reads 8 planes and
        # writes 8 planes for each of the S+1 surfaces; record-last reads 8, writes 8.
        if args.mode in ("record", "gen"):
            alg_bytes = 8 * b * (S + 2) * n
        elif opd_mode:
            alg_bytes = (2 + 5) * b * n  # pupil planes in; OPD, intensity, pupil point out
        elif spot:
            alg_bytes = 2 * b * n  # the two pupil planes; everything else stays in registers
        else:
            alg_bytes = 16 * b * n
        if pol:
            alg_bytes += 2 * 9 * b * n  # PRT read-modify-write (SURVEY figure)
        # bytes this launch really has to move: with the zero-copy object row, row 0
        # of the record block IS the input, so only S rows are written; the PRT of a fresh
        # trace is write-only (starts from I in-kernel)
        moved_bytes = alg_bytes - (8 * b * n if alias else 0) - (9 * b * n if pol else 0)
        if gen:  # two pupil planes are read instead of eight ray planes; row 0 IS written
            moved_bytes -= 6 * b * n
        alg_GBps = alg_bytes / (kern_ms * 1e-3) / 1e9
        moved_GBps = moved_bytes / (kern_ms * 1e-3) / 1e9
        traffic = load_traffic(args.workload, args.dtype,
                               args.mode + (":alias" if alias else ""))
        traffic_rays = 10_000_000  # every committed PMC pass ran 1e7 rays per launch
        if traffic is not None and n != traffic_rays:
            traffic = traffic * n / traffic_rays
        # write-only yardstick of THIS run: a non-temporal fill of the launch's own write
        # footprint with its own store width
        written = moved_bytes - (2 * b * n if (gen or spot) else 8 * b * n)
        fill_big = fill_plain = None
        if bw is not None and args.mode in ("record", "gen") and written >= (1 << 28):
            fill_plain = stream_fill_bandwidth(hip, device, min(written, 16 << 30), b,
                                               max(int(round(written / (b * n))), 1))
            if record is not None and record.is_contiguous():
                # ... and on the record block of the timed steps ITSELF (its trace legs are
                # done; the fill overwrites it): the ceiling of the place the kernel wrote to
                rb = record.view(-1).view(torch.uint8)
                fill_big = stream_fill_bandwidth(hip, device, rb.numel(), b,
                                                 int(record.shape[0]) * 8, block=rb)
        live = None
        if args.traffic == "live" and world == 1 and not args.plumbing_check and args.steps:
            live = measure_traffic_live(
                args, "opd_trace_kernel" if opd_mode else
                ("spot_trace_kernel" if spot else "trace_kernel"))
            if live is not None:
                traffic = live["bytes"]
        if steady is not None:
            steady["achieved"] = moved_bytes / (steady["kernel_ms"] * 1e-3) / 1e9
            steady["frac"] = steady["achieved"] / HBM_PEAK_GBS
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
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "plumbing-check (CPU, not a measurement)" if args.plumbing_check
                    else "synthetic",
            "config": {
                "workload": f"{desc} ({S} traced surfaces incl. image plane), "
                            + (f"{job_rays:.3g} rays in total over {world} GPU(s) "
                               f"({n:.4g} on rank 0) " if strong else f"{n:.3g} rays/GPU ")
                            + f"{args.dtype}, lambda={wavelength} um, "
                            f"uniform-disc pupil, Hy={hy}, "
                            f"mode={MODE_NAMES[args.mode]}",
                "baseline_config": "configs[2]" if strong else
                                   ("configs[1]" if (args.workload, args.dtype, n) ==
                                    ("double_gauss", "f32", 10_000_000)
                                    and args.mode in ("gen", "record") else None),
                "rays_per_gpu": n,
                "rays_per_rank": [r_["rays"] for r_ in per_rank],
                "rays_total": job_rays,
                "surfaces": S,
                "mode": args.mode,
                "object_row": ("written by the tracing launch itself (generated there)" if gen else
                               ("zero-copy (rays generated into record row 0)" if alias
                                else "copied")) if args.mode in ("record", "gen") else None,
                "exchange": configured_exchange,
                "parallelism": f"ray-shard x{world}",
            },
            "exchange": (dict(exchange_info or {}, rccl_world=rccl_world,
                              backend=dist.get_backend() if have_pg else None)
                         if (exchange_info or have_pg) else None),
            "roofline": {
                "bound": "hbm",
                "kernel": "opd_trace_kernel" if opd_mode else
                          ("spot_trace_kernel" if spot else
                           (gen_kernel_label(table, args.dtype, pol) if gen else "trace_kernel")),
                # the contract's `achieved` / `frac`: bytes this launch really moves (equal
                # to the PMC traffic within 0.1 %) over the HIP-event kernel time
                "achieved": moved_GBps,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": moved_GBps / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": (
                    "measured in THIS run on this box: two child runs of this command (3 steps, "
                    "plain block) under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc "
                    "WRITE_SIZE`, mean over the kernel's launches after the first; "
                    f"2 x {live['FETCH_SIZE_KiB']:.1f} KiB (gfx950 halves wide reads) + "
                    f"{live['WRITE_SIZE_KiB']:.1f} KiB") if live is not None else (
                    "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE / "
                    "WRITE_SIZE passes of this command, committed (not re-measured "
                    "in this run" + (f"; scaled from {traffic_rays} to {n} rays)"
                                     if n != traffic_rays else ")")),
                "traffic_box": "this run's" if live is not None else (
                    traffic_meta().get("box") or
                    "the builder's MI355X box of the round the PMC passes ran in "
                    "(tools/collect_profiles.py); another box than this run's"),
                "traffic_over_moved": (traffic / moved_bytes) if traffic else None,
                "kernel_ms": kern_ms,
                "kernel_us_minmax": [min(kern_each) * 1e3, max(kern_each) * 1e3]
                                    if kern_each else None,
                "kernel_us_each": [round(t * 1e3, 1) for t in kern_each[:64]],
                "steady_state": steady,
                "record_placement": placement,
                "record_placement_ranks": [r_["placement"] for r_ in per_rank],
                "kernel_ms_ranks": [r_["kernel_ms"] for r_ in per_rank],
                "moved_bytes": moved_bytes,
                "algorithmic_bytes": alg_bytes,
                "achieved_algorithmic": alg_GBps,
                "frac_algorithmic": alg_GBps / HBM_PEAK_GBS,
                "bytes_per_ray_surface": alg_bytes / (float(n) * S),
                "device_copy_GBps": bw and bw["copy_GBps"],
                "device_fill_1GiB_GBps": bw and bw["fill_GBps"],
                "stream_fill_GBps": fill_big,
                "stream_fill_where": "the record block of the timed steps itself" if fill_big
                                     else None,
                "stream_fill_plain_block_GBps": fill_plain,
                "frac_of_write_ceiling": (moved_GBps / fill_big) if fill_big else None,
                "note": ("fused spot kernel: only the two pupil planes touch HBM, the kernel is "
                         "vector-ALU bound by construction -- the HBM fraction is reported for "
                         "the contract, not as its limiter" if spot else
                         "frac = moved_bytes / kernel time / 8 TB/s spec peak; "
                         "frac_algorithmic credits SURVEY 8d's figure (the object row the ray "
                         "generator wrote into the record block outside the timed region and, "
                         "for polarised runs, the PRT read a fresh trace never does); "
                         "frac_of_write_ceiling = moved GB/s over `ol_stream_fill` -- this "
                         "launch's own store pattern (as many planes at the same stride, one "
                         "non-temporal store per lane and plane) with the arithmetic taken out, "
                         "timed in this run INTO THE SAME record block (the fill writes the "
                         "stride padding too and issues 4-byte stores where the packed-pair "
                         "kernel issues 8-byte ones: a few % either way); "
                         "steady_state = the same launch after the clock transient of the "
                         "first ~100 ms (outside the reported region)"),
            },
        }
        gc.enable()  # (the GPU legs of this line are done)
        if world == 1 and not args.no_cpu_baseline:
            port = cpu_baseline(table, hy, "last" if spot else ("record" if gen else args.mode),
                                args.cpu_seconds, wl)
            ref = None if args.no_ref_baselines else reference_baselines(
                args.workload, args.dtype, wavelength, hy, S, args.ref_seconds, device,
                full_rays=n)
            if ref is not None and ref.get("numpy"):
                # the north-star comparator: Optiland's own NumPy path on this node's host
                out["cpu_baseline"] = dict(ref["numpy"], port=port)
                out["gpu_baseline"] = {"torch": ref.get("torch")}
                drop = ref.get("dropin")
                if drop:
                    # whole reference call (ray generation + trace + objects + read-back)
                    # over the trace kernel alone of this bench line
                    # (a drop-in trace writes into an ordinary allocation: its kernel is
                    # the plain-block one, not the placed block of this line's steps)
                    plain_ms = (placement or {}).get("kernel_ms_plain_block") or kern_ms
                    drop["kernel_ms_plain_block"] = plain_ms
                    drop["over_trace_kernel"] = drop["ms_per_call"] / plain_ms
                out["dropin"] = drop
            else:
                out["cpu_baseline"] = port
                out["gpu_baseline"] = None
        else:
            out["cpu_baseline"] = None
            out["gpu_baseline"] = None
        print(json.dumps(out))
    hip.close()
    if have_pg:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
