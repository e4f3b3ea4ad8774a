"""Round 6: what the in-window spread of the fp64 record-all kernel IS.

ROUNDS x (1 s idle, then LAUNCHES launches of one configuration back to back, HIP events around
each).  Run under `rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` (tools/gpu_r06.sh
clock_transient): per dispatch, GRBM_GUI_ACTIVE / (end - start) is the clock the graphics
engine really ran at during that launch.  A sampler thread reads the part's own account next to
it (sclk / mclk / fclk levels, average power, temperatures from sysfs) as fast as sysfs answers.
Slow launches at a LOWER engine clock: power management (and the power column says whether the
part sits at its limit); slow launches at the SAME clock: not the shader clock (memory side)."""
import glob
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402

LAUNCHES = int(os.environ.get("LAUNCHES", "40"))
ROUNDS = int(os.environ.get("ROUNDS", "3"))
CONFIGS = [c for c in os.environ.get("CONFIGS", "dg_f64,dg_f32,zf_f32").split(",") if c]
WORK = {"dg_f64": ("double_gauss", torch.float64), "dg_f32": ("double_gauss", torch.float32),
        "zf_f32": ("zernike_fresnel", torch.float32), "zf_f64": ("zernike_fresnel", torch.float64),
        "rc_f32": ("rc_asphere", torch.float32), "rc_f64": ("rc_asphere", torch.float64)}
dev = torch.device("cuda", 0)
n = 10_000_000
samples, stop = [], threading.Event()


def first(pattern):
    g = sorted(glob.glob(pattern))
    return g[0] if g else None


CARD = os.path.dirname(first("/sys/class/drm/card*/device/pp_dpm_sclk") or "/nonexistent/x")
HW = first(CARD + "/hwmon/hwmon*") if os.path.isdir(CARD) else None


def star(path):
    try:
        for ln in open(path):
            if "*" in ln:
                return ln.split(":")[1].strip().rstrip("*").strip()
    except OSError:
        pass
    return "?"


def read(path):
    try:
        return open(path).read().strip()
    except OSError:
        return "?"


def sampler():
    while not stop.is_set():
        t = time.perf_counter()
        row = [t, star(CARD + "/pp_dpm_sclk"), star(CARD + "/pp_dpm_mclk"),
               star(CARD + "/pp_dpm_fclk")]
        if HW:
            row += [read(HW + "/power1_average"), read(HW + "/temp1_input"),
                    read(HW + "/freq1_input")]
        samples.append(row)


th = threading.Thread(target=sampler, daemon=True)
th.start()
for cfg in CONFIGS:
    workload, dtype = WORK[cfg]
    table, hy, _desc, wavelength = bench.load_workload(workload)
    wl = table.wavelength_index(wavelength)
    hip = HipSystem(table, dev)
    px, py = bench.make_pupil(n, dtype, 1234, dev)
    rec, info = hip.alloc_record_placed(n, dtype)
    prt = torch.empty((9, n), dtype=dtype, device=dev) if table.uses_polarization else None
    for r in range(ROUNDS):
        torch.cuda.synchronize()
        time.sleep(1.0)
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
               for _ in range(LAUNCHES)]
        t0 = time.perf_counter()
        for a, b in evs:
            a.record()
            hip.trace_generate(px, py, wl, field=(0.0, hy), record=rec, prt=prt,
                               zero_status=False, defer_status=True)
            b.record()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        ms = [a.elapsed_time(b) for a, b in evs]
        print(f"ROUND {cfg} {r} placed={info.get('placed')} t0={t0:.6f} t1={t1:.6f} ms " +
              " ".join(f"{v:.4f}" for v in ms), flush=True)
    hip.close()
    del rec, prt
stop.set()
th.join(timeout=1.0)
print(f"SAMPLER card={CARD} hwmon={HW} samples={len(samples)}")
for row in samples:
    print("S " + " ".join(str(v) for v in row))
