#!/usr/bin/env python
"""Why does the C5 generating kernel show 257-430 us call to call (profiles/r03_zf_f32_gen_rocprof.txt)
when the double-Gauss one stays within +-5 %?

Launches a kernel configuration back to back (HIP events around every launch) while a host
thread samples the GPU's shader clock and socket power from sysfs (hwmon freq1_input /
power1_average, ~1 kHz; `rocm-smi` output at the start and the end as a cross-check), then
prints per-launch duration next to the clock / power sampled during that launch, the
correlation of the two, and duration x clock (constant if the spread IS the clock).

    python tools/gpu_clock_spread.py --workload zernike_fresnel [--dtype f32] [--launches 300]
                                      [--idle-ms 0] [--mode gen]
`--idle-ms K`: sleep K ms between launches (does the clock drop when the queue drains?).
"""
import argparse
import glob
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402
from optiland_amd.rays import _state_dict  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="zernike_fresnel")
ap.add_argument("--dtype", default="f32")
ap.add_argument("--mode", default="gen", choices=("gen", "gen_epi"))
ap.add_argument("--rays", type=float, default=1e7)
ap.add_argument("--launches", type=int, default=300)
ap.add_argument("--idle-ms", type=float, default=0.0)
ap.add_argument("--out", default=None)
a = ap.parse_args()


def _sysfs():
    """(freq file, power file) of the first GPU that exposes them, or (None, None)."""
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        f = os.path.join(hw, "freq1_input")
        p = next((os.path.join(hw, k) for k in ("power1_average", "power1_input")
                  if os.path.exists(os.path.join(hw, k))), None)
        if os.path.exists(f):
            return f, p
    return None, None


def _smi():
    try:
        return subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showperflevel"],
                              capture_output=True, text=True, timeout=20).stdout
    except Exception as exc:  # noqa: BLE001
        return f"rocm-smi unavailable: {exc}"


FREQ, POWER = _sysfs()
samples = []  # (t, MHz, W)
stop = threading.Event()


def _read(path, scale):
    try:
        with open(path) as fh:
            return float(fh.read().strip()) * scale
    except Exception:  # noqa: BLE001
        return float("nan")


def sampler():
    while not stop.is_set():
        samples.append((time.perf_counter(), _read(FREQ, 1e-6) if FREQ else float("nan"),
                        _read(POWER, 1e-6) if POWER else float("nan")))
        time.sleep(0.0005)


dev = torch.device("cuda", 0)
table, hy, _desc, wavelength = bench.load_workload(a.workload)
wl = table.wavelength_index(wavelength)
hip = HipSystem(table, dev)
dtype = torch.float32 if a.dtype == "f32" else torch.float64
n = int(a.rays)
pol = table.uses_polarization
px, py = bench.make_pupil(n, dtype, 1234, dev)
record = hip.alloc_record(n, dtype)
prt = torch.empty((18 if table.needs_complex_prt else 9, n), dtype=dtype, device=dev) if pol \
    else None
kw = {}
if a.mode == "gen_epi" and pol:
    kw["update_intensity"] = _state_dict(table.polarization)


def launch():
    hip.trace_generate(px, py, wl, field=(0.0, hy), record=record, prt=prt, defer_status=True,
                       **kw)


print("== rocm-smi before ==")
print(_smi())
for _ in range(5):
    launch()
torch.cuda.synchronize()
th = threading.Thread(target=sampler, daemon=True)
th.start()
time.sleep(0.05)
evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
       for _ in range(a.launches)]
marks = []
t_begin = time.perf_counter()
for e0, e1 in evs:
    if a.idle_ms > 0:
        torch.cuda.synchronize()
        time.sleep(a.idle_ms * 1e-3)
    marks.append(time.perf_counter())
    e0.record()
    launch()
    e1.record()
torch.cuda.synchronize()
t_end = time.perf_counter()
time.sleep(0.05)
stop.set()
th.join()
print("== rocm-smi after ==")
print(_smi())

dur = np.array([e0.elapsed_time(e1) * 1e3 for e0, e1 in evs])  # us
# where in host time each launch ran: with a full queue the launches execute back to back
# from the first one; cumulative device time from the first event places them
start = np.array([evs[0][0].elapsed_time(e0) * 1e-3 for e0, _ in evs])  # s after first launch
t0 = marks[0]
st = np.array(samples)
clk = np.full(len(dur), np.nan)
pw = np.full(len(dur), np.nan)
if a.idle_ms > 0:
    base = np.array(marks)  # every launch starts right after its host mark
else:
    base = t0 + start
for k in range(len(dur)):
    m = (st[:, 0] >= base[k]) & (st[:, 0] <= base[k] + dur[k] * 1e-6 + 1e-3)
    if m.any():
        clk[k] = np.nanmean(st[m, 1])
        pw[k] = np.nanmean(st[m, 2])

ok = np.isfinite(clk)
rep = {
    "workload": a.workload, "dtype": a.dtype, "mode": a.mode, "rays": n, "launches": len(dur),
    "idle_ms": a.idle_ms,
    "us": {"mean": float(dur.mean()), "min": float(dur.min()), "max": float(dur.max()),
           "p05": float(np.percentile(dur, 5)), "p50": float(np.percentile(dur, 50)),
           "p95": float(np.percentile(dur, 95)), "std": float(dur.std())},
    "sclk_MHz": {"min": float(np.nanmin(st[:, 1])), "max": float(np.nanmax(st[:, 1])),
                 "mean": float(np.nanmean(st[:, 1]))} if FREQ else None,
    "power_W": {"min": float(np.nanmin(st[:, 2])), "max": float(np.nanmax(st[:, 2])),
                "mean": float(np.nanmean(st[:, 2]))} if POWER else None,
    "samples": len(st), "sysfs": [FREQ, POWER],
    "wall_s": t_end - t_begin,
}
if ok.sum() > 10:
    rep["corr_duration_vs_inverse_clock"] = float(np.corrcoef(dur[ok], 1.0 / clk[ok])[0, 1])
    prod = dur[ok] * clk[ok]  # us x MHz = cycles
    rep["cycles_per_launch"] = {"mean": float(prod.mean()), "std_over_mean":
                                float(prod.std() / prod.mean())}
    rep["duration_std_over_mean"] = float(dur[ok].std() / dur[ok].mean())
print(json.dumps(rep))
print("launch  us      MHz     W")
for k in list(range(0, min(40, len(dur)))) + list(range(len(dur) - 10, len(dur))):
    print(f"{k:5d} {dur[k]:8.1f} {clk[k]:7.0f} {pw[k]:7.1f}")
# decimated clock trace over the whole run
print("t_ms  MHz  W   (every 20th sample)")
for row in st[::20][:200]:
    print(f"{(row[0] - t0) * 1e3:8.2f} {row[1]:6.0f} {row[2]:6.1f}")
if a.out:
    with open(a.out, "w") as fh:
        json.dump(rep, fh)
hip.close()
