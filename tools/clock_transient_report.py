"""Post-processing of tools/gpu_clock_transient.py under rocprofv3 (tools/gpu_r06.sh
clock_transient): per round and launch, HIP-event ms, the profiler's duration, the effective
engine clock GRBM_GUI_ACTIVE / duration, and the sysfs samples that fall inside the round."""
import csv
import glob
import sys

import numpy as np

out = sys.argv[1]
log = open(out + "/run.log").read().splitlines()
rounds = []
for ln in log:
    if ln.startswith("ROUND "):
        p = ln.split()
        rounds.append((p[1], int(p[2]), float(p[4][3:]), float(p[5][3:]),
                       [float(v) for v in p[7:]]))
samples = [ln.split()[1:] for ln in log if ln.startswith("S ")]
disp = {}
for f in glob.glob(out + "/a/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "trace_kernel" in r["Kernel_Name"]:
            disp[int(r["Dispatch_Id"])] = [int(r["Start_Timestamp"]), int(r["End_Timestamp"]), None]
for f in glob.glob(out + "/a/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = int(r["Dispatch_Id"])
        if d in disp and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            disp[d][2] = float(r["Counter_Value"])
order = sorted(disp)
print("# effective engine clock per launch = GRBM_GUI_ACTIVE / (end - start) of the dispatch "
      "(rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE), next to the launch's HIP-event time")
k = 0
for cfg, r, t0, t1, ms in rounds:
    rows = order[k:k + len(ms)]
    k += len(ms)
    if len(rows) < len(ms):
        break
    dur = np.array([(disp[d][1] - disp[d][0]) * 1e-3 for d in rows])      # us
    act = np.array([disp[d][2] or np.nan for d in rows])
    mhz = act / dur
    inside = [s for s in samples if t0 - 0.05 <= float(s[0]) <= t1 + 0.05]
    print(f"\n## {cfg} round {r}: {len(ms)} launches after 1 s idle; max/min of launches 6-25 = "
          f"{max(ms[5:25]) / min(ms[5:25]):.3f}")
    print("launch   event_ms  prof_us   GUI_ACTIVE   eff_MHz")
    for i in range(len(ms)):
        print(f"{i:5d}   {ms[i]:8.4f} {dur[i]:8.1f} {act[i]:12.0f} {mhz[i]:9.1f}")
    if inside:
        print("sysfs samples in this round (t - t0 [ms], sclk, mclk, fclk, power uW, temp mC, freq1 Hz):")
        step = max(len(inside) // 40, 1)
        for s in inside[::step]:
            print(f"   {(float(s[0]) - t0) * 1e3:8.2f}  " + "  ".join(s[1:]))
    corr = np.corrcoef(np.array(ms), mhz)[0, 1] if np.isfinite(mhz).all() else float("nan")
    print(f"correlation(event_ms, eff_MHz) = {corr:.3f};  ms x MHz (work in cycles) min/max = "
          f"{np.nanmin(np.array(ms) * mhz):.0f} / {np.nanmax(np.array(ms) * mhz):.0f}")
