#!/bin/bash
# What sits in the 35-53 ms between the first timed launch's two events (torchrun, --exchange none,
# 60 steps)?  HIP API + kernel + memory-copy trace of that run.
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
export RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561
timeout 600 rocprofv3 --kernel-trace --hip-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/prof_r04_stall -o stall -- python $R/bench.py --gpus 1 --steps 60 --warmup 20 --settle 0 --no-cpu-baseline --exchange none > $R/gpurun_out/r04_stall_bench.log 2>&1
cd $R
tail -1 gpurun_out/r04_stall_bench.log | cut -c1-300
python - <<'PY'
import csv, glob
base = glob.glob("gpurun_out/prof_r04_stall/**/", recursive=True)
k = [f for f in glob.glob("gpurun_out/prof_r04_stall/**/*kernel_trace.csv", recursive=True)]
h = [f for f in glob.glob("gpurun_out/prof_r04_stall/**/*hip_api_trace.csv", recursive=True)]
m = [f for f in glob.glob("gpurun_out/prof_r04_stall/**/*memory_copy_trace.csv", recursive=True)]
print(k, h, m)
rows = list(csv.DictReader(open(k[0])))
tk = [r for r in rows if "trace_kernel" in r["Kernel_Name"]]
# the timed region: the last 60 trace_kernel dispatches
tk.sort(key=lambda r: int(r["Start_Timestamp"]))
reg = tk[-60:]
t0 = int(reg[0]["Start_Timestamp"]); t_prev_end = int(tk[-61]["End_Timestamp"])
out = open("gpurun_out/r04_stall_trace.txt", "w")
def P(*a):
    s = " ".join(str(x) for x in a); print(s); out.write(s + "\n")
P("# last warm-up kernel ended at", 0, "us; first timed kernel", (t0 - t_prev_end) / 1e3, "us later, ran", (int(reg[0]["End_Timestamp"]) - t0) / 1e3, "us; second started", (int(reg[1]["Start_Timestamp"]) - t_prev_end) / 1e3)
lo, hi = t_prev_end - 2_000_000, int(reg[1]["End_Timestamp"]) + 1_000_000
P("# kernels in the window:")
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if lo <= s <= hi:
        P(f"  K {(s - t_prev_end) / 1e3:10.1f} us +{(e - s) / 1e3:9.1f} us  {r['Kernel_Name'][:90]}")
if m:
    for r in csv.DictReader(open(m[0])):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if lo <= s <= hi:
            P(f"  M {(s - t_prev_end) / 1e3:10.1f} us +{(e - s) / 1e3:9.1f} us  {r.get('Direction', '')} {r.get('Bytes', '')}")
if h:
    P("# HIP API calls longer than 200 us in the window:")
    for r in csv.DictReader(open(h[0])):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if lo <= s <= hi and e - s > 200_000:
            P(f"  A {(s - t_prev_end) / 1e3:10.1f} us +{(e - s) / 1e3:9.1f} us  {r['Function']}")
PY
