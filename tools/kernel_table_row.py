#!/usr/bin/env python
"""One summary row from a gpu_kernel_table.sh output directory."""
import csv
import glob
import json
import os
import sys

out, tag, args = sys.argv[1], sys.argv[2], sys.argv[3]
KERNELS = ("trace_kernel", "spot_trace_kernel", "opd_trace_kernel")


def find(sub, pat):
    g = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return g[0] if g else None


line = None
try:
    with open(os.path.join(out, "stats.log")) as f:
        for ln in f:
            if ln.startswith("{"):
                line = json.loads(ln)
except OSError:
    pass
row = {"tag": tag, "args": args.strip()}
f = find("stats", "*kernel_stats.csv")
if f:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if any(r["Name"].startswith(("void ol::" + k, "ol::" + k)) or ("ol::" + k + "<") in r["Name"]
                   for k in KERNELS):
                row["kernel"] = r["Name"].split("(")[0].replace("void ", "").replace("ol::", "")
                row["calls"] = int(r["Calls"])
                row["avg_us"] = float(r["AverageNs"]) / 1e3
                break
f = find("sq", "*counter_collection.csv")
if f and "kernel" in row:
    vals = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if row["kernel"].split("<")[0] in r["Kernel_Name"] and "<" in r["Kernel_Name"]:
                vals.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for k, v in vals.items():
        row[k] = sum(v) / len(v)
if line:
    n = line["config"]["rays_per_gpu"]
    row["rays"] = n
    row["event_ms"] = line["roofline"]["kernel_ms"]
    row["moved_GB"] = line["roofline"]["moved_bytes"] / 1e9
    if "avg_us" in row:
        row["TBps_moved"] = line["roofline"]["moved_bytes"] / (row["avg_us"] * 1e-6) / 1e12
        row["frac"] = row["TBps_moved"] / 8.0
    for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_SMEM"):
        if k in row:
            # wave-level instruction counts: x 64 lanes / rays = instructions per ray
            row[k.replace("SQ_INSTS_", "") + "_per_ray"] = row[k] * 64.0 / n
    if "SQ_INSTS_VALU" in row and "avg_us" in row:
        # issue bound: a SIMD issues one wave-instruction per 4 cycles at best (64 lanes / 16)
        simds = 256 * 4
        row["valu_issue_ms"] = row["SQ_INSTS_VALU"] * 4.0 / simds / 2.4e9 * 1e3
print(json.dumps({k: (round(v, 4) if isinstance(v, float) else v) for k, v in row.items()}))
