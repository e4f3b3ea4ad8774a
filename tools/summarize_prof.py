#!/usr/bin/env python
"""Condense rocprofv3 CSV output (kernel stats + PMC passes) into a short summary."""
import csv
import glob
import os
import sys

out, tag = sys.argv[1], sys.argv[2]


def find(sub, pat):
    g = glob.glob(os.path.join(out, sub, "**", pat), recursive=True)
    return g[0] if g else None


f = find("stats", "*kernel_stats.csv")
if f:
    print("== kernel stats (rocprofv3 --kernel-trace --stats) ==")
    with open(f) as fh:
        rows = list(csv.DictReader(fh))
    for r in rows[:8]:
        print({k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs",
                                             "Percentage", "MinNs", "MaxNs")})
for sub, ctr in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE")):
    f = find(sub, "*counter_collection.csv")
    if not f:
        print(f"no counter file for {ctr}")
        continue
    vals = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            if r.get("Counter_Name") != ctr:
                continue
            vals.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    print(f"== {ctr} per dispatch (raw counter units, KiB per rocprofv3 docs) ==")
    for k, v in vals.items():
        print(f"{k[:60]:60s} n={len(v):3d} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
