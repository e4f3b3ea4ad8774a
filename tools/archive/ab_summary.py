#!/usr/bin/env python
"""Per-arm mean / min of the interleaved A/B logs written by tools/gpu_ab_*.sh:
    python tools/ab_summary.py gpurun_out/r03_ab_fetch_levels.txt ...
(percentages relative to the arm called "product" = the library as it was built for that run)."""
import sys, re, collections
for path in sys.argv[1:]:
    d = collections.OrderedDict()
    for ln in open(path):
        m = re.match(r"(\S+) (\S+)\s+kernel_ms=([\d.]+) min=([\d.]+)", ln)
        if m:
            d.setdefault(m.group(1), collections.OrderedDict()).setdefault(m.group(2), []).append((float(m.group(3)), float(m.group(4))))
    print("#", path)
    for tag, arms in d.items():
        base = None
        out = []
        for arm, v in arms.items():
            mean = sum(a for a, _ in v) / len(v); mn = min(b for _, b in v)
            if arm == "product": base = mean
            out.append((arm, mean, mn, len(v)))
        print(f"{tag:<12}", "  ".join(f"{a}={m:.4f}(min {mn:.4f},n{n})" + (f"[{(m/base-1)*100:+.1f}%]" if base and a != 'product' else "") for a, m, mn, n in out))
