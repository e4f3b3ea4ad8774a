#!/usr/bin/env python
"""Write ceiling of the record-all store pattern, measured with `ol_stream_fill` (the pattern
with the arithmetic taken out): planes x elements-per-plane x store width, plane stride aligned
to 2 MiB or not.  Interleaved, two rounds.  Prints one line per configuration.

    python tools/gpu_fill_sweep.py
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optiland_amd import _capi  # noqa: E402

lib = _capi.load()
dev = torch.device("cuda", 0)
buf = torch.empty(9 << 30, dtype=torch.uint8, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def run(n, planes, width, reps=8):
    nbytes = n * planes * width
    assert nbytes <= buf.numel()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for k in range(3 + reps):
        if k == 3:
            e0.record()
        rc = lib.ol_stream_fill(C.c_void_p(buf.data_ptr()), nbytes, width, planes, 0x3f800000,
                                stream)
        assert rc == 0, lib.ol_last_error()
    e1.record()
    torch.cuda.synchronize()
    return nbytes / (e0.elapsed_time(e1) / reps * 1e-3) / 1e12


CFG = []
for planes in (104, 41, 13, 1):
    for width in (4, 8, 16):
        for n, tag in ((10_000_000, "stride 4e7 B x w/4 (256 B aligned)"),
                       (10_485_760, "stride 2 MiB-aligned")):
            if n * planes * width <= buf.numel() and n * planes * width >= (1 << 28):
                CFG.append((planes, width, n, tag))
res = {c: [] for c in CFG}
for rnd in range(2):
    for c in (CFG if rnd == 0 else CFG[::-1]):
        res[c].append(run(c[2], c[0], c[1]))
print(f"{'planes':>6} {'width':>5} {'elements/plane':>14} {'GB':>6}  TB/s (two rounds)   layout")
for c in CFG:
    planes, width, n, tag = c
    print(f"{planes:6d} {width:5d} {n:14d} {n * planes * width / 1e9:6.2f}  "
          f"{res[c][0]:.3f} {res[c][1]:.3f}        {tag}")
