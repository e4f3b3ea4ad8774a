#!/usr/bin/env python
"""Scan windows of one big device slab: the record-all kernel and the arithmetic-free store
pattern (`ol_stream_fill`) timed on the SAME 4 GiB window, every `step` GiB -- which windows
are fast, does the cheap probe find them, do they stay fast?

    python tools/gpu_window_scan.py [f32|f64] [slab_GiB] [step_GiB]
"""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402

dtype = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == "f32") else torch.float64
slab_gib = int(sys.argv[2]) if len(sys.argv) > 2 else 96
step = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0
b = 4 if dtype == torch.float32 else 8
dev = torch.device("cuda", 0)
table, hy, _d, wavelength = bench.load_workload("double_gauss")
wl = table.wavelength_index(wavelength)
hip = HipSystem(table, dev)
n = 10_000_000
px, py = bench.make_pupil(n, dtype, 1234, dev)
rows = hip.num_surfaces
stride = hip.record_stride(n, b)
need = rows * 8 * stride * b
slab = torch.empty(slab_gib << 30, dtype=torch.uint8, device=dev)
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
print(f"{dtype} slab {slab_gib} GiB at 0x{slab.data_ptr():x}, window {need / 2**30:.3f} GiB, "
      f"free {torch.cuda.mem_get_info()[0] / 2**30:.1f} GiB of {torch.cuda.mem_get_info()[1] / 2**30:.1f}")


def trace_ms(view, launches=16, warm=4):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(warm + launches)]
    for e0, e1 in ev:
        e0.record()
        hip.trace_generate(px, py, wl, field=(0.0, hy), record=view, defer_status=True)
        e1.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(c) for a, c in ev[warm:]]))


def fill_ms(off, reps=4):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    nb = rows * 8 * stride * b
    for k in range(1 + reps):
        if k == 1:
            e0.record()
        hip.lib.ol_stream_fill(C.c_void_p(slab.data_ptr() + off), nb, b, rows * 8, 0, stream)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


# settle the clocks first
v0 = slab[:need].view(dtype).view(rows, 8, stride)
trace_ms(v0, launches=100)
res = []
off = 0
while off + need <= slab.numel():
    view = slab[off: off + need].view(dtype).view(rows, 8, stride)
    f = fill_ms(off)
    t = trace_ms(view)
    res.append((off, t, f))
    print(f"window +{off / 2**30:7.3f} GiB  trace {t:.4f} ms ({4.24e9 * (b / 4) / (t * 1e-3) / 1e12:.2f} TB/s)"
          f"   fill {f:.4f} ms ({need / (f * 1e-3) / 1e12:.2f} TB/s)", flush=True)
    off += int(step * (1 << 30))
t = np.array([r[1] for r in res]); f = np.array([r[2] for r in res])
print(f"trace: min {t.min():.4f} median {np.median(t):.4f} max {t.max():.4f};  "
      f"fill: min {f.min():.4f} median {np.median(f):.4f};  corr(trace, fill) = {np.corrcoef(t, f)[0, 1]:.3f}")
order = np.argsort(t)
for tag, idx in (("fastest", order[0]), ("second", order[1]), ("slowest", order[-1])):
    off = res[idx][0]
    view = slab[off: off + need].view(dtype).view(rows, 8, stride)
    again = [trace_ms(view, launches=40) for _ in range(2)]
    print(f"{tag} window +{off / 2**30:.3f} GiB again: {again[0]:.4f} {again[1]:.4f} (first pass {res[idx][1]:.4f})")
hip.close()
