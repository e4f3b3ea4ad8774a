#!/usr/bin/env python
"""Does the speed of the record-all kernel depend on WHICH block of device memory it writes?

One process, one system, one pupil; the record block is allocated again and again (sizes: as the
engine allocates it, padded to a multiple of 1 GiB, carved out of one 64 GiB slab at different
offsets) and the same 60 sustained launches are timed on each.  Prints the virtual address, its
alignment, and the median kernel time per block.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402

dtype = torch.float32 if (len(sys.argv) < 2 or sys.argv[1] == "f32") else torch.float64
b = 4 if dtype == torch.float32 else 8
dev = torch.device("cuda", 0)
table, hy, _d, wavelength = bench.load_workload("double_gauss")
wl = table.wavelength_index(wavelength)
hip = HipSystem(table, dev)
n = 10_000_000
px, py = bench.make_pupil(n, dtype, 1234, dev)
rows = hip.num_surfaces


def timed(record, launches=60, warm=30):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
          for _ in range(warm + launches)]
    for e0, e1 in ev:
        e0.record()
        hip.trace_generate(px, py, wl, field=(0.0, hy), record=record, defer_status=True)
        e1.record()
    torch.cuda.synchronize()
    t = [a.elapsed_time(c) for a, c in ev[warm:]]
    return float(np.median(t)), float(np.min(t))


def align_of(p):
    k = 0
    while p % (1 << (k + 1)) == 0 and k < 40:
        k += 1
    return k


def report(tag, record):
    med, mn = timed(record)
    p = record.data_ptr()
    print(f"{tag:<44} va=0x{p:x} (2^{align_of(p)}-aligned)  stride={record.shape[2]:>9d}  "
          f"median {med:.4f} ms  min {mn:.4f}  -> {4.24e9 * (b / 4) / (med * 1e-3) / 1e12:.2f} TB/s",
          flush=True)


stride = hip.record_stride(n, b)
print(f"dtype {dtype}, engine stride {stride} elements ({stride * b} B), block {rows * 8 * stride * b / 2**30:.3f} GiB")
# 1. the engine's own allocation, again and again (cache emptied in between)
for k in range(4):
    rec = hip.alloc_record(n, dtype)
    report(f"engine block #{k}", rec)
    del rec
    torch.cuda.empty_cache()
# 2. the same with other allocations alive in between (moves the block elsewhere)
ballast = []
for k in range(4):
    ballast.append(torch.empty((3 << 30) + k * (700 << 20), dtype=torch.uint8, device=dev))
    rec = hip.alloc_record(n, dtype)
    report(f"engine block after {len(ballast)} ballast allocation(s)", rec)
    del rec
torch.cuda.empty_cache()
del ballast
torch.cuda.empty_cache()
# 3. carved out of ONE big slab at different offsets (same physical neighbourhood, other alignments)
slab_bytes = 40 << 30
slab = torch.empty(slab_bytes, dtype=torch.uint8, device=dev)
need = rows * 8 * stride * b
for off in (0, 1 << 30, (1 << 30) + (2 << 20), 7 << 30, (13 << 30) + (512 << 20), 20 << 30, 31 << 30):
    view = slab[off: off + need].view(dtype).view(rows, 8, stride)
    report(f"slab + {off / 2**30:.3f} GiB", view)
# 4. unpadded stride (n elements) inside the slab
need2 = rows * 8 * n * b
for off in (0, 5 << 30, 17 << 30):
    view = slab[off: off + need2].view(dtype).view(rows, 8, n)
    report(f"slab + {off / 2**30:.3f} GiB, stride = n", view)
del slab
torch.cuda.empty_cache()
# 5. back to a plain engine block
rec = hip.alloc_record(n, dtype)
report("engine block (end)", rec)
hip.close()
