"""The C5 launch of bench.py (polarised generating launch, every row recorded, write-only PRT,
no epilogue), LAUNCHES times -- the target of a PC-sampling run (tools/gpu_r06.sh pcsamp)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from optiland_amd import load_system  # noqa: E402
from optiland_amd.engine import HipSystem  # noqa: E402

LAUNCHES = int(os.environ.get("LAUNCHES", "60"))
dtype = torch.float64 if os.environ.get("DTYPE") == "f64" else torch.float32
dev = torch.device("cuda", 0)
t = load_system(os.environ.get("SYSTEM", "zernike_fresnel_fringe"))
n = 10_000_000
g = torch.Generator(device=dev).manual_seed(1)
r = torch.rand(n, generator=g, device=dev).sqrt()
th = 2 * np.pi * torch.rand(n, generator=g, device=dev)
px, py = (r * th.cos()).to(dtype), (r * th.sin()).to(dtype)
hip = HipSystem(t, dev)
rec = hip.alloc_record(n, dtype)
prt = torch.empty((9, n), dtype=dtype, device=dev) if t.uses_polarization else None
for _ in range(LAUNCHES):
    hip.trace_generate(px, py, 0, field=(0.0, 1.0), record=rec, prt=prt, defer_status=True)
torch.cuda.synchronize()
print("done", LAUNCHES)
