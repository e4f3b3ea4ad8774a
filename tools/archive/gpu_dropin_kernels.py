"""GPU box, under `rocprofv3 --kernel-trace --stats`: 300 calls of a reference-built
DoubleGauss's `Optic.trace_generic` (1e6 rays, scalar field, device tensors) through
`integration.enable()` and nothing else -- the kernel list shows what one drop-in call
launches (ol::raygen_kernel, ol::trace_kernel, one torch fill for `rays.w`)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tests import _live
be = _live.import_reference()
from optiland_amd import integration
be.set_backend("torch"); be.set_device("cuda"); be.set_precision("float32")
integration.enable()
lens, w = _live.build_system("DoubleGauss")
n = 1_000_000
g = torch.Generator(device="cuda").manual_seed(5)
r = torch.rand(n, generator=g, device="cuda").sqrt(); th = 2 * np.pi * torch.rand(n, generator=g, device="cuda")
px, py = r * th.cos(), r * th.sin()
for _ in range(300):
    lens.trace_generic(0.0, 0.7, px, py, w)
torch.cuda.synchronize()
c = lens.ray_tracer._hip_companion
print("calls 300, packs", c.pack_count, "speculative hits", c.speculative_hits, "path", c.last_path)
