import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiland_amd import load_system, tracer as tr
from optiland_amd.analysis import SpotDiagram
for name in ("cooke_generic", "double_gauss"):
    t = tr.HipRayTracer(load_system(name), "cuda:0", dtype=torch.float32)
    for ref in ("chief_ray", "centroid"):
        for _ in range(3):
            s = SpotDiagram(t, reference=ref)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(50):
            s = SpotDiagram(t, reference=ref)
        dt = (time.perf_counter() - t0) / 50
        print(f"{name} SpotDiagram({ref}), {len(s.fields)} fields x {len(s.wavelengths)} wavelengths, 6 rings: {dt*1e3:.3f} ms; rms[0]={s.rms_spot_radius()[0]}")
