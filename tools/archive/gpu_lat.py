import os, sys, cProfile, pstats, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from optiland_amd import load_system, tracer as tr
t = tr.HipRayTracer(load_system("double_gauss"), "cuda:0", dtype=torch.float32)
px = torch.rand(100, device="cuda:0") * 0.5
py = torch.rand(100, device="cuda:0") * 0.5
npx, npy = px.cpu().numpy().astype(np.float64), py.cpu().numpy().astype(np.float64)


def timeit(label, fn, n=500):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / n * 1e6:.1f} us per call")


timeit("trace_generic 100 rays, device pupil", lambda: t.trace_generic(0.0, 0.7, px, py, 0.5876))
timeit("trace_generic 100 rays, numpy pupil", lambda: t.trace_generic(0.0, 0.7, npx, npy, 0.5876))
timeit("trace (hexapolar 6 rings)", lambda: t.trace(0.0, 0.7, 0.5876, 6, "hexapolar"))
timeit("trace_spot (hexapolar 6 rings)", lambda: t.trace_spot(0.0, 0.7, 0.5876, 6, "hexapolar"))
t.record_all = False
timeit("trace_generic record_all=False", lambda: t.trace_generic(0.0, 0.7, px, py, 0.5876))
t.record_all = True
pr = cProfile.Profile()
pr.enable()
for _ in range(300):
    t.trace_generic(0.0, 0.7, px, py, 0.5876)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
