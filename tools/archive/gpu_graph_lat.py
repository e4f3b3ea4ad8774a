import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiland_amd import load_system, tracer as tr
from optiland_amd.graph import GraphedTrace
t = tr.HipRayTracer(load_system("double_gauss"), "cuda:0", dtype=torch.float32)
n = 100
px = torch.rand(n, device="cuda:0") * 0.5
py = torch.rand(n, device="cuda:0") * 0.5


def timeit(label, fn, reps=1000):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / reps * 1e6:.1f} us per call")


timeit("eager trace_generic (100 rays, record-all)", lambda: t.trace_generic(0.0, 0.7, px, py, 0.5876))
g = GraphedTrace(t.engine, n, torch.float32, wavelength_index=1)
g.px.copy_(px), g.py.copy_(py), g.hy.fill_(0.7)
timeit("graph replay + status read-back", lambda: g.replay())
timeit("graph replay, status deferred", lambda: g.replay(check=False))
g.check_status()
for n2 in (1000, 100000):
    g2 = GraphedTrace(t.engine, n2, torch.float32, wavelength_index=1)
    timeit(f"graph replay, status deferred, {n2} rays", lambda: g2.replay(check=False))
