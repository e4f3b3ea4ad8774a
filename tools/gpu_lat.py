import os, sys, cProfile, pstats, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optiland_amd import load_system, tracer as tr
t = tr.HipRayTracer(load_system("double_gauss"), "cuda:0", dtype=torch.float32)
px = torch.rand(100, device="cuda:0") * 0.5
py = torch.rand(100, device="cuda:0") * 0.5
for _ in range(20):
    t.trace_generic(0.0, 0.7, px, py, 0.5876)
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    t.trace_generic(0.0, 0.7, px, py, 0.5876)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
