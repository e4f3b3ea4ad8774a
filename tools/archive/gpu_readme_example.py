import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from optiland_amd import load_system
from optiland_amd.tracer import HipRayTracer
from optiland_amd.analysis import SpotDiagram
from optiland_amd.wavefront import FFTPSF
from optiland_amd.graph import GraphedTrace

t = HipRayTracer(load_system("double_gauss"), "cuda:0", dtype=torch.float32)
rays = t.trace(0.0, 0.7, 0.5876, num_rays=64, distribution="hexapolar")
x_all = t.surfaces.x
print(len(rays), tuple(x_all.shape))
spot = SpotDiagram(t)
print(spot.rms_spot_radius()[0], spot.geometric_spot_radius()[0])
g = GraphedTrace(t.engine, n=512)
g.px.uniform_(-0.5, 0.5); g.py.uniform_(-0.5, 0.5); g.hy.fill_(0.7); res = g.replay()
print(tuple(res.record.shape))
psf = FFTPSF(HipRayTracer(load_system("cooke_generic"), "cuda:0", dtype=torch.float64),
             field=(0.0, 0.0), wavelength=0.55, num_rays=128)
print(psf.strehl_ratio())
