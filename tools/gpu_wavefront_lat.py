import os, sys, time, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiland_amd import load_system, tracer as tr
from optiland_amd.wavefront import FFTPSF, OPD, Wavefront
t = tr.HipRayTracer(load_system("cooke_generic"), "cuda:0", dtype=torch.float64)


def timeit(label, fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{label}: {(time.perf_counter() - t0) / reps * 1e3:.3f} ms")


timeit("OPD(15 rings).rms()", lambda: OPD(t, (0.0, 1.0), 0.55).rms())
timeit("FFTPSF(num_rays=128)", lambda: FFTPSF(t, (0.0, 1.0), 0.55, num_rays=128))
timeit("FFTPSF(num_rays=512)", lambda: FFTPSF(t, (0.0, 1.0), 0.55, num_rays=512))
timeit("FFTPSF(num_rays=2048)", lambda: FFTPSF(t, (0.0, 1.0), 0.55, num_rays=2048), reps=5)
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    FFTPSF(t, (0.0, 1.0), 0.55, num_rays=128)
pr.disable(); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
