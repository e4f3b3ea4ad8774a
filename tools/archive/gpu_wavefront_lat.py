"""GPU box: wall time of the wavefront consumers, fused (`ol_trace_opd` + `ol_pupil_fill`)
against the un-fused chain (ol_generate_rays -> record-all ol_trace -> ol_wavefront_opd ->
torch reductions / scatter), same process, alternating.  Output -> gpurun_out/wavefront_lat.txt"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from optiland_amd import load_system, tracer as tr
from optiland_amd.wavefront import FFTPSF, OPD

out = []


def timeit(label, fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / reps * 1e3
    out.append(f"{label:58s} {ms:9.3f} ms")
    print(out[-1])
    return ms


for name, field, wl in (("cooke_generic", (0.0, 1.0), 0.55), ("double_gauss", (0.0, 0.7), 0.5876),
                        ("rc_asphere", (0.0, 1.0), 0.55)):
    t = tr.HipRayTracer(load_system(name), "cuda:0", dtype=torch.float64)
    out.append(f"# {name}")
    for rounds in range(2):
        for fused in (True, False):
            tag = "fused" if fused else "un-fused"
            timeit(f"OPD(15 rings).rms()                 [{tag}]",
                   lambda: OPD(t, field, wl, fused=fused).rms())
            timeit(f"OPD(15 rings, remove_tilt).rms()    [{tag}]",
                   lambda: OPD(t, field, wl, remove_tilt=True, fused=fused).rms())
            timeit(f"FFTPSF(num_rays=128)                [{tag}]",
                   lambda: FFTPSF(t, field, wl, num_rays=128, fused=fused))
            timeit(f"FFTPSF(num_rays=512)                [{tag}]",
                   lambda: FFTPSF(t, field, wl, num_rays=512, fused=fused), reps=15)
            timeit(f"FFTPSF(num_rays=2048: 1024^2 samples) [{tag}]",
                   lambda: FFTPSF(t, field, wl, num_rays=2048, fused=fused), reps=5)
    t.engine.close()
os.makedirs("gpurun_out", exist_ok=True)
open("gpurun_out/wavefront_lat.txt", "w").write("\n".join(out) + "\n")
