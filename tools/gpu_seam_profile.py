"""Round 6: where the wall time of the reference's analysis constructors goes THROUGH the seams
(torch backend on the device, integration.enable()): wall time per call and a cProfile of 20
calls each -- SpotDiagram is tools/gpu_spotdiag.py; here EncircledEnergy, OPD (the three
reference strategies), FFTPSF, on the double Gauss.  Writes gpurun_out/r06_seam_profile.txt."""
import cProfile
import io
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from tests import _live  # noqa: E402


def wall(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def main():
    be = _live.import_reference()
    from optiland import analysis, wavefront
    from optiland.psf import FFTPSF
    from optiland.samples.objectives import DoubleGauss
    from optiland_amd import integration
    be.set_backend("torch")
    be.set_device("cuda")
    out = []
    for precision in ("float32", "float64"):
        be.set_precision(precision)
        integration.enable()
        lens = DoubleGauss()
        cases = {
            "EncircledEnergy(num_rays=64)": lambda: analysis.EncircledEnergy(lens, num_rays=64),
            "OPD(num_rays=256) chief_ray": lambda: wavefront.OPD(lens, (0, 1), 0.5876, num_rays=256),
            "OPD(num_rays=256) centroid": lambda: wavefront.OPD(lens, (0, 1), 0.5876, num_rays=256,
                                                                strategy="centroid_sphere"),
            "OPD(num_rays=256) best_fit": lambda: wavefront.OPD(lens, (0, 1), 0.5876, num_rays=256,
                                                                strategy="best_fit_sphere"),
            "FFTPSF(num_rays=128, grid 512)": lambda: FFTPSF(lens, (0, 1), 0.5876, num_rays=128,
                                                             grid_size=512),
            "Optic.trace(1024 rays)": lambda: lens.trace(Hx=0, Hy=0.7, wavelength=0.5876,
                                                         num_rays=1024, distribution="uniform"),
        }
        for name, fn in cases.items():
            try:
                ms = wall(fn, 20)
            except Exception as exc:  # noqa: BLE001
                out.append(f"{precision} {name}: {type(exc).__name__}: {exc}")
                continue
            out.append(f"{precision:8s} {name:34s} {ms:8.3f} ms per call")
            print(out[-1], flush=True)
            if precision == "float32":
                pr = cProfile.Profile()
                pr.enable()
                for _ in range(20):
                    fn()
                pr.disable()
                s = io.StringIO()
                pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22)
                out.append(s.getvalue())
        integration.disable()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_seam_profile.txt"), "w") as f:
        f.write("\n".join(out))


if __name__ == "__main__":
    main()
