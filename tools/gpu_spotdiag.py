#!/usr/bin/env python
"""GPU box (rounds 5-6): the reference's `SpotDiagram` / `EncircledEnergy` through the analysis seams,
the fields x wavelengths grid as ONE launch (`ol_trace_spot_batch`) against one launch per cell
(round 4's seam): wall clock per construction, 6 and 400 rings, fp32 and fp64, and where the
time of the small case goes (cProfile).  Writes gpurun_out/r06_spotdiag.json / .txt."""
import cProfile
import io
import json
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from tests import _live  # noqa: E402


def wall(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 4)


def main():
    be = _live.import_reference()
    from optiland import analysis
    from optiland.samples.objectives import CookeTriplet, DoubleGauss
    from optiland_amd import analysis_seams, integration
    be.set_backend("torch")
    be.set_device("cuda")
    doc = {"device": torch.cuda.get_device_name(0)}
    for precision in ("float32", "float64"):
        be.set_precision(precision)
        integration.enable()
        for name, build in (("DoubleGauss", DoubleGauss), ("CookeTriplet", CookeTriplet)):
            lens = build()
            row = {}
            grid = analysis_seams._spot_grid
            for label, fn in (("grid", grid), ("per_cell", lambda self: None)):
                analysis_seams._spot_grid = fn
                row[f"SpotDiagram_6_rings_ms_{label}"] = wall(lambda: analysis.SpotDiagram(lens), 20)
                row[f"SpotDiagram_400_rings_rms_ms_{label}"] = wall(
                    lambda: analysis.SpotDiagram(lens, num_rings=400).rms_spot_radius(), 5)
                row[f"EncircledEnergy_64_rings_ms_{label}"] = wall(
                    lambda: analysis.EncircledEnergy(lens, num_rays=64), 5)
            analysis_seams._spot_grid = grid
            a = analysis.SpotDiagram(lens, num_rings=400)
            row["cells"] = f"{len(a.fields)} fields x {len(a.wavelengths)} wavelengths"
            row["rays_per_cell_400_rings"] = int(a.data[0][0].x.numel())
            row["stats"] = dict(analysis_seams.STATS)
            doc[f"{name}:{precision}"] = row
            print(name, precision, json.dumps(row), flush=True)
        if precision == "float32":
            lens = DoubleGauss()
            for _ in range(5):
                analysis.SpotDiagram(lens)
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(50):
                analysis.SpotDiagram(lens)
            pr.disable()
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
            doc["profile_6_rings_x50"] = s.getvalue()
            print(s.getvalue()[:6000])
            for _ in range(3):
                analysis.SpotDiagram(lens, num_rings=400).rms_spot_radius()
            pr = cProfile.Profile()
            pr.enable()
            for _ in range(20):
                analysis.SpotDiagram(lens, num_rings=400).rms_spot_radius()
            pr.disable()
            s = io.StringIO()
            pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(40)
            doc["profile_400_rings_rms_x20"] = s.getvalue()
            print(s.getvalue()[:9000])
            # ... and where the DEVICE time of one such call goes (kernels, by name)
            from torch.profiler import ProfilerActivity, profile
            with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
                for _ in range(5):
                    analysis.SpotDiagram(lens, num_rings=400).rms_spot_radius()
                torch.cuda.synchronize()
            doc["device_400_rings_rms_x5"] = prof.key_averages().table(
                sort_by="cuda_time_total", row_limit=14, max_name_column_width=60)
            print(doc["device_400_rings_rms_x5"][:6000])
        integration.disable()
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_spotdiag.json"), "w") as f:
        json.dump(doc, f, indent=1)


if __name__ == "__main__":
    main()
