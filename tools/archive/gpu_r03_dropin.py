#!/usr/bin/env python
"""GPU box, round 3: the drop-in end to end after (a) ray generation fused into the record-all
launch (`ol_trace_generate`), (b) lazy records, (c) the host first-order model / read-back
cache in the packer, (d) the fused kernels behind the reference's own analysis classes.
Writes gpurun_out/r03_dropin.json.

    python tools/gpu_r03_dropin.py [--rays 1e7]
"""
import argparse
import json
import os
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
    ap = argparse.ArgumentParser()
    ap.add_argument("--rays", type=float, default=1e7)
    args = ap.parse_args()
    n = int(args.rays)
    be = _live.import_reference()
    from optiland_amd import analysis_seams, integration, packer
    doc = {"device": torch.cuda.get_device_name(0), "rays": n}
    be.set_backend("torch")
    be.set_device("cuda")

    def pupil(m, dt):
        g = torch.Generator(device="cuda").manual_seed(5)
        r = torch.rand(m, generator=g, device="cuda", dtype=torch.float32).sqrt()
        th = 2 * np.pi * torch.rand(m, generator=g, device="cuda", dtype=torch.float32)
        return (r * th.cos()).to(dt), (r * th.sin()).to(dt)

    # ------------------------------------------------ 1. Optic.trace_generic at full size
    for precision, dt in (("float32", torch.float32), ("float64", torch.float64)):
        be.set_precision(precision)
        px, py = pupil(n, dt)
        row = {}
        for label, env, kw in (("two_launch", "0", {}), ("fused_generate", "1", {}),
                               ("lazy_records", "1", {"lazy_records": True})):
            os.environ["OPTILAND_HIP_FUSE_GENERATE"] = env
            integration.enable(**kw)
            lens, w = _live.build_system("DoubleGauss")
            row[label + "_ms"] = wall(lambda: lens.trace_generic(0.0, 0.7, px, py, w), 12)
            if kw:
                # a consumer that does read an interior surface afterwards
                def read():
                    lens.trace_generic(0.0, 0.7, px, py, w)
                    return lens.surfaces.surfaces[3].x
                row["lazy_then_read_surface_ms"] = wall(read, 6)
                # one that reads the image surface through the returned rays only
                def img():
                    r = lens.trace_generic(0.0, 0.7, px, py, w)
                    return r.x.sum()
                row["lazy_image_plane_consumer_ms"] = wall(img, 12)
            integration.disable()
        os.environ["OPTILAND_HIP_FUSE_GENERATE"] = "1"
        doc["trace_generic_1e7_" + precision] = row

    # ------------------------------------------------ 2. edit-then-trace loop, 100 rays
    be.set_precision("float32")
    pxs, pys = pupil(100, torch.float32)
    loop = {}
    for label, env in (("reference_paraxial", "0"), ("host_first_order", "1")):
        os.environ["OPTILAND_HIP_HOST_PARAXIAL"] = env
        packer._TENSOR_VALUES.clear()
        integration.enable()
        lens, w = _live.build_system("DoubleGauss")
        lens.trace_generic(0.0, 0.7, pxs, pys, w)
        k = [0]

        def step():
            k[0] += 1
            lens.updater.set_radius(50.0 + 1e-4 * k[0], 2)
            lens.trace_generic(0.0, 0.7, pxs, pys, w)
        loop[label + "_ms_per_iteration"] = wall(step, 100, warm=5)
        comp = lens.ray_tracer._hip_companion
        loop[label + "_packs"] = comp.pack_count
        loop["unchanged_optic_trace_ms"] = wall(lambda: lens.trace_generic(0.0, 0.7, pxs, pys, w), 200)
        integration.disable()
    os.environ["OPTILAND_HIP_HOST_PARAXIAL"] = "1"
    doc["set_radius_then_trace_100_rays"] = loop

    # ------------------------------------------------ 3. the reference's analyses on device
    from optiland import analysis
    from optiland.psf import FFTPSF
    from optiland.wavefront import OPD
    be.set_precision("float64")
    ana = {}
    for label, flag in (("without_seams", False), ("with_seams", True)):
        integration.enable(analyses=flag)
        for k_ in analysis_seams.STATS:
            analysis_seams.STATS[k_] = 0
        lens, w = _live.build_system("CookeTriplet")
        row = {}
        row["SpotDiagram_6_rings_ms"] = wall(lambda: analysis.SpotDiagram(lens), 10)
        row["SpotDiagram_400_rings_ms"] = wall(
            lambda: analysis.SpotDiagram(lens, num_rings=400).rms_spot_radius(), 5)
        row["EncircledEnergy_1e6_ms"] = wall(
            lambda: analysis.EncircledEnergy(lens, num_rays=1_000_000).centroid(), 5)
        row["OPD_256_rings_ms"] = wall(lambda: OPD(lens, (0.0, 1.0), w, num_rings=256).rms(), 5)
        row["FFTPSF_1024_ms"] = wall(
            lambda: FFTPSF(lens, (0.0, 1.0), w, num_rays=512, grid_size=1024).strehl_ratio(), 5)
        row["seam_calls"] = dict(analysis_seams.STATS)
        ana[label] = row
        if flag:  # where the remaining host time of the seamed analyses goes
            import cProfile
            import io
            import pstats
            buf = io.StringIO()
            for nm, fn in (("SpotDiagram(400 rings).rms_spot_radius()",
                            lambda: analysis.SpotDiagram(lens, num_rings=400).rms_spot_radius()),
                           ("OPD(256 rings).rms()",
                            lambda: OPD(lens, (0.0, 1.0), w, num_rings=256).rms()),
                           ("EncircledEnergy(1e6 rays).centroid()",
                            lambda: analysis.EncircledEnergy(lens, num_rays=1_000_000).centroid()),
                           ("FFTPSF(512, 1024).strehl_ratio()",
                            lambda: FFTPSF(lens, (0.0, 1.0), w, num_rays=512,
                                           grid_size=1024).strehl_ratio())):
                pr = cProfile.Profile()
                pr.enable()
                for _ in range(5):
                    fn()
                torch.cuda.synchronize()
                pr.disable()
                buf.write(f"\n===== {nm} x5\n")
                pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(28)
            with open(os.path.join(ROOT, "gpurun_out", "r03_analyses_profile.txt"), "w") as fh:
                fh.write(buf.getvalue())
        integration.disable()
    doc["reference_analyses_cooke_fp64"] = ana

    be.set_precision("float64")
    be.set_device("cpu")
    be.set_backend("numpy")
    out = os.path.join(ROOT, "gpurun_out", "r03_dropin.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
