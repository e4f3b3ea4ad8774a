#!/usr/bin/env python
"""GPU box, round 4: what the drop-in gained this round, end to end through the LIVE reference.

 1. `Optic.trace_generic` with per-ray field planes (three fields x the pupil, 1e7 rays) and on
    an apodised optic: one launch (`ol_trace_generate`, ABI 8) vs the two launches of round 3
    (`OPTILAND_HIP_FUSE_GENERATE=0`), fp32 and fp64, bit-identical results checked.
 2. First-call latency of `Optic.trace(num_rays=1825, "hexapolar")` (1e7 points) and of
    "uniform" with a 3568-point side: pupil planes sampled on the device (`ol_pupil_points`)
    vs on the host + upload (`OPTILAND_HIP_DEVICE_PUPIL=0`); then the cached call.
 3. The object protocol on the device: a traced optic pickles, the copy traces bit-identically;
    deep copy; `to_dict` / `from_dict`.
 4. The reference's own analyses with the seams (as round 3) for continuity.
Writes gpurun_out/r04_dropin.json.
"""
import copy
import json
import os
import pickle
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


def once(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) * 1e3, 3), out


def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
    be = _live.import_reference()
    from optiland_amd import analysis_seams, integration
    from optiland_amd import tracer as tr
    doc = {"device": torch.cuda.get_device_name(0), "rays": n}
    be.set_backend("torch")
    be.set_device("cuda")

    def pupil(m, dt):
        g = torch.Generator(device="cuda").manual_seed(5)
        r = torch.rand(m, generator=g, device="cuda", dtype=torch.float32).sqrt()
        th = 2 * np.pi * torch.rand(m, generator=g, device="cuda", dtype=torch.float32)
        return (r * th.cos()).to(dt), (r * th.sin()).to(dt)

    # ------------------------------------------------ 1. per-ray fields / apodisation
    for precision, dt in (("float32", torch.float32), ("float64", torch.float64)):
        be.set_precision(precision)
        px, py = pupil(n, dt)
        hy = torch.tensor([0.0, 0.7, 1.0], device="cuda", dtype=dt).repeat_interleave(
            (n + 2) // 3)[:n].contiguous()
        hx = torch.zeros_like(hy)
        row = {}
        keep = {}
        for label, env in (("two_launch", "0"), ("one_launch", "1")):
            os.environ["OPTILAND_HIP_FUSE_GENERATE"] = env
            integration.enable()
            lens, w = _live.build_system("DoubleGauss")
            row[f"field_planes_{label}_ms"] = wall(
                lambda: lens.trace_generic(hx, hy, px, py, w), 10)
            r = lens.trace_generic(hx, hy, px, py, w)
            keep[label] = (r.x.clone(), r.y.clone(), lens.surfaces.surfaces[5].y.clone())
            # scalar field for reference (the round-3 single launch)
            row[f"scalar_field_{label}_ms"] = wall(
                lambda: lens.trace_generic(0.0, 0.7, px, py, w), 10)
            # apodised pupil
            lens.updater.set_apodization("GaussianApodization", sigma=0.8)
            row[f"apodized_{label}_ms"] = wall(lambda: lens.trace_generic(0.0, 0.7, px, py, w), 10)
            r = lens.trace_generic(0.0, 0.7, px, py, w)
            keep[label + "_apod"] = (r.x.clone(), r.i.clone())
            comp = integration.hip_tracer_of(lens)
            row[f"path_{label}"] = comp.last_path
            integration.disable()
            del lens, r
            torch.cuda.empty_cache()
        os.environ["OPTILAND_HIP_FUSE_GENERATE"] = "1"
        row["bit_identical"] = bool(
            all(torch.equal(a.nan_to_num(), b.nan_to_num())
                for a, b in zip(keep["two_launch"], keep["one_launch"]))
            and all(torch.equal(a.nan_to_num(), b.nan_to_num())
                    for a, b in zip(keep["two_launch_apod"], keep["one_launch_apod"])))
        doc["trace_generic_1e7_" + precision] = row
        del keep, px, py, hx, hy
        torch.cuda.empty_cache()

    # ------------------------------------------------ 2. first-call latency of named samplers
    be.set_precision("float32")
    lat = {}
    for name, num in (("hexapolar", 1825), ("uniform", 3568)):
        for label, env in (("host_sampling_and_upload", "0"), ("device_sampling", "1")):
            os.environ["OPTILAND_HIP_DEVICE_PUPIL"] = env
            tr._PUPIL_PLANES.clear()
            integration.enable()
            lens, w = _live.build_system("DoubleGauss")
            lens.trace(0.0, 0.7, w, 6, "hexapolar")        # engine, tables, caches warm
            first, rays = once(lambda: lens.trace(0.0, 0.7, w, num, name))
            lat[f"{name}_{num}_{label}_first_call_ms"] = first
            lat[f"{name}_{num}_points"] = int(rays.x.numel())
            lat[f"{name}_{num}_{label}_cached_call_ms"] = wall(
                lambda: lens.trace(0.0, 0.7, w, num, name), 8, warm=1)
            integration.disable()
            del lens, rays
            torch.cuda.empty_cache()
    os.environ["OPTILAND_HIP_DEVICE_PUPIL"] = "1"
    tr._PUPIL_PLANES.clear()
    doc["first_call_latency"] = lat

    # ------------------------------------------------ 3. object protocol on the device
    be.set_precision("float64")
    proto = {}
    for mode in ("install", "enable", "enable_lazy"):
        lens, w = _live.build_system("CookeTriplet")
        if mode == "install":
            integration.install(lens)
        else:
            integration.enable(lazy_records=(mode == "enable_lazy"))
        r0 = lens.trace(0.0, 0.7, w, 32, "hexapolar")
        t_p, blob = once(lambda: pickle.dumps(lens))
        twin = pickle.loads(blob)
        r1 = twin.trace(0.0, 0.7, w, 32, "hexapolar")
        deep = copy.deepcopy(lens)
        r2 = deep.trace(0.0, 0.7, w, 32, "hexapolar")
        from optiland.optic import Optic
        rebuilt = Optic.from_dict(lens.to_dict())
        r3 = rebuilt.trace(0.0, 0.7, w, 32, "hexapolar")
        proto[mode] = {
            "pickle_bytes": len(blob), "pickle_ms": t_p,
            "pickle_copy_bit_identical": bool(torch.equal(r0.y, r1.y) and torch.equal(r0.opd, r1.opd)),
            "deepcopy_bit_identical": bool(torch.equal(r0.y, r2.y)),
            "from_dict_max_abs_diff": float((r0.y - r3.y).abs().max()),
            "copy_path": integration.hip_tracer_of(twin).last_path,
            "surfaces_recorded_in_copy": int(twin.surfaces.y.shape[0]),
        }
        integration.disable()
        integration.uninstall(lens) if mode == "install" else None
    doc["object_protocol_on_device"] = proto

    # ------------------------------------------------ 4. the reference's analyses (continuity)
    from optiland import analysis
    from optiland.psf import FFTPSF
    from optiland.wavefront import OPD
    integration.enable()
    for k_ in analysis_seams.STATS:
        analysis_seams.STATS[k_] = 0
    lens, w = _live.build_system("CookeTriplet")
    row = {}
    row["SpotDiagram_6_rings_ms"] = wall(lambda: analysis.SpotDiagram(lens), 10)
    row["SpotDiagram_400_rings_ms"] = wall(
        lambda: analysis.SpotDiagram(lens, num_rings=400).rms_spot_radius(), 5)
    row["EncircledEnergy_1e6_ms"] = wall(
        lambda: analysis.EncircledEnergy(lens, num_rays=1_000_000).centroid(), 5)
    # (round 3 wrote `num_rings=256` here: `OPD` takes `num_rays` -- the keyword fell into
    # **kwargs and the run used the default 15 rings.  Both sizes now, under their own names.)
    row["OPD_15_rings_ms"] = wall(lambda: OPD(lens, (0.0, 1.0), w).rms(), 5)
    row["OPD_256_rings_ms"] = wall(lambda: OPD(lens, (0.0, 1.0), w, num_rays=256).rms(), 5)
    for strat in ("centroid", "best_fit"):
        try:
            row[f"OPD_256_rings_{strat}_ms"] = wall(
                lambda: OPD(lens, (0.0, 1.0), w, num_rays=256, strategy=strat).rms(), 5)
        except Exception as exc:  # noqa: BLE001
            row[f"OPD_256_rings_{strat}_ms"] = repr(exc)
    row["FFTPSF_1024_ms"] = wall(
        lambda: FFTPSF(lens, (0.0, 1.0), w, num_rays=512, grid_size=1024).strehl_ratio(), 5)
    row["seam_calls"] = dict(analysis_seams.STATS)
    row["seams_skipped"] = dict(analysis_seams.SKIPPED)
    doc["reference_analyses_cooke_fp64_with_seams"] = row
    import cProfile
    import io
    import pstats
    buf = io.StringIO()
    for nm, fn in (("OPD(256 rings, chief_ray).rms()",
                    lambda: OPD(lens, (0.0, 1.0), w, num_rays=256).rms()),
                   ("OPD(256 rings, centroid).rms()",
                    lambda: OPD(lens, (0.0, 1.0), w, num_rays=256, strategy="centroid").rms()),
                   ("OPD(256 rings, best_fit).rms()",
                    lambda: OPD(lens, (0.0, 1.0), w, num_rays=256, strategy="best_fit").rms())):
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        pr.disable()
        buf.write(f"\n===== {nm} x5\n")
        pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(35)
    with open(os.path.join(ROOT, "gpurun_out", "r04_opd_profile.txt"), "w") as fh:
        fh.write(buf.getvalue())
    integration.disable()

    be.set_precision("float64")
    be.set_device("cpu")
    be.set_backend("numpy")
    out = os.path.join(ROOT, "gpurun_out", "r04_dropin.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
