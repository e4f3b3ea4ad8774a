#!/usr/bin/env python
"""CPU, build container only (round 5): the reference's OWN classes on random lenses, with and
without the drop-in.

For every seed the lens of tests/test_reference_fuzz.py:build_random_lens is built twice through
the reference's public API: once under the NumPy backend (the reference as it is), once under the
torch backend (cpu) with `integration.enable(force=True)` and the product's engine class on the
HOST build of the kernel source (tests/_hostmath.py).  Families:
  trace, trace_generic (64-element field / pupil arrays), trace_distributions (every deterministic
  pupil sampler), trace_wavelengths (every wavelength in turn, twice round), aimed (iterative and
  robust ray aiming), sg_trace (the caller's own rays through SurfaceGroup.trace, whole and with
  skip), edit_loop (edit through the updater / trace, six rounds), spot, ee (SpotDiagram,
  EncircledEnergy), opd / opd_centroid / opd_best_fit / opd_detrended (OPD with each reference
  strategy, tilt removal), fftpsf;
  `others`: RayFan, PupilAberration, Distortion, GridDistortion, FieldCurvature,
  RmsSpotSizeVsField, RmsWavefrontErrorVsField, ThroughFocusSpotDiagram.
Prints, per family, how many lenses were compared, the worst relative difference (to the largest
value of the reference's result), every lens over the limit -- 1e-6; wavefront families on lenses
with a Newton surface 2e-3 (the reference stops its iteration at 1e-6 mm) -- and which seams
declined.  Results and triage of every flag: profiles/r05_seam_fuzz.txt.

    python tools/seam_fuzz.py LO HI [polarised] [others]
    OL_FUZZ_KINDS=standard      conic surfaces only (no stop tolerance between the two sides)
    OL_FUZZ_PRECISION=float32   the torch backend at float32 (limit 1e-4)
    OL_FUZZ_FAMILIES=a,b,...    only these families
    OL_FUZZ_REFERENCE_NEWTON=1  `enable(reference_newton=True)` (round 6): the reference's own
                                batch-global Newton stop rule -- Newton lenses are then held to
                                the conic lenses' limit, 1e-6, in every family
"""
import os
import sys
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "refshim"), "/root/reference"]
warnings.filterwarnings("ignore")
import importlib.util  # noqa: E402

import optiland.backend as be  # noqa: E402

be.set_backend("numpy")
spec = importlib.util.spec_from_file_location(
    "rf", os.path.join(ROOT, "tests", "test_reference_fuzz.py"))
rf = importlib.util.module_from_spec(spec)
spec.loader.exec_module(rf)
if os.environ.get("OL_FUZZ_KINDS"):
    # e.g. OL_FUZZ_KINDS=standard: conic surfaces only (decentres, tilts, mirrors, apertures,
    # coatings, finite objects as before) -- no Newton stop tolerance between the two sides, so
    # every family is held to 1e-6
    rf.KINDS = os.environ["OL_FUZZ_KINDS"].split(",")
import optiland_amd.tracer as tr  # noqa: E402
from optiland_amd import analysis_seams, integration  # noqa: E402
from tests import _hostmath as hm  # noqa: E402

_cls = hm.make_engine_class()
tr._make_engine = lambda table, device: _cls(table, device)


def _np(a):
    return np.asarray(be.to_numpy(a), dtype=np.float64)


PRECISION = os.environ.get("OL_FUZZ_PRECISION", "float64")   # float32: the fp32 kernels (1e-4)
OTHERS = "others" in sys.argv[3:]   # the reference's other analyses instead of the ten families
POLARISED = "polarised" in sys.argv[3:]   # only the polarised lenses: spot, ee
REFERENCE_NEWTON = os.environ.get("OL_FUZZ_REFERENCE_NEWTON", "0") == "1"


def families(lens, polarised=False, only=None):
    from optiland import analysis
    from optiland.psf import FFTPSF
    from optiland.wavefront import OPD
    out = {}
    w = lens.primary_wavelength

    def spot():
        s = analysis.SpotDiagram(lens, num_rings=5)
        return np.array([[float(_np(v)) for v in f] for f in s.rms_spot_radius()] +
                        [[float(_np(v)) for v in f] for f in s.geometric_spot_radius()])

    def ee():
        # (the curve itself is only formed while plotting, encircled_energy.py:135-168: the
        # centroids and the hits it is formed from are compared)
        e = analysis.EncircledEnergy(lens, num_rays=6, distribution="hexapolar", num_points=16)
        cen = np.array([[float(_np(c[0])), float(_np(c[1]))] for c in e.centroid()]).ravel()
        hits = np.concatenate([np.concatenate([_np(d.x), _np(d.y), _np(d.intensity)])
                               for f in e.data for d in f])
        return np.concatenate([cen, hits])

    def opd(**kw):
        return lambda: _np(OPD(lens, (0.0, 0.7), w, num_rays=5, **kw).get_data((0.0, 0.7), w).opd)

    def psf():
        return _np(FFTPSF(lens, (0.0, 0.5), w, num_rays=32, grid_size=64).psf)

    def edit_loop():
        # an optimiser's pattern: edit the prescription through the reference's updater, trace,
        # edit, trace ... (the drop-in's change detector, re-packing and ol_system_update)
        rng_ = np.random.default_rng(4242)
        surfs = lens.surface_group.surfaces
        parts = []
        for _step in range(6):
            kind = int(rng_.integers(0, 4))
            idx = int(rng_.integers(1, len(surfs) - 1)) if len(surfs) > 2 else 1
            g_ = surfs[idx].geometry
            if kind == 0 and np.isfinite(float(_np(getattr(g_, "radius", np.inf)))):
                lens.updater.set_radius(float(_np(g_.radius)) * (1 + 1e-3 * rng_.standard_normal()), idx)
            elif kind == 1 and hasattr(g_, "k"):
                lens.updater.set_conic(float(_np(g_.k)) + 1e-3 * rng_.standard_normal(), idx)
            elif kind == 2:
                th = float(_np(surfs[idx + 1].geometry.cs.z)) - float(_np(g_.cs.z)) if idx + 1 < len(surfs) else 0.0
                if np.isfinite(th):
                    lens.updater.set_thickness(th * (1 + 1e-3 * rng_.standard_normal()) + 1e-4, idx)
            else:
                lens.updater.scale_system(1.0 + 1e-3 * rng_.standard_normal())
            r = lens.trace(0.2, 0.6, w, 3, "hexapolar")
            parts.append(np.nan_to_num(np.stack([_np(getattr(r, k)) for k in
                                                 ("x", "y", "z", "L", "M", "N", "i", "opd")]),
                                       nan=-7.0).ravel())
            parts.append(np.nan_to_num(_np(lens.surfaces.z)[1:], nan=-7.0, posinf=-8.0, neginf=-9.0).ravel())
        return np.concatenate(parts)

    def aimed():
        # iterative / robust ray aiming: the reference builds the rays (its aimers trace to the
        # stop surface by surface), the surface loop enters the HIP path through SurfaceGroup.trace
        parts = []
        try:
            for mode in ("iterative", "robust"):
                lens.ray_tracer.set_aiming(mode, 10, 1e-9)
                r = lens.trace(0.0, 0.7, w, 3, "hexapolar")
                parts.append(np.nan_to_num(np.stack([_np(getattr(r, k)) for k in
                                                     ("x", "y", "z", "L", "M", "N", "i", "opd")]),
                                           nan=-7.0).ravel())
        finally:
            lens.ray_tracer.set_aiming("paraxial", 10, 1e-6)
        return np.concatenate(parts)

    def trace_wavelengths():
        # every wavelength of the lens in turn, twice round (the drop-in keeps one packed table
        # and one device system per wavelength)
        parts = []
        for _round in range(2):
            for wl in lens.wavelengths.get_wavelengths():
                r = lens.trace(0.1, 0.8, wl, 4, "hexapolar")
                parts.append(np.nan_to_num(np.stack([_np(getattr(r, k)) for k in
                                                     ("x", "y", "z", "L", "M", "N", "i", "opd")]),
                                           nan=-7.0).ravel())
        return np.concatenate(parts)

    def trace():
        r = lens.trace(0.3, -0.5, w, 5, "hexapolar")
        rec = np.stack([_np(getattr(lens.surfaces, k))[1:] for k in ("x", "y", "z", "L", "M", "N", "intensity", "opd")])
        return np.concatenate([np.nan_to_num(np.stack([_np(getattr(r, k)) for k in ("x", "y", "z", "L", "M", "N", "i", "opd")]), nan=-7.0).ravel(),
                               np.nan_to_num(rec, nan=-7.0, posinf=-8.0, neginf=-9.0).ravel()])

    def trace_distributions():
        # (the deterministic samplers of distribution.py; "random" / "sobol" draw)
        parts = []
        for dist, n_ in (("uniform", 9), ("line_x", 11), ("line_y", 8), ("cross", 7), ("ring", 12),
                         ("positive_line_x", 5), ("hexapolar", 1)):
            r = lens.trace(-0.4, 0.6, w, n_, dist)
            parts.append(np.nan_to_num(np.stack([_np(getattr(r, k)) for k in
                                                 ("x", "y", "z", "L", "M", "N", "i", "opd")]),
                                       nan=-7.0).ravel())
        return np.concatenate(parts)

    def sg_trace():
        # the caller's own rays through SurfaceGroup.trace (in place), whole and with skip
        rng_ = np.random.default_rng(9)
        n = 48
        rr, th = np.sqrt(rng_.random(n)) * 0.8, 2 * np.pi * rng_.random(n)
        parts = []
        for skip in (0, 2):
            # (scalar field: with per-ray field ARRAYS the reference's torch backend fails in the
            # object-space-telecentric branch of the aimer, `be.full_like(Px, z)`, by itself)
            rays = lens.ray_tracer.ray_generator.generate_rays(0.2, -0.4, be.array(rr * np.cos(th)),
                                                               be.array(rr * np.sin(th)), w)
            if skip:
                # (start where a trace of the first `skip` surfaces leaves the rays)
                for surf in lens.surface_group.surfaces[:skip]:
                    surf.trace(rays)
            out = lens.surface_group.trace(rays, skip) if skip else lens.surface_group.trace(rays)
            assert out is rays
            parts.append(np.nan_to_num(np.stack([_np(getattr(rays, k)) for k in
                                                 ("x", "y", "z", "L", "M", "N", "i", "opd")]),
                                       nan=-7.0).ravel())
            parts.append(np.nan_to_num(_np(lens.surfaces.x)[max(skip, 1):], nan=-7.0, posinf=-8.0,
                                       neginf=-9.0).ravel())
        return np.concatenate(parts)

    def trace_generic():
        rng_ = np.random.default_rng(5)
        n = 64
        hx, hy = be.array(rng_.uniform(-0.5, 0.5, n)), be.array(rng_.uniform(-1, 1, n))
        rr, th = np.sqrt(rng_.random(n)) * 0.9, 2 * np.pi * rng_.random(n)
        r = lens.trace_generic(hx, hy, be.array(rr * np.cos(th)), be.array(rr * np.sin(th)), w)
        return np.nan_to_num(np.stack([_np(getattr(r, k)) for k in ("x", "y", "z", "L", "M", "N", "i", "opd")]), nan=-7.0).ravel()

    def flat(v):
        """Every number of an analysis's `data` (dicts in key order, lists, arrays), as one vector."""
        if isinstance(v, dict):
            return np.concatenate([flat(v[k]) for k in v] or [np.zeros(0)])
        if isinstance(v, (list, tuple)):
            return np.concatenate([flat(x) for x in v] or [np.zeros(0)])
        if hasattr(v, "intensity") and hasattr(v, "x"):      # SpotData
            return flat((v.x, v.y, v.intensity))
        try:
            return np.nan_to_num(_np(v), nan=-7.0, posinf=-8.0, neginf=-9.0).ravel()
        except Exception:  # noqa: BLE001
            try:
                return np.nan_to_num(np.array([float(v)]), nan=-7.0, posinf=-8.0, neginf=-9.0)
            except Exception:  # noqa: BLE001 - a non-numeric leaf (names, enums)
                return np.zeros(0)

    # the reference's other consumers of Optic.trace / trace_generic (their own call patterns:
    # scalar and array coordinates, line distributions, the image surface moved between traces)
    others = {
        "RayFan": lambda: flat(analysis.RayFan(lens, num_points=17).data),
        "Distortion": lambda: flat(analysis.Distortion(lens, num_points=16).data),
        "GridDistortion": lambda: flat(analysis.GridDistortion(lens, num_points=5).data),
        "FieldCurvature": lambda: flat(analysis.FieldCurvature(lens, num_points=16).data),
        "RmsSpotSizeVsField": lambda: flat(analysis.RmsSpotSizeVsField(lens, num_fields=8, num_rings=4).data),
        "RmsWavefrontErrorVsField": lambda: flat(analysis.RmsWavefrontErrorVsField(
            lens, num_fields=6, num_rays=5).data),
        "PupilAberration": lambda: flat(analysis.PupilAberration(lens, num_points=17).data),
        # (moves the image surface between its traces: the change detector's business)
        "ThroughFocusSpot": lambda: flat([[[(d.x, d.y, d.intensity) for d in f] for f in step]
                                          for step in analysis.ThroughFocusSpotDiagram(
                                              lens, delta_focus=0.05, num_steps=3, num_rings=3).results]),
    }

    todo = {"trace": trace, "trace_wavelengths": trace_wavelengths, "aimed": aimed, "sg_trace": sg_trace,
            "edit_loop": edit_loop, "trace_distributions": trace_distributions,
            "trace_generic": trace_generic, "spot": spot, "ee": ee, "opd": opd(), "opd_centroid": opd(strategy="centroid"),
            "opd_best_fit": opd(strategy="best_fit"), "opd_detrended": opd(remove_tilt=True),
            "fftpsf": psf}
    todo["edit_loop"] = todo.pop("edit_loop")   # last: it edits the lens
    if OTHERS:
        todo = others
    if polarised:  # (wavefronts of polarised systems are not part of the seams)
        todo = {k: v for k, v in todo.items() if k in ("trace", "aimed", "edit_loop", "trace_wavelengths", "sg_trace", "trace_distributions", "trace_generic", "spot", "ee")
                or (OTHERS and not k.startswith("RmsWavefront"))}
    if only is None and os.environ.get("OL_FUZZ_FAMILIES"):
        only = set(os.environ["OL_FUZZ_FAMILIES"].split(","))
    if only:
        todo = {k: v for k, v in todo.items() if k in only}
    for k, fn in todo.items():
        try:
            with np.errstate(all="ignore"):
                out[k] = fn()
        except Exception as e:  # noqa: BLE001
            out[k] = e
    return out


lo, hi = int(sys.argv[1]), int(sys.argv[2])
stats = {}
bad = []
declined = {}
for seed in range(lo, hi):
    be.set_backend("numpy")
    lens, _rng = rf.build_random_lens(seed, be)
    polarised = lens.polarization != "ignore"
    if polarised and not POLARISED:
        continue
    if POLARISED and not polarised:
        continue
    want = families(lens, polarised)
    from optiland.geometries.newton_raphson import NewtonRaphsonGeometry
    nr = sum(1 for s in lens.surface_group.surfaces
             if isinstance(s.geometry, NewtonRaphsonGeometry))
    be.set_backend("torch")
    be.set_device("cpu")
    be.set_precision(PRECISION)
    integration.enable(force=True, analyses=True, reference_newton=REFERENCE_NEWTON)
    for k in analysis_seams.STATS:
        analysis_seams.STATS[k] = 0
    try:
        lens2, _ = rf.build_random_lens(seed, be)
        got = families(lens2, polarised)
        for k, v in analysis_seams.STATS.items():
            if k.endswith("_fallback") and v:
                declined[k] = declined.get(k, 0) + v
    finally:
        integration.disable()
        be.set_backend("numpy")
    for k in want:
        a, b = got[k], want[k]
        if isinstance(a, Exception) or isinstance(b, Exception):
            # (both raising is agreement: the reference's two backends name the same failure
            # differently -- max() of an empty spot is a ValueError in NumPy, a RuntimeError in torch)
            if isinstance(a, Exception) != isinstance(b, Exception):
                bad.append((seed, k, nr, "raised on one side: %r | %r" % (a, b)))
            else:
                stats.setdefault((k + " (both raise)", nr > 0), [0, 0.0])[0] += 1
            continue
        if a.size == 0 and b.size == 0:
            continue
        if a.shape != b.shape or not np.isfinite(b).all():
            if a.shape != b.shape:
                bad.append((seed, k, nr, f"shapes {a.shape} {b.shape}"))
            continue
        if not np.isfinite(a).all():
            bad.append((seed, k, nr, "NaN with the drop-in only"))
            continue
        e = float(np.max(np.abs(a - b)) / max(1.0, float(np.abs(b).max())))
        st = stats.setdefault((k, nr > 0), [0, 0.0])
        st[0] += 1
        st[1] = max(st[1], e)
        # ray-level families: 1e-6 whatever the lens (the Newton stop tolerance, 1e-6 mm, is 1e-8 of
        # these maps); wavefront families on Newton lenses: 2e-3 (1e-6 mm are 2e-3 waves per surface)
        limit = 1e-6 if (nr == 0 or REFERENCE_NEWTON
                         or not k.startswith(("opd", "fftpsf", "RmsWavefront"))) else 2e-3
        if PRECISION == "float32":
            limit = max(limit, 1e-4)       # BASELINE.json: fp32 within 1e-4
        if e > limit:
            bad.append((seed, k, nr, e))
print("seeds", lo, hi)
for (k, newton), (n, worst) in sorted(stats.items()):
    print(f"  {k:14s} {'Newton lenses' if newton else 'conic lenses ':13s} compared {n:4d}  worst {worst:.3e}")
print("seam fall-backs:", declined or "none")
print("over 1e-6 (conic lenses) / 2e-3 (Newton lenses), or one-sided:", len(bad))
for b_ in bad[:400]:
    print("   ", b_)
