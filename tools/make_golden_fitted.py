#!/usr/bin/env python
"""Golden fixtures of the FITTED wavefront references: CentroidStrategy / BestFitStrategy of
the reference (wavefront/strategy.py:287-620), run on bundles the reference itself traced.

Build container only (imports /root/reference):

    python tools/make_golden_fitted.py          # writes tests/golden/wavefront_fitted.npz

Per case: the traced bundle at the image surface (x, y, z, L, M, N, opd, intensity), the pupil
points, and for strategy x reference type x backend what the reference made of it -- centre /
plane point, radius, normal, OPD map in waves, pupil intersection points.  Two bundles per lens:
"clean" (as traced) and "dirty" -- the same rays with NaN positions, zero and negative
intensities and a handful of displaced outliers, so that the validity mask, the clamping of the
weights and the k-sigma trimming all act.  Backends: numpy (its std has ddof 0) and torch
(torch.std: ddof 1).  Pins oracle.wavefront_fit (tests/test_oracle_golden.py) and through it
ol_wavefront_fit (tests/test_wavefront_fit.py).
"""

from __future__ import annotations

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = os.environ.get("OPTILAND_REFERENCE", "/root/reference")
sys.dont_write_bytecode = True
sys.path[:0] = [os.path.join(ROOT, "tests", "refshim"), REF, ROOT]

import numpy as np  # noqa: E402

import optiland.backend as be  # noqa: E402
from optiland.distribution import create_distribution  # noqa: E402
from optiland.rays import RealRays  # noqa: E402
from optiland.samples.objectives import CookeTriplet, DoubleGauss  # noqa: E402
from optiland.wavefront.strategy import BestFitStrategy, CentroidStrategy  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
CASES = (("cooke", CookeTriplet, (0.0, 1.0), 0.55), ("dgauss", DoubleGauss, (0.0, 0.7), 0.5876))
KEYS = ("x", "y", "z", "L", "M", "N", "opd", "i")


def _np(v):
    return np.asarray(be.to_numpy(v), dtype=np.float64)


def dirty(planes, rng):
    """The traced bundle with invalid, unlit, negative-weight and outlying rays."""
    p = {k: v.copy() for k, v in planes.items()}
    n = p["x"].size
    idx = rng.permutation(n)
    p["x"][idx[0:5]] = np.nan            # invalid: not finite
    p["L"][idx[5:7]] = np.inf
    p["i"][idx[7:14]] = 0.0              # invalid: no intensity
    p["i"][idx[14:17]] = -0.25           # valid, weight clamped to 0
    p["i"][idx[17:60]] *= rng.uniform(0.2, 0.9, 43)   # uneven weights
    p["y"][idx[60:66]] += 0.4            # outliers the 3-sigma rule removes
    p["x"][idx[66:68]] -= 0.6
    return p


def run(backend, optic_cls, field, wl, planes, px, py, strategy_cls, reference_type):
    be.set_backend(backend)
    if backend == "torch":
        be.set_precision("float64")
    optic = optic_cls()
    dist = create_distribution("hexapolar")
    dist.generate_points(8)
    dist.x, dist.y = be.array(px), be.array(py)
    strat = strategy_cls(optic, dist, reference_type=reference_type)

    def bundle(*_a, **_k):
        r = RealRays(*[be.array(planes[k]) for k in ("x", "y", "z", "L", "M", "N", "i")], wl)
        r.opd = be.array(planes["opd"])
        return r

    optic.trace = bundle  # compute_wavefront_data traces through the optic: hand it ours
    geo = strat._create_reference_geometry(_tilted(strat, field, bundle()))
    d = strat.compute_wavefront_data(field, wl)
    out = {"opd": _np(d.opd), "pupil": np.stack([_np(d.pupil_x), _np(d.pupil_y), _np(d.pupil_z)]),
           "radius": np.float64(d.radius)}
    if reference_type == "sphere":
        out["center"] = np.array(geo.center, dtype=np.float64)
    else:
        out["center"] = np.array(geo.point, dtype=np.float64)
        out["normal"] = np.array(geo.normal, dtype=np.float64)
    return out


def _tilted(strat, field, rays):
    rays.opd = strat._correct_tilt(field, rays.opd)
    return rays


def main():
    out = {}
    rng = np.random.default_rng(20260926)
    for tag, cls, field, wl in CASES:
        be.set_backend("numpy")
        optic = cls()
        dist = create_distribution("hexapolar")
        dist.generate_points(8)
        rays = optic.trace(*field, wl, None, dist)
        px, py = _np(dist.x), _np(dist.y)
        clean = {k: _np(getattr(rays, k)) for k in KEYS}
        out[f"{tag}_px"], out[f"{tag}_py"] = px, py
        out[f"{tag}_field"], out[f"{tag}_wl"] = np.array(field), np.float64(wl)
        for variant, planes in (("clean", clean), ("dirty", dirty(clean, rng))):
            out[f"{tag}_{variant}_rays"] = np.stack([planes[k] for k in KEYS])
            for sname, scls in (("centroid", CentroidStrategy), ("best_fit", BestFitStrategy)):
                for rtype in ("sphere", "plane"):
                    for backend in ("numpy", "torch"):
                        got = run(backend, cls, field, wl, planes, px, py, scls, rtype)
                        for k, v in got.items():
                            out[f"{tag}_{variant}_{sname}_{rtype}_{backend}_{k}"] = v
                        print(f"{tag} {variant} {sname} {rtype} {backend}: radius="
                              f"{got['radius']:.9g} centre={got['center']} "
                              f"rms={np.sqrt(np.nanmean(got['opd'] ** 2)):.6f}")
    be.set_backend("numpy")
    np.savez_compressed(os.path.join(GOLD, "wavefront_fitted.npz"), **out)


if __name__ == "__main__":
    main()
