"""GPU box: the stand-alone analyses on the real engine vs the same host code on the
oracle-backed engine (tests/_fake_engine.py), over every table under fuzz_tables/:
SpotDiagram (fused spot kernel / planes path when polarised), EncircledEnergy, OPD.
fp64; prints the worst relative differences."""
import glob
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from optiland_amd.analysis import EncircledEnergy, SpotDiagram  # noqa: E402
from optiland_amd.system import SystemTable  # noqa: E402
from optiland_amd.tracer import HipRayTracer  # noqa: E402
from optiland_amd.wavefront import FFTPSF, OPD  # noqa: E402
from oracle import oracle  # noqa: E402
from tests._fake_engine import OracleEngine  # noqa: E402

DEV = os.environ.get("OL_FUZZ_DEVICE", "cuda:0")
_HOST = None
if DEV == "cpu":  # the same comparison with the HOST build of the kernel source as "real" engine
    from tests import _hostmath as _hm  # noqa: E402
    _HOST = _hm.make_engine_class()


def lost_alive_rays(table):
    """Round 5 triage of a flagged lens: rays of the spot diagram's own bundles that one side
    carries to the image ALIVE although its Newton iteration lost them -- the recorded hit is
    not on the surface (|sag(x, y) - z| > 1e-3 mm in the surface's frame) -- or that one side
    reports as NaN at a Newton surface while the other keeps them.  A ray with no root (it
    misses the asphere) wanders for max_iter steps on both sides; whether that ends on a
    finite point or in the square root of a negative number is rounding noise (DESIGN
    section 7): seed 8288, field 1 ray 467 -- the reference hands on a point 1.7 mm off the
    surface, the kernel says NaN; same lens, on-axis ray 62 -- the other way round.
    Returns (reference side, kernel side)."""
    nr_rows = np.nonzero(table.surfaces["max_iter"] > 0)[0]
    if not nr_rows.size:
        return 0, 0
    mf = table.raygen.get("max_field", 0.0) or 1.0
    counts = []
    for real in (False, True):
        eng = OracleEngine(table, "cpu") if not real else \
            (None if _HOST is None else _HOST(table, "cpu"))
        t = HipRayTracer(table, DEV if real else "cpu", dtype=torch.float64, engine=eng)
        lost = 0
        for f in table.fields:
            for w in table.wavelengths:
                with np.errstate(all="ignore"):
                    t.trace(float(f[0] / mf), float(f[1] / mf), float(w), 5, "hexapolar")
                xs, ys, zs = (getattr(t.surfaces, k).double().cpu().numpy() for k in ("x", "y", "z"))
                inten = t.surfaces.intensity.double().cpu().numpy()
                gone = np.zeros(xs.shape[1], dtype=bool)
                for s_i in nr_rows:
                    sf = table.surfaces[s_i]
                    Rm, o_ = np.array(sf["rot"]).reshape(3, 3), np.array(sf["origin"])
                    loc = Rm @ (np.stack([xs[s_i], ys[s_i], zs[s_i]]) - o_[:, None])
                    # alive when it arrives: the intensity recorded at the surface before
                    came = inten[s_i - 1] > 0
                    for j in np.nonzero(came & ~gone)[0]:
                        f_ = oracle.sag(table, int(s_i), float(loc[0, j]), float(loc[1, j])) - loc[2, j]
                        if not abs(f_) < 1e-3:
                            lost += 1
                            gone[j] = True
        counts.append(lost)
        if real:
            t.engine.close()
    return tuple(counts)


worst = {"spot_rms": 0.0, "spot_geo": 0.0, "spot_centroid": 0.0, "ee": 0.0, "opd": 0.0,
         "opd_centroid_sphere": 0.0, "opd_best_fit_sphere": 0.0, "opd_detrended": 0.0,
         "opd_afocal": 0.0, "fftpsf": 0.0}
count = {"spot": 0, "ee": 0, "opd": 0, "opd_centroid_sphere": 0, "opd_best_fit_sphere": 0,
         "opd_detrended": 0, "opd_afocal": 0, "fftpsf": 0, "raised_both": 0}
bad = []


CONVERGED = os.environ.get("OL_FUZZ_CONVERGED", "0") == "1"


def both(fn, table):
    out = []
    ref_table = table
    if CONVERGED and bool(np.any(table.surfaces["max_iter"] > 0)):
        # OL_FUZZ_CONVERGED=1: the oracle's Newton loops run to 1e-13 mm instead of the lens's
        # own tolerance (1e-6 mm when built by `surfaces.add`), so that what is compared is the
        # kernel's arithmetic and not where the reference chose to stop
        import copy
        ref_table = copy.deepcopy(table)
        ref_table.surfaces["tol"] = np.where(ref_table.surfaces["max_iter"] > 0, 1e-13,
                                             ref_table.surfaces["tol"])
    for real in (True, False):
        eng = (None if _HOST is None else _HOST(table, "cpu")) if real \
            else OracleEngine(ref_table, "cpu")
        t = HipRayTracer(table if real else ref_table, DEV if real else "cpu",
                         dtype=torch.float64, engine=eng)
        try:
            with np.errstate(all="ignore"):
                out.append(fn(t))
        except (ValueError, NotImplementedError, RuntimeError) as e:
            out.append(e)
        finally:
            if real:
                t.engine.close()
    return out


ONLY = {int(a) for a in sys.argv[1:]}   # optional: seeds to look at (triage of flagged lenses)
for path in sorted(glob.glob(os.path.join(ROOT, "fuzz_tables", "*.json"))):
    if ONLY and int(os.path.basename(path)[5:9]) not in ONLY:
        continue
    table = SystemTable.load(path)
    name = os.path.basename(path)

    def spot(t):
        s = SpotDiagram(t, num_rings=5)
        return (np.array(s.rms_spot_radius()), np.array(s.geometric_spot_radius()),
                np.array(s.centroid(), dtype=np.float64))
    a, b = both(spot, table)
    if isinstance(a, Exception) or isinstance(b, Exception):
        assert type(a) is type(b), (name, a, b)
        count["raised_both"] += 1
        continue
    if np.isfinite(b[0]).all() and np.isfinite(b[2]).all():
        scale = max(1.0, float(np.abs(b[2]).max()))
        for k, (u, v) in zip(("spot_rms", "spot_geo", "spot_centroid"), zip(a, b)):
            e = float(np.max(np.abs(u - v)) / scale)
            worst[k] = max(worst[k], e)
            if e > 1e-6:
                bad.append((name, k, e))
        count["spot"] += 1
    if table.polarization is None and not table.uses_polarization:
        def ee(t):
            e = EncircledEnergy(t, num_rays=6, distribution="hexapolar", num_points=32)
            return e.r_step, e.ee
        a, b = both(ee, table)
        if not (isinstance(a, Exception) or isinstance(b, Exception)) and np.isfinite(b[0]).all():
            d = np.abs(a[1] - b[1])[:, 1:]
            # 1e-7: the Newton stop tolerance moves hits (and absorbed energy) by ~1e-8
            frac = float((d > 1e-7 * max(1.0, b[1].max())).mean())
            worst["ee"] = max(worst["ee"], frac)
            if frac > 0.05:
                bad.append((name, "ee", frac))
            count["ee"] += 1

        def opd(t):
            o = OPD(t, (0.0, 0.7), float(table.wavelengths[0]), num_rays=5)
            return o.data.opd.double().cpu().numpy()
        a, b = both(opd, table)
        if not (isinstance(a, Exception) or isinstance(b, Exception)) and np.isfinite(b).all():
            e = float(np.max(np.abs(a - b)) / max(1.0, float(np.abs(b).max())))
            worst["opd"] = max(worst["opd"], e)
            if e > 1e-5:
                bad.append((name, "opd", e))
            count["opd"] += 1

        # round 5: the fitted reference spheres (ol_wavefront_fit + ol_wavefront_opd_fitted on
        # the real engine, oracle.wavefront_fit behind the stand-in)
        for strat in ("centroid_sphere", "best_fit_sphere"):
            def opd_fit(t, strat=strat):
                o = OPD(t, (0.0, 0.7), float(table.wavelengths[0]), num_rays=5, strategy=strat)
                return o.data.opd.double().cpu().numpy()
            a, b = both(opd_fit, table)
            if not (isinstance(a, Exception) or isinstance(b, Exception)) and np.isfinite(b).all() \
                    and np.isfinite(a).all():
                e = float(np.max(np.abs(a - b)) / max(1.0, float(np.abs(b).max())))
                worst["opd_" + strat] = max(worst["opd_" + strat], e)
                if e > (1e-4 if strat == "best_fit_sphere" else 1e-5):
                    bad.append((name, "opd_" + strat, e))
                count["opd_" + strat] += 1
            elif isinstance(a, Exception) != isinstance(b, Exception):
                bad.append((name, "opd_" + strat + " raised on one side only",
                            str(a if isinstance(a, Exception) else b)[:80]))
        # round 5: tilt removal, the planar reference, and the FFT PSF (pupil scatter + rocFFT)
        extra = {"opd_detrended": lambda t: OPD(t, (0.3, -0.6), float(table.wavelengths[-1]), num_rays=5,
                                               remove_tilt=True).data.opd.double().cpu().numpy(),
                 "opd_afocal": lambda t: OPD(t, (0.0, 0.7), float(table.wavelengths[0]), num_rays=5,
                                            afocal=True).data.opd.double().cpu().numpy(),
                 "fftpsf": lambda t: FFTPSF(t, (0.0, 0.5), float(table.wavelengths[0]), num_rays=32,
                                            grid_size=64).psf.double().cpu().numpy()}
        for key, fn in extra.items():
            a, b = both(fn, table)
            if not (isinstance(a, Exception) or isinstance(b, Exception)) and np.isfinite(b).all() \
                    and np.isfinite(a).all():
                e = float(np.max(np.abs(a - b)) / max(1.0, float(np.abs(b).max())))
                worst[key] = max(worst[key], e)
                if e > 1e-5:
                    bad.append((name, key, e))
                count[key] += 1
            elif isinstance(a, Exception) != isinstance(b, Exception) \
                    or (not isinstance(b, Exception) and np.isfinite(b).all() != np.isfinite(a).all()):
                bad.append((name, key + " one side only",
                            str(a if isinstance(a, Exception) else b)[:80]))
print("compared:", count)
print("worst:", {k: f"{v:.3e}" for k, v in worst.items()})
print("flagged:", len(bad))
notes = {}
_nr = {}
for b_ in bad:
    if b_[0] not in _nr:
        _nr[b_[0]] = int((SystemTable.load(os.path.join(ROOT, "fuzz_tables", b_[0]))
                          .surfaces["max_iter"] > 0).sum())
print("flagged on lenses WITHOUT a Newton surface:",
      [b_ for b_ in bad if _nr[b_[0]] == 0] or "none")
import collections  # noqa: E402
print("flagged, by family (Newton lenses: the reference stops at 1e-6 mm = 2e-3 waves per surface):",
      dict(collections.Counter(b_[1] for b_ in bad)))
for b_ in bad[:40]:
    nm = b_[0]
    if nm not in notes:
        tb = SystemTable.load(os.path.join(ROOT, "fuzz_tables", nm))
        nr = int((tb.surfaces["max_iter"] > 0).sum())
        notes[nm] = f"{nr} Newton surfaces" + (
            f", tol {sorted(set(tb.surfaces['tol'][tb.surfaces['max_iter'] > 0].tolist()))} mm" if nr else "")
        if b_[1].startswith("spot"):
            a_, b_k = lost_alive_rays(tb)
            notes[nm] += (f"; rays that reach a Newton surface alive and are not on it afterwards "
                          f"(no root): {a_} in the reference, {b_k} in the kernel")
    print("   ", b_, "--", notes[nm])
