"""Locate / import the LIVE reference package for tests that need it.

Build container: /root/reference.  GPU box: the copy staged by
`oracle/stage_reference.py` under the git-ignored `oracle/_ref/` (test infrastructure,
travels with the gpurun snapshot).  Import stubs for numba / vtk / seaborn, which the
image lacks, are this repo's tests/refshim/.
"""

from __future__ import annotations

import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SHIM = os.path.join(HERE, "refshim")


def reference_root() -> str | None:
    for cand in (os.environ.get("OPTILAND_REFERENCE"), "/root/reference",
                 os.path.join(ROOT, "oracle", "_ref")):
        if cand and os.path.isdir(os.path.join(cand, "optiland")):
            return cand
    return None


def import_reference():
    """`optiland.backend` of the live reference (sys.path extended on first use)."""
    root = reference_root()
    if root is None:
        raise ImportError("reference package not present (run oracle/stage_reference.py)")
    sys.dont_write_bytecode = True
    for p in (root, SHIM):
        if p not in sys.path:
            sys.path.insert(0, p)
    import optiland.backend as be
    return be


# ------------------------------------------------------------------ the BASELINE configs
def rc_asphere():
    """Config C4 (BASELINE.json configs[3]): Ritchey-Chretien mirrors + even-asphere
    corrector plate; the same system as tools/make_golden.py:rc_asphere."""
    import optiland.backend as be
    from optiland import optic as optic_mod
    from optiland import physical_apertures
    lens = optic_mod.Optic(name="RCAsphere")
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, thickness=4910.01016)
    obsc = physical_apertures.RadialAperture(r_max=be.inf, r_min=177.80035)
    lens.surfaces.add(index=2, radius=-11040.02286, thickness=-4910.01016,
                      material="mirror", is_stop=True, conic=-1.001152, aperture=obsc)
    lens.surfaces.add(index=3, radius=-1349.31166, thickness=6265.20955,
                      material="mirror", conic=-1.483014)
    lens.surfaces.add(index=4, surface_type="even_asphere", radius=be.inf, thickness=5.0,
                      material="N-BK7", conic=0.0, coefficients=[0.0, 1e-12, -1e-18])
    lens.surfaces.add(index=5, thickness=96.7)
    lens.surfaces.add(index=6)
    lens.set_aperture(aperture_type="EPD", value=2400)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=0.1)
    lens.wavelengths.add(value=0.55, is_primary=True)
    return lens


def zernike_fresnel(polarization="unpolarized", zernike_type="fringe"):
    """Config C5 (BASELINE.json configs[4]): Zernike freeform singlet, Fresnel coatings,
    polarisation on; the same system as tools/make_golden.py:zernike_fresnel."""
    import optiland.backend as be
    from optiland import optic as optic_mod
    from optiland.rays import PolarizationState
    lens = optic_mod.Optic(name=f"ZernikeFresnel_{zernike_type}")
    coeffs = [0.0, 2e-4, -3e-4, 5e-4, 1e-3, -4e-4, 2.5e-4, -1.5e-4, 3e-4, 1e-4,
              -2e-4, 1.2e-4]
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, surface_type="zernike", radius=50.0, thickness=5.0,
                      material="N-BK7", is_stop=True, zernike_type=zernike_type,
                      norm_radius=15.0, coefficients=coeffs)
    lens.surfaces.add(index=2, radius=-200.0, thickness=75.0)
    lens.surfaces.add(index=3)
    lens.set_aperture(aperture_type="EPD", value=20)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=3)
    lens.wavelengths.add(value=0.55, is_primary=True)
    if polarization is not None:
        lens.surfaces.set_fresnel_coatings()
        if polarization == "unpolarized":
            st = PolarizationState(is_polarized=False)
        else:
            st = PolarizationState(is_polarized=True, Ex=1.0, Ey=0.5, phase_x=0.0,
                                   phase_y=0.7)
        lens.updater.set_polarization(st)
    return lens


def build_system(name: str):
    """One of the BASELINE.json systems, built under the CURRENT reference backend."""
    if name == "DoubleGauss":
        from optiland.samples.objectives import DoubleGauss
        return DoubleGauss(), 0.5876
    if name == "CookeTriplet":
        from optiland.samples.objectives import CookeTriplet
        return CookeTriplet(), 0.55
    if name == "RCAsphere":
        return rc_asphere(), 0.55
    if name == "ZernikeFresnelUnpolarized":
        return zernike_fresnel("unpolarized"), 0.55
    if name == "ZernikeFresnelPolarized":
        return zernike_fresnel("elliptical"), 0.55
    raise KeyError(name)


SYSTEMS = ("CookeTriplet", "DoubleGauss", "RCAsphere", "ZernikeFresnelUnpolarized",
           "ZernikeFresnelPolarized")
