#!/usr/bin/env python
"""Generate golden fixtures by importing the REFERENCE (NumPy backend, fp64).

Run in the build container only (the GPU box has no /root/reference):

    python tools/make_golden.py            # writes tests/golden/*.json|*.npz
                                           # and optiland_amd/data/*.json

For every case it (1) builds the reference `Optic`, (2) packs it with
`optiland_amd.packer.pack_optic` into a JSON surface table, (3) generates rays and
traces them with the reference's own `SurfaceGroup.trace`, and (4) stores inputs,
the per-surface recorded arrays and the final ray state.  The fixtures pin the
oracle (tests/test_oracle_golden.py) and the HIP path (tests/test_gpu_parity.py).
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
from optiland import optic as optic_mod  # noqa: E402
from optiland import physical_apertures  # noqa: E402
from optiland.coatings import (FresnelCoating, PolarizerCoating, RetarderCoating,  # noqa: E402
                               SimpleCoating)
from optiland.rays import PolarizationState  # noqa: E402
from optiland.samples.objectives import CookeTriplet, DoubleGauss  # noqa: E402
from optiland.samples.simple import AsphericSinglet  # noqa: E402
from optiland.samples.telescopes import HubbleTelescope  # noqa: E402

from optiland_amd.packer import pack_optic  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "optiland_amd", "data")

be.set_backend("numpy")


# ----------------------------------------------------------------- systems
def rc_asphere():
    """Config C4: Ritchey-Chretien (Hubble mirrors) + even-asphere corrector plate
    (SURVEY.md section 8d)."""
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
    """Config C5: Zernike freeform singlet + Fresnel coatings."""
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


def tilted_fold():
    """Edge case: decentered + tilted surfaces, a fold mirror, rectangular and
    elliptical apertures, a SimpleCoating, finite object distance."""
    lens = optic_mod.Optic(name="TiltedFold")
    lens.surfaces.add(index=0, radius=be.inf, thickness=80.0)
    lens.surfaces.add(index=1, radius=40.0, thickness=6.0, material="N-BK7", is_stop=True,
                      aperture=physical_apertures.EllipticalAperture(a=5.5, b=4.5),
                      coating=SimpleCoating(transmittance=0.97, reflectance=0.02))
    lens.surfaces.add(index=2, radius=-60.0, thickness=20.0, dx=0.3, dy=-0.2, rx=0.02,
                      ry=-0.015)
    lens.surfaces.add(index=3, radius=be.inf, thickness=-25.0, material="mirror",
                      rx=0.35, rz=0.1,
                      aperture=physical_apertures.RectangularAperture(-6, 6, -5, 7),
                      coating=SimpleCoating(transmittance=0.0, reflectance=0.9))
    lens.surfaces.add(index=4, radius=be.inf, rx=0.7, conic=0.0)
    lens.set_aperture(aperture_type="EPD", value=12)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=2.0)
    lens.wavelengths.add(value=0.6328, is_primary=True)
    return lens


def nr_family():
    """f3 geometries: odd asphere + x^i y^j polynomial freeform + offset aperture."""
    lens = optic_mod.Optic(name="NRFamily")
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, surface_type="odd_asphere", radius=35.0, thickness=4.0,
                      material="N-SF11", is_stop=True, conic=-0.3,
                      coefficients=[0.0, 1e-4, -2e-5, 1e-6])
    lens.surfaces.add(index=2, surface_type="polynomial", radius=-90.0, thickness=30.0,
                      conic=0.1,
                      coefficients=[[0.0, 1e-3, 2e-4], [-5e-4, 1e-4, 1e-6], [3e-4, -1e-5, 0]],
                      aperture=physical_apertures.OffsetRadialAperture(
                          r_max=9.0, r_min=0.5, offset_x=0.4, offset_y=-0.3))
    lens.surfaces.add(index=3)
    lens.set_aperture(aperture_type="EPD", value=16)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=5)
    lens.wavelengths.add(value=0.5876, is_primary=True)
    return lens


def boolean_apertures():
    """Edge case: nested Union / Intersection / Difference apertures
    (physical_apertures/base.py:259-340) on two surfaces."""
    pa = physical_apertures
    lens = optic_mod.Optic(name="BooleanApertures")
    d_shape = pa.IntersectionAperture(pa.RadialAperture(r_max=7.0),
                                      pa.RectangularAperture(-5.0, 5.5, -8.0, 8.0))
    spider = pa.UnionAperture(pa.RectangularAperture(-0.4, 0.4, -9.0, 9.0),
                              pa.OffsetRadialAperture(r_max=1.5, r_min=0.0, offset_x=1.0,
                                                      offset_y=-0.5))
    ap1 = pa.DifferenceAperture(d_shape, spider)
    ap2 = pa.UnionAperture(pa.EllipticalAperture(a=3.0, b=2.0, offset_x=-1.0),
                           pa.DifferenceAperture(pa.RadialAperture(r_max=6.0, r_min=0.0),
                                                 pa.RadialAperture(r_max=4.5, r_min=0.0)))
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, radius=60.0, thickness=5.0, material="N-BK7", is_stop=True,
                      aperture=ap1)
    lens.surfaces.add(index=2, radius=-80.0, thickness=40.0, aperture=ap2)
    lens.surfaces.add(index=3)
    lens.set_aperture(aperture_type="EPD", value=16)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=3)
    lens.wavelengths.add(value=0.55, is_primary=True)
    return lens


def polygon_apertures():
    """PolygonAperture (physical_apertures/polygon.py; matplotlib point-in-polygon on the
    NumPy backend): a concave L-shaped stop, and a hexagon minus a triangle inside a
    boolean tree on the second surface."""
    pa = physical_apertures
    lens = optic_mod.Optic(name="PolygonApertures")
    ell = pa.PolygonAperture(x=[-6.0, 6.0, 6.0, 1.5, 1.5, -6.0], y=[-6.0, -6.0, -1.0, -1.0, 6.0, 6.0])
    th = np.linspace(0, 2 * np.pi, 7)[:-1] + 0.2
    hexa = pa.PolygonAperture(x=5.5 * np.cos(th), y=5.5 * np.sin(th))
    tri = pa.PolygonAperture(x=[-1.0, 2.0, 0.5], y=[-1.0, -0.5, 2.0])
    ap2 = pa.DifferenceAperture(pa.IntersectionAperture(hexa, pa.RadialAperture(r_max=5.2)), tri)
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, radius=60.0, thickness=5.0, material="N-BK7", is_stop=True,
                      aperture=ell)
    lens.surfaces.add(index=2, radius=-80.0, thickness=40.0, aperture=ap2)
    lens.surfaces.add(index=3)
    lens.set_aperture(aperture_type="EPD", value=16)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=3)
    lens.wavelengths.add(value=0.55, is_primary=True)
    return lens


def f3_family():
    """f3 geometries: biconic, toroidal (with y^2i terms) and Chebyshev."""
    lens = optic_mod.Optic(name="F3Family")
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, surface_type="biconic", radius_x=40.0, radius_y=55.0,
                      conic_x=-0.5, conic_y=0.2, thickness=4.0, material="N-BK7", is_stop=True)
    lens.surfaces.add(index=2, surface_type="toroidal", radius_x=-70.0, radius_y=-45.0,
                      conic=0.3, toroidal_coeffs_poly_y=[1e-5, -2e-8], thickness=6.0)
    lens.surfaces.add(index=3, surface_type="chebyshev", radius=80.0, conic=-0.2,
                      coefficients=[[0.0, 1e-3, -2e-3], [5e-4, 2e-3, 0.0], [-1e-3, 0.0, 4e-4]],
                      norm_x=12.0, norm_y=12.0, thickness=3.0, material="N-SF11")
    lens.surfaces.add(index=4, surface_type="biconic", radius_x=be.inf, radius_y=-30.0,
                      thickness=25.0)
    lens.surfaces.add(index=5, surface_type="toroidal", radius_x=be.inf, radius_y=60.0,
                      thickness=0.0)
    lens.surfaces.add(index=6)
    lens.set_aperture(aperture_type="EPD", value=12)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=4)
    lens.wavelengths.add(value=0.5876, is_primary=True)
    return lens


def polarizer_retarder(with_retarder=True):
    """f3 Jones elements: linear polarizer plate, (tilted) quarter-wave-ish retarder
    plate, then a Fresnel-coated singlet; fully polarised input state."""
    lens = optic_mod.Optic(name="PolarizerRetarder")
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, radius=be.inf, thickness=2.0, is_stop=True,
                      coating=PolarizerCoating(axis=(1.0, 0.6, 0.0)))
    if with_retarder:
        lens.surfaces.add(index=2, radius=be.inf, thickness=3.0, rx=0.12, material="N-BK7",
                          coating=RetarderCoating(retardance=1.3, axis=(0.2, 1.0, 0.1)))
        lens.surfaces.add(index=3, radius=be.inf, thickness=2.0, rx=0.12,
                          coating=PolarizerCoating(axis=(0.0, 1.0, 0.0)))
    else:
        lens.surfaces.add(index=2, radius=be.inf, thickness=3.0, rx=0.12, material="N-BK7")
        lens.surfaces.add(index=3, radius=be.inf, thickness=2.0, rx=0.12,
                          coating=PolarizerCoating(axis=(0.3, 1.0, 0.0)))
    lens.surfaces.add(index=4, radius=35.0, thickness=5.0, material="N-SF11")
    lens.surfaces.add(index=5, radius=-80.0, thickness=30.0)
    lens.surfaces.add(index=6)
    for i in (4, 5):
        s_ = lens.surfaces[i]
        s_.coating = FresnelCoating(s_.material_pre, s_.material_post)
    lens.set_aperture(aperture_type="EPD", value=12)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=5)
    lens.wavelengths.add(value=0.55, is_primary=True)
    lens.updater.set_polarization(PolarizationState(is_polarized=True, Ex=1.0, Ey=0.3,
                                                    phase_x=0.0, phase_y=0.4))
    return lens


def coated_mirror_polarised():
    """Polarised rays through every way a surface can (not) touch the PRT: an UNCOATED
    curved refractor (rays.update() with the identity Jones matrix), a SimpleCoating
    refractor (intensity only -- SimpleCoating never calls rays.update()), a
    Fresnel-coated tilted MIRROR (reflection Jones, j22 = -1), a SimpleCoating mirror, a
    Fresnel refractor and the image plane."""
    lens = optic_mod.Optic(name="CoatedMirrorPolarised")
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, radius=60.0, thickness=4.0, material="N-BK7", is_stop=True)
    lens.surfaces.add(index=2, radius=-90.0, thickness=12.0,
                      coating=SimpleCoating(transmittance=0.9, reflectance=0.05))
    lens.surfaces.add(index=3, radius=-150.0, thickness=-10.0, material="mirror", rx=0.08)
    lens.surfaces.add(index=4, radius=be.inf, thickness=10.0, material="mirror", rx=0.08,
                      coating=SimpleCoating(transmittance=0.0, reflectance=0.8))
    lens.surfaces.add(index=5, radius=45.0, thickness=3.0, material="SF6")
    lens.surfaces.add(index=6, radius=be.inf, thickness=25.0)
    lens.surfaces.add(index=7)
    m3 = lens.surfaces[3]
    m3.coating = FresnelCoating(m3.material_pre, m3.material_post)
    for i in (5, 6):
        s_ = lens.surfaces[i]
        s_.coating = FresnelCoating(s_.material_pre, s_.material_post)
    lens.set_aperture(aperture_type="EPD", value=8)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=2)
    lens.wavelengths.add(value=0.55, is_primary=True)
    lens.updater.set_polarization(PolarizationState(is_polarized=True, Ex=0.8, Ey=0.6,
                                                    phase_x=0.1, phase_y=-0.5))
    return lens


def vignetted_cooke():
    """Cooke triplet with vignetting factors on the off-axis fields and an x field:
    exercises FieldGroup.get_vig_factor (nearest field) in trace() and the double
    (1 - v) scaling of trace_generic (real_ray_tracer.py:134-137)."""
    lens = optic_mod.Optic(name="VignettedCooke")
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, radius=22.01359, thickness=3.25896, material="SK16")
    lens.surfaces.add(index=2, radius=-435.76044, thickness=6.00755)
    lens.surfaces.add(index=3, radius=-22.21328, thickness=0.99997, material=("F2", "schott"))
    lens.surfaces.add(index=4, radius=20.29192, thickness=4.75041, is_stop=True)
    lens.surfaces.add(index=5, radius=79.68360, thickness=2.95208, material="SK16")
    lens.surfaces.add(index=6, radius=-18.39533, thickness=42.20778)
    lens.surfaces.add(index=7)
    lens.set_aperture(aperture_type="EPD", value=10)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=14, vx=0.05, vy=0.15)
    lens.fields.add(y=20, x=5, vx=0.1, vy=0.3)
    lens.wavelengths.add(value=0.55, is_primary=True)
    return lens


def finite_conjugate(field_type="object_height", telecentric=False):
    """Cooke triplet used at finite conjugates (object 150 mm in front): object-height
    fields on a planar object (fields/field_types/object_height.py), optionally
    object-space telecentric with an object-NA aperture (ray_aiming/paraxial.py:82-87);
    the same layout with angle fields covers AngleField's finite-object branch
    (angle.py:48-58).  Vignetting on the outer field exercises the (1 - v) factors of
    both aiming branches."""
    lens = optic_mod.Optic(name=f"FiniteCooke_{field_type}{'_tele' if telecentric else ''}")
    lens.surfaces.add(index=0, radius=be.inf, thickness=150.0)
    lens.surfaces.add(index=1, radius=22.01359, thickness=3.25896, material="SK16")
    lens.surfaces.add(index=2, radius=-435.76044, thickness=6.00755)
    lens.surfaces.add(index=3, radius=-22.21328, thickness=0.99997, material=("F2", "schott"))
    lens.surfaces.add(index=4, radius=20.29192, thickness=4.75041, is_stop=True)
    lens.surfaces.add(index=5, radius=79.68360, thickness=2.95208, material="SK16")
    lens.surfaces.add(index=6, radius=-18.39533, thickness=62.0)
    lens.surfaces.add(index=7)
    if telecentric:
        lens.set_aperture(aperture_type="objectNA", value=0.03)
        lens.obj_space_telecentric = True
    else:
        lens.set_aperture(aperture_type="EPD", value=8)
    lens.fields.set_type(field_type=field_type)
    if field_type == "angle":
        lens.fields.add(y=0)
        lens.fields.add(y=4, vx=0.05, vy=0.1)
        lens.fields.add(y=6, x=2)
    elif telecentric:
        # chief rays leave parallel to the axis: keep the object small enough that the
        # bundle still clears the (non-telecentric) triplet without grazing its rims
        lens.fields.add(y=0)
        lens.fields.add(y=2, vx=0.05, vy=0.1)
        lens.fields.add(y=3, x=1)
    else:
        lens.fields.add(y=0)
        lens.fields.add(y=10, vx=0.05, vy=0.1)
        lens.fields.add(y=15, x=5)
    lens.wavelengths.add(value=0.55, is_primary=True)
    return lens


def tir_prism():
    """Edge case: steep glass->air exit so part of the bundle is totally internally
    reflected (NaN directions, real_rays.py:179-180) and part misses a small
    sphere (NaN distance, standard.py:132-137)."""
    lens = optic_mod.Optic(name="TIRMiss")
    lens.surfaces.add(index=0, radius=be.inf, thickness=be.inf)
    lens.surfaces.add(index=1, radius=be.inf, thickness=10.0, material="N-SF11", is_stop=True)
    lens.surfaces.add(index=2, radius=-7.4, thickness=1.5)
    lens.surfaces.add(index=3, radius=2.6, thickness=5.0, material="N-BK7")
    lens.surfaces.add(index=4)
    lens.set_aperture(aperture_type="EPD", value=10)
    lens.fields.set_type(field_type="angle")
    lens.fields.add(y=0)
    lens.fields.add(y=4)
    lens.wavelengths.add(value=0.55, is_primary=True)
    return lens


# ------------------------------------------------------------------ helpers
def disc_points(n, seed):
    rng = np.random.default_rng(seed)
    r = np.sqrt(rng.random(n))
    th = 2 * np.pi * rng.random(n)
    return r * np.cos(th), r * np.sin(th)


def stack_record(optic):
    sg = optic.surfaces
    return np.stack([np.asarray(a, dtype=np.float64) for a in
                     (sg.x, sg.y, sg.z, sg.L, sg.M, sg.N, sg.intensity, sg.opd)], axis=1)


def run_case(name, optic, Hx, Hy, Px, Py, wavelength, use_trace=None, save_json_to=(GOLD,)):
    """Trace and dump.  `use_trace`: dict(num_rays=, distribution=) to go through
    Optic.trace() (fields x pupil expansion) instead of trace_generic."""
    table = pack_optic(optic, wavelengths=[wavelength], name=name)
    for d in save_json_to:
        if d == GOLD:
            table.save(os.path.join(d, f"{name}.json"))
        else:  # shipped sample systems carry every wavelength of the optic
            pack_optic(optic, wavelengths=None, name=name).save(os.path.join(d, f"{name}.json"))

    tracer = optic.ray_tracer
    if use_trace is not None:
        from optiland.distribution import create_distribution
        dist = create_distribution(use_trace["distribution"])
        dist.generate_points(use_trace["num_rays"])
        Px, Py = np.asarray(dist.x, dtype=float), np.asarray(dist.y, dtype=float)
        Hx = np.atleast_1d(np.asarray(Hx, dtype=float))
        Hy = np.atleast_1d(np.asarray(Hy, dtype=float))
        nf, npup = len(Hx), len(Px)
        Hx_full, Hy_full = np.repeat(Hx, npup), np.repeat(Hy, npup)
        Px_full, Py_full = np.tile(Px, nf), np.tile(Py, nf)
    else:
        Hx_full, Hy_full, Px_full, Py_full = (np.broadcast_to(
            np.asarray(a, dtype=float), np.broadcast(Hx, Hy, Px, Py).shape).copy()
            for a in (Hx, Hy, Px, Py))
    # what trace_generic does before generate_rays (real_ray_tracer.py:134-137)
    vx, vy = optic.fields.get_vig_factor(Hx_full, Hy_full)
    if use_trace is None:
        Px_gen, Py_gen = Px_full * (1 - vx), Py_full * (1 - vy)
    else:
        Px_gen, Py_gen = Px_full, Py_full
    rays = tracer.ray_generator.generate_rays(Hx_full, Hy_full, Px_gen, Py_gen, wavelength)
    rays_in = np.stack([np.array(getattr(rays, k), dtype=np.float64)
                        for k in ("x", "y", "z", "L", "M", "N", "i")])
    optic.surfaces.trace(rays)
    record = stack_record(optic)
    last = optic.surfaces[-1]
    last.material_post.propagation_model.propagate(rays, last.thickness)
    out = dict(Hx=Hx_full, Hy=Hy_full, Px=Px_full, Py=Py_full,
               vx=np.asarray(vx, dtype=float) * np.ones_like(Px_full),
               vy=np.asarray(vy, dtype=float) * np.ones_like(Px_full),
               wavelength=np.float64(wavelength), rays_in=rays_in, record=record,
               via_trace=np.bool_(use_trace is not None))
    polarized = hasattr(rays, "p")
    if polarized:
        out["prt"] = np.array(rays.p, dtype=np.complex128)
        out["i_before_update"] = np.array(rays.i, dtype=np.float64)
        rays.update_intensity(optic.polarization_state)
        out["i_updated"] = np.array(rays.i, dtype=np.float64)
    out["final"] = np.stack([np.array(getattr(rays, k), dtype=np.float64)
                             for k in ("x", "y", "z", "L", "M", "N", "i", "opd")])
    out["pre_dir"] = np.stack([np.array(rays.L0), np.array(rays.M0), np.array(rays.N0)])

    # cross-check against the reference's public entry point on a fresh optic state
    if use_trace is None:
        chk = optic.trace_generic(Hx_full, Hy_full, Px_full, Py_full, wavelength)
    else:
        chk = optic.trace(Hx, Hy, wavelength, use_trace["num_rays"], use_trace["distribution"])
    for k, row in zip(("x", "y", "z", "L", "M", "N", "opd"), (0, 1, 2, 3, 4, 5, 7)):
        a, b = np.asarray(getattr(chk, k)), out["final"][row]
        assert np.array_equal(a, b, equal_nan=True), (name, k)
    if polarized and use_trace is not None:
        assert np.array_equal(np.asarray(chk.i), out["i_updated"], equal_nan=True)

    np.savez_compressed(os.path.join(GOLD, f"{name}.npz"), **out)
    nan_frac = float(np.mean(np.isnan(out["final"][0])))
    clip_frac = float(np.mean(out["final"][6] == 0))
    print(f"{name:28s} N={rays_in.shape[1]:6d} S={record.shape[0]-1:2d} "
          f"nan={nan_frac:.3f} clipped={clip_frac:.3f}")
    return table


def wavefront_goldens():
    """f4: the reference's own OPD / FFT PSF numbers for two sample lenses."""
    from optiland.psf import FFTPSF
    from optiland.wavefront import OPD
    out = {}
    for tag, optic, field, wl in (("cooke", CookeTriplet(), (0.0, 1.0), 0.55),
                                  ("dgauss", DoubleGauss(), (0.0, 0.7), 0.5876)):
        opd = OPD(optic, field, wl)  # 15 hexapolar rings, chief-ray sphere
        d = opd.get_data(field, wl)
        out[f"{tag}_opd"] = np.asarray(d.opd, dtype=np.float64)
        out[f"{tag}_pupil"] = np.stack([np.asarray(v, dtype=np.float64)
                                        for v in (d.pupil_x, d.pupil_y, d.pupil_z)])
        out[f"{tag}_radius"] = np.float64(d.radius)
        out[f"{tag}_rms"] = np.float64(opd.rms())
        psf = FFTPSF(optic, field, wl, num_rays=64)
        full = np.asarray(psf.psf, dtype=np.float64)
        c = full.shape[0] // 2
        out[f"{tag}_psf_center"] = full[c - 16:c + 16, c - 16:c + 16]
        out[f"{tag}_psf_sum"] = np.float64(full.sum())
        out[f"{tag}_strehl"] = np.float64(psf.strehl_ratio())
        out[f"{tag}_grid"] = np.array([psf.num_rays, psf.grid_size])
        print(f"wavefront {tag}: rms={out[tag + '_rms']:.6f} waves strehl={out[tag + '_strehl']:.5f} "
              f"grid={psf.num_rays}/{psf.grid_size}")
    np.savez_compressed(os.path.join(GOLD, "wavefront.npz"), **out)


def zemax_toroid_tables():
    """System tables (no traces) of the two single-toroid lenses whose Zemax ray data the
    reference's own tests hard-code (tests/test_geometries.py:1483-1840); the Zemax
    numbers themselves live in tests/test_external_known_answers.py."""
    from optiland.materials import IdealMaterial
    for name, kw, t1, t2, mat, epd in (
            ("zemax_toroid_posRx", dict(radius_x=100.0, radius_y=50.0, conic=-0.5,
                                        toroidal_coeffs_poly_y=[0.05, 0.0002]), 5.0, 10.0,
             IdealMaterial(n=1.5, k=0), 10.0),
            ("zemax_toroid_negRx", dict(radius_x=-50.0, radius_y=40.0, conic=-0.5,
                                        toroidal_coeffs_poly_y=[5e-5, 5e-6]), 7.0, 70.0,
             "N-BK7", 20.0)):
        lens = optic_mod.Optic(name=name)
        lens.surfaces.add(index=0, thickness=be.inf)
        lens.surfaces.add(index=1, surface_type="toroidal", thickness=t1, material=mat,
                          is_stop=True, **kw)
        lens.surfaces.add(index=2, thickness=t2, material="air")
        lens.surfaces.add(index=3)
        lens.set_aperture(aperture_type="EPD", value=epd)
        lens.wavelengths.add(value=0.550, is_primary=True)
        lens.fields.set_type("angle")
        lens.fields.add(y=0)
        pack_optic(lens, wavelengths=[0.55], name=name).save(os.path.join(GOLD, f"{name}.json"))
        print(f"{name}: table only")


def sample_goldens():
    """EVERY lens of optiland.samples (29 systems, 3 to 43 surfaces: photographic
    objectives, microscopes, eyepieces, IR triplets, wide-angle projection lenses,
    telescopes, an eye model, a UV lithography lens): 3 field points x 2 hexapolar
    rings at the primary wavelength through the reference's ray generator and
    SurfaceGroup.trace.  Small on purpose -- breadth over real prescriptions (glass
    catalogue indices, stops in odd places, steep fields), not ray count."""
    import importlib
    import inspect
    from optiland.distribution import create_distribution
    dist = create_distribution("hexapolar")
    dist.generate_points(2)
    px1, py1 = np.asarray(dist.x, dtype=float) * 0.95, np.asarray(dist.y, dtype=float) * 0.95
    n_ok = 0
    for m in ("eyepieces", "infrared", "lithography", "microscopes", "miscellaneous",
              "objectives", "simple", "telescopes"):
        mod = importlib.import_module("optiland.samples." + m)
        for cname, cls in inspect.getmembers(mod, inspect.isclass):
            if cls.__module__ != mod.__name__ or not issubclass(cls, optic_mod.Optic):
                continue
            lens = cls()
            hy = np.repeat([0.0, 0.7, 1.0], px1.size)
            hx = np.zeros_like(hy)
            w = float(np.asarray(lens.primary_wavelength).reshape(-1)[0])
            run_case(f"sample_{cname}", lens, hx, hy, np.tile(px1, 3), np.tile(py1, 3), w)
            n_ok += 1
    print(f"{n_ok} sample systems")


def fuzz_goldens(count=24):
    """Random lenses of the differential fuzz (tests/test_reference_fuzz.py:
    `build_random_lens`, seeds 0..count-1) frozen as fixtures, so that the GPU box -- where
    the reference does not exist -- holds the KERNEL to reference results on random
    systems too (tests/test_gpu_parity.py picks every *.npz up)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "_ref_fuzz", os.path.join(HERE, "..", "tests", "test_reference_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for seed in range(count):
        lens, rng = mod.build_random_lens(seed, be)
        n = 160
        r, th = np.sqrt(rng.random(n)) * 0.9, 2 * np.pi * rng.random(n)
        hx, hy = float(rng.uniform(-0.6, 0.6)), float(rng.uniform(-1, 1))
        try:
            with np.errstate(all="ignore"):
                run_case(f"fuzz_{seed:02d}", lens, hx, hy, r * np.cos(th), r * np.sin(th),
                         float(lens.primary_wavelength))
        except ValueError as e:   # Zernike / Chebyshev range error: not a trace fixture
            print(f"fuzz_{seed:02d}: skipped ({e})")


def main():
    if "--samples-only" in sys.argv:
        sample_goldens()
        return
    if "--fuzz-only" in sys.argv:
        fuzz_goldens()
        return
    if "--zemax-only" in sys.argv:
        zemax_toroid_tables()
        return
    os.makedirs(GOLD, exist_ok=True)
    os.makedirs(DATA, exist_ok=True)

    # C1: Cooke triplet through Optic.trace (fields x hexapolar pupil)
    run_case("cooke_trace_hexapolar6", CookeTriplet(), [0.0, 0.0, 0.0], [0.0, 0.7, 1.0],
             None, None, 0.55, use_trace=dict(num_rays=6, distribution="hexapolar"))
    px, py = disc_points(1500, 0)
    run_case("cooke_generic", CookeTriplet(), 0.0, 1.0, px, py, 0.55,
             save_json_to=(GOLD, DATA))
    # full-size C1 system JSON for the plumbing run is the same table.

    # C2/C3: double Gauss
    px, py = disc_points(2000, 0)
    run_case("double_gauss", DoubleGauss(), 0.0, 0.7, px, py, 0.5876,
             save_json_to=(GOLD, DATA))
    hx = np.repeat([0.0, 0.3, -0.5], 300)
    hy = np.repeat([0.0, 1.0, 0.6], 300)
    px, py = disc_points(900, 3)
    run_case("double_gauss_multifield", DoubleGauss(), hx, hy, px, py, 0.4861)

    # C4: RC + even asphere corrector; pure-NR singlet
    px, py = disc_points(2000, 1)
    run_case("rc_asphere", rc_asphere(), 0.0, 1.0, px, py, 0.55, save_json_to=(GOLD, DATA))
    run_case("hubble", HubbleTelescope(), 0.0, 0.5, px, py, 0.55)
    px, py = disc_points(1500, 2)
    run_case("aspheric_singlet", AsphericSinglet(), 0.0, 0.0, px, py, 0.587)

    # C5: Zernike + Fresnel, unpolarised, through Optic.trace (update_intensity)
    for zt in ("fringe", "standard", "noll"):
        run_case(f"zernike_fresnel_{zt}", zernike_fresnel("unpolarized", zt), [0.0, 0.0],
                 [0.0, 1.0], None, None, 0.55,
                 use_trace=dict(num_rays=24, distribution="uniform"),
                 save_json_to=(GOLD, DATA) if zt == "fringe" else (GOLD,))
    px, py = disc_points(1200, 5)
    run_case("zernike_fresnel_polarized", zernike_fresnel("polarized"), 0.0, 1.0, px, py, 0.55)
    run_case("zernike_nopol", zernike_fresnel(None), 0.0, 0.5, px, py, 0.55)

    # edge cases
    px, py = disc_points(1500, 7)
    run_case("tilted_fold", tilted_fold(), 0.0, 1.0, px, py, 0.6328)
    run_case("nr_family", nr_family(), 0.0, 1.0, px * 0.9, py * 0.9, 0.5876)
    run_case("tir_miss", tir_prism(), 0.0, 1.0, px, py, 0.55)
    run_case("boolean_apertures", boolean_apertures(), 0.0, 0.5, px, py, 0.55)
    run_case("polygon_apertures", polygon_apertures(), 0.0, 0.5, px, py, 0.55)
    run_case("f3_family", f3_family(), 0.0, 1.0, px, py, 0.5876)
    run_case("polarizer_retarder", polarizer_retarder(True), [0.0, 0.0], [0.0, 1.0], None, None,
             0.55, use_trace=dict(num_rays=20, distribution="uniform"))
    run_case("polarizer_only", polarizer_retarder(False), 0.0, 1.0, px, py, 0.55)
    run_case("coated_mirror_polarised", coated_mirror_polarised(), [0.0, 0.0], [0.0, 1.0], None,
             None, 0.55, use_trace=dict(num_rays=12, distribution="uniform"))
    run_case("vignetted_trace", vignetted_cooke(), [0.0, 0.0, 0.25], [0.0, 0.7, 1.0], None, None,
             0.55, use_trace=dict(num_rays=5, distribution="hexapolar"))
    hx = np.repeat([0.0, 0.05, 0.25, 0.2], 200)
    hy = np.repeat([0.1, 0.6, 1.0, 0.9], 200)
    px8, py8 = disc_points(800, 11)
    run_case("vignetted_generic", vignetted_cooke(), hx, hy, px8, py8, 0.55)
    # finite conjugates: object-height fields (plain and telecentric), finite angle field
    hx = np.repeat([0.0, 0.1, 1 / 3, -0.2], 150)
    hy = np.repeat([0.0, 0.6, 1.0, 0.9], 150)
    px6, py6 = disc_points(600, 13)
    for nm, lens in (("finite_object_height", finite_conjugate("object_height")),
                     ("finite_object_height_telecentric", finite_conjugate("object_height", True)),
                     ("finite_angle", finite_conjugate("angle"))):
        run_case(nm + "_generic", lens, hx, hy, px6, py6, 0.55)
    run_case("finite_object_height_trace", finite_conjugate("object_height"),
             [0.0, 0.0, 1 / 3], [0.0, 0.7, 1.0], None, None, 0.55,
             use_trace=dict(num_rays=4, distribution="hexapolar"))
    # paraxial image height fields: object at infinity (slope scale) and finite (height scale)
    lens = CookeTriplet()
    lens.fields.set_type(field_type="paraxial_image_height")
    lens.fields.fields.clear()
    lens.fields.add(y=0)
    lens.fields.add(y=12.0, vx=0.05, vy=0.1)
    lens.fields.add(y=18.0, x=4.0)
    run_case("image_height_infinite_generic", lens, hx, hy, px6, py6, 0.55)
    lens = finite_conjugate("object_height")
    lens.fields.set_type(field_type="paraxial_image_height")
    lens.fields.fields.clear()
    lens.fields.add(y=0)
    lens.fields.add(y=3.0, vx=0.05, vy=0.1)
    lens.fields.add(y=5.0, x=1.5)
    run_case("image_height_finite_trace", lens, [0.0, 0.0, 0.3], [0.0, 0.6, 1.0], None, None, 0.55,
             use_trace=dict(num_rays=4, distribution="hexapolar"))
    run_case("finite_telecentric_trace", finite_conjugate("object_height", True),
             [0.0, 0.0, 1 / 3], [0.0, 0.7, 1.0], None, None, 0.55,
             use_trace=dict(num_rays=4, distribution="hexapolar"))
    # pupil apodization (ray_generator.py:81-85): every class of optiland/apodization
    from optiland import apodization as apod_mod
    for nm, ap in (("gaussian", apod_mod.GaussianApodization(sigma=0.7)),
                   ("cosine_squared", apod_mod.CosineSquaredApodization(R=0.9)),
                   ("hann", apod_mod.HannApodization(D=1.8)),
                   ("polynomial", apod_mod.PolynomialApodization(R=0.95, p=1.5)),
                   ("super_gaussian", apod_mod.SuperGaussianApodization(w=0.8, n=4.0)),
                   ("tukey", apod_mod.TukeyApodization(R=0.9, alpha=0.4))):
        lens = vignetted_cooke()
        lens.updater.set_apodization(ap)
        run_case(f"apodized_{nm}_trace", lens, [0.0, 0.0], [0.0, 0.7], None, None, 0.55,
                 use_trace=dict(num_rays=4, distribution="hexapolar"))
    lens = vignetted_cooke()
    lens.updater.set_apodization(apod_mod.GaussianApodization(sigma=0.7))
    hx = np.repeat([0.0, 0.25], 150)
    hy = np.repeat([0.6, 1.0], 150)
    px3, py3 = disc_points(300, 17)
    run_case("apodized_gaussian_generic", lens, hx, hy, px3, py3, 0.55)
    wavefront_goldens()
    zemax_toroid_tables()
    sample_goldens()
    fuzz_goldens()


if __name__ == "__main__":
    main()
