"""First-order (paraxial) quantities of a packed system, on the host in plain floats.

The device ray generator needs a handful of first-order scalars -- entrance pupil location
and diameter, exit pupil location, the launch-plane offset of an infinite object -- which
the reference obtains from `optic.paraxial` (optiland/paraxial.py:74-262): three or four
y-u traces, each one a Python loop over the surfaces made of backend array operations
(raytrace/paraxial_ray_tracer.py:58-140), with `SurfaceGroup.positions`
(surfaces/surface_group.py:155-161) rebuilt from coordinate-system objects on every call.  On
the `cuda` device that is ~400 kernel launches and a dozen blocking read-backs per re-pack --
an order of magnitude more than the trace of a small ray bundle it prepares.

Everything those traces read is already in the packed table (radii, axial vertex positions,
indices at the primary wavelength, mirror flags, the stop index): this module restates the
same recurrences on Python floats.  `first_order()` returns None whenever the system has
something the restatement does not cover (thin-lens "paraxial" surfaces never reach the packer;
geometries whose `radius` the reference's `SurfaceGroup.radii` would not simply read;
aperture types other than the four of optiland/aperture/) and the packer then asks the
reference as before.  tests/test_paraxial_host.py holds it to `optic.paraxial` on the
reference's samples and on random lenses.
"""

from __future__ import annotations

import math


def _trace(R, n, pos, reflect, y, u, z, reverse=False, skip=0):
    """raytrace/paraxial_ray_tracer.py:58-140 for one ray.  R, n, pos, reflect: per surface
    (object first).  Returns (heights, slopes) as the reference appends them: one entry per
    surface from `skip` on, the object surface repeating the incoming ray."""
    m = len(R)
    is_obj = [k == 0 for k in range(m)]
    if reverse:
        R = [-r for r in reversed(R)]
        rolled = [n[-1]] + list(n[:-1])          # be.roll(n, shift=1)
        n = list(reversed(rolled))
        last = pos[-1]
        pos = [last - p for p in reversed(pos)]
        reflect = list(reversed(reflect))
        is_obj = list(reversed(is_obj))
    power = [0.0] * m
    for k in range(1, m):
        power[k] = (n[k] - n[k - 1]) / R[k] if R[k] != 0.0 else math.copysign(math.inf, 1.0)
    hs, us = [], []
    for k in range(skip, m):
        if is_obj[k]:
            hs.append(y)
            us.append(u)
            continue
        t = pos[k] - z
        z = pos[k]
        y = y + t * u
        if reflect[k]:
            u = -u - 2.0 * y / R[k]
        else:
            u = (n[k - 1] * u - y * power[k]) / n[k]
        hs.append(y)
        us.append(u)
    return hs, us


def first_order(radii, n, pos, reflect, stop_index, aperture_kind, aperture_value,
                object_infinite, obj_z):
    """dict(EPL, EPD, XPL, f2) or None (not covered).  Arguments: per-surface lists (object
    first) of radius, index after the surface at the primary wavelength, axial vertex
    position, mirror flag; index of the stop surface; the system aperture (class name of
    optiland/aperture/*.py, value)."""
    m = len(radii)
    if m < 3 or stop_index is None or not (0 < stop_index < m):
        return None
    R = [float(r) for r in radii]
    n = [float(v) for v in n]
    pos = [float(p) for p in pos]
    if any(math.isnan(v) for v in R + n + pos[1:]):
        return None

    def trace(y, u, z, reverse=False, skip=0):
        return _trace(R, n, pos, reflect, y, u, z, reverse, skip)

    try:
        # paraxial.py:206-229
        if stop_index == 1:
            EPL = pos[1]
        else:
            z0 = pos[-1] - pos[stop_index]
            y, u = trace(0.0, 0.1, z0, reverse=True, skip=m - stop_index)
            EPL = y[-1] / u[-1]
        # paraxial.py:244-256
        y, u = trace(0.0, 0.1, pos[stop_index], skip=stop_index + 1)
        XPL = -y[-1] / u[-1]
        # paraxial.py:74-86
        y, u = trace(1.0, 0.0, pos[1] - 1.0)
        f2 = -y[0] / u[-1]
        v = float(aperture_value)
        if aperture_kind == "EPDAperture":            # aperture/epd.py
            EPD = v
        elif aperture_kind == "ImageFNOAperture":     # aperture/image_fno.py:47-58
            EPD = f2 / v
        elif aperture_kind == "ObjectNAAperture":     # aperture/object_na.py:50-76
            EPD = 2.0 * (EPL - obj_z) * math.tan(math.asin(v / n[0]))
        elif aperture_kind == "FloatByStopAperture":  # aperture/float_by_stop.py:50-83
            if object_infinite:
                y, _ = trace(1.0, 0.0, -1.0)
                EPD = v / y[stop_index]
            else:
                y, _ = trace(0.0, 0.1, obj_z)
                EPD = (0.1 * v / y[stop_index]) * (EPL - obj_z)
        else:
            return None
    except (ZeroDivisionError, ValueError, IndexError):
        return None
    out = {"EPL": EPL, "EPD": EPD, "XPL": XPL, "f2": f2}
    if any(math.isnan(x) for x in out.values()):
        return None
    return out
