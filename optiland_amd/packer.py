"""Pack a live reference `Optic` into a `SystemTable`.

This is the only module that touches reference objects.  It reads exactly the
attributes the reference's own trace reads (citations inline) and refuses --
raising `UnsupportedSystem` -- anything the fused kernel does not implement, so
the caller can leave such systems on the reference's own array path
(SURVEY.md section 8b "must-not-intercept cases").
"""

from __future__ import annotations

import math
import weakref

import numpy as np

from . import system as S
from .system import SystemTable


class UnsupportedSystem(Exception):
    """The optic contains something the fused HIP path does not implement."""


# Values of device tensors already read back: id(tensor) -> (weakref, _version, float).  On the
# `cuda` device every scalar of the prescription is a tensor and reading one is a BLOCKING
# device-to-host copy (~4 us each, ~700 per pack: the whole cost of a re-pack); a re-pack
# after `set_radius` finds all but the changed tensor here.  Keyed like the change detector
# (fingerprint.py): object identity + in-place version; tensors that require grad are never
# cached (their `.data` can be rewritten without a version bump).
_TENSOR_VALUES: dict = {}
# The cache is only consulted during an INCREMENTAL pack (`pack_surfaces(tokens=..., cache=...)`
# under the change detector): a forced full pack -- no tokens, OPTILAND_HIP_PACK_CACHE=0, or
# after `tracer.invalidate()` -- reads every tensor live, so the documented escape hatches for
# the detector's blind spot (a `.data` write bumps no version) really re-read the prescription.
_TENSOR_CACHE_ACTIVE = False


def clear_tensor_values() -> None:
    """Forget every device scalar read back so far (`tracer.invalidate()`)."""
    _TENSOR_VALUES.clear()


def _f(v) -> float:
    """Backend scalar / 0-d array / python number -> float."""
    if type(v) is float or type(v) is int:  # the common case, ~160 calls per pack
        return float(v)
    if hasattr(v, "detach"):
        cacheable = _TENSOR_CACHE_ACTIVE and not v.requires_grad and v.numel() == 1
        if cacheable:
            hit = _TENSOR_VALUES.get(id(v))
            if hit is not None and hit[1] == v._version and hit[0]() is v:
                return hit[2]
        val = float(v.detach().reshape(-1)[0].item()) if v.numel() else float("nan")
        if cacheable:
            if len(_TENSOR_VALUES) > 20000:
                _TENSOR_VALUES.clear()
            try:
                _TENSOR_VALUES[id(v)] = (weakref.ref(v), v._version, val)
            except TypeError:
                pass
        return val
    return float(np.asarray(v).reshape(-1)[0]) if np.ndim(v) else float(v)


def _rot_x(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]], dtype=np.float64)


def _rot_y(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]], dtype=np.float64)


def _rot_z(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], dtype=np.float64)


_EYE3 = np.eye(3)
_EYE3.setflags(write=False)


def cs_to_affine(cs):
    """Fold a (possibly nested) reference CoordinateSystem into (R, origin).

    Reference: `CoordinateSystem.localize` (optiland/coordinate_system.py:73-89)
    translates by (-x,-y,-z) then applies rotate_z(-rz), rotate_y(-ry),
    rotate_x(-rx) (rays/real_rays.py:112-152), each only when the angle is
    non-zero; a parent `reference_cs` is localized first.  Hence
    local = R (global - origin), R = Rx(-rx) Ry(-ry) Rz(-rz) R_parent.
    """
    rx, ry, rz = _f(cs.rx), _f(cs.ry), _f(cs.rz)
    t = np.array([_f(cs.x), _f(cs.y), _f(cs.z)], dtype=np.float64)
    if not (rx or ry or rz) and cs.reference_cs is None:
        return _EYE3, t  # the common untilted, un-nested surface
    R = np.eye(3)
    if rz:
        R = _rot_z(-rz) @ R
    if ry:
        R = _rot_y(-ry) @ R
    if rx:
        R = _rot_x(-rx) @ R
    if cs.reference_cs is not None:
        Rp, tp = cs_to_affine(cs.reference_cs)
        # child origin expressed in the global frame
        with np.errstate(invalid="ignore"):
            t = tp + Rp.T @ t
        R = R @ Rp
    return R, t


def _pack_geometry(geom, row, coeffs: list):
    name = type(geom).__name__
    row["coeff_offset"] = len(coeffs)
    row["n_coeff"] = 0
    row["radius"] = math.inf
    row["conic"] = 0.0
    row["tol"] = 0.0
    row["max_iter"] = 0
    row["norm_radius"] = 1.0
    if name == "Plane":
        row["geom_kind"] = S.GEOM_PLANE
        return
    if name == "StandardGeometry":
        row["geom_kind"] = S.GEOM_STANDARD
        row["radius"] = _f(geom.radius)
        row["conic"] = _f(geom.k)
        return
    if name in ("EvenAsphere", "OddAsphere"):
        row["geom_kind"] = (
            S.GEOM_EVEN_ASPHERE if name == "EvenAsphere" else S.GEOM_ODD_ASPHERE
        )
        row["radius"] = _f(geom.radius)
        row["conic"] = _f(geom.k)
        row["tol"] = float(geom.tol)
        row["max_iter"] = int(geom.max_iter)
        cs_ = [_f(c) for c in geom.coefficients]
        row["n_coeff"] = len(cs_)
        coeffs.extend(cs_)
        return
    if name == "PolynomialGeometry":
        row["geom_kind"] = S.GEOM_POLYNOMIAL
        row["radius"] = _f(geom.radius)
        row["conic"] = _f(geom.k)
        row["tol"] = float(geom.tol)
        row["max_iter"] = int(geom.max_iter)
        c = np.atleast_2d(np.asarray(_to_np(geom.coefficients), dtype=np.float64))
        row["n_coeff"] = c.size
        row["poly_cols"] = c.shape[1]
        coeffs.extend(c.reshape(-1).tolist())
        return
    if name == "ChebyshevPolynomialGeometry":
        row["geom_kind"] = S.GEOM_CHEBYSHEV
        row["radius"] = _f(geom.radius)
        row["conic"] = _f(geom.k)
        row["tol"] = float(geom.tol)
        row["max_iter"] = int(geom.max_iter)
        c = np.atleast_2d(np.asarray(_to_np(geom.coefficients), dtype=np.float64))
        row["n_coeff"] = c.size
        row["poly_cols"] = c.shape[1]
        coeffs.extend([_f(geom.norm_x), _f(geom.norm_y)])
        coeffs.extend(c.reshape(-1).tolist())
        return
    if name == "BiconicGeometry":
        row["geom_kind"] = S.GEOM_BICONIC
        row["radius"] = _f(geom.Rx)
        row["conic"] = _f(geom.kx)
        row["tol"] = float(geom.tol)
        row["max_iter"] = int(geom.max_iter)
        row["n_coeff"] = 2
        coeffs.extend([_f(geom.Ry), _f(geom.ky)])
        _base_conic_is_the_profiles(name, geom, _f(geom.Rx), _f(geom.kx))
        return
    if name == "ToroidalGeometry":
        _base_conic_is_the_profiles(name, geom, _f(geom.R_yz), 0.0)
        row["geom_kind"] = S.GEOM_TOROIDAL
        row["radius"] = _f(geom.R_yz)
        row["conic"] = 0.0  # base conic handed to NewtonRaphsonGeometry (toroidal.py:67-69)
        row["tol"] = float(geom.tol)
        row["max_iter"] = int(geom.max_iter)
        poly = [float(v) for v in np.asarray(_to_np(geom.coeffs_poly_y), dtype=np.float64)]
        row["n_coeff"] = 2 + len(poly)
        coeffs.extend([_f(geom.R_rot), _f(geom.k_yz)] + poly)
        return
    if name == "ZernikePolynomialGeometry":
        row["geom_kind"] = S.GEOM_ZERNIKE
        row["radius"] = _f(geom.radius)
        row["conic"] = _f(geom.k)
        row["tol"] = float(geom.tol)
        row["max_iter"] = int(geom.max_iter)
        row["norm_radius"] = _f(geom.norm_radius)
        z = geom.zernike
        cj = np.asarray(_to_np(z.coeffs), dtype=np.float64).reshape(-1)
        row["n_coeff"] = cj.size
        # (n, m) come from the scheme's own index table and N_j from its
        # _norm_constant (optiland/zernike/base.py:42-68, 145-193).
        for c, (n, m) in zip(cj, z.indices):
            coeffs.extend([float(c), float(n), float(m), _f(z._norm_constant(n, m))])
        return
    raise UnsupportedSystem(f"geometry {name} is not on the fused path")


def _base_conic_is_the_profiles(name, geom, radius, conic):
    """Biconic / toroidal surfaces under the opt-in reference Newton rule: the kernels start the
    iteration from the conic of the profile (Rx, kx / R_yz, 0), the reference from
    `geometry.radius` / `geometry.k` (newton_raphson.py:119-135) -- the same numbers unless
    `updater.set_radius / set_conic` moved the base apart from the profile
    (optic_updater.py:38-70).  The converged intersection does not depend on the start; the
    batch-global iteration COUNT does, so with that option such a surface goes to the
    reference's own loop."""
    if not S.OPTIONS["reference_newton"]:
        return
    r, k = _f(geom.radius), _f(geom.k)
    if not ((r == radius or (math.isinf(r) and math.isinf(radius))) and k == conic):
        raise UnsupportedSystem(f"{name}: base conic edited apart from the profile "
                                "(reference Newton rule)")


def _to_np(v):
    if hasattr(v, "detach"):
        return v.detach().cpu().numpy()
    return np.asarray(v)


def _leaf_token(ap, coeffs: list):
    name = type(ap).__name__
    if name in ("PolygonAperture", "FileAperture"):
        # polygon.py:37-41: vertices (n, 2); stored x0, y0, x1, y1, ... in the coefficients
        v = np.asarray(_to_np(ap.vertices), dtype=np.float64).reshape(-1, 2)
        off = len(coeffs)
        coeffs.extend(float(c) for c in v.reshape(-1))
        return [S.AP_POLYGON, float(off), float(v.shape[0]), 0.0, 0.0]
    if name == "RadialAperture":
        return [S.AP_RADIAL, _f(ap.r_min), _f(ap.r_max), 0.0, 0.0]
    if name == "OffsetRadialAperture":
        return [S.AP_OFFSET_RADIAL, _f(ap.r_min), _f(ap.r_max), _f(ap.offset_x), _f(ap.offset_y)]
    if name == "RectangularAperture":
        return [S.AP_RECTANGULAR, _f(ap.x_min), _f(ap.x_max), _f(ap.y_min), _f(ap.y_max)]
    if name == "EllipticalAperture":
        return [S.AP_ELLIPTICAL, _f(ap.a), _f(ap.b), _f(ap.offset_x), _f(ap.offset_y)]
    return None


_BOOL_OPS = {"UnionAperture": S.AP_OP_UNION, "IntersectionAperture": S.AP_OP_INTERSECTION,
             "DifferenceAperture": S.AP_OP_DIFFERENCE}


def _flatten_aperture(ap, out: list, coeffs: list):
    """Boolean aperture tree (physical_apertures/base.py:259-340) -> reverse-Polish
    tokens.  Returns the stack depth needed."""
    leaf = _leaf_token(ap, coeffs)
    if leaf is not None:
        out.append(leaf)
        return 1
    name = type(ap).__name__
    if name in _BOOL_OPS:
        da = _flatten_aperture(ap.a, out, coeffs)
        db = _flatten_aperture(ap.b, out, coeffs)
        out.append([_BOOL_OPS[name], 0.0, 0.0, 0.0, 0.0])
        return max(da, 1 + db)
    raise UnsupportedSystem(f"aperture {name} is not on the fused path")


def _pack_aperture(ap, row, coeffs: list):
    row["aperture_kind"] = S.AP_NONE
    if ap is None:
        return
    leaf = _leaf_token(ap, coeffs)
    if leaf is not None:
        row["aperture_kind"] = int(leaf[0])
        row["aperture"] = leaf[1:]
        return
    tokens: list = []
    depth = _flatten_aperture(ap, tokens, coeffs)
    if depth > 16:
        raise UnsupportedSystem("boolean aperture nesting deeper than 16")
    row["aperture_kind"] = S.AP_COMPOSITE
    row["aperture"] = [float(len(coeffs)), float(len(tokens)), 0.0, 0.0]
    for t in tokens:
        coeffs.extend(float(v) for v in t)


def _pack_coating(coating, row, coeffs: list):
    row["coating_kind"] = S.COAT_NONE
    if coating is None:
        return
    name = type(coating).__name__
    if name == "PolarizerCoating":
        row["coating_kind"] = S.COAT_POLARIZER
        row["coat"] = [float(len(coeffs)), 0.0]
        coeffs.extend(float(v) for v in _to_np(coating.jones.axis).reshape(-1))
        return
    if name == "RetarderCoating":
        row["coating_kind"] = S.COAT_RETARDER
        row["coat"] = [float(len(coeffs)), 0.0]
        coeffs.extend(float(v) for v in _to_np(coating.jones.axis).reshape(-1))
        coeffs.append(_f(coating.jones.retardance))
        return
    if name == "SimpleCoating":
        row["coating_kind"] = S.COAT_SIMPLE
        row["coat"] = [_f(coating.transmittance), _f(coating.reflectance)]
    elif name == "FresnelCoating":
        row["coating_kind"] = S.COAT_FRESNEL
    else:
        raise UnsupportedSystem(f"coating {name} is not on the fused path")


_INDEX_MEMO: dict = {}  # (id(material), wavelength, which) -> value; cleared per pack


def _scalar_index(material, w: float, which: str) -> float:
    """material.n(lambda) / material.k(lambda) with a *scalar* wavelength.

    The reference calls these with the whole N-vector `rays.w` and builds a
    cache key out of all N values (optiland/materials/base.py:73-79), the
    dominant CPU cost in its profile; per trace every ray shares one wavelength
    (rays/ray_generator.py:87), so a scalar call returns the same number.
    """
    key = (id(material), w, which)
    hit = _INDEX_MEMO.get(key)
    if hit is None:  # consecutive surfaces share materials (pre of one = post of the last)
        hit = _INDEX_MEMO[key] = _f(getattr(material, which)(w))
    return hit


def _pack_surface_local(is_object: bool, surf, wl):
    """One surface, packed by itself: (desc row, its coefficient block as a list with every
    offset RELATIVE to the block, optics rows per wavelength).  Raises `UnsupportedSystem`."""
    row = np.zeros((), dtype=S.SURFACE_DESC_DTYPE)
    opt = np.zeros(wl.size, dtype=S.SURFACE_OPTICS_DTYPE)
    coeffs: list = []
    geom = surf.geometry
    R, t = cs_to_affine(geom.cs)
    row["origin"] = t
    row["rot"] = R.reshape(-1)
    row["flags"] = 0 if (R is _EYE3 or np.array_equal(R, _EYE3)) else S.SURF_ROTATED
    if S.OPTIONS["reference_root"]:   # (the C side honours it on curved standard surfaces only)
        row["flags"] |= S.SURF_REFERENCE_ROOT
    im = surf.interaction_model
    if type(im).__name__ != "RefractiveReflectiveModel":
        raise UnsupportedSystem(
            f"interaction model {type(im).__name__} is not on the fused path"
        )
    if getattr(im, "bsdf", None) is not None:
        raise UnsupportedSystem("BSDF scatter is not on the fused path")
    _pack_geometry(geom, row, coeffs)
    if S.OPTIONS["reference_newton"] and int(row["geom_kind"]) not in (S.GEOM_PLANE,
                                                                      S.GEOM_STANDARD):
        # the reference's batch-global stop rule on this Newton-Raphson surface (opt-in)
        row["flags"] |= S.SURF_REFERENCE_NEWTON
    _pack_aperture(surf.aperture, row, coeffs)
    _pack_coating(surf.coating, row, coeffs)
    if is_object:
        # ObjectSurface.trace only records (surfaces/object_surface.py:56-93)
        row["interaction"] = S.INTERACT_RECORD_ONLY
        opt[:] = (1.0, 1.0, 0.0)
        opt["n2"] = [_scalar_index(surf.material_post, float(w), "n") for w in wl]
        return row, coeffs, opt
    row["interaction"] = S.INTERACT_REFLECT if im.is_reflective else S.INTERACT_REFRACT
    pre, post = surf.material_pre, surf.material_post
    for m in (pre, post):
        pm = type(m.propagation_model).__name__
        if pm != "HomogeneousPropagation":
            raise UnsupportedSystem(f"propagation model {pm} is not on the fused path")
    for j, w in enumerate(wl):
        w = float(w)
        n1 = _scalar_index(pre, w, "n")
        n2 = _scalar_index(post, w, "n")
        k1 = _scalar_index(pre, w, "k")
        # propagation/homogeneous.py:44-53: alpha = 4 pi k / lambda, applied
        # as exp(-alpha * t * 1e3) only when k > 0.
        absorb = (4.0 * math.pi * k1 / w) * 1e3 if k1 > 0 else 0.0
        opt[j] = (n1, n2, absorb)
    return row, coeffs, opt


def _relocate(row, local: list, base: int):
    """The row and the coefficient block of `_pack_surface_local` moved to offset `base` of
    the table's coefficient buffer: every stored offset shifted (the geometry block, a
    polygon's vertex block, a boolean aperture's token list and the polygon leaves inside
    it, a polarizer / retarder axis block)."""
    if base == 0:
        return row, local
    row = row.copy()
    row["coeff_offset"] = int(row["coeff_offset"]) + base
    ak = int(row["aperture_kind"])
    if ak == S.AP_POLYGON:
        ap = np.array(row["aperture"])
        ap[0] += base
        row["aperture"] = ap
    elif ak == S.AP_COMPOSITE:
        ap = np.array(row["aperture"])
        off, cnt = int(ap[0]), int(ap[1])
        ap[0] += base
        row["aperture"] = ap
        fixed = None
        for k in range(cnt):
            j = off + 5 * k
            if int(local[j]) == S.AP_POLYGON:
                if fixed is None:
                    fixed = list(local)
                fixed[j + 1] = local[j + 1] + base
        if fixed is not None:
            local = fixed
    if int(row["coating_kind"]) in (S.COAT_POLARIZER, S.COAT_RETARDER):
        c = np.array(row["coat"])
        c[0] += base
        row["coat"] = c
    return row, local


def pack_surfaces(surfaces, wavelengths, name: str = "surfaces",
                  tolerate: bool = False, tokens=None, cache: dict | None = None,
                  keep=None) -> SystemTable:
    """Flatten a sequence of reference `Surface` objects (a `SurfaceGroup`'s list) for
    the given wavelengths (microns): everything `SurfaceGroup.trace`
    (surfaces/surface_group.py:245-257) needs -- no ray-generator scalars, no
    polarisation state (those belong to the `Optic`, see `pack_optic`).

    Raises `UnsupportedSystem` for anything outside the fused path -- or, with
    `tolerate`, packs a never-traced placeholder row for every such surface (except the
    object surface) and lists their indices in `table.unsupported`: the caller then
    launches the fused trace on the `[first, last]` runs between them and leaves those
    surfaces to the reference (integration._hip_surface_group_trace).

    `tokens` + `cache` (both or neither): per-surface change-detector tokens
    (fingerprint.surface_token, i.e. element [1] of `optic_token`) and a dict the caller
    keeps -- a surface whose token is the one it was last packed under is NOT read again,
    its row and coefficient block are taken from the cache and only relocated.  A re-pack
    after `set_radius` then touches one surface instead of all of them.
    `keep`: the objects the tokens' `id()`s belong to (second element of `optic_token`): the
    cache pins them -- and the surfaces -- for as long as it holds rows stored under those
    tokens, so that no freed object can be recycled at the same address with the same version
    and make a stale row look current (the contract of fingerprint.py: ids are only valid
    while `keep` is held).
    """
    _INDEX_MEMO.clear()
    surfaces = list(surfaces)
    unsupported: list = []
    wl = np.array([float(w) for w in np.atleast_1d(wavelengths)], dtype=np.float64)
    wl_key = wl.tobytes() + (b"|reference_root" if S.OPTIONS["reference_root"] else b"") \
        + (b"|reference_newton" if S.OPTIONS["reference_newton"] else b"")
    n_s = len(surfaces)
    use_cache = cache is not None and tokens is not None and len(tokens) == n_s
    if use_cache:
        # rows of surfaces that left the optic go (their ids are free to be recycled); what
        # stays is pinned: the surfaces themselves and everything the current tokens name
        live = {id(s_) for s_ in surfaces}
        for k in [k for k in cache if (k if isinstance(k, int) else
                                       k[1] if isinstance(k, tuple) else None) not in live
                  and not isinstance(k, str)]:
            del cache[k]
        if keep is None:  # a caller without the keep list: rows without pins are not kept
            for k in [k for k in cache if not isinstance(k, str)]:
                del cache[k]
            cache.pop("assembled", None)
        cache["pinned"] = (surfaces, keep)

    if use_cache:
        table = _patched_table(surfaces, wl, wl_key, name, tokens, cache)
        if table is not None:
            return table
    desc = np.zeros(n_s, dtype=S.SURFACE_DESC_DTYPE)
    optics = np.zeros((n_s, wl.size), dtype=S.SURFACE_OPTICS_DTYPE)
    coeffs: list = []
    bases = []

    for i, surf in enumerate(surfaces):
        packed = None
        if use_cache:
            hit = cache.get(id(surf))
            if hit is not None and hit[0] == (tokens[i], i == 0, wl_key):
                packed = hit[1]
        if packed is None:
            try:
                packed = _pack_surface_local(i == 0, surf, wl)
            except UnsupportedSystem:
                if not tolerate or i == 0:
                    raise
                packed = None
            if use_cache and packed is not None:
                cache[id(surf)] = ((tokens[i], i == 0, wl_key), packed)
        if packed is None:
            desc[i]["rot"] = _EYE3.reshape(-1)  # placeholder row, never traced
            desc[i]["interaction"] = S.INTERACT_RECORD_ONLY
            optics[i, :] = (1.0, 1.0, 0.0)
            unsupported.append(i)
            continue
        row, local, opt = packed
        bases.append((len(coeffs), len(local)))
        row, local = _relocate(row, local, len(coeffs))
        desc[i] = row
        optics[i, :] = opt
        coeffs.extend(local)

    table = SystemTable(
        surfaces=desc,
        coeffs=np.asarray(coeffs, dtype=np.float64),
        optics=optics,
        wavelengths=wl,
        name=name,
    )
    table.last_thickness = _f(surfaces[-1].thickness) if n_s else 0.0
    table.unsupported = tuple(unsupported)
    if use_cache and not unsupported:
        cache["assembled"] = (tuple(id(s_) for s_ in surfaces), wl_key, list(tokens), bases,
                              table)
    return table


def _patched_table(surfaces, wl, wl_key, name, tokens, cache):
    """The table of the previous `pack_surfaces` call with the rows of the CHANGED surfaces
    replaced, or None when that does not apply (other surfaces, other wavelengths, a
    coefficient block that changed its length, an unsupported surface).  An optimiser's
    `set_radius` re-pack then costs one surface, not one surface plus the re-assembly of all."""
    prev = cache.get("assembled")
    if prev is None:
        return None
    ids, p_wl, p_tokens, bases, p_table = prev
    n_s = len(surfaces)
    if p_wl != wl_key or len(ids) != n_s or any(id(s_) != k for s_, k in zip(surfaces, ids)):
        return None
    changed = [i for i in range(n_s) if tokens[i] != p_tokens[i]]
    desc = p_table.surfaces.copy()
    optics = p_table.optics.copy()
    coeffs = p_table.coeffs.copy()
    for i in changed:
        try:
            packed = _pack_surface_local(i == 0, surfaces[i], wl)
        except UnsupportedSystem:
            return None  # the full path decides (tolerate / raise)
        row, local, opt = packed
        base, length = bases[i]
        if len(local) != length:
            return None
        cache[id(surfaces[i])] = ((tokens[i], i == 0, wl_key), packed)
        row, local = _relocate(row, local, base)
        desc[i] = row
        optics[i, :] = opt
        if length:
            coeffs[base:base + length] = local
    table = SystemTable(surfaces=desc, coeffs=coeffs, optics=optics, wavelengths=wl, name=name)
    table.last_thickness = _f(surfaces[-1].thickness) if n_s else 0.0
    table.unsupported = ()
    cache["assembled"] = (ids, wl_key, list(tokens), bases, table)
    return table


def pack_optic(optic, wavelengths=None, name: str | None = None, tokens=None,
               cache: dict | None = None, keep=None) -> SystemTable:
    """Flatten `optic` (a reference `Optic`) for the given wavelengths (microns):
    `pack_surfaces` + the ray-generator scalars + the polarisation state.
    `tokens`, `cache`: see `pack_surfaces` (incremental re-pack).

    Raises `UnsupportedSystem` for anything outside the fused path.
    """
    global _TENSOR_CACHE_ACTIVE
    from . import fingerprint as _fp

    before = _TENSOR_CACHE_ACTIVE
    _TENSOR_CACHE_ACTIVE = bool(_fp.ENABLED and tokens is not None and cache is not None)
    try:
        return _pack_optic(optic, wavelengths, name, tokens, cache, keep)
    finally:
        _TENSOR_CACHE_ACTIVE = before


def _pack_optic(optic, wavelengths, name, tokens, cache, keep) -> SystemTable:
    if wavelengths is None:
        wavelengths = [_f(w.value) for w in optic.wavelengths.wavelengths]
    table = pack_surfaces(optic.surfaces, wavelengths,
                          name or (optic.name or type(optic).__name__), tokens=tokens,
                          cache=cache, keep=keep)
    _pack_raygen(optic, table, tokens, cache)
    table.primary_wavelength = _f(optic.primary_wavelength)
    pol = optic.polarization
    if pol != "ignore":
        st = optic.polarization_state
        table.polarization = {
            "is_polarized": bool(st.is_polarized),
            "Ex": None if st.Ex is None else _f(st.Ex),
            "Ey": None if st.Ey is None else _f(st.Ey),
            "phase_x": None if st.phase_x is None else _f(st.phase_x),
            "phase_y": None if st.phase_y is None else _f(st.phase_y),
        }
    return table


# The paraxial scalars below are produced by the REFERENCE's own paraxial tracer
# (Paraxial.EPL / EPD / XPL: three paraxial traces, each walking every surface through
# `position_in_gcs`); on the drop-in path they cost more than the whole GPU trace of a
# small ray batch.  They only depend on the first-order layout at the PRIMARY
# wavelength, so they are memoised per optic against a fingerprint of everything they
# can depend on -- analyses that call Optic.trace() per field and per wavelength
# (spot diagrams, ray fans) then pay for them once.
_RAYGEN_CACHE: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


def _raygen_fingerprint(optic, table: SystemTable):
    prim = _f(optic.primary_wavelength)
    ap = optic.aperture
    fd = optic.fields.field_definition
    surfaces = list(optic.surfaces)
    idx = tuple(_scalar_index(s.material_post, prim, "n") for s in surfaces)
    stop = tuple(bool(getattr(s, "is_stop", False)) for s in surfaces)
    fields = tuple((_f(f.x), _f(f.y), _f(f.vx), _f(f.vy)) for f in optic.fields.fields)
    mode = getattr(optic.ray_tracer, "ray_aiming_config", {}).get("mode", "paraxial")
    # `SurfaceGroup.radii` (surface_group.py: geometry.radius of every surface) is what the
    # paraxial tracer reads.  For a biconic / toroidal surface that is NOT the table's radius:
    # the sag has its own Rx / R_yz and `updater.set_radius` moves `geometry.radius` alone
    # (optic_updater.py:38-54, biconic.py:56-66, toroidal.py:67-82) -- the packed bytes stay
    # what they were while EPL / EPD move (found by tools/seam_fuzz.py edit_loop, round 6)
    radii = tuple(_f(getattr(s.geometry, "radius", math.inf)) for s in surfaces)
    return hash((
        table.surfaces.tobytes(), table.coeffs.tobytes(), radii, prim, idx, stop,
        type(ap).__name__, None if ap is None else _f(ap.value),
        type(fd).__name__, fields, bool(optic.object_surface.is_infinite),
        bool(optic.obj_space_telecentric), _pack_apodization(optic.apodization), mode,
    ))


def _pack_raygen(optic, table: SystemTable, tokens=None, cache=None) -> None:
    # host first-order model (paraxial_host.py): as cheap as the memo's own fingerprint --
    # computed outright, no memo
    if _compute_raygen(optic, table, host_only=True, tokens=tokens, cache=cache):
        return
    try:
        key = _raygen_fingerprint(optic, table)
        hit = _RAYGEN_CACHE.get(optic)
    except TypeError:  # an unhashable / non-weakref-able Optic subclass: no memo
        key, hit = None, None
    if hit is not None and hit[0] == key:
        table.raygen, table.fields = dict(hit[1]), list(hit[2])
        return
    _compute_raygen(optic, table)
    if key is not None:
        try:
            _RAYGEN_CACHE[optic] = (key, dict(table.raygen), list(table.fields))
        except TypeError:
            pass


def _pack_apodization(ap):
    """(kind, a, b) of optiland/apodization/*.py, or None for a class the device
    generator does not know."""
    if ap is None:
        return S.APOD_NONE, 0.0, 0.0
    name = type(ap).__name__
    if name == "UniformApodization":
        return S.APOD_NONE, 0.0, 0.0
    if name == "GaussianApodization":
        return S.APOD_GAUSSIAN, _f(ap.sigma), 0.0
    if name == "CosineSquaredApodization":
        return S.APOD_COSINE_SQUARED, _f(ap.R), 0.0
    if name == "HannApodization":
        return S.APOD_HANN, _f(ap.D), 0.0
    if name == "PolynomialApodization":
        return S.APOD_POLYNOMIAL, _f(ap.R), _f(ap.p)
    if name == "SuperGaussianApodization":
        return S.APOD_SUPER_GAUSSIAN, _f(ap.w), _f(ap.n)
    if name == "TukeyApodization":
        return S.APOD_TUKEY, _f(ap.R), _f(ap.alpha)
    return None


def _compute_raygen(optic, table: SystemTable, host_only: bool = False, tokens=None,
                    cache=None) -> bool:
    """Scalars for on-device ray generation.  `host_only`: give up
    (return False, table untouched) as soon as the reference's paraxial tracer would have to
    be asked; otherwise True.

    Packed: paraxial aiming, any apodization of optiland/apodization/, for AngleField
    (object at infinity or finite, fields/field_types/angle.py:17-58), ObjectHeightField
    on a planar object (object_height.py:19-47) and ParaxialImageHeightField
    (paraxial_image_height.py:19-60), incl. the object-space-telecentric branch of the
    aimer (rays/ray_aiming/paraxial.py:33-106).  Otherwise (iterative / robust aiming,
    real image height fields) `table.raygen` stays empty and callers generate rays with
    the reference's own RayGenerator.
    """
    fd = optic.fields.field_definition
    kind = {"AngleField": S.FIELD_ANGLE, "ObjectHeightField": S.FIELD_OBJECT_HEIGHT,
            "ParaxialImageHeightField": S.FIELD_PARAXIAL_IMAGE_HEIGHT}.get(type(fd).__name__)
    if kind is None:
        return True
    apod = _pack_apodization(optic.apodization)
    if apod is None:
        return True
    mode = getattr(optic.ray_tracer, "ray_aiming_config", {}).get("mode", "paraxial")
    if mode != "paraxial":
        return True
    obj = optic.object_surface
    # surfaces/object_surface.py:48-50 `is_infinite` = isinf(cs.z), read through the packer's
    # read-back cache instead of a backend reduction
    infinite = math.isinf(_f(obj.geometry.cs.z))
    if kind == S.FIELD_OBJECT_HEIGHT or (kind == S.FIELD_PARAXIAL_IMAGE_HEIGHT and not infinite):
        # object_height.py:36-47: z0 = obj.geometry.sag(x0, y0) + obj z -- planar objects only
        if infinite or table.surfaces[0]["geom_kind"] != S.GEOM_PLANE:
            return True
    tele_dz = 0.0
    if optic.obj_space_telecentric:
        # ray_aiming/paraxial.py:82-87, 108-123: object-height fields with an
        # object-NA aperture only; z1 - z0 = sqrt(1 - sin^2) / sin
        if kind == S.FIELD_ANGLE or type(optic.aperture).__name__ != "ObjectNAAperture":
            return True
        sin = _f(optic.aperture.value)
        if not 0.0 < sin < 1.0:
            return True
        tele_dz = math.sqrt(1.0 - sin * sin) / sin
    # SurfaceGroup.positions (surface_group.py:155-161) = z of every vertex in the
    # global frame = the origins already folded by cs_to_affine
    pos = np.asarray(table.surfaces["origin"][:, 2], dtype=np.float64).reshape(-1)
    fo = _host_first_order(optic, table, pos, infinite, tokens, cache) \
        if kind != S.FIELD_PARAXIAL_IMAGE_HEIGHT else None
    if fo is not None:
        # the reference's own recurrences on the packed table (paraxial_host.py): no backend
        # array operation, no read-back
        EPL, EPD = fo["EPL"], fo["EPD"]
    elif host_only:
        return False
    else:
        EPL = _f(optic.paraxial.EPL())
        EPD = _f(optic.paraxial.EPD())
    if infinite:
        # angle.py:102-118 (the same expression in paraxial_image_height.py:124-140)
        offset = (EPD - float(np.min(pos[1:-1]))) if fo is not None \
            else _f(fd._get_starting_z_offset(optic))
        z_first = float(pos[1])
    else:
        offset = 0.0
        z_first = float(pos[0])
    if fo is not None:   # fields/field_group.py:63-67
        fxy = [(_f(f.x), _f(f.y)) for f in optic.fields.fields]
        max_field = max((math.hypot(a, b) for a, b in fxy), default=0.0)
    else:
        max_field = _f(optic.fields.max_field)
    field_scale = max_field
    if kind == S.FIELD_PARAXIAL_IMAGE_HEIGHT:
        # paraxial_image_height.py:36-60: two unit paraxial chief-ray traces from the stop
        # give the linear map from image height to object slope (infinite) / height
        y_img_unit = _f(fd._trace_unit_chief_ray(optic, plane="image")[0])
        y_obj_unit, u_obj_unit = (_f(v) for v in fd._trace_unit_chief_ray(optic, plane="object"))
        field_scale = (u_obj_unit if infinite else y_obj_unit) * max_field / y_img_unit
    table.raygen = {
        "object_infinite": 1.0 if infinite else 0.0,
        "field_kind": float(kind),
        "field_scale": field_scale,
        "EPL": EPL,
        "EPD": EPD,
        "max_field": max_field,
        "offset": offset,
        "z_first": z_first,
        "tele_dz": tele_dz,
        "apod_kind": float(apod[0]),
        "apod_a": apod[1],
        "apod_b": apod[2],
    }
    table.fields = [
        (_f(f.x), _f(f.y), _f(f.vx), _f(f.vy)) for f in optic.fields.fields
    ]
    # wavefront analysis (wavefront/strategy.py:157-160, 63): exit pupil z and the
    # image-space index at the PRIMARY wavelength
    try:
        if fo is not None:
            table.raygen["pupil_z"] = fo["XPL"] + float(pos[-1])
            table.raygen["n_image"] = fo["n_image"]
        else:
            table.raygen["pupil_z"] = _f(optic.paraxial.XPL()) + float(pos[-1])
            table.raygen["n_image"] = _f(optic.surfaces.n(optic.primary_wavelength)[-1])
    except Exception:  # systems without a well-defined exit pupil: no wavefront data
        pass
    return True


_HOST_PARAXIAL_GEOMS = frozenset((S.GEOM_PLANE, S.GEOM_STANDARD, S.GEOM_EVEN_ASPHERE,
                                  S.GEOM_ODD_ASPHERE, S.GEOM_POLYNOMIAL, S.GEOM_CHEBYSHEV,
                                  S.GEOM_ZERNIKE))


def _primary_indices(surfaces, prim: float, tokens, cache):
    """n(primary wavelength) behind every surface.  With the change-detector tokens of an
    incremental re-pack (`pack_surfaces`), a surface whose token stands keeps the value it
    had: the token covers both material objects, so the ~13 `material.n()` calls of a
    re-pack (a cache-key build each, materials/base.py:73-100) shrink to the edited one."""
    if cache is None or tokens is None or len(tokens) != len(surfaces):
        return [_scalar_index(s.material_post, prim, "n") for s in surfaces]
    out = []
    for s, tok in zip(surfaces, tokens):
        key = ("n_primary", id(s))
        hit = cache.get(key)
        if hit is not None and hit[0] == prim and hit[1] == tok:
            out.append(hit[2])
            continue
        v = _scalar_index(s.material_post, prim, "n")
        cache[key] = (prim, tok, v)
        out.append(v)
    return out


def _host_first_order(optic, table: SystemTable, pos, infinite, tokens=None, cache=None):
    """EPL / EPD / XPL / n_image from the packed table (paraxial_host.first_order), or None
    when the system has something that restatement does not cover."""
    import os

    if os.environ.get("OPTILAND_HIP_HOST_PARAXIAL", "1") == "0":
        return None
    from . import paraxial_host

    surf = table.surfaces
    if not _HOST_PARAXIAL_GEOMS.issuperset(surf["geom_kind"].tolist()):
        return None
    surfaces = list(optic.surfaces)
    stop = None
    for i, s in enumerate(surfaces):
        if getattr(s, "surface_type", None) == "paraxial":
            return None
        if stop is None and getattr(s, "is_stop", False):
            stop = i
    obj = surfaces[0]
    if getattr(obj.geometry.cs, "reference_cs", None) is not None:
        return None
    ap = optic.aperture
    if stop is None or ap is None:
        return None
    prim = _f(optic.primary_wavelength)
    n = _primary_indices(surfaces, prim, tokens, cache)
    reflect = (surf["interaction"] == S.INTERACT_REFLECT).tolist()
    fo = paraxial_host.first_order(surf["radius"].tolist(), n, pos.tolist(), reflect, stop,
                                   type(ap).__name__, _f(ap.value), infinite,
                                   _f(obj.geometry.cs.z))
    if fo is not None:
        fo["n_image"] = n[-1]
    return fo
