"""ctypes wrapper around oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

May be imported by tests/, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
bench.py -- never by anything under optiland_amd/.
"""

from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (seconds)."""
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "trace_oracle.c")
    hdr = os.path.join(_HERE, "..", "include", "optiland_hip.h")
    stale = (
        force
        or not os.path.exists(so)
        or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr))
    )
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"],
                              stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.oracle_trace.restype = C.c_uint32
        _LIB.oracle_polarized_intensity.restype = C.c_uint32
        _LIB.oracle_sag.restype = C.c_double
    return _LIB


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class PolState(C.Structure):
    _fields_ = [
        ("is_polarized", C.c_int32),
        ("reserved_", C.c_int32),
        ("Ex", C.c_double),
        ("Ey", C.c_double),
        ("phase_x", C.c_double),
        ("phase_y", C.c_double),
    ]


def pol_state(polarization: dict | None) -> PolState:
    if not polarization or not polarization.get("is_polarized"):
        return PolState(0, 0, 0.0, 0.0, 0.0, 0.0)
    return PolState(1, 0, polarization["Ex"], polarization["Ey"],
                    polarization["phase_x"], polarization["phase_y"])


def trace(table, rays, wavelength_index=0, record=True, polarized=False,
          first=0, last=None, record_out=None):
    """Trace `rays` (dict of float64 arrays x,y,z,L,M,N,i[,opd]) through `table`.

    Returns dict(final rays..., record=(rows,8,n) or None, prt=(n,3,3) complex or
    None, pre_dir=(3,n), status=int).  Inputs are not modified.  `record_out`: an
    optional preallocated C-contiguous float64 (rows, 8, n) array to record into (the
    timing harness reuses one so that it measures the trace, not page faults).
    """
    n = int(np.asarray(rays["x"]).size)
    last = table.num_surfaces - 1 if last is None else last
    planes = []
    for k in ("x", "y", "z", "L", "M", "N", "i", "opd"):
        if k == "opd" and k not in rays:
            planes.append(np.zeros(n))
        else:
            planes.append(np.ascontiguousarray(np.array(rays[k], dtype=np.float64)).copy())
    arr = (C.c_void_p * 8)(*[_ptr(p) for p in planes])
    rows = last - first + 1
    rec = None
    if record:
        if record_out is not None:
            assert record_out.shape == (rows, 8, n) and record_out.dtype == np.float64 \
                and record_out.flags.c_contiguous
            rec = record_out
        else:
            rec = np.zeros((rows, 8, n))
    prt = None
    if polarized:
        prt = np.tile(np.eye(3, dtype=np.complex128), (n, 1, 1))
    pre = np.zeros((3, n))
    surf = np.ascontiguousarray(table.surfaces)
    optics = np.ascontiguousarray(table.optics)
    coeffs = np.ascontiguousarray(table.coeffs, dtype=np.float64)
    status = lib().oracle_trace(
        _ptr(surf), C.c_int32(table.num_surfaces), _ptr(coeffs), _ptr(optics),
        C.c_int32(optics.shape[1]), C.c_int32(wavelength_index), C.c_int64(n), arr,
        _ptr(rec) if rec is not None else None,
        _ptr(prt) if prt is not None else None, _ptr(pre),
        C.c_int32(first), C.c_int32(last),
    )
    out = dict(zip(("x", "y", "z", "L", "M", "N", "i", "opd"), planes))
    out.update(record=rec, prt=prt, pre_dir=pre, status=int(status))
    return out


def generate_rays(raygen: dict, hx, hy, px, py, vx=None, vy=None):
    from optiland_amd.system import RAYGEN_DTYPE

    p = np.zeros(1, dtype=RAYGEN_DTYPE)
    p["object_infinite"] = int(raygen["object_infinite"])
    p["field_kind"] = int(raygen.get("field_kind", 0))
    p["tele_dz"] = float(raygen.get("tele_dz", 0.0))
    p["apod_kind"] = int(raygen.get("apod_kind", 0))
    p["apod_a"] = float(raygen.get("apod_a", 0.0))
    p["apod_b"] = float(raygen.get("apod_b", 0.0))
    for k in ("EPL", "EPD", "max_field", "offset", "z_first"):
        p[k] = raygen[k]
    p["max_field"] = float(raygen.get("field_scale", raygen["max_field"]))
    hx, hy, px, py = (np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64),
                      np.broadcast(hx, hy, px, py).shape)).reshape(-1).copy()
                      for a in (hx, hy, px, py))
    n = hx.size
    outs = [np.zeros(n) for _ in range(7)]
    arr = (C.c_void_p * 7)(*[_ptr(o) for o in outs])
    vxp = _ptr(np.ascontiguousarray(vx, dtype=np.float64)) if vx is not None else None
    vyp = _ptr(np.ascontiguousarray(vy, dtype=np.float64)) if vy is not None else None
    lib().oracle_generate_rays(_ptr(p), C.c_int64(n), _ptr(hx), _ptr(hy), _ptr(px),
                               _ptr(py), vxp, vyp, arr)
    return dict(zip(("x", "y", "z", "L", "M", "N", "i"), outs))


def polarized_intensity(prt, L0, M0, N0, i0, polarization):
    n = int(np.asarray(L0).size)
    prt = np.ascontiguousarray(prt, dtype=np.complex128)
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in (L0, M0, N0, i0)]
    out = np.zeros(n)
    st = pol_state(polarization)
    status = lib().oracle_polarized_intensity(C.c_int64(n), _ptr(prt), *[_ptr(v) for v in a],
                                              C.byref(st), _ptr(out))
    return out, int(status)


def polygon_contains(vertices, x, y):
    """Restated matplotlib crossings test for an (n, 2) vertex array and point arrays."""
    v = np.ascontiguousarray(np.asarray(vertices, dtype=np.float64).reshape(-1, 2))
    f = lib().oracle_polygon_contains
    f.restype = C.c_int
    xs, ys = np.atleast_1d(np.asarray(x, float)), np.atleast_1d(np.asarray(y, float))
    return np.array([bool(f(_ptr(v), C.c_int(v.shape[0]), C.c_double(a), C.c_double(b)))
                     for a, b in zip(xs, ys)])


def sag(table, surface_index, x, y):
    surf = np.ascontiguousarray(table.surfaces[surface_index:surface_index + 1])
    coeffs = np.ascontiguousarray(table.coeffs, dtype=np.float64)
    return lib().oracle_sag(_ptr(surf), _ptr(coeffs), C.c_double(x), C.c_double(y))


def normal(table, surface_index, x, y):
    surf = np.ascontiguousarray(table.surfaces[surface_index:surface_index + 1])
    coeffs = np.ascontiguousarray(table.coeffs, dtype=np.float64)
    out = np.zeros(3)
    lib().oracle_normal(_ptr(surf), _ptr(coeffs), C.c_double(x), C.c_double(y), _ptr(out))
    return out


def distance(table, surface_index, x, y, z, L, M, N):
    surf = np.ascontiguousarray(table.surfaces[surface_index:surface_index + 1])
    coeffs = np.ascontiguousarray(table.coeffs, dtype=np.float64)
    a = [np.ascontiguousarray(np.atleast_1d(v), dtype=np.float64) for v in (x, y, z, L, M, N)]
    t = np.zeros(a[0].size)
    lib().oracle_distance(_ptr(surf), _ptr(coeffs), C.c_int64(a[0].size),
                          *[_ptr(v) for v in a], _ptr(t))
    return t


def wavefront_fit(kind: str, params: dict, rays8, px, py, *, trim_std=3.0, flavour="torch",
                  planar=False) -> dict:
    """CentroidStrategy / BestFitStrategy of the reference restated in NumPy, operation for
    operation (wavefront/strategy.py:287-620): from the bundle at the image surface `rays8` =
    x, y, z, L, M, N, opd, intensity to the reference sphere / plane, the piston, the OPD map in
    waves and the reference-surface intersection points.  `params`: n_image, wavelength_um,
    ux, uy, half_epd (the launch-plane tilt, strategy.py:88-139).  `flavour`: which backend's
    reductions -- "torch": std with n - 1 (torch.std) and a NaN-ignoring mean
    (backend/torch_backend.py:969-1001), "numpy": np.std / np.mean.  Raises the reference's
    ValueErrors.  Pinned by tests/golden/wavefront_fitted.npz (tools/make_golden_fitted.py)."""
    x, y, z, L, M, N, opd, inten = (np.asarray(v, dtype=np.float64) for v in rays8)
    px, py = np.asarray(px, dtype=np.float64), np.asarray(py, dtype=np.float64)
    ni = float(params["n_image"])
    wl = float(params["wavelength_um"])
    ux, uy, half = (float(params.get(k, 0.0)) for k in ("ux", "uy", "half_epd"))
    torch_like = flavour == "torch"

    def mean(v):
        if torch_like:
            ok = ~np.isnan(v)
            return v[ok].sum() / ok.sum() if ok.any() else np.nan
        return np.mean(v)

    with np.errstate(all="ignore"):
        opd = opd + (ux * (px * half) + uy * (py * half))             # :318-319 _correct_tilt
        valid = (np.isfinite(x) & np.isfinite(y) & np.isfinite(z) & np.isfinite(L)
                 & np.isfinite(M) & np.isfinite(N) & np.isfinite(opd) & (inten != 0))  # :376-385
        if not valid.any():
            raise ValueError("No valid ray samples found for best-fit geometry.")
        P = np.stack((x, y, z), axis=1)[valid]
        D = np.stack((L, M, N), axis=1)[valid]
        pts = P - (opd[valid] / ni)[:, None] * D                      # :389-393
        normal = None
        if kind == "centroid":
            w = inten[valid]
            w = np.where(w < 0.0, 0.0, w)                             # :406-408
            total = w.sum()
            if total == 0:
                w = np.ones_like(w)
                total = w.sum()
            if trim_std and trim_std > 0:                             # :417-429
                c0 = (P * w[:, None]).sum(axis=0) / total
                dist = np.linalg.norm(P - c0, axis=1)
                mean_d = mean(dist)
                std_d = np.std(dist, ddof=1 if torch_like else 0)
                if std_d > 0:
                    keep = dist <= mean_d + trim_std * std_d
                    if keep.sum() >= 4:
                        w = w * keep
            total = w.sum()
            center = (P * w[:, None]).sum(axis=0) / total             # :447-450
            if planar:                                                # :485-517
                normal = (D * w[:, None]).sum(axis=0) / w.sum()
                nrm = np.linalg.norm(normal)
                if nrm > 0:
                    normal = normal / nrm
                radius = np.inf
            else:                                                     # :457-474
                radius = float((w * np.linalg.norm(pts - center, axis=1)).sum() / w.sum())
        elif kind == "best_fit":
            if pts.shape[0] < 4:
                raise ValueError("Need at least 4 valid ray samples for best-fit.")
            if planar:                                                # :584-605
                center = pts.mean(axis=0)
                normal = np.linalg.svd(pts - center, full_matrices=False)[2][-1, :]
                radius = np.inf
            else:                                                     # :556-582
                A = np.stack([pts[:, 0], pts[:, 1], pts[:, 2], np.ones(len(pts))], axis=1)
                b = (pts ** 2).sum(axis=1)
                c = np.linalg.lstsq(A, b, rcond=None)[0]
                center = c[:3] / 2
                radius = float(np.sqrt(c[3] + (center ** 2).sum()))
        else:
            raise ValueError(kind)
        # reference_geometry.py:55-83 / 99-124 path_length, strategy.py:321-345
        bl, bm, bn = -L, -M, -N
        if normal is None:
            xc, yc, zc = center
            a_ = bl ** 2 + bm ** 2 + bn ** 2
            b_ = 2 * (bl * (x - xc) + bm * (y - yc) + bn * (z - zc))
            c_ = (x ** 2 + y ** 2 + z ** 2 - 2 * (x * xc + y * yc + z * zc)
                  + xc ** 2 + yc ** 2 + zc ** 2 - radius ** 2)
            d_ = b_ ** 2 - 4 * a_ * c_
            d_ = np.where(d_ < 0, 0, d_)
            t1 = (-b_ - np.sqrt(d_)) / (2 * a_)
            t2 = (-b_ + np.sqrt(d_)) / (2 * a_)
            t = np.where(t1 < 0, t2, t1)
        else:
            num = (x - center[0]) * normal[0] + (y - center[1]) * normal[1] \
                + (z - center[2]) * normal[2]
            den = bl * normal[0] + bm * normal[1] + bn * normal[2]
            den = np.where(np.abs(den) < 1e-12, 1e-12, den)
            t = -num / den
        opd_img = ni * t
        o = opd - opd_img
        alive = inten > 0
        if not alive.any():
            raise ValueError("No valid rays with non-zero intensity for OPD calculation.")
        mean_opd = mean(o[alive])
        waves = (mean_opd - o) / (wl * 1e-3)
        tt = opd_img / ni
        pupil = np.stack([x - tt * L, y - tt * M, z - tt * N])
    return dict(center=np.asarray(center, dtype=np.float64), radius=radius, normal=normal,
                opd_ref=float(mean_opd), opd=waves, pupil=pupil)


class WavefrontParams(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("xc", "yc", "zc", "R", "n_image", "opd_ref", "ux",
                                          "uy", "half_epd", "wavelength_um", "nx", "ny", "nz", "last_thickness", "last_absorb")]


def wavefront_opd(params: dict, rays7, px, py):
    """rays7: x,y,z,L,M,N,opd at the image surface.  Returns (opd_waves, pupil(3,n))."""
    a = [np.ascontiguousarray(v, dtype=np.float64) for v in rays7]
    px, py = (np.ascontiguousarray(v, dtype=np.float64) for v in (px, py))
    n = a[0].size
    p = WavefrontParams(**{k: float(params.get(k, 0.0)) for k, _ in WavefrontParams._fields_})
    out = np.zeros(n)
    pupil = np.zeros((3, n))
    rp = (C.c_void_p * 7)(*[_ptr(v) for v in a])
    pp = (C.c_void_p * 3)(*[_ptr(pupil[k]) for k in range(3)])
    lib().oracle_wavefront_opd(C.byref(p), C.c_int64(n), rp, _ptr(px), _ptr(py), _ptr(out), pp)
    return out, pupil
