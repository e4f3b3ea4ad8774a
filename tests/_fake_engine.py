"""Oracle-backed stand-in for `optiland_amd.engine.HipSystem` (tests only).

Lets the `-m "not gpu"` suite exercise the HOST logic (tracer mirror, reference
integration, sharding) on CPU tensors.  It lives under tests/ and is injected by
monkeypatching `optiland_amd.tracer._make_engine`; the product has no such path.
"""

from __future__ import annotations

import numpy as np
import torch

from optiland_amd import system as S
from optiland_amd.engine import PLANES, TraceResult
from oracle import oracle


class OracleEngine:
    def __init__(self, table, device=None):
        self.table = table
        self.device = torch.device("cpu")
        self.calls = 0

    @property
    def num_surfaces(self):
        return self.table.num_surfaces

    def close(self):
        pass

    def trace(self, rays, wavelength_index=0, record=True, prt=None, first=0, last=None,
              write_rays=None, check_status=True, prt_identity=False,
              nonunit_directions=False):
        # (nonunit_directions: the oracle restates polarized_rays.py:136-202 literally -- k as
        # it comes -- so it needs no word about it)
        self.calls += 1
        rays = list(rays)
        n = int(rays[0].numel())
        dtype = rays[0].dtype
        last = self.num_surfaces - 1 if last is None else last
        if prt is None and any(self.table.surfaces["coating_kind"][first:last + 1] >= S.COAT_FRESNEL):
            raise ValueError("Polarization must be set when surfaces have "
                             "polarization-dependent coatings.")
        d = {k: rays[j].double().numpy() for j, k in enumerate(PLANES)}
        out = oracle.trace(self.table, d, wavelength_index, record=bool(record is not False
                           and record is not None), polarized=prt is not None,
                           first=first, last=last)
        if out["status"] & S.STATUS_ZERNIKE_RANGE:
            raise ValueError(
                "Zernike coordinates must be normalized "
                "to [-1, 1]. Consider updating the normalization "
                "radius to 1.1x the surface aperture.")
        if out["status"] & S.STATUS_CHEBYSHEV_RANGE:
            raise ValueError(
                "Chebyshev input coordinates must be normalized "
                "to [-1, 1]. Consider updating the normalization "
                "factors.")
        rec = None
        if out["record"] is not None:
            if isinstance(record, torch.Tensor):
                rec = record
                rec[: out["record"].shape[0], :, :n].copy_(torch.as_tensor(out["record"], dtype=dtype))
            else:
                rec = torch.as_tensor(out["record"], dtype=dtype)
        if write_rays is None:
            write_rays = rec is None
        if write_rays:
            for j, k in enumerate(PLANES):
                rays[j].copy_(torch.as_tensor(out[k], dtype=dtype))
        if prt is not None:
            if prt_identity:
                prt.zero_()
                prt[0].fill_(1), prt[4].fill_(1), prt[8].fill_(1)
            p = out["prt"]  # (n,3,3) complex; oracle started from identity -> multiply in
            start = prt[:9].t().reshape(n, 3, 3).double().numpy().astype(np.complex128)
            if prt.shape[0] == 18:
                start = start + 1j * prt[9:].t().reshape(n, 3, 3).double().numpy()
            newp = np.einsum("nij,njk->nik", p, start)
            prt[:9].copy_(torch.as_tensor(newp.real.reshape(n, 9).T.copy(), dtype=dtype))
            if prt.shape[0] == 18:
                prt[9:].copy_(torch.as_tensor(newp.imag.reshape(n, 9).T.copy(), dtype=dtype))
        return TraceResult(n, rays, rec, prt, out["status"], first, last)

    def row0_planes(self, record, n):
        return [record[0, k, :n] for k in range(8)]

    def alloc_record(self, n, dtype, rows=None):
        rows = self.num_surfaces if rows is None else rows
        return torch.empty((rows, 8, max(n, 1)), dtype=dtype)

    def generate_rays(self, hx, hy, px, py, vx=None, vy=None, out=None, *, flags=0,
                      zero_status=True):
        n = int(px.numel())

        def plane(v, default):
            if v is None:
                v = default
            if isinstance(v, torch.Tensor):
                return v.double().numpy()
            return np.full(n, float(v))

        H = [plane(hx, 0.0), plane(hy, 0.0)]
        P = [plane(px, 0.0), plane(py, 0.0)]
        V = [plane(vx, 1.0), plane(vy, 1.0)]
        bad = lambda a: not bool(np.all((a >= -1) & (a <= 1)))  # noqa: E731
        if flags & 0x1 and (bad(H[0]) or bad(H[1])):
            raise ValueError("Normalized field coordinates must be within (-1, 1)")
        if flags & 0x2 and (bad(P[0]) or bad(P[1])):
            raise ValueError("Normalized pupil coordinates must be within (-1, 1)")
        if flags & 0x4:  # trace_generic: pupil pre-scaled by (1 - v)
            P = [P[0] * V[0], P[1] * V[1]]
        g = oracle.generate_rays(self.table.raygen, H[0], H[1], P[0], P[1], V[0], V[1])
        planes = [torch.as_tensor(g[k], dtype=px.dtype) for k in ("x", "y", "z", "L", "M", "N", "i")]
        if out is not None:
            for dst, src in zip(out, planes):
                dst.copy_(src)
            if len(out) > 7:
                out[7].zero_()
            return list(out[:7])
        return planes

    def polarized_intensity(self, prt, k0, i0, polarization):
        n = int(i0.numel())
        p = prt[:9].t().reshape(n, 3, 3).double().numpy().astype(np.complex128)
        if prt.shape[0] == 18:
            p = p + 1j * prt[9:].t().reshape(n, 3, 3).double().numpy()
        out, status = oracle.polarized_intensity(p, *[k.double().numpy() for k in k0],
                                                 i0.double().numpy(), polarization)
        if status & S.STATUS_K_PARALLEL_X:
            raise ValueError("k-vector parallel to x-axis is not currently supported.")
        return torch.as_tensor(out, dtype=i0.dtype)

    def wavefront_opd(self, params, rays7, px, py, want_pupil=True):
        out, pupil = oracle.wavefront_opd(params, [t.double().numpy() for t in rays7],
                                          px.double().numpy(), py.double().numpy())
        return (torch.as_tensor(out, dtype=px.dtype),
                torch.as_tensor(pupil, dtype=px.dtype) if want_pupil else None)

    # ol_wavefront_fit / ol_wavefront_opd_fitted through oracle.wavefront_fit (the NumPy
    # restatement of strategy.py:287-620): same reference layout as the product's engine
    def can_wavefront_fit(self):
        return True

    def wavefront_fit(self, kind, params, rays8, px, py, *, trim_std=3.0, flavour="torch",
                      planar=False):
        ref = torch.zeros(15, dtype=torch.float64)
        bits = 0
        try:
            got = oracle.wavefront_fit(kind, params, [t.double().numpy() for t in rays8],
                                       px.double().numpy(), py.double().numpy(),
                                       trim_std=trim_std, flavour=flavour, planar=planar)
        except ValueError as exc:
            msg = str(exc)
            bits = 1 if "No valid ray samples" in msg else 2 if "at least 4" in msg else 4
            got = None
        if got is not None:
            ref[0:3] = torch.as_tensor(got["center"])
            ref[3] = 0.0 if planar else got["radius"]
            ref[9] = got["opd_ref"]
            if got["normal"] is not None:
                ref[10:13] = torch.as_tensor(got["normal"])
        ref[4] = params["n_image"]
        ref[5] = 1.0 / (params["wavelength_um"] * 1e-3)
        ref[6], ref[7], ref[8] = params.get("ux", 0.0), params.get("uy", 0.0), params["half_epd"]
        ref[14:].view(torch.int32)[0] = bits
        return ref

    @staticmethod
    def fit_result(reference):
        return float(reference[3]), int(reference[-1:].view(torch.int32)[0])

    @staticmethod
    def raise_for_fit_status(bits):
        from optiland_amd.engine import HipSystem
        HipSystem.raise_for_fit_status(bits)

    def wavefront_opd_fitted(self, reference, rays7, px, py, want_pupil=True):
        r = reference.tolist()
        params = dict(xc=r[0], yc=r[1], zc=r[2], R=r[3], n_image=r[4], opd_ref=r[9], ux=r[6],
                      uy=r[7], half_epd=r[8], wavelength_um=1.0 / r[5] * 1e3, nx=r[10], ny=r[11],
                      nz=r[12])
        return self.wavefront_opd(params, rays7, px, py, want_pupil)

    def trace_opd(self, params, px, py, wl_index, *, field, vig=(1.0, 1.0), want_pupil=True,
                  moments=None, check_status=True):
        """ol_trace_opd as the composition of the oracle's generate -> trace -> OPD."""
        n = int(px.numel())
        rays = self.generate_rays(float(field[0]), float(field[1]), px, py, float(vig[0]),
                                  float(vig[1])) + [torch.zeros(n, dtype=px.dtype)]
        self.trace(rays, wl_index, record=False)
        r7 = [rays[k] for k in (0, 1, 2, 3, 4, 5, 7)]
        opd, pupil = self.wavefront_opd(params, r7, px, py, want_pupil=True)
        inten = rays[6].clone()
        w, o, X, Y = inten.double(), opd.double(), pupil[0].double(), pupil[1].double()
        alive = w > 0
        got = torch.stack([w.sum(), (w * X).sum(), (w * Y).sum(), (w * X * X).sum(),
                           (w * X * Y).sum(), (w * Y * Y).sum(), (w * o).sum(),
                           (w * o * X).sum(), (w * o * Y).sum(), alive.double().sum(),
                           o[alive].sum(), (o[alive] ** 2).sum()])
        if moments is None:
            moments = torch.zeros(12, dtype=torch.float64)
        moments += got
        return opd, inten, (pupil if want_pupil else None), moments

    def pupil_fill(self, opd, intensity, cell, n_side, grid_size, pupil_xy=None, plane=None):
        o = opd.double()
        if pupil_xy is not None:
            o = o - (plane[0] + plane[1] * pupil_xy[0].double() + plane[2] * pupil_xy[1].double())
        val = torch.sqrt(intensity.double()) * torch.exp(-2j * np.pi * o)
        grid = torch.zeros((grid_size, grid_size), dtype=torch.complex128)
        pad = (grid_size - n_side) // 2
        r = torch.div(cell.long(), n_side, rounding_mode="floor")
        c = cell.long() - r * n_side
        grid[r + pad, c + pad] = val
        return grid

    def trace_spot(self, px, py, wl_index, *, field=None, hx=None, hy=None, vig=(1.0, 1.0),
                   vx=None, vy=None, center=(0.0, 0.0), hits=None, out=None, check_status=True,
                   flags=0):
        n = int(px.numel())
        dtype = px.dtype
        if field is not None:
            hx, hy = float(field[0]), float(field[1])
        if vx is None:
            vx, vy = float(vig[0]), float(vig[1])
        rays = self.generate_rays(hx, hy, px, py, vx, vy) + [torch.zeros(n, dtype=dtype)]
        self.trace(rays, wl_index, record=False)
        x, y, i = rays[0].double(), rays[1].double(), rays[6].double()
        if hits is not None:
            for dst, src in zip(hits, (rays[0], rays[1], rays[6])):
                dst.copy_(src)
        m = i > 0
        dx, dy = x[m] - center[0], y[m] - center[1]
        r2 = dx * dx + dy * dy
        r2 = r2[~torch.isnan(r2)]
        got = torch.tensor([float(m.sum()), dx.sum(), dy.sum(), (dx * dx).sum(), (dy * dy).sum(),
                            i[m].sum(), r2.max() if r2.numel() else 0.0], dtype=torch.float64)
        if out is None:
            return got
        out[:6] += got[:6]
        out[6] = torch.maximum(out[6], got[6])
        return out

    def spot_moments(self, x, y, intensity):
        m = intensity > 0
        xd, yd = x[m].double(), y[m].double()
        c = float(m.sum())
        return torch.tensor([c, xd.sum(), yd.sum(), (xd * xd).sum(), (yd * yd).sum(), c],
                            dtype=torch.float64)

    def irradiance(self, x, y, power, x_edges, y_edges, out=None):
        xn, yn, pn = (v.double().numpy() for v in (x, y, power))
        valid = pn > 0.0
        with np.errstate(invalid="ignore"):
            h, _, _ = np.histogram2d(xn[valid], yn[valid], bins=[x_edges.numpy(), y_edges.numpy()],
                                     weights=pn[valid])
        h = torch.as_tensor(h)
        if out is None:
            return h
        out += h
        return out

    def radial_energy(self, x, y, intensity, cx, cy, r_step, out=None):
        r = torch.sqrt((x.double() - cx) ** 2 + (y.double() - cy) ** 2).numpy()
        e = intensity.double().numpy()
        rs = r_step.numpy()
        cum = np.array([np.nansum(e[r <= v]) for v in rs])
        bins = torch.as_tensor(np.diff(cum, prepend=0.0))
        if out is None:
            return bins
        out += bins
        return out

    def spot_max_r2(self, x, y, intensity, cx, cy):
        m = intensity > 0
        r2 = (x[m].double() - cx) ** 2 + (y[m].double() - cy) ** 2
        return r2.max().reshape(1) if r2.numel() else torch.zeros(1, dtype=torch.float64)
