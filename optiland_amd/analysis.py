"""Spot-diagram statistics on device (the step AFTER the path).

Mirrors the numeric part of the reference's `SpotDiagram`
(optiland/analysis/spot_diagram/core.py:329-372, 420-481 and reference.py:60-105):
for every (field, wavelength) trace `num_rings` hexapolar rings, keep rays with
intensity > 0, centre on the chief ray of the reference wavelength (default) or on
the centroid, and report RMS and geometric (maximum) radius.  For unpolarised
systems each spot is ONE fused kernel (`ol_trace_spot`: generate -> trace -> reduce)
that returns seven doubles -- no ray or hit plane is ever written; polarised systems
trace into planes and reduce them with `ol_spot_moments` / `ol_spot_max_r2`.
"""

from __future__ import annotations

import numpy as np


class SpotDiagram:
    def __init__(self, tracer, fields="all", wavelengths="all", num_rings: int = 6,
                 distribution: str = "hexapolar", coordinates: str = "local",
                 reference: str = "chief_ray", primary_index: int | None = None):
        if coordinates not in ("global", "local"):  # core.py:105-106
            raise ValueError("Coordinates must be 'global' or 'local'.")
        if reference not in ("chief_ray", "centroid"):
            raise ValueError(f"Invalid reference '{reference}'. Must be 'chief_ray' or 'centroid'.")
        self.tracer = tracer
        self.coordinates = coordinates
        table = tracer.table
        s = table.surfaces[-1]
        if s["flags"] & 1 and coordinates == "local":
            raise NotImplementedError("local spot coordinates on a tilted image surface")
        # radii are frame-independent for an untilted image surface; only the centroid
        # moves by the surface origin between the two coordinate systems
        self._origin = (np.asarray(s["origin"], dtype=np.float64) if coordinates == "local"
                        else np.zeros(3))
        mf = table.raygen.get("max_field", 0.0) or 1.0
        if fields == "all":
            fields = [(f[0] / mf, f[1] / mf) for f in table.fields]
        self.fields = [tuple(map(float, f)) for f in fields]
        self.wavelengths = (list(map(float, table.wavelengths)) if wavelengths == "all"
                            else [float(w) for w in wavelengths])
        self.num_rings, self.distribution, self.reference = num_rings, distribution, reference
        if primary_index is None:  # core.py:114-119: the optic's primary wavelength, else 0
            primary_index = table.reference_wavelength_index(self.wavelengths)
        self.ref_index = primary_index
        self._moments = None
        self._centers = None
        self._run()

    # Internal form: per [field][wavelength] seven doubles ABOUT the field's centre
    # (cx, cy): {count, sum dx, sum dy, sum dx^2, sum dy^2, sum i, max r^2}.
    def _run(self):
        t = self.tracer
        fused = t.table.polarization is None and not t.table.uses_polarization
        self.fused = fused
        if fused:
            self._run_fused()
        else:
            self._run_planes()

    def _chief_centers(self):
        """Chief-ray image points of ALL fields at the reference wavelength: one
        trace_generic launch (Px = Py = 0 per field), one read-back."""
        import torch
        t = self.tracer
        hx = torch.tensor([f[0] for f in self.fields], dtype=t.dtype, device=t.device)
        hy = torch.tensor([f[1] for f in self.fields], dtype=t.dtype, device=t.device)
        z = torch.zeros_like(hx)
        r = t.trace_generic(hx, hy, z, z, self.wavelengths[self.ref_index])
        xy = torch.stack([r.x, r.y]).double().cpu().numpy()
        return [(float(xy[0, i]), float(xy[1, i])) for i in range(len(self.fields))]

    def _run_fused(self):
        """One `ol_trace_spot` launch per (field, wavelength): generate -> trace ->
        reduce in a single kernel; no ray or hit planes are materialised.  All launches
        are queued back to back; the 7-double results and the status word are read
        back once at the end."""
        import torch
        t = self.tracer
        if self.reference == "chief_ray":
            centers = self._chief_centers()
            t.reset_status()
        else:
            t.reset_status()  # centroid of the reference wavelength: one extra reduction pass per field
            first = [t.trace_spot(hx, hy, self.wavelengths[self.ref_index], self.num_rings,
                                  self.distribution, check_status=False)[0]
                     for hx, hy in self.fields]
            m0 = torch.stack(first).cpu().numpy()
            centers = [(m[1] / m[0], m[2] / m[0]) for m in m0]
        dev = [t.trace_spot(hx, hy, w, self.num_rings, self.distribution, center=c,
                            check_status=False)[0]
               for (hx, hy), c in zip(self.fields, centers) for w in self.wavelengths]
        mom = torch.stack(dev).cpu().numpy().reshape(len(self.fields), len(self.wavelengths), 7)
        t.check_status()
        self._moments = [[mom[fi, wi] for wi in range(len(self.wavelengths))]
                         for fi in range(len(self.fields))]
        self._centers = centers

    def _chief_center(self, hx, hy):
        r = self.tracer.trace_generic(hx, hy, 0.0, 0.0, self.wavelengths[self.ref_index])
        return float(r.x[0]), float(r.y[0])

    def _run_planes(self):
        """Polarised systems: trace() (with its update_intensity epilogue) into planes,
        then the two reduction kernels."""
        t, eng = self.tracer, self.tracer.engine
        old = t.record_all
        t.record_all = True
        try:
            raw = [[None] * len(self.wavelengths) for _ in self.fields]
            hits = [[None] * len(self.wavelengths) for _ in self.fields]
            for fi, (hx, hy) in enumerate(self.fields):
                for wi, w in enumerate(self.wavelengths):
                    t.trace(hx, hy, w, self.num_rings, self.distribution)
                    # core.py:455-459 reads the RECORDED image-plane state
                    x, y, inten = (v[-1].contiguous().clone() for v in
                                   (t.surfaces.x, t.surfaces.y, t.surfaces.intensity))
                    raw[fi][wi] = eng.spot_moments(x, y, inten).cpu().numpy()
                    hits[fi][wi] = (x, y, inten)
            centers, mom = [], []
            for fi, (hx, hy) in enumerate(self.fields):
                if self.reference == "chief_ray":
                    cx, cy = self._chief_center(hx, hy)
                else:
                    m = raw[fi][self.ref_index]
                    cx, cy = m[1] / m[0], m[2] / m[0]
                centers.append((cx, cy))
                row = []
                for wi in range(len(self.wavelengths)):
                    m = raw[fi][wi]
                    n = m[0]
                    r2 = float(eng.spot_max_r2(*hits[fi][wi], cx, cy)[0])
                    row.append(np.array([n, m[1] - n * cx, m[2] - n * cy,
                                         m[3] - 2 * cx * m[1] + n * cx * cx,
                                         m[4] - 2 * cy * m[2] + n * cy * cy, n, r2]))
                mom.append(row)
        finally:
            t.record_all = old
        self._moments, self._centers = mom, centers

    def centroid(self):
        """(x, y) centroid per field at the reference wavelength, image-local."""
        out = []
        for fi, (cx, cy) in enumerate(self._centers):
            m = self._moments[fi][self.ref_index]
            out.append((cx + m[1] / m[0] - self._origin[0], cy + m[2] / m[0] - self._origin[1]))
        return out

    def rms_spot_radius(self):
        """sqrt(mean((x-cx)^2 + (y-cy)^2)) per [field][wavelength]."""
        return [[float(max((m[3] + m[4]) / m[0], 0.0)) ** 0.5 for m in row]
                for row in self._moments]

    def geometric_spot_radius(self):
        """max sqrt((x-cx)^2 + (y-cy)^2) per [field][wavelength]."""
        return [[float(m[6]) ** 0.5 for m in row] for row in self._moments]


class EncircledEnergy:
    """Encircled-energy curves on device (analysis/encircled_energy.py:20-160).

    Mirrors the reference's numerics: per field `num_rays` pupil points of `distribution`
    (default 100 000 random points) at one wavelength (a number / 'primary') or at every
    wavelength ('all'), image-plane hits WITHOUT the
    intensity mask (its `_generate_field_data`, :163-185), centred on the chief ray of
    that wavelength (the inherited `SpotDiagram` reference), radius steps
    `linspace(0, 1.2 * max geometric radius over all fields, num_points)` and
    `ee(r) = nansum(energy[radii <= r])` (:147-160).  Each field is one fused
    `ol_trace_spot` launch that also writes the three hit planes, plus one
    `ol_radial_energy` histogram pass; only the `num_points` doubles come back.
    """

    def __init__(self, tracer, fields="all", wavelength="primary", num_rays: int = 100_000,
                 distribution: str = "random", num_points: int = 256):
        import torch
        table = tracer.table
        if table.polarization is not None or table.uses_polarization:
            raise NotImplementedError("encircled energy of polarised systems")
        # encircled_energy.py:53-64: a number, 'primary' or 'all'
        if isinstance(wavelength, (int, float)):
            wls = [float(wavelength)]
        elif wavelength == "primary":
            wls = [float(table.primary_wavelength) if table.primary_wavelength is not None
                   else float(table.wavelengths[len(table.wavelengths) // 2])]
        elif wavelength == "all":
            wls = [float(w) for w in table.wavelengths]
        else:
            raise TypeError(f"Unsupported wavelength: {wavelength}. "
                            "Expected 'primary', 'all', or a number.")
        mf = table.raygen.get("max_field", 0.0) or 1.0
        if fields == "all":
            fields = [(f[0] / mf, f[1] / mf) for f in table.fields]
        self.fields = [tuple(map(float, f)) for f in fields]
        self.wavelengths, self.num_points = wls, int(num_points)
        self.ref_index = table.reference_wavelength_index(wls)
        self.wavelength = wls[0]  # the curve `view()` titles / `centroid()` reads
        t, eng = tracer, tracer.engine
        hx = torch.tensor([f[0] for f in self.fields], dtype=t.dtype, device=t.device)
        hy = torch.tensor([f[1] for f in self.fields], dtype=t.dtype, device=t.device)
        z = torch.zeros_like(hx)
        chief = t.trace_generic(hx, hy, z, z, wls[self.ref_index])
        cxy = torch.stack([chief.x, chief.y]).double().cpu().numpy()
        self._centers = [(float(cxy[0, i]), float(cxy[1, i])) for i in range(len(self.fields))]
        hits, rmax2 = [], []
        for (fx, fy), c in zip(self.fields, self._centers):
            row = []
            for w in wls:
                _, h = t.trace_spot(fx, fy, w, num_rays, distribution, center=c, hits=True,
                                    check_status=False)
                row.append(h)
                dx, dy = h[0].double() - c[0], h[1].double() - c[1]
                rmax2.append(torch.max(dx * dx + dy * dy))  # NaN-propagating, like be.max
            hits.append(row)
        t.check_status()
        self._hits_all = hits
        self._hits = [row[0] for row in hits]
        axis_lim = float(torch.sqrt(torch.stack(rmax2).max()))
        self.r_step = np.linspace(0.0, axis_lim * 1.2, self.num_points)
        r_dev = torch.as_tensor(self.r_step, dtype=torch.float64, device=t.device)
        curves = [torch.stack([torch.cumsum(eng.radial_energy(h[0], h[1], h[2], c[0], c[1], r_dev), 0)
                               for h in row])
                  for row, c in zip(hits, self._centers)]
        # (n_fields, n_wavelengths, num_points): every curve `view()` draws; `ee` = the first
        # wavelength's, the only one when `wavelength` is a number or 'primary'
        self.ee_all = torch.stack(curves).cpu().numpy()
        self.ee = self.ee_all[:, 0, :]

    def centroid(self):
        """encircled_energy.py:117-131: plain mean of the hit coordinates per field."""
        return [(float(h[0].double().mean()), float(h[1].double().mean())) for h in self._hits]


def _aperture_extent(row):
    """`aperture.extent` of the reference's leaf apertures (physical_apertures/
    radial.py:47-54, offset_radial.py:34-46, rectangular.py:33-40, elliptical.py:33-40)
    from a packed surface row, or None."""
    from . import system as S
    k, a = int(row["aperture_kind"]), [float(v) for v in row["aperture"]]
    if k == S.AP_RECTANGULAR:
        return a[0], a[1], a[2], a[3]
    if k == S.AP_RADIAL:
        return -a[1], a[1], -a[1], a[1]
    if k == S.AP_OFFSET_RADIAL:
        return a[2] - a[1], a[2] + a[1], a[3] - a[1], a[3] + a[1]
    if k == S.AP_ELLIPTICAL:
        return -a[0], a[0], -a[1], a[1]
    return None


class IncoherentIrradiance:
    """Detector irradiance maps on device (analysis/irradiance.py:79-353, the
    non-differentiable path), same arguments as the reference: for every field and
    wavelength trace `num_rays` pupil points of `distribution`, bin the detector-plane
    hits with `numpy.histogram2d` semantics weighted by the ray power (rays with power > 0
    only), divide by the pixel area.  `data[field][wavelength] = (irradiance, x_edges,
    y_edges)` as in the reference, `irradiance` a device tensor of shape (npix_x, npix_y).

    The detector must be the image surface and carry a physical aperture whose `extent`
    gives the pixel grid (irradiance.py:136-145); `extent=(x_min, x_max, y_min, y_max)`
    overrides it (boolean / polygon apertures have no packed extent).  `px_size=(dx, dy)`
    derives the resolution from the pixel size like the reference (:302-313);
    `user_initial_rays` (a `rays.RealRays` of one wavelength) is traced instead of a field's
    pupil sampling (:276-280).  Not taken over: `source` / `skip_trace` and the autograd
    branch.
    The hits stay on the GPU: one trace + one `ol_irradiance` histogram pass per map;
    sharded runs add their `power_map`s (an all-reduce of H x W bins instead of an
    all-gather of hits)."""

    def __init__(self, tracer, num_rays: int = 5, res=(128, 128), px_size=None,
                 detector_surface: int = -1, *, fields="all", wavelengths="all",
                 distribution: str = "random", user_initial_rays=None, extent=None):
        from .rays import RealRays
        table = tracer.table
        if user_initial_rays is not None and not isinstance(user_initial_rays, RealRays):
            raise TypeError("user_initial_rays must be a RealRays object.")  # irradiance.py:121
        self.user_initial_rays = user_initial_rays
        n_s = table.num_surfaces
        if int(detector_surface) not in (-1, n_s - 1):
            raise NotImplementedError("the detector must be the image surface")
        s = table.surfaces[-1]
        if s["flags"] & 1:
            raise NotImplementedError("irradiance on a tilted detector surface")
        if extent is None:
            from . import system as S
            if int(s["aperture_kind"]) == S.AP_NONE:
                raise ValueError("Detector surface has no physical aperture - set one "
                                 "(e.g. RectangularAperture) so that the irradiance "
                                 "grid can be defined.")
            extent = _aperture_extent(s)
            if extent is None:
                raise NotImplementedError("pass extent= for a boolean / polygon detector aperture")
        self.tracer = tracer
        self.num_rays, self.distribution = num_rays, distribution
        self.npix_x, self.npix_y = int(res[0]), int(res[1])
        self.px_size = None if px_size is None else (float(px_size[0]), float(px_size[1]))
        self.detector_surface = int(detector_surface)
        mf = table.raygen.get("max_field", 0.0) or 1.0
        if fields == "all":
            fields = [(f[0] / mf, f[1] / mf) for f in table.fields]
        self.fields = [tuple(map(float, f)) for f in fields]
        self.wavelengths = (list(map(float, table.wavelengths)) if wavelengths == "all"
                            else [float(w) for w in wavelengths])
        x_min, x_max, y_min, y_max = (float(v) for v in extent)
        if self.px_size is None:
            self.x_edges = np.linspace(x_min, x_max, self.npix_x + 1, dtype=float)
            self.y_edges = np.linspace(y_min, y_max, self.npix_y + 1, dtype=float)
            self.pixel_area = ((self.x_edges[1] - self.x_edges[0])
                               * (self.y_edges[1] - self.y_edges[0]))
        else:
            dx, dy = self.px_size
            self.x_edges = np.arange(x_min, x_max + 0.5 * dx, dx, dtype=float)
            self.y_edges = np.arange(y_min, y_max + 0.5 * dy, dy, dtype=float)
            self.pixel_area = dx * dy
            self.npix_x, self.npix_y = len(self.x_edges) - 1, len(self.y_edges) - 1
        self.power_maps = [[self._power_map(f, w) for w in self.wavelengths] for f in self.fields]
        self.data = [[(pm / self.pixel_area, self.x_edges, self.y_edges) for pm in row]
                     for row in self.power_maps]
        # first field / wavelength, for the single-map case
        self.power_map = self.power_maps[0][0]
        self.irradiance = self.data[0][0][0]

    def _power_map(self, field, wavelength):
        import torch
        tracer = self.tracer
        s = tracer.table.surfaces[-1]
        ox, oy = float(s["origin"][0]), float(s["origin"][1])
        if self.user_initial_rays is not None:
            # irradiance.py:276-280: the caller's bundle (copied: traced in place), straight
            # through the surface loop -- field and pupil sampling play no part
            u = self.user_initial_rays
            planes = [p.detach().to(device=tracer.device, dtype=tracer.dtype).reshape(-1).clone()
                      for p in (u.x, u.y, u.z, u.L, u.M, u.N, u.i)]
            planes.append(torch.zeros_like(planes[0]))
            wi = tracer.table.wavelength_index(float(u.w.reshape(-1)[0]))
            tracer.engine.trace(planes, wi, record=False)
            rays = type("Hits", (), dict(x=planes[0], y=planes[1], i=planes[6]))
        else:
            old = tracer.record_all
            tracer.record_all = False
            try:
                rays = tracer.trace(field[0], field[1], wavelength, self.num_rays,
                                    self.distribution)
            finally:
                tracer.record_all = old
        dev = tracer.device
        # detector-local = global - vertex (untilted): shift the EDGES instead of the hits
        xe = torch.as_tensor(self.x_edges + ox, dtype=torch.float64, device=dev)
        ye = torch.as_tensor(self.y_edges + oy, dtype=torch.float64, device=dev)
        return tracer.engine.irradiance(rays.x.contiguous(), rays.y.contiguous(),
                                        rays.i.contiguous(), xe, ye)

    def peak_irradiance(self):
        """irradiance.py: maximum of every map, per [field][wavelength]."""
        return [[float(irr.max()) for irr, _, _ in row] for row in self.data]
