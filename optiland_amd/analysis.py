"""Spot-diagram statistics on device (the step AFTER the path).

Mirrors the numeric part of the reference's `SpotDiagram`
(optiland/analysis/spot_diagram/core.py:329-372, 420-481 and reference.py:60-105):
for every (field, wavelength) trace `num_rings` hexapolar rings, keep rays with
intensity > 0, centre on the chief ray of the reference wavelength (default) or on
the centroid, and report RMS and geometric (maximum) radius.  The image-plane hits
never leave the GPU: each spot is reduced by `ol_spot_moments` / `ol_spot_max_r2`
to seven doubles.
"""

from __future__ import annotations

import numpy as np


class SpotDiagram:
    def __init__(self, tracer, fields="all", wavelengths="all", num_rings: int = 6,
                 distribution: str = "hexapolar", reference: str = "chief_ray",
                 primary_index: int | None = None):
        if reference not in ("chief_ray", "centroid"):
            raise ValueError(f"Invalid reference '{reference}'. Must be 'chief_ray' or 'centroid'.")
        self.tracer = tracer
        table = tracer.table
        s = table.surfaces[-1]
        if s["flags"] & 1:
            raise NotImplementedError("spot statistics on a tilted image surface")
        self._origin = np.asarray(s["origin"], dtype=np.float64)
        mf = table.raygen.get("max_field", 0.0) or 1.0
        if fields == "all":
            fields = [(f[0] / mf, f[1] / mf) for f in table.fields]
        self.fields = [tuple(map(float, f)) for f in fields]
        self.wavelengths = (list(map(float, table.wavelengths)) if wavelengths == "all"
                            else [float(w) for w in wavelengths])
        self.num_rings, self.distribution, self.reference = num_rings, distribution, reference
        if primary_index is None:
            primary_index = len(self.wavelengths) // 2
        self.ref_index = primary_index
        self._moments = None
        self._centers = None
        self._geo = None
        self._run()

    # image-plane local coordinates: global minus the image vertex (untilted)
    def _run(self):
        t, eng = self.tracer, self.tracer.engine
        old = t.record_all
        t.record_all = True
        try:
            mom = [[None] * len(self.wavelengths) for _ in self.fields]
            hits = [[None] * len(self.wavelengths) for _ in self.fields]
            for fi, (hx, hy) in enumerate(self.fields):
                for wi, w in enumerate(self.wavelengths):
                    t.trace(hx, hy, w, self.num_rings, self.distribution)
                    x, y, inten = (t.surfaces.x[-1], t.surfaces.y[-1], t.surfaces.intensity[-1])
                    x, y, inten = x.contiguous().clone(), y.contiguous().clone(), inten.contiguous().clone()
                    mom[fi][wi] = eng.spot_moments(x, y, inten).cpu().numpy()
                    hits[fi][wi] = (x, y, inten)
            centers = []
            for fi, (hx, hy) in enumerate(self.fields):
                if self.reference == "chief_ray":
                    r = t.trace_generic(hx, hy, 0.0, 0.0, self.wavelengths[self.ref_index])
                    centers.append((float(r.x[0]), float(r.y[0])))
                else:
                    m = mom[fi][self.ref_index]
                    centers.append((m[1] / m[0], m[2] / m[0]))
            geo = [[float(eng.spot_max_r2(*hits[fi][wi], *centers[fi])[0]) ** 0.5
                    for wi in range(len(self.wavelengths))] for fi in range(len(self.fields))]
        finally:
            t.record_all = old
        self._moments, self._centers, self._geo = mom, centers, geo

    def centroid(self):
        """(x, y) centroid per field at the reference wavelength, image-local."""
        out = []
        for fi in range(len(self.fields)):
            m = self._moments[fi][self.ref_index]
            out.append((m[1] / m[0] - self._origin[0], m[2] / m[0] - self._origin[1]))
        return out

    def rms_spot_radius(self):
        """sqrt(mean((x-cx)^2 + (y-cy)^2)) per [field][wavelength]."""
        out = []
        for fi, (cx, cy) in enumerate(self._centers):
            row = []
            for m in self._moments[fi]:
                n = m[0]
                v = (m[3] - 2 * cx * m[1] + n * cx * cx + m[4] - 2 * cy * m[2] + n * cy * cy) / n
                row.append(float(max(v, 0.0)) ** 0.5)
            out.append(row)
        return out

    def geometric_spot_radius(self):
        """max sqrt((x-cx)^2 + (y-cy)^2) per [field][wavelength]."""
        return [list(r) for r in self._geo]
