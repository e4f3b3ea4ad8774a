"""Wavefront OPD and scalar FFT PSF on device.

Mirrors the numerics of the reference's chief-ray wavefront strategy
(optiland/wavefront/strategy.py:142-243), `Wavefront` / `OPD`
(wavefront/wavefront.py:28-176, wavefront/opd.py:145-159) and the scalar FFT PSF
(psf/fft.py:42-262, psf/base.py:418-438):

    chief ray -> reference sphere (centre = chief-ray image point, radius to the
    paraxial exit pupil) -> per-ray OPD in waves (`ol_wavefront_opd`) ->
    pupil function A exp(-i 2 pi OPD) on a uniform grid -> zero-pad -> FFT -> |.|^2.

The trace runs in fp64 (an OPD good to lambda/1000 over a 200 mm path needs 1e-9
relative); the FFT is torch.fft on the ROCm device (rocFFT) -- a plain library
transform, as MFMA GEMMs would be hipBLASLt's job.
"""

from __future__ import annotations

import math

from . import _capi

import numpy as np
import torch

from .distribution import create_distribution


class WavefrontData:
    """wavefront/wavefront_data.py:10-38 (+ the device-side sums of the fused kernel)."""

    def __init__(self, pupil_x, pupil_y, pupil_z, opd, intensity, radius, moments=None):
        self.pupil_x, self.pupil_y, self.pupil_z = pupil_x, pupil_y, pupil_z
        self.opd, self.intensity, self.radius = opd, intensity, radius
        # `ol_trace_opd`'s 12 sums (tilt-fit moments; count / sum / sum of squares of the OPD
        # over rays with i > 0) of the OPD as first computed -- None on the un-fused path.
        # They describe `_moments_of` (that exact tensor) only: consumers check identity, so a
        # caller who replaces or detrends `opd` gets sums recomputed from the data it passes
        self.moments = moments
        self._moments_of = opd if moments is not None else None


class Wavefront:
    """OPD map for ONE field and wavelength (chief-ray strategy, spherical reference)."""

    def __init__(self, tracer, field, wavelength, num_rays: int = 12,
                 distribution="hexapolar", strategy: str = "chief_ray",
                 remove_tilt: bool = False, robust_trim_std: float = 3.0, afocal: bool = False,
                 fused: bool | None = None):
        # wavefront/strategy.py:606-616 (incl. the backward-compatible aliases)
        strategy = {"centroid_sphere": "centroid", "best_fit_sphere": "best_fit"}.get(strategy,
                                                                                       strategy)
        if strategy not in ("chief_ray", "centroid", "best_fit"):
            raise ValueError(f"Unknown wavefront strategy: {strategy}")
        self.strategy = strategy
        self.robust_trim_std = float(robust_trim_std)
        self.remove_tilt = bool(remove_tilt)
        # wavefront.py:74: afocal systems measure against a PLANE (reference_geometry.py:87-128)
        self.afocal = bool(afocal)
        if tracer.dtype != torch.float64:
            raise ValueError("wavefront analysis needs an fp64 tracer (OPD in waves)")
        rg = tracer.table.raygen
        if "pupil_z" not in rg or "n_image" not in rg:
            raise ValueError("this SystemTable carries no exit-pupil data")
        self.tracer = tracer
        # fused generate -> trace -> OPD kernel (`ol_trace_opd`): chief-ray strategy on an
        # unpolarised system; None = use it whenever it applies
        can_fuse = (strategy == "chief_ray" and tracer.table.polarization is None
                    and not tracer.table.uses_polarization
                    and hasattr(tracer.engine, "trace_opd"))
        if fused and not can_fuse:
            raise ValueError("fused OPD needs the chief-ray strategy on an unpolarised system")
        self.fused = can_fuse if fused is None else bool(fused)
        self.field = (float(field[0]), float(field[1]))
        self.wavelength = float(wavelength)
        self.num_rays = num_rays
        if isinstance(distribution, str):
            distribution = create_distribution(distribution)
            distribution.generate_points(num_rays)
        self.distribution = distribution
        self.data = self._compute()
        if self.remove_tilt:  # wavefront.py:173-174
            self.data.opd = self.fit_and_remove_tilt(self.data)
            # the device sums describe the OPD as the kernel produced it, not the detrended
            # map now stored: a later fit (or rms) must take its sums from the data it is
            # given, as the reference's always does
            self.data.moments = None

    @staticmethod
    def fit_and_remove_tilt(data, remove_piston: bool = False, ridge: float = 1e-12):
        """Weighted plane fit of the OPD over the pupil, subtracted (wavefront.py:103-148;
        the piston only with `remove_piston`).  Formulated as what it is on a device: the
        nine weighted moments
            S = sum w [1, x, y, xx, xy, yy],   T = sum w opd [1, x, y]
        -- straight out of the fused kernel's epilogue (`data.moments`), or one (9, n) x (n,)
        reduction on the un-fused path -- the symmetric 3x3 normal system (+ ridge on the
        diagonal, as the reference regularises) solved on the host in closed form, and one
        elementwise pass for the residual."""
        x, y, w, opd = data.pupil_x, data.pupil_y, data.intensity, data.opd
        mom = getattr(data, "moments", None)
        if mom is not None and getattr(data, "_moments_of", None) is opd:
            # the kernel's own sums -- valid only for the very OPD tensor they were taken of
            m = mom[:9].double().cpu().numpy()
        else:
            basis = torch.stack([torch.ones_like(x), x, y, x * x, x * y, y * y,
                                 opd, opd * x, opd * y])
            m = (basis @ w.to(basis.dtype)).double().cpu().numpy()   # one read-back
        A = np.array([[m[0], m[1], m[2]], [m[1], m[3], m[4]], [m[2], m[4], m[5]]]) \
            + ridge * np.eye(3)
        a0, bx, cy = np.linalg.solve(A, m[6:9])
        if not remove_piston:
            a0 = 0.0
        return opd - (float(a0) + float(bx) * x + float(cy) * y)

    # strategy.py:83-139: tilt of the launch plane for angle fields at infinity
    def _tilt_cosines(self):
        rg = self.tracer.table.raygen
        # only AngleField at infinity is corrected (strategy.py:113-117); other field
        # types keep their launch-plane tilt, as in the reference
        if not rg.get("object_infinite") or int(rg.get("field_kind", 0)) != 0:
            return 0.0, 0.0
        fx = self.field[0] * rg["max_field"]
        fy = self.field[1] * rg["max_field"]
        tx, ty = math.tan(math.radians(fx)), math.tan(math.radians(fy))
        uz = 1.0 / math.sqrt(1.0 + tx * tx + ty * ty)
        return tx * uz, ty * uz

    def chief_reference(self):
        """(params, R, fused_status): the `ol_wavefront_params` of the chief-ray reference
        sphere / plane for this field and wavelength (strategy.py:176-184, 228-243), from one
        traced chief ray."""
        t, rg = self.tracer, self.tracer.table.raygen
        hx, hy = self.field
        # The seven numbers of the traced chief ray come back in ONE device-to-host copy (its
        # status word is read with the fused launch's) and the reference geometry is worked
        # out on the host in double -- the same expressions as wavefront_device.h -- instead
        # of a one-ray kernel launch and four more read-backs
        fused_status = self.fused and hasattr(t, "defer_checks")
        if fused_status:
            t.defer_checks = True
        try:
            chief = t.trace_generic(hx, hy, 0.0, 0.0, self.wavelength)
        finally:
            if fused_status:
                t.defer_checks = False
        c = torch.stack([chief.x, chief.y, chief.z, chief.L, chief.M, chief.N, chief.opd]) \
            .reshape(7, -1)[:, 0].double().cpu().tolist()
        xc, yc, zc, Lc, Mc, Nc, opd_c = c
        ux, uy = self._tilt_cosines()
        params = dict(xc=xc, yc=yc, zc=zc, n_image=rg["n_image"], opd_ref=0.0, ux=ux,
                      uy=uy, half_epd=rg["EPD"] / 2.0, wavelength_um=self.wavelength)
        if self.afocal:  # strategy.py:260-284: plane through the chief-ray hit, normal to it
            R = math.inf
            params.update(R=0.0, nx=Lc, ny=Mc, nz=Nc)
            t_back = 0.0  # the chief ray starts ON its own plane
        else:
            R = math.sqrt(xc * xc + yc * yc + (zc - rg["pupil_z"]) ** 2)
            params.update(R=R)
            # back-propagation distance of the chief ray from the sphere's centre (its own
            # image point) to the sphere: the quadratic of reference_geometry.py:41-79
            # with r = centre -> c = -R^2, b = 0
            a_ = Lc * Lc + Mc * Mc + Nc * Nc
            sq = math.sqrt(max(4.0 * a_ * R * R, 0.0))
            t1, t2 = -sq / (2.0 * a_), sq / (2.0 * a_)
            t_back = t2 if t1 < 0.0 else t1
        # chief-ray OPD to the reference (pupil point (0, 0): no tilt term)
        params["opd_ref"] = opd_c - rg["n_image"] * t_back
        return params, R, fused_status

    def _compute(self) -> WavefrontData:
        if self.strategy != "chief_ray":
            return self._compute_fitted()
        t = self.tracer
        hx, hy = self.field
        # 1. chief ray alone -> reference sphere
        params, R, fused_status = self.chief_reference()
        # 2. the full pupil (strategy.py:190-205)
        if self.fused:  # one launch: pupil points -> OPD map + its reductions, no ray planes
            px, py = t._dev(self.distribution.x), t._dev(self.distribution.y)
            wl, _ = t._wavelength_index(self.wavelength)
            # (check_status=False keeps the chief-ray launches' status bits: one read-back
            # for the three launches)
            opd, intensity, pupil, mom = t.engine.trace_opd(
                params, px, py, wl, field=(hx, hy), vig=t._vig_scalar(hx, hy), want_pupil=True,
                check_status=not fused_status)
            if fused_status:
                t.check_status()
            return WavefrontData(pupil[0], pupil[1], pupil[2], opd, intensity, R, moments=mom)
        # the intensity the reference reads is the RECORDED image-plane row
        # (wavefront/strategy.py:198: surfaces.intensity[-1], i.e. before a polarised
        # update_intensity): record-all is forced for this trace whatever the caller set
        keep_record = t.record_all
        t.record_all = True
        try:
            rays = t.trace(hx, hy, self.wavelength, None, self.distribution)
            intensity = t.surfaces.intensity[-1].clone()
        finally:
            t.record_all = keep_record
        px = t._dev(self.distribution.x)
        py = t._dev(self.distribution.y)
        r7 = [v.contiguous() for v in (rays.x, rays.y, rays.z, rays.L, rays.M, rays.N, rays.opd)]
        opd, pupil = t.engine.wavefront_opd(params, r7, px, py, want_pupil=True)
        return WavefrontData(pupil[0], pupil[1], pupil[2], opd, intensity, R)


    def _compute_fitted(self) -> WavefrontData:
        """CentroidStrategy / BestFitStrategy (wavefront/strategy.py:287-620): the reference
        sphere / plane comes from the traced bundle itself -- centred on the intensity-weighted
        (3-sigma trimmed) centroid of the image points with the weighted mean wavefront
        distance as radius, or the least-squares sphere / plane through the wavefront points
        -- and the piston is the mean OPD of the rays with intensity > 0.  Three steps on the
        device: the trace, `ol_wavefront_fit` (a chain of reductions that leaves the reference
        in device memory) and `ol_wavefront_opd_fitted`; ONE read-back (radius + fit status).
        This stand-alone class follows the reference's NumPy backend where its two backends
        differ (np.std, plain mean)."""
        t, rg = self.tracer, self.tracer.table.raygen
        hx, hy = self.field
        rays = t.trace(hx, hy, self.wavelength, None, self.distribution)
        px, py = t._dev(self.distribution.x), t._dev(self.distribution.y)
        ux, uy = self._tilt_cosines()
        params = dict(n_image=rg["n_image"], wavelength_um=self.wavelength, ux=ux, uy=uy,
                      half_epd=rg["EPD"] / 2.0)
        r8 = [v.contiguous() for v in (rays.x, rays.y, rays.z, rays.L, rays.M, rays.N, rays.opd,
                                       rays.i)]
        px, py = px.contiguous(), py.contiguous()
        ref = t.engine.wavefront_fit(self.strategy, params, r8, px, py,
                                     trim_std=self.robust_trim_std, flavour="numpy",
                                     planar=self.afocal)
        opd, pupil = t.engine.wavefront_opd_fitted(ref, r8[:7], px, py, want_pupil=True)
        R, bits = t.engine.fit_result(ref)
        if bits & _capi.FIT_SINGULAR and self.strategy == "best_fit" and not self.afocal \
                and not bits & (_capi.FIT_NO_VALID | _capi.FIT_TOO_FEW):
            return self._best_fit_rank_deficient(r8, px, py, params)
        t.engine.raise_for_fit_status(bits)
        return WavefrontData(pupil[0], pupil[1], pupil[2], opd, r8[7].clone(),
                             math.inf if self.afocal else R)

    def _best_fit_rank_deficient(self, r8, px, py, params) -> WavefrontData:
        """The least-squares sphere through wavefront points that do not span space -- a
        collimated beam: the points of an afocal system lie in one plane -- the way the
        reference gets it (wavefront/strategy.py:556-582 on its NumPy backend):
        `np.linalg.lstsq` of the RAW system `[x y z 1] c = |p|^2`, i.e. the rank-3 minimum-norm
        solution of the SVD.  That sphere is an artefact of where the origin lies (4.7 mm of
        radius for a flat wavefront, hundreds of waves of "OPD"), but it is what the reference
        returns, so it is what this returns: the device fit reports the cloud as singular
        (`ol_wavefront_fit`: its centred normal equations have no fourth pivot), the points
        come back once, and the OPD map is taken on the device against the host's sphere
        (`ol_wavefront_opd`), piston = the mean over the rays with intensity > 0
        (strategy.py:331).  Rare by construction; `afocal=True` is the fit such a beam wants."""
        import numpy as np
        t = self.tracer
        x, y, z, L, M, N, opd, inten = (v.double().cpu().numpy() for v in r8)
        pxn, pyn = px.double().cpu().numpy(), py.double().cpu().numpy()
        ni, half = float(params["n_image"]), float(params["half_epd"])
        with np.errstate(all="ignore"):
            opd_t = opd + (params["ux"] * (pxn * half) + params["uy"] * (pyn * half))
            valid = (np.isfinite(x) & np.isfinite(y) & np.isfinite(z) & np.isfinite(L)
                     & np.isfinite(M) & np.isfinite(N) & np.isfinite(opd_t) & (inten != 0))
            pts = np.stack((x, y, z), axis=1)[valid] \
                - (opd_t[valid] / ni)[:, None] * np.stack((L, M, N), axis=1)[valid]
            A = np.concatenate([pts, np.ones((len(pts), 1))], axis=1)
            c = np.linalg.lstsq(A, (pts ** 2).sum(axis=1), rcond=None)[0]
            centre = c[:3] / 2
            R = float(np.sqrt(c[3] + (centre ** 2).sum()))
        sphere = dict(params, xc=float(centre[0]), yc=float(centre[1]), zc=float(centre[2]), R=R,
                      opd_ref=0.0)
        opd_w, pupil = t.engine.wavefront_opd(sphere, r8[:7], px, py, want_pupil=True)
        alive = r8[7] > 0
        if not bool(alive.any()):
            raise ValueError("No valid rays with non-zero intensity for OPD calculation.")
        opd_w = opd_w - opd_w[alive].mean()   # (piston - opd) / lambda, piston = mean(opd[i > 0])
        return WavefrontData(pupil[0], pupil[1], pupil[2], opd_w, r8[7].clone(), R)


class OPD(Wavefront):
    """wavefront/opd.py:72-93: OPD wavefront; `num_rays` = number of hexapolar rings for
    the default distribution (15), as in the reference (`num_rings` is kept as an alias)."""

    def __init__(self, tracer, field, wavelength, num_rays: int = 15,
                 distribution="hexapolar", strategy: str = "chief_ray",
                 remove_tilt: bool = False, num_rings: int | None = None, **kwargs):
        super().__init__(tracer, field, wavelength,
                         num_rays=num_rays if num_rings is None else num_rings,
                         distribution=distribution, strategy=strategy, remove_tilt=remove_tilt,
                         **kwargs)

    def rms(self) -> float:
        """opd.py:145-159."""
        d = self.data
        if d.moments is not None and d._moments_of is d.opd:
            cnt, _s1, s2 = d.moments[9:12].tolist()  # epilogue of the fused kernel, one read-back
            if cnt == 0:
                raise ValueError("No valid rays with non-zero intensity for RMS calculation.")
            return math.sqrt(s2 / cnt)
        mask = d.intensity > 0
        if not bool(mask.any()):
            raise ValueError("No valid rays with non-zero intensity for RMS calculation.")
        o = d.opd[mask]
        return float(torch.sqrt(torch.mean(o * o)))


def calculate_grid_size(num_rays: int) -> tuple[int, int]:
    """psf/fft.py:20-39."""
    eff = int(np.floor(32 * 2 ** ((np.log2(num_rays) - 5) / 2)))
    return eff, num_rays * 2


class FFTPSF:
    """Scalar FFT PSF (psf/fft.py:42-262) for one field and wavelength."""

    def __init__(self, tracer, field, wavelength, num_rays: int = 128, grid_size=None,
                 strategy: str = "chief_ray", remove_tilt: bool = False, **kwargs):
        if grid_size is None:
            if num_rays < 32:
                raise ValueError("num_rays must be at least 32 if grid_size is not specified.")
            num_rays, grid_size = calculate_grid_size(num_rays)
        elif grid_size < num_rays:
            raise ValueError(f"Grid size ({grid_size}) must be greater than or equal to the "
                             f"number of rays ({num_rays}).")
        self.num_rays, self.grid_size = num_rays, grid_size
        self.wavefront = Wavefront(tracer, field, wavelength, num_rays, "uniform",
                                   strategy=strategy, remove_tilt=remove_tilt, **kwargs)
        self.pupil = self._generate_pupil()
        self.psf = self._compute_psf()

    def _generate_pupil(self) -> torch.Tensor:
        """psf/fft.py:101-137: A exp(-i 2 pi OPD) on the num_rays^2 grid, 0 off-disc --
        written straight into the zero-padded FFT grid by `ol_pupil_fill` (self._padded);
        the n x n block is returned as a view."""
        d = self.wavefront.data
        n, gsz = self.num_rays, self.grid_size
        # the disc mask comes from the SAME arithmetic that placed the traced samples
        # (distribution._uniform: np.linspace + np.meshgrid, row-major ravel) -- a
        # torch.linspace grid differs by an ulp on boundary points such as (0.6, 0.8)
        # and would select a different number of cells for many odd n
        g = np.linspace(-1.0, 1.0, n)
        xg, yg = np.meshgrid(g, g)
        cells = np.flatnonzero((xg**2 + yg**2 <= 1).reshape(-1)).astype(np.int32)
        before = (gsz - n) // 2
        eng = self.wavefront.tracer.engine
        if self.wavefront.fused and hasattr(eng, "pupil_fill"):
            cell = torch.from_numpy(cells).to(d.opd.device)
            self._padded = eng.pupil_fill(d.opd, d.intensity, cell, n, gsz)
        else:  # un-fused path (polarised systems, other strategies, A/B): plain tensor ops
            P = torch.zeros(n * n, dtype=torch.complex128, device=d.opd.device)
            P[torch.from_numpy(cells.astype(np.int64)).to(d.opd.device)] = \
                torch.sqrt(d.intensity) * torch.exp(-2j * math.pi * d.opd)
            after = before + (gsz - n) % 2
            self._padded = torch.nn.functional.pad(P.reshape(n, n), (before, after, before, after))
        return self._padded[before:before + n, before:before + n]

    def _compute_psf(self) -> torch.Tensor:
        """psf/fft.py:139-200: (padded pupil) FFT (rocFFT), |.|^2, Strehl normalisation."""
        amp = torch.fft.fftshift(torch.fft.fft2(self._padded))
        norm = float((self.pupil.abs() > 0).sum()) ** 2
        return (amp * amp.conj()).real / norm * 100

    def strehl_ratio(self) -> float:
        """psf/base.py:418-438."""
        c = self.psf.shape[0] // 2
        return float(self.psf[c, self.psf.shape[1] // 2]) / 100
