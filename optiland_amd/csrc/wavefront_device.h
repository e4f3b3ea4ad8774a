// wavefront_device.h -- OPD of ONE image-plane ray against the reference sphere / plane.
// ONE definition shared by the stand-alone ol_wavefront_opd kernel (aux_kernels.hip) and
// the fused generate -> trace -> OPD kernel (trace_kernel.hip), so both produce the same
// numbers.  Reference: wavefront/strategy.py:163-215 (ChiefRayStrategy),
// wavefront/reference_geometry.py:41-79 (sphere) / 87-128 (plane), strategy.py:83-139
// (_correct_tilt).
#pragma once
#include <hip/hip_runtime.h>

#include "device_table.h"
#include "trace_launch.h"
#include "surface_math.h"  // Math<T>

// OL_WAVEFRONT_FAST (default on): the reference-sphere intersection takes its square root
// and its quotients from Math<T> (fp64: hardware seeds + two refinement steps, ~1 ulp; ONE
// reciprocal of 2a serves both roots) instead of the IEEE library sequences -- in fp64 an
// IEEE quotient is ~30 vector instructions and a square root ~30, and the fused OPD kernel
// (every instance fp64) is bound by vector issue.  A/B knob, tools/build_variants.py.
#ifndef OL_WAVEFRONT_FAST
#define OL_WAVEFRONT_FAST 1
#endif

namespace ol {

template <typename T>
OL_DEV WavefrontConsts<T> load_consts(cptr<WavefrontConsts<T>> p) {
  WavefrontConsts<T> w;
  w.xc = p->xc; w.yc = p->yc; w.zc = p->zc; w.R = p->R; w.ni = p->ni; w.inv_w = p->inv_w;
  w.ux = p->ux; w.uy = p->uy; w.half_epd = p->half_epd; w.opd_ref = p->opd_ref;
  w.nx = p->nx; w.ny = p->ny; w.nz = p->nz; w.planar = p->planar;
  w.last_t = p->last_t; w.last_absorb = p->last_absorb;
  return w;
}

// What ends Optic.trace / trace_generic (real_ray_tracer.py:104-110, 145-149;
// propagation/homogeneous.py:30-57): the rays go on by the last surface's thickness through
// its post-medium -- position and, in an absorbing medium, intensity; the optical path is NOT
// extended.  0 in every sample lens (the last surface is the image plane); the reference's own
// test_finite_conjugate_angle_field_opd ends 95 mm behind its last surface.
// INTENSITY: the returned rays' intensity is attenuated too (what trace_generic hands back: the
// chief ray); the intensity a wavefront REPORTS is `surfaces.intensity[-1]` (strategy.py:188),
// the recorded last row, which the reference's write-back of `rays.i` never reaches (SURVEY.md
// Appendix D: it assigns into a temporary stack) -- so the OPD kernels leave it alone.
template <typename T, bool INTENSITY>
OL_DEV void final_propagate(const WavefrontConsts<T>& w, Ray<T>& g) {
  if (w.last_t != T(0)) {   // launch-uniform
    g.x = g.x + w.last_t * g.L;
    g.y = g.y + w.last_t * g.M;
    g.z = g.z + w.last_t * g.N;
    if constexpr (INTENSITY) {
      if (w.last_absorb > T(0)) g.i = g.i * Math<T>::exp(-w.last_absorb * w.last_t);
    }
  }
}

// (xr, yr, zr), (Ld, Md, Nd), opd_in: the ray at the image surface (global frame);
// (px, py): its normalised pupil coordinates.  Returns the ray's optical path to the reference
// surface in mm; pu = the point where the back-propagated ray meets that surface.
// TILT_FIRST: the launch-plane tilt is added to the ray's path BEFORE the image-to-reference
// path is subtracted (CentroidStrategy / BestFitStrategy, strategy.py:318-325) instead of
// after it (ChiefRayStrategy, :190-195) -- the same number but for the last bit.
template <typename T, bool TILT_FIRST = false>
OL_DEV T wavefront_opd_mm(const WavefrontConsts<T>& w, T xr, T yr, T zr, T Ld, T Md, T Nd,
                          T opd_in, T px, T py, T (&pu)[3]) {
  const T L = -Ld, M = -Md, N = -Nd;  // trace backwards from the image
  T t;
#if OL_WAVEFRONT_FAST
  using m = Math<T>;
  if (w.planar) {  // reference_geometry.py:104-124
    const T num = (xr - w.xc) * w.nx + (yr - w.yc) * w.ny + (zr - w.zc) * w.nz;
    T den = L * w.nx + M * w.ny + N * w.nz;
    den = m::abs(den) < T(1e-12) ? T(1e-12) : den;
    t = -m::div(num, den);
  } else {
    const T a = L * L + M * M + N * N;
    const T b = T(2) * (L * (xr - w.xc) + M * (yr - w.yc) + N * (zr - w.zc));
    const T c = xr * xr + yr * yr + zr * zr - T(2) * (xr * w.xc + yr * w.yc + zr * w.zc) +
                w.xc * w.xc + w.yc * w.yc + w.zc * w.zc - w.R * w.R;
    T d = b * b - T(4) * a * c;
    d = d < T(0) ? T(0) : d;
    const T sq = m::sqrt(d);
    const T i2a = m::rcp(T(2) * a);
    const T t1 = (-b - sq) * i2a, t2 = (-b + sq) * i2a;
    t = t1 < T(0) ? t2 : t1;
  }
  const T opd_img = w.ni * t;
  const T tilt = w.ux * (px * w.half_epd) + w.uy * (py * w.half_epd);
  const T opd = TILT_FIRST ? (opd_in + tilt) - opd_img : opd_in - opd_img + tilt;
  const T tt = m::div(opd_img, w.ni);
#else
  if (w.planar) {  // reference_geometry.py:104-124
    const T num = (xr - w.xc) * w.nx + (yr - w.yc) * w.ny + (zr - w.zc) * w.nz;
    T den = L * w.nx + M * w.ny + N * w.nz;
    den = fabs(den) < T(1e-12) ? T(1e-12) : den;
    t = -num / den;
  } else {
    const T a = L * L + M * M + N * N;
    const T b = T(2) * (L * (xr - w.xc) + M * (yr - w.yc) + N * (zr - w.zc));
    const T c = xr * xr + yr * yr + zr * zr - T(2) * (xr * w.xc + yr * w.yc + zr * w.zc) +
                w.xc * w.xc + w.yc * w.yc + w.zc * w.zc - w.R * w.R;
    T d = b * b - T(4) * a * c;
    d = d < T(0) ? T(0) : d;
    const T sq = sqrt(d);
    const T t1 = (-b - sq) / (T(2) * a), t2 = (-b + sq) / (T(2) * a);
    t = t1 < T(0) ? t2 : t1;
  }
  const T opd_img = w.ni * t;
  const T tilt = w.ux * (px * w.half_epd) + w.uy * (py * w.half_epd);
  const T opd = TILT_FIRST ? (opd_in + tilt) - opd_img : opd_in - opd_img + tilt;
  const T tt = opd_img / w.ni;
#endif
  pu[0] = xr - tt * Ld;
  pu[1] = yr - tt * Md;
  pu[2] = zr - tt * Nd;
  return opd;
}

// ... and the OPD in waves against the reference's own path `opd_ref`
template <typename T, bool TILT_FIRST = false>
OL_DEV T wavefront_one(const WavefrontConsts<T>& w, T xr, T yr, T zr, T Ld, T Md, T Nd, T opd_in,
                       T px, T py, T (&pu)[3]) {
  const T opd = wavefront_opd_mm<T, TILT_FIRST>(w, xr, yr, zr, Ld, Md, Nd, opd_in, px, py, pu);
  return (w.opd_ref - opd) * w.inv_w;
}

}  // namespace ol
