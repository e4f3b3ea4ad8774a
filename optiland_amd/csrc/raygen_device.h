// raygen_device.h -- per-ray launch state from normalised field / pupil coordinates.
// ONE definition shared by the stand-alone ray generator (aux_kernels.hip) and the
// fused generate -> trace -> reduce spot kernel (trace_kernel.hip), so both produce
// the same rays.  Reference: rays/ray_generator.py:47-99,
// rays/ray_aiming/paraxial.py:33-106, fields/field_types/angle.py:17-58.
#pragma once
#include <hip/hip_runtime.h>

#include "device_table.h"
#include "trace_launch.h"
#include "surface_math.h"  // Math<T>: the hardware reciprocal square root

// OL_RAYGEN_RSQ (default on): the launch direction is normalised with ONE reciprocal square
// root (v_rsq_f32; in fp64 the hardware seed + two refinement steps, Math<double>::rsqrt)
// instead of an IEEE square root and three IEEE quotients -- 37 instead of 72 vector
// instructions per generated ray in fp32, ~40 instead of ~150 in fp64, in every generating
// kernel (trace, spot, OPD, the update_intensity epilogue regenerates the ray once more).
// Within 2 ulp of the correctly rounded direction cosines.  A/B knob, tools/build_variants.py.
#ifndef OL_RAYGEN_RSQ
#define OL_RAYGEN_RSQ 1
#endif

namespace ol {

template <typename T>
OL_DEV T tan_deg(T deg);
template <>
OL_DEV float tan_deg<float>(float deg) {
  // formed in double: the field angle is a per-field constant in practice and
  // fp32 tanf of a degree->radian product would cost 1e-7 of a 20 mm offset.
  return (float)tan((double)deg * 0.017453292519943295);
}
template <>
OL_DEV double tan_deg<double>(double deg) {
  return tan(deg * 0.017453292519943295);
}

// RaygenConsts (trace_launch.h) read out of the kernarg segment, field by field (scalar loads)
template <typename T>
OL_DEV RaygenConsts<T> load_consts(cptr<RaygenConsts<T>> p) {
  RaygenConsts<T> c;
  c.EPL = p->EPL; c.EPD = p->EPD; c.maxf = p->maxf; c.off_epl = p->off_epl;
  c.z_inf = p->z_inf; c.z_fin = p->z_fin; c.epl_z = p->epl_z; c.tele_dz = p->tele_dz;
  c.apod_a = p->apod_a; c.apod_b = p->apod_b; c.apod_kind = p->apod_kind;
  c.infinite = p->infinite; c.height = p->height; c.linear = p->linear;
  c.telecentric = p->telecentric;
  return c;
}

// initial intensity from the pupil apodization (ray_generator.py:81-85,
// optiland/apodization/*.py); launch-uniform switch
template <typename T>
OL_DEV T raygen_apodize(const RaygenConsts<T>& c, T px, T py) {
  if (c.apod_kind == 0) return T(1);
  const T pi = T(3.14159265358979323846);
  const T r2 = px * px + py * py;
  const T r = sqrt(r2);
  const T a = c.apod_a, b = c.apod_b;
  switch (c.apod_kind) {
    case 1: return exp(-r2 / (T(2) * a * a));                                  // gaussian.py
    case 2: {                                                                  // cosine_squared.py
      const T cs = cos(pi * r / (T(2) * a));
      return r < a ? cs * cs : T(0);
    }
    case 3: return r < a / T(2) ? T(0.5) * (T(1) - cos(T(2) * pi * r / a)) : T(0);  // hann.py
    case 4: {                                                                  // polynomial.py
      const T q = r / a;
      return r < a ? pow(T(1) - q * q, b) : T(0);
    }
    case 5: return exp(-pow(r / a, b));                                        // super_gaussian.py
    default: {                                                                 // tukey.py
      const T flat = a * (T(1) - b / T(2));
      const T taper = T(0.5) * (T(1) + cos(pi * (r - flat) / (a * b / T(2))));
      T i = r <= flat ? T(1) : T(0);
      return (r > flat && r < a) ? taper : i;
    }
  }
}

// field quantity per axis: tan(field angle) (angle.py:40-47), the object height
// (object_height.py:38-41), or the paraxially scaled slope / height of an image-height
// field (paraxial_image_height.py:36-60); hoistable when the field is launch-uniform
template <typename T>
OL_DEV void raygen_field(const RaygenConsts<T>& c, T hx, T hy, T& tx, T& ty) {
  if (c.linear) {  // object height, or paraxial image height (slope or height scale)
    tx = c.maxf * hx;
    ty = c.maxf * hy;
  } else {
    tx = tan_deg<T>(c.maxf * hx);
    ty = tan_deg<T>(c.maxf * hy);
  }
}

// range checks (real_ray_tracer.py:156-173: all((v >= -1) & (v <= 1)); NaN fails) and
// the trace_generic pre-scaling of the pupil (real_ray_tracer.py:134-137)
template <typename T>
OL_DEV bool outside_unit(T v) { return !(v >= T(-1) && v <= T(1)); }

template <typename T>
OL_DEV void raygen_pupil(uint32_t flags, T vx, T vy, T& px, T& py,
                                             uint32_t& status) {
  if ((flags & kRaygenCheckPupil) && (outside_unit(px) || outside_unit(py)))
    status |= kStatusPupilRange;
  if (flags & kRaygenPrescalePupil) {
    px *= vx;
    py *= vy;
  }
}

// o[0..5] = x, y, z, L, M, N  (intensity is 1, ray_generator.py:81-85)
template <typename T>
OL_DEV void raygen_one(const RaygenConsts<T>& c, T tx, T ty, T px, T py, T vx,
                                           T vy, T (&o)[6]) {
  T x0, y0, z0;
  if (c.height) {  // object_height.py:36-47 (planar object surface)
    x0 = tx;
    y0 = ty;
    z0 = c.z_fin;
  } else if (c.infinite) {
    x0 = px * c.EPD / T(2) * vx + (-tx * c.off_epl);
    y0 = py * c.EPD / T(2) * vy + (-ty * c.off_epl);
    z0 = c.z_inf;
  } else {
    x0 = -tx * c.epl_z;
    y0 = -ty * c.epl_z;
    z0 = c.z_fin;
  }
  T x1, y1, z1;
  if (c.telecentric) {  // ray_aiming/paraxial.py:82-87
    x1 = px * vx + x0;
    y1 = py * vy + y0;
    z1 = c.tele_dz + z0;
  } else {              // :88-94
    x1 = px * c.EPD * vx / T(2);
    y1 = py * c.EPD * vy / T(2);
    z1 = c.EPL;
  }
  const T dx = x1 - x0, dy = y1 - y0, dz = z1 - z0;
  o[0] = x0;
  o[1] = y0;
  o[2] = z0;
#if OL_RAYGEN_RSQ
  using m = Math<T>;
  const T m2 = m::fma(dx, dx, m::fma(dy, dy, dz * dz));
  // paraxial.py:96-104: mag < 1e-9  <=>  mag^2 < 1e-18 (NaN compares false either way)
  const bool is_zero = m2 < T(1e-18);
  const T inv = m::rsqrt(m2);
  o[3] = is_zero ? T(0) : dx * inv;
  o[4] = is_zero ? T(0) : dy * inv;
  o[5] = is_zero ? T(1) : dz * inv;
#else
  T mag = sqrt(dx * dx + dy * dy + dz * dz);
  const bool is_zero = mag < T(1e-9);  // paraxial.py:96-104
  mag = is_zero ? T(1) : mag;
  o[3] = is_zero ? T(0) : dx / mag;
  o[4] = is_zero ? T(0) : dy / mag;
  o[5] = is_zero ? T(1) : dz / mag;
#endif
}

}  // namespace ol
