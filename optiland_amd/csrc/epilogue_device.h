// epilogue_device.h -- per-ray arithmetic of the epilogue kernels in aux_kernels.hip:
// `update_intensity` of a polarised trace and one sample of the FFT-PSF pupil function.
// OL_DEV like surface_math.h: device code in the product; tests/hostmath compiles it for
// the host as a checker.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "trace_launch.h"
#include "surface_math.h"  // Math<T>: the hardware reciprocal square root

namespace ol {

// (PolFields<T>: trace_launch.h)

// rays/polarized_rays.py:68-133, 204-233.  Real PRT (P): |P E0|^2 = |P Re E0|^2 +
// |P Im E0|^2; complex PRT (CPLX: P + i Q): full complex product.  (kx, ky, kz): the
// direction the ray was launched with; flag gets OL_STATUS_K_PARALLEL_X (:221-222).
// One incident state E0 = (ar + i ai) s_hat + (br + i bi) p_hat:
template <typename T, bool CPLX>
OL_DEV T pol_state_term(T ar, T ai, T br, T bi, const T (&s)[3], const T (&p)[3],
                        const T (&P)[9], const T (&Q)[9]) {
  const T er[3] = {ar * s[0] + br * p[0], ar * s[1] + br * p[1], ar * s[2] + br * p[2]};
  const T ei[3] = {ai * s[0] + bi * p[0], ai * s[1] + bi * p[1], ai * s[2] + bi * p[2]};
  T acc = T(0);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    T vr = P[3 * a] * er[0] + P[3 * a + 1] * er[1] + P[3 * a + 2] * er[2];
    T vi = P[3 * a] * ei[0] + P[3 * a + 1] * ei[1] + P[3 * a + 2] * ei[2];
    if (CPLX) {
      vr -= Q[3 * a] * ei[0] + Q[3 * a + 1] * ei[1] + Q[3 * a + 2] * ei[2];
      vi += Q[3 * a] * er[0] + Q[3 * a + 1] * er[1] + Q[3 * a + 2] * er[2];
    }
    acc += vr * vr + vi * vi;
  }
  return acc;
}

// The same term for an incident state with REAL amplitudes (ai = bi = 0, a launch-uniform
// fact: both states of the unpolarised mean -- E0 = s_hat and E0 = p_hat,
// polarized_rays.py:122-133 -- and every linear state): Im E0 = 0 and half of the products
// above are products with zero.  (They only differ where the matrix holds an infinity, which
// 0 turns into NaN and this form leaves as it is.)  Round 6: the epilogue of the C5 launch was
// 132 vector instructions per ray, a sixth of the kernel (profiles/r06_phase_costs_before.txt).
template <typename T, bool CPLX>
OL_DEV T pol_state_term_real(T ar, T br, const T (&s)[3], const T (&p)[3], const T (&P)[9],
                             const T (&Q)[9]) {
  const T er[3] = {ar * s[0] + br * p[0], ar * s[1] + br * p[1], ar * s[2] + br * p[2]};
  T acc = T(0);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const T vr = P[3 * a] * er[0] + P[3 * a + 1] * er[1] + P[3 * a + 2] * er[2];
    acc += vr * vr;
    if (CPLX) {
      const T vi = Q[3 * a] * er[0] + Q[3 * a + 1] * er[1] + Q[3 * a + 2] * er[2];
      acc += vi * vi;
    }
  }
  return acc;
}

// +-0 without a vector compare (the amplitudes are launch-uniform: scalar registers)
OL_DEV bool amplitude_is_zero(float v) { return (hw::float_bits(v) & 0x7fffffff) == 0; }
OL_DEV bool amplitude_is_zero(double v) {
  return (hw::double_bits(v) & 0x7fffffffffffffffll) == 0;
}

template <typename T, bool CPLX>
OL_DEV T pol_intensity_one(const PolFields<T>& f, T kx, T ky, T kz, const T (&P)[9],
                           const T (&Q)[9], T i0, uint32_t& flag) {
  using m = Math<T>;
  // p = k x x_hat = (0, kz, -ky), normalised; s = p x k.  One reciprocal square root for the
  // norm (v_rsq_f32; fp64: the seed + refinement of Math<double>) instead of an IEEE square
  // root and two IEEE quotients -- as the ray generator normalises the direction itself
  const T n2 = m::fma(kz, kz, ky * ky);
  if (n2 == T(0)) flag |= 0x2u;  // OL_STATUS_K_PARALLEL_X
  const T inv = m::rsqrt(n2);
  const T p[3] = {T(0), kz * inv, -(ky * inv)};
  const T s[3] = {p[1] * kz - p[2] * ky, p[2] * kx, -(p[1] * kx)};
  // (the one or two incident states by CONSTANT index: a run-time index into the amplitude
  // arrays sends them to LDS / scratch)
  const bool real0 = amplitude_is_zero(f.ai[0]) && amplitude_is_zero(f.bi[0]);
  T acc = real0 ? pol_state_term_real<T, CPLX>(f.ar[0], f.br[0], s, p, P, Q)
                : pol_state_term<T, CPLX>(f.ar[0], f.ai[0], f.br[0], f.bi[0], s, p, P, Q);
  if (f.nf > 1) {
    const bool real1 = amplitude_is_zero(f.ai[1]) && amplitude_is_zero(f.bi[1]);
    acc += real1 ? pol_state_term_real<T, CPLX>(f.ar[1], f.br[1], s, p, P, Q)
                 : pol_state_term<T, CPLX>(f.ar[1], f.ai[1], f.br[1], f.bi[1], s, p, P, Q);
  }
  // (1 / nf: 1 or 1/2, exact)
  return acc * i0 * (f.nf > 1 ? T(0.5) : T(1));
}

// analysis/spot_diagram/core.py:329-372, 440-481: what one image-plane hit adds to the
// masked (i > 0) spot moments about (cx, cy) -- count, sum dx, sum dy, sum dx^2, sum dy^2,
// sum i -- and to the largest squared radius (NaN hits compare false and are skipped)
template <typename T>
OL_DEV void spot_accumulate(double (&s)[6], double& rmax, T x, T y, T i, double cx, double cy) {
  if (i > T(0)) {
    const double dx = (double)x - cx, dy = (double)y - cy;
    const double dx2 = dx * dx, dy2 = dy * dy;
    s[0] += 1.0;
    s[1] += dx;
    s[2] += dy;
    s[3] += dx2;
    s[4] += dy2;
    s[5] += (double)i;
    const double r2 = dx2 + dy2;
    rmax = r2 > rmax ? r2 : rmax;
  }
}

// what one ray adds to the twelve moments of ol_trace_opd (trace_launch.h: kOpdMoments):
// the nine intensity-weighted moments of the tilt fit (wavefront/wavefront.py:103-148) and
// count / sum / sum of squares of the OPD over rays with i > 0 (wavefront/opd.py:145-159)
OL_DEV void opd_accumulate(double (&s)[kOpdMoments], double wi, double od, double X, double Y,
                           bool alive) {
  s[0] += wi; s[1] += wi * X; s[2] += wi * Y;
  s[3] += wi * X * X; s[4] += wi * X * Y; s[5] += wi * Y * Y;
  s[6] += wi * od; s[7] += wi * od * X; s[8] += wi * od * Y;
  if (alive) {
    s[9] += 1.0;
    s[10] += od;
    s[11] += od * od;
  }
}

// analysis/encircled_energy.py:147-160: the radius step a hit's energy goes to -- the first
// index with r <= steps[idx] (the reference's own `radii <= r` comparisons), -1 for a NaN
// energy, a NaN radius or one beyond the last step
OL_DEV int radial_step_index(const double* steps, int n_steps, double r, double e) {
  if (!(e == e) || !(r <= steps[n_steps - 1])) return -1;
  int lo = 0, hi = n_steps - 1;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (r <= steps[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}

// analysis/irradiance.py:341-353: numpy.histogram2d's bin of v = searchsorted(edges, v,
// "right") - 1 with the right-most edge folded into the last bin; -1 outside or NaN
OL_DEV int edge_bin(const double* __restrict__ e, int nb, double v) {
  if (!(v >= e[0]) || !(v <= e[nb])) return -1;
  if (v == e[nb]) return nb - 1;
  int lo = 0, hi = nb;  // invariant: e[lo] <= v < e[hi]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (e[mid] <= v) lo = mid; else hi = mid;
  }
  return lo;
}

// psf/fft.py:101-137: one sample A exp(-i 2 pi OPD) of the pupil function, and where it
// goes in the zero-padded grid (re, im interleaved doubles)
OL_DEV void pupil_sample(double opd_waves, double intensity, double& re, double& im) {
  const double amp = sqrt(intensity);
  double sn, cs;
  sincos(-6.283185307179586476925286766559 * opd_waves, &sn, &cs);
  re = amp * cs;
  im = amp * sn;
}

OL_DEV int64_t pupil_cell_offset(int32_t cell, int32_t n_side, int32_t grid, int32_t pad) {
  const int64_t r = cell / n_side, cc = cell - r * n_side;
  return ((r + pad) * (int64_t)grid + (cc + pad)) * 2;
}

}  // namespace ol
