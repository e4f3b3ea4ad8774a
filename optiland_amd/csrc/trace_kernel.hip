// trace_kernel.hip -- fused whole-system sequential ray trace for gfx950 (MI355X).
//
// One launch replaces the reference's Python loop over surfaces
// (optiland/surfaces/surface_group.py:245-257) and the ~100 N-element array
// operations it runs per surface.  Each thread owns RPT rays
// (struct-of-arrays in HBM; RPT = 1, or the 16-byte vector of consecutive rays per
// plane per lane -- chosen per mode from measurements, see launch_trace()), keeps
// their state in VGPRs across ALL surfaces and streams the recorded per-surface
// state out.  Surface constants are wave-uniform -> scalar loads / SGPR operands
// (one s_load_dwordx16 per surface, prefetched one surface ahead).  No MFMA, no
// LDS: this is a streaming vector-ALU path bounded by HBM write bandwidth in
// record-all mode.
//
// Per-surface arithmetic follows SURVEY.md Appendix A; each device function
// cites the reference lines it implements.  Differences that are deliberate:
//   * conic intersection uses the cancellation-free root  t = C / q  (the
//     reference's (-b +- sqrt(d)) / 2a loses digits for near-flat surfaces;
//     same root selection rule, see conic_distance());
//   * ray state is carried in the LOCAL frame of the last surface and moved to
//     the next frame with a host-precomputed relative transform; the global
//     coordinates the reference records are formed only for the store;
//   * the Newton-Raphson loop is re-based on the conic hit, stops per ray (the
//     reference's test is a global max over the batch, newton_raphson.py:148)
//     and hands its last gradient to the normal -- see newton_iterate().
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdlib.h>

#include <type_traits>

#include "device_table.h"
#include "raygen_device.h"
#include "wavefront_device.h"
#include "trace_launch.h"

#ifndef OL_TABLE_IN_LDS
#define OL_TABLE_IN_LDS 0  // 1: stage the surface table in LDS (measured slower)
#endif

namespace ol {

// --------------------------------------------------------------------------
// arithmetic helpers
// --------------------------------------------------------------------------
template <typename T>
struct Math;

// lane-wise compare / select vocabulary shared by the scalar types and the packed
// pair (Math<f32x2>): the lean (conic-only, unpolarised) path is written once in these
// terms and instantiated for T and for f32x2.
#define OL_SCALAR_LANE_OPS(T)                                                             \
  using scalar = T;                                                                       \
  using mask = bool;                                                                      \
  static constexpr int lanes = 1;                                                         \
  static __device__ __forceinline__ T splat(T v) { return v; }                            \
  static __device__ __forceinline__ bool lt(T a, T b) { return a < b; }                   \
  static __device__ __forceinline__ bool le(T a, T b) { return a <= b; }                  \
  static __device__ __forceinline__ bool gt(T a, T b) { return a > b; }                   \
  static __device__ __forceinline__ bool ge(T a, T b) { return a >= b; }                  \
  static __device__ __forceinline__ bool eq(T a, T b) { return a == b; }                  \
  static __device__ __forceinline__ bool ne(T a, T b) { return a != b; }                  \
  static __device__ __forceinline__ bool all(bool v) { return v; }                        \
  static __device__ __forceinline__ bool mnot(bool m) { return !m; }                      \
  static __device__ __forceinline__ bool mand(bool p, bool q) { return p && q; }          \
  static __device__ __forceinline__ bool same(bool p, bool q) { return p == q; }          \
  static __device__ __forceinline__ T select(bool m, T a, T b) { return m ? a : b; }      \
  static __device__ __forceinline__ bool mselect(bool m, bool a, bool b) { return m ? a : b; }

template <>
struct Math<float> {
  // v_rcp_f32 / v_sqrt_f32 / v_rsq_f32: 1 ulp, quarter rate, no denormal
  // fix-up sequences -- well inside the 1e-4 fp32 parity budget.
  static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
  static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
  static __device__ __forceinline__ float rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
  static __device__ __forceinline__ float div(float a, float b) { return a * rcp(b); }
  static __device__ __forceinline__ float exp(float x) { return __expf(x); }
  static __device__ __forceinline__ float abs(float x) { return __builtin_fabsf(x); }
  static __device__ __forceinline__ float copysign(float a, float b) {
    return __builtin_copysignf(a, b);
  }
  static __device__ __forceinline__ float fma(float a, float b, float c) {
    return __builtin_fmaf(a, b, c);
  }
  static __device__ __forceinline__ float eps() { return 1.1920929e-7f; }
  static __device__ __forceinline__ float guard() { return 1e-14f; }
  OL_SCALAR_LANE_OPS(float)
};

template <>
struct Math<double> {
  static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
  static __device__ __forceinline__ double sqrt(double x) { return __builtin_sqrt(x); }
  static __device__ __forceinline__ double rsqrt(double x) { return 1.0 / __builtin_sqrt(x); }
  static __device__ __forceinline__ double div(double a, double b) { return a / b; }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double abs(double x) { return __builtin_fabs(x); }
  static __device__ __forceinline__ double copysign(double a, double b) {
    return __builtin_copysign(a, b);
  }
  static __device__ __forceinline__ double fma(double a, double b, double c) {
    return __builtin_fma(a, b, c);
  }
  static __device__ __forceinline__ double eps() { return 2.220446049250313e-16; }
  static __device__ __forceinline__ double guard() { return 1e-14; }
  OL_SCALAR_LANE_OPS(double)
};

// Two fp32 rays in one 64-bit register pair.  add / mul / fma on this type compile to
// the packed instructions v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32, which issue at
// the same rate as their scalar forms and so do two rays' worth of arithmetic per
// issue slot -- the only way to the full fp32 vector rate on CDNA3/4.  sqrt / rcp /
// compares / selects stay one instruction per ray (there are no packed forms), and a
// mask is a pair of bools so that a compare still lands in an SGPR pair and feeds
// v_cndmask directly, exactly like the scalar code.
typedef float f32x2 __attribute__((ext_vector_type(2)));
struct Mask2 {
  bool a, b;
};

template <>
struct Math<f32x2> {
  using scalar = float;
  using mask = Mask2;
  static constexpr int lanes = 2;
  using V = f32x2;
  static __device__ __forceinline__ V rcp(V x) {
    return V{__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
  }
  static __device__ __forceinline__ V sqrt(V x) {
    return V{__builtin_amdgcn_sqrtf(x.x), __builtin_amdgcn_sqrtf(x.y)};
  }
  static __device__ __forceinline__ V rsqrt(V x) {
    return V{__builtin_amdgcn_rsqf(x.x), __builtin_amdgcn_rsqf(x.y)};
  }
  static __device__ __forceinline__ V div(V a, V b) { return a * rcp(b); }
  static __device__ __forceinline__ V exp(V x) { return V{__expf(x.x), __expf(x.y)}; }
  static __device__ __forceinline__ V abs(V x) {
    return V{__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
  }
  static __device__ __forceinline__ V copysign(V a, V b) {
    return V{__builtin_copysignf(a.x, b.x), __builtin_copysignf(a.y, b.y)};
  }
  static __device__ __forceinline__ V fma(V a, V b, V c) {
    return __builtin_elementwise_fma(a, b, c);
  }
  static __device__ __forceinline__ float eps() { return 1.1920929e-7f; }
  static __device__ __forceinline__ float guard() { return 1e-14f; }
  static __device__ __forceinline__ V splat(float v) { return V{v, v}; }
  static __device__ __forceinline__ mask lt(V a, V b) { return {a.x < b.x, a.y < b.y}; }
  static __device__ __forceinline__ mask le(V a, V b) { return {a.x <= b.x, a.y <= b.y}; }
  static __device__ __forceinline__ mask gt(V a, V b) { return {a.x > b.x, a.y > b.y}; }
  static __device__ __forceinline__ mask ge(V a, V b) { return {a.x >= b.x, a.y >= b.y}; }
  static __device__ __forceinline__ mask eq(V a, V b) { return {a.x == b.x, a.y == b.y}; }
  static __device__ __forceinline__ mask ne(V a, V b) { return {a.x != b.x, a.y != b.y}; }
  static __device__ __forceinline__ mask all(bool v) { return {v, v}; }
  static __device__ __forceinline__ mask mnot(mask m) { return {!m.a, !m.b}; }
  static __device__ __forceinline__ mask mand(mask p, mask q) { return {p.a && q.a, p.b && q.b}; }
  static __device__ __forceinline__ mask same(mask p, mask q) { return {p.a == q.a, p.b == q.b}; }
  static __device__ __forceinline__ V select(mask m, V a, V b) {
    return V{m.a ? a.x : b.x, m.b ? a.y : b.y};
  }
  static __device__ __forceinline__ mask mselect(mask m, mask a, mask b) {
    return {m.a ? a.a : b.a, m.b ? a.b : b.b};
  }
};

template <typename T>
struct Ray {
  T x, y, z, L, M, N, i, opd;
};

// 3x3 real polarisation ray-tracing matrix (the imaginary part
// is identically zero for uncoated / Fresnel surfaces unless the ray is already
// NaN through total internal reflection).
// POLK: 0 = no polarisation, 1 = real PRT (9 values), 2 = complex PRT (18 values:
// real part then imaginary part; needed only behind a retarder, jones.py:331-393).
template <typename T, int POLK>
struct Prt {
  T m[POLK == 2 ? 18 : 9];
};

// Surface Jones matrix in the local (s, p) basis: 2x2 block A + iB, and the
// k-component factor j22 (jones.py:109-117: +-1).
template <typename T>
struct Jones {
  T a00, a01, a10, a11, b00, b01, b10, b11, j22;
};

// --------------------------------------------------------------------------
// geometry: conic
// --------------------------------------------------------------------------
// standard.py:97-148.  Reference quadratic a t^2 + b t + c with
//   a = R*A, b = R*B, c = R*C;  A = cv(L^2+M^2+(1+k)N^2),
//   B/2 = E = cv(xL+yM+(1+k)zN) - N,  C = cv(x^2+y^2+(1+k)z^2) - 2z.
// Roots: t_a = C/q (no cancellation), t_b = q/A with q = -(E + sgn(E) sqrt(E^2-AC)).
// Reference picks t1 if |z+t1 N| <= |z+t2 N| else t2 where
//   t1 = (-E + sgn(R) sqrt(disc))/A, t2 = (-E - sgn(R) sqrt(disc))/A;
// a == 0 -> -c/b which is exactly t_a.
template <typename V>
__device__ __forceinline__ V flat_distance(V z, V N) {  // standard.py:108-111
  using m = Math<V>;
  const V g = m::splat(m::guard());
  V Ns = m::select(m::gt(m::abs(N), g), N, g);
  return -m::div(z, Ns);
}

template <typename V>
__device__ __forceinline__ V curved_distance(typename Math<V>::scalar cv,
                                             typename Math<V>::scalar kp1, V x, V y, V z, V L, V M,
                                             V N) {
  using m = Math<V>;
  const V zero = m::splat(0);
  const V kz = kp1 * z, kN = kp1 * N;
  V E = m::fma(m::splat(cv), m::fma(x, L, m::fma(y, M, kz * N)), -N);
  V A = cv * m::fma(L, L, m::fma(M, M, kN * N));
  V C = m::fma(m::splat(cv), m::fma(x, x, m::fma(y, y, kz * z)), m::splat(-2) * z);
  V disc = m::fma(E, E, -A * C);
  V sq = m::sqrt(disc);  // NaN when the ray misses (standard.py:132-137)
  V q = -(E + m::copysign(sq, E));
  V ta = m::div(C, q);
  V tb = m::div(q, A);
  // t_b is the reference's t1 iff -sgn(E) == sgn(R); the reference keeps t1 when
  // |z + t1 N| <= |z + t2 N| and t2 otherwise (also when the comparison is NaN)
  const auto b_is_t1 = m::same(m::lt(E, zero), m::all(cv > 0));
  const V t1 = m::select(b_is_t1, tb, ta), t2 = m::select(b_is_t1, ta, tb);
  const V z1 = m::abs(m::fma(t1, N, z)), z2 = m::abs(m::fma(t2, N, z));
  V t = m::select(m::le(z1, z2), t1, t2);
  t = m::select(m::eq(A, zero), ta, t);
  return t;
}

template <typename T>
__device__ __forceinline__ T conic_distance(const DevSurf<T>& s, T x, T y, T z, T L, T M, T N) {
  if (s.flags & kSurfRadiusInf) return flat_distance(z, N);
  return curved_distance(s.cv, s.kp1, x, y, z, L, M, N);
}

// Unit normal of the conic at the hit point (x, y, z).
// Reference (standard.py:150-175): (fx, fy) = cv (x, y) / sqrt(D), D = 1 - (1+k) cv^2 r^2,
// n = (fx, fy, -1) / sqrt(fx^2 + fy^2 + 1) -- two reciprocal square roots.  Multiplying
// through by sqrt(D):  n = (cv x, cv y, -sqrt(D)) / sqrt(cv^2 r^2 + D), and ON the
// surface sqrt(D) = |1 - cv (1+k) z|  (z = cv r^2 / (1 + sqrt(D))), so the hit point's
// own z replaces the first square root.  For a sphere (k = 0) the denominator is
// cv^2 (x^2 + y^2 + z^2) - 2 cv z + 1 = 1 identically: NO transcendental at all; other
// conics keep one rsq.  |.| keeps the reference's sign when the selected root lies on
// the far sheet.  (v_rsq / v_sqrt are quarter rate: on the VALU-bound record-last and
// fused-spot kernels the two removed rsq were 11 % of the issue cycles.)
template <typename V>
__device__ __forceinline__ void conic_normal(typename Math<V>::scalar cv,
                                             typename Math<V>::scalar kp1, V x, V y, V z, V& nx,
                                             V& ny, V& nz) {
  using m = Math<V>;
  const V w = m::abs(m::fma(m::splat(-cv * kp1), z, m::splat(1)));
  nx = cv * x;
  ny = cv * y;
  nz = -w;
  const V n2 = m::fma(nx, nx, m::fma(ny, ny, w * w));
  V h;
  if (kp1 != typename m::scalar(1)) {  // surface-uniform
    h = m::rsqrt(n2);
  } else {
    // sphere: |n|^2 = 1 + e with e = O(rounding of the hit point); one Newton step of
    // 1/sqrt at 1 (1 - e/2) restores the unit length to O(e^2) without a transcendental
#ifndef OL_SPHERE_RENORM
#define OL_SPHERE_RENORM 1
#endif
    h = OL_SPHERE_RENORM ? m::fma(m::splat(-0.5), n2, m::splat(1.5)) : m::splat(1);
  }
  nx = nx * h;
  ny = ny * h;
  nz = nz * h;
}

// --------------------------------------------------------------------------
// Newton-Raphson geometries: sag + gradient at (x, y)
// --------------------------------------------------------------------------
// even_asphere.py:93-140 (Horner in r^2 instead of r2**(i+1))
template <typename T>
__device__ __forceinline__ void even_asphere_eval(const DevSurf<T>& s, const T* __restrict__ c,
                                                  T x, T y, T& sag, T& fx, T& fy) {
  using m = Math<T>;
  T r2 = m::fma(x, x, y * y);
  T g = m::sqrt(m::fma(-s.kp1 * s.cv * s.cv, r2, T(1)));
  sag = m::div(s.cv * r2, T(1) + g);
  T f = m::div(s.cv, g);
  // P(r2) = sum C_i r2^(i+1);  P'(r2) = sum (i+1) C_i r2^i
  T p = T(0), dp = T(0);
  for (int i = s.n_coeff - 1; i >= 0; --i) {
    T ci = c[i];
    dp = m::fma(dp, r2, T(i + 1) * ci);
    p = m::fma(p, r2, ci);
  }
  sag = m::fma(p, r2, sag);
  f = m::fma(T(2), dp, f);
  fx = x * f;
  fy = y * f;
}

// odd_asphere.py:86-143: sum C_i r^(i+1); gradient terms (i+1) x C_i r^(i-1),
// non-finite terms (i == 0 at r == 0) zeroed.
template <typename T>
__device__ __forceinline__ void odd_asphere_eval(const DevSurf<T>& s, const T* __restrict__ c,
                                                 T x, T y, T& sag, T& fx, T& fy) {
  using m = Math<T>;
  T r2 = m::fma(x, x, y * y);
  T r = m::sqrt(r2);
  T g = m::sqrt(m::fma(-s.kp1 * s.cv * s.cv, r2, T(1)));
  sag = m::div(s.cv * r2, T(1) + g);
  T f = m::div(s.cv, g);
  // Q(r) = sum_{i>=1} (i+1) C_i r^(i-1);  P(r) = sum C_i r^(i+1)
  T p = T(0), q = T(0);
  for (int i = s.n_coeff - 1; i >= 0; --i) {
    T ci = c[i];
    p = m::fma(p, r, ci);
    if (i >= 1) q = m::fma(q, r, T(i + 1) * ci);
  }
  sag = m::fma(p, r, sag);
  T c0 = s.n_coeff > 0 ? c[0] : T(0);
  T t0 = r > T(0) ? m::div(c0, r) : T(0);  // i = 0 term: x C_0 / r, 0 at r == 0
  f = f + q + t0;
  fx = x * f;
  fy = y * f;
}

// polynomial.py:105-155: sum c[i][j] x^i y^j (row i = x power), nested Horner.
template <typename T>
__device__ __forceinline__ void polynomial_eval(const DevSurf<T>& s, const T* __restrict__ c,
                                                T x, T y, T& sag, T& fx, T& fy) {
  using m = Math<T>;
  T r2 = m::fma(x, x, y * y);
  T g = m::sqrt(m::fma(-s.kp1 * s.cv * s.cv, r2, T(1)));
  sag = m::div(s.cv * r2, T(1) + g);
  T f = m::div(s.cv, g);
  fx = x * f;
  fy = y * f;
  const int cols = s.cold->poly_cols;
  const int rows = cols > 0 ? s.n_coeff / cols : 0;
  // outer Horner in x over rows; inner Horner in y gives q_i(y) and q_i'(y)
  T P = T(0), dPdx = T(0), dPdy = T(0);
  for (int i = rows - 1; i >= 0; --i) {
    T qi = T(0), dqi = T(0);
    for (int j = cols - 1; j >= 0; --j) {
      T cij = c[i * cols + j];
      dqi = m::fma(dqi, y, qi);
      qi = m::fma(qi, y, cij);
    }
    dPdx = m::fma(dPdx, x, P);
    P = m::fma(P, x, qi);
    dPdy = m::fma(dPdy, x, dqi);
  }
  sag += P;
  fx += dPdx;
  fy += dPdy;
}

// zernike.py:153-252 + zernike/base.py:42-137.  Terms regrouped on the host per
// azimuthal order m into radial polynomials in u = rho^2 with the factor rho^m taken out
// (capi.hip:build_zernike_block, layout in device_table.h).  Evaluated in CARTESIAN form:
// with the harmonic polynomials
//     A_m + i B_m = (x_n + i y_n)^m   (= rho^m (cos m phi + i sin m phi)),
// advanced by one complex multiply per order, the cos and sin terms of one order are
// Qc(u) A_m + Qs(u) B_m and their gradient follows from
//     d(A_m, B_m)/dx_n = m (A_{m-1}, B_{m-1}),   d(A_m, B_m)/dy_n = m (-B_{m-1}, A_{m-1}),
// so there is no atan2 / cos / sin, no sqrt, no reciprocal and no polar chain rule in
// the loop.  The (cos, sin) pair of every quantity lives in one 2-vector: in fp32 the
// three Horner chains (sag polynomial, normal polynomial, its u-derivative from host-made
// derivative coefficients) and the harmonic recurrence are packed v_pk_fma_f32 /
// v_pk_mul_f32 -- one issue slot for both kinds -- and the order / length headers are
// integer bit patterns, so the loop control stays on the scalar unit.  (The polar form
// the reference writes down costs 4 transcendentals per evaluation and ~20 vector
// operations per (m, kind) group; this kernel is VALU-issue bound: profiles/r02_zf_*.)
// Away from the vertex the two forms are the same polynomial.  AT the vertex the
// reference's chain rule is regularised with eps = 1e-14 (zernike.py:206-231:
// drho/dx = x_n / (rho + eps) / norm, dphi/dx = -y_n / (rho^2 + eps) / norm), which damps
// the radial part of the gradient by rho / (rho + eps), the azimuthal part by
// rho^2 / (rho^2 + eps) and makes it exactly zero at rho == 0 (the tilt terms' true
// gradient is not): reproduced below for the rays that need it (rho^2 < 1e-8) by
// splitting the Cartesian gradient into those two parts.
template <typename T>
using vec2 = T __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int slot_int(const float* p) { return __float_as_int(*p); }
__device__ __forceinline__ int slot_int(const double* p) { return (int)__double_as_longlong(*p); }

template <typename T>
__device__ __forceinline__ void zernike_eval(const DevSurf<T>& s, const T* __restrict__ c, T x,
                                             T y, T& sag, T& fx, T& fy, uint32_t& status) {
  using m = Math<T>;
  using V2 = vec2<T>;
  T r2 = m::fma(x, x, y * y);
  T g = m::sqrt(m::fma(-s.kp1 * s.cv * s.cv, r2, T(1)));
  sag = m::div(s.cv * r2, T(1) + g);
  T f = m::div(s.cv, g);
  fx = x * f;
  fy = y * f;
  const T inv = s.cold->inv_norm;
  const T xn = x * inv, yn = y * inv;
  if (m::abs(xn) > T(1) || m::abs(yn) > T(1)) status |= 0x1u;  // OL_STATUS_ZERNIKE_RANGE
  const T u = m::fma(xn, xn, yn * yn);
  const V2 uu = {u, u};

  T zsum = T(0), gx = T(0), gy = T(0);  // gradient w.r.t. (x_n, y_n)
  // H = (A_m, B_m) of the current order, Hp of the one below (weighted by m = 0 at order 0)
  V2 H = {T(1), T(0)}, Hp = {T(0), T(0)};
  int mcur = 0;
  const T* p = c;
  for (int lv = 0; lv < s.n_coeff; ++lv) {
    const int mg = slot_int(p), K = slot_int(p + 1);
    p += kZernLevelHeader;
    for (; mcur < mg; ++mcur) {  // levels come sorted by ascending m (uniform trip count)
      Hp = H;
      // (A, B) <- (x A - y B, y A + x B)
      const V2 sw = {-Hp.y, Hp.x};
      H = xn * Hp + yn * sw;
    }
    // per power of u, (cos, sin) pairs of: a = sag coefficients (normalisation constant
    // included), b = coefficients of the NORMAL (the reference forms it without the
    // constant, zernike.py:234-240), d = (k + 1) b_{k+1} (dQn/du)
    // (K >= 1: the chains start from the highest coefficients instead of from zero)
    const T* e = p + kZernLevelStride * (K - 1);
    V2 qs = {e[0], e[1]}, qn = {e[2], e[3]}, dq = {e[4], e[5]};
    for (int k = K - 2; k >= 0; --k) {
      e -= kZernLevelStride;
      const V2 ak = {e[0], e[1]}, bk = {e[2], e[3]}, dk = {e[4], e[5]};
      qs = qs * uu + ak;
      qn = qn * uu + bk;
      dq = dq * uu + dk;
    }
    p += kZernLevelStride * K;
    const V2 zs = qs * H;
    zsum += zs.x + zs.y;
    const V2 t1 = dq * H;                     // dQn/dx_n = 2 x_n Qn'
    const T t1s = (t1.x + t1.y) * T(2);
    const V2 w = qn * T(mg);
    const V2 hx = w * Hp;                     // cos: m Qc A_{m-1}   sin: m Qs B_{m-1}
    const V2 swp = {-Hp.y, Hp.x};
    const V2 hy = w * swp;                    // cos: -m Qc B_{m-1}  sin: m Qs A_{m-1}
    gx = m::fma(t1s, xn, gx + (hx.x + hx.y));
    gy = m::fma(t1s, yn, gy + (hy.x + hy.y));
  }
  if (u < T(1e-8)) {  // the reference's eps-regularised chain rule near / at the vertex
    const T eps = m::guard();
    const T Rr = m::fma(xn, gx, yn * gy);     // rho dZ/drho
    const T Az = m::fma(xn, gy, -(yn * gx));  // dZ/dphi
    const T rho = m::sqrt(u);
    const T d1 = u > T(0) ? m::rcp(m::fma(eps, rho, u)) : T(0);  // 1 / (rho (rho + eps))
    const T d2 = m::rcp(u + eps);
    gx = m::fma(Rr * d1, xn, -(Az * d2 * yn));
    gy = m::fma(Rr * d1, yn, Az * d2 * xn);
  }
  sag += zsum;
  fx = m::fma(gx, inv, fx);
  fy = m::fma(gy, inv, fy);
}

// chebyshev.py:126-225.  T_n by the three-term recurrence instead of
// cos(n acos x); T_n'(x) = n U_{n-1}(x) instead of n sin(n acos x)/sqrt(1-x^2)
// (identical for |x| < 1; at |x| == 1 the reference divides by zero).  As in the
// reference the derivative is taken w.r.t. the NORMALISED coordinate and is not
// divided by norm_x / norm_y (chebyshev.py:176-186).
template <typename T>
__device__ __forceinline__ void chebyshev_eval(const DevSurf<T>& s, const T* __restrict__ c, T x,
                                               T y, T& sag, T& fx, T& fy, uint32_t& status) {
  using m = Math<T>;
  T r2 = m::fma(x, x, y * y);
  T g = m::sqrt(m::fma(-s.kp1 * s.cv * s.cv, r2, T(1)));
  sag = m::div(s.cv * r2, T(1) + g);
  T f = m::div(s.cv, g);
  fx = x * f;
  fy = y * f;
  const T xn = x * c[0], yn = y * c[1];
  if (m::abs(xn) > T(1) || m::abs(yn) > T(1)) status |= 0x4u;  // OL_STATUS_CHEBYSHEV_RANGE
  const int cols = s.cold->poly_cols;
  const int rows = cols > 0 ? s.n_coeff / cols : 0;
  const T* grid = c + 2;
  // Ti, Ui1 = T_i(xn), U_{i-1}(xn)
  T Ti = T(1), Tim = T(0), Ui1 = T(0), Ui2 = T(0);
  T S = T(0), Sx = T(0), Sy = T(0);
  for (int i = 0; i < rows; ++i) {
    // row polynomial in y: sum_j c_ij T_j(yn) and sum_j c_ij j U_{j-1}(yn)
    T Tj = T(1), Tjm = T(0), Uj1 = T(0), Uj2 = T(0);
    T q = T(0), dq = T(0);
    for (int j = 0; j < cols; ++j) {
      const T cij = grid[i * cols + j];
      q = m::fma(cij, Tj, q);
      dq = m::fma(cij * T(j), Uj1, dq);
      // advance: T_{j+1} = 2 y T_j - T_{j-1};  U_j = 2 y U_{j-1} - U_{j-2}
      const T Tn = j == 0 ? yn : m::fma(T(2) * yn, Tj, -Tjm);
      const T Un = j == 0 ? T(1) : m::fma(T(2) * yn, Uj1, -Uj2);
      Tjm = Tj; Tj = Tn; Uj2 = Uj1; Uj1 = Un;
    }
    S = m::fma(Ti, q, S);
    Sx = m::fma(T(i) * Ui1, q, Sx);
    Sy = m::fma(Ti, dq, Sy);
    const T Tn = i == 0 ? xn : m::fma(T(2) * xn, Ti, -Tim);
    const T Un = i == 0 ? T(1) : m::fma(T(2) * xn, Ui1, -Ui2);
    Tim = Ti; Ti = Tn; Ui2 = Ui1; Ui1 = Un;
  }
  sag += S;
  fx += Sx;
  fy += Sy;
}

// biconic.py:69-158: z = zx(x) + zy(y), each a conic profile; clamps kept.
template <typename T>
__device__ __forceinline__ void biconic_eval(const DevSurf<T>& s, const T* __restrict__ c, T x,
                                             T y, T& sag, T& fx, T& fy) {
  using m = Math<T>;
  const T cx = s.cv, kx1 = s.kp1, cy = c[0], ky1 = c[1];
  const T lim = m::guard();
  T zx = T(0), zy = T(0);
  fx = T(0);
  fy = T(0);
  if (cx != T(0)) {
    T v = m::fma(-kx1 * cx * cx, x * x, T(1));
    T st0 = v < lim ? T(0) : v;   // sag: clamp to 0
    T st1 = v < lim ? lim : v;    // gradient: clamp to 1e-14
    zx = m::div(cx * x * x, T(1) + m::sqrt(st0));
    fx = m::div(cx * x, m::sqrt(st1));
  }
  if (cy != T(0)) {
    T v = m::fma(-ky1 * cy * cy, y * y, T(1));
    T st0 = v < lim ? T(0) : v;
    T st1 = v < lim ? lim : v;
    zy = m::div(cy * y * y, T(1) + m::sqrt(st0));
    fy = m::div(cy * y, m::sqrt(st1));
  }
  sag = zx + zy;
}

// toroidal.py:86-242: Y-Z profile z_y(y) (conic + even polynomial) rotated about an
// axis parallel to Y at distance R_rot.  Invalid domain ((R - z_y)^2 < x^2): sag is
// NaN and the reference's normal is (0, 0, -1), i.e. zero gradient.
template <typename T>
__device__ __forceinline__ void toroidal_eval(const DevSurf<T>& s, const T* __restrict__ c, T x,
                                              T y, T& sag, T& fx, T& fy) {
  using m = Math<T>;
  const T R = c[0], invR = c[1], k1 = c[2], cyz = c[3];
  const T* a = c + 4;
  const T y2 = y * y;
  const T lim = m::guard();
  T zy = T(0), dzy = T(0);
  if (cyz != T(0)) {
    T v = m::fma(-k1 * cyz * cyz, y2, T(1));
    T r0 = v < T(0) ? T(0) : v;
    T r1 = v < lim ? lim : v;
    zy = m::div(cyz * y2, T(1) + m::sqrt(r0));
    dzy = m::div(cyz * y, m::sqrt(r1));
  }
  T p = T(0), dp = T(0);  // sum a_i y2^(i+1), sum 2(i+1) a_i y^(2i+1)
  for (int i = s.n_coeff - 1; i >= 0; --i) {
    dp = m::fma(dp, y2, T(2 * (i + 1)) * a[i]);
    p = m::fma(p, y2, a[i]);
  }
  zy = m::fma(p, y2, zy);
  dzy = m::fma(dp, y, dzy);
  if (invR == T(0)) {  // cylinder extruded along x
    sag = zy;
    fx = T(0);
    fy = dzy;
    return;
  }
  const T d = R - zy;
  const T term = m::fma(d, d, -x * x);
  const bool valid = term >= T(0);
  const T sq = m::sqrt(valid ? term : lim);
  const T ssq = m::abs(sq) < lim ? lim : sq;
  const T sgd = d > T(0) ? T(1) : (d < T(0) ? T(-1) : T(0));
  const T sgR = R > T(0) ? T(1) : T(-1);
  // z_y + (d - sign(d) sqrt(term)) = R - sign(d) sqrt(term)
  sag = valid ? zy + (d - sgd * sq) : (term < T(0) ? T(__builtin_nanf("")) : term);
  const T isq = m::rcp(ssq);
  fx = valid ? sgR * x * isq : T(0);
  fy = valid ? sgR * d * dzy * isq : T(0);
}

template <typename T>
__device__ __forceinline__ void nr_eval(const DevSurf<T>& s, const T* __restrict__ c, T x, T y,
                                        T& sag, T& fx, T& fy, uint32_t& status) {
  switch (s.geom) {
    case kGeomEvenAsphere: even_asphere_eval(s, c, x, y, sag, fx, fy); break;
    case kGeomOddAsphere: odd_asphere_eval(s, c, x, y, sag, fx, fy); break;
    case kGeomPolynomial: polynomial_eval(s, c, x, y, sag, fx, fy); break;
    case kGeomChebyshev: chebyshev_eval(s, c, x, y, sag, fx, fy, status); break;
    case kGeomBiconic: biconic_eval(s, c, x, y, sag, fx, fy); break;
    case kGeomToroidal: toroidal_eval(s, c, x, y, sag, fx, fy); break;
    default: zernike_eval(s, c, x, y, sag, fx, fy, status); break;
  }
}

// newton_raphson.py:119-168.  f(t) = sag(x(t), y(t)) - z(t); with the unit
// normal n = (fx, fy, -1)/|.| the reference's  -nx/nz, -ny/nz  are just (fx, fy),
// so f'(t) = fx L + fy M - N (guard on f' kept; |nz| > 1e-14 always holds for
// finite gradients).  Differences from the reference, all deliberate:
//  * re-based iteration: the ray is first moved to the conic/plane hit
//    P_b = P + t0 D and Newton runs on the small correction dt, so that
//    z_b + dt N does not cancel a metres-long path against itself (fp32 on
//    a telescope: ulp(6265 mm) = 5e-4 mm >> tol = 1e-6 mm);
//  * per-ray stop rule: the reference stops the whole batch when max |f| < tol
//    (newton_raphson.py:148); here a ray that sees |f| < tol takes ONE more
//    update and leaves (quadratic convergence => its residual is far below the
//    reference's own), NaN rays leave at once, and a ray whose residual no longer
//    halves (rounding floor reached -- fp32 with tol below the noise of sag - z)
//    leaves too instead of spinning to max_iter;
//  * the gradient of the LAST evaluation is returned and reused for the surface
//    normal: the hit point moved by |f|/|f'| < tol since, which changes the
//    normal by < curvature * tol.
// Returns t = t0 + dt and leaves the hit point in (x, y, z).
template <typename T>
struct NewtonRay {
  T xb, yb, zb, dt, fprev, gx, gy;
  bool active;
};

template <typename T>
__device__ __forceinline__ void newton_iterate(const DevSurf<T>& s, const T* __restrict__ c,
                                               NewtonRay<T>& q, T L, T M, T N, int it,
                                               uint32_t& status) {
  using m = Math<T>;
  T xi = m::fma(q.dt, L, q.xb), yi = m::fma(q.dt, M, q.yb), zi = m::fma(q.dt, N, q.zb);
  T sag, fx, fy;
  nr_eval(s, c, xi, yi, sag, fx, fy, status);
  T f = sag - zi;
  T af = m::abs(f);
  bool done = !(af >= s.cold->tol);                       // converged, or NaN
  done = done || (it > 0 && !(af < T(0.5) * q.fprev));  // residual stopped halving
  T df = m::fma(fx, L, m::fma(fy, M, -N));
  T dfs = m::abs(df) > m::guard() ? df : m::guard();
  q.dt = q.dt - m::div(f, dfs);
  q.fprev = af;
  q.gx = fx;
  q.gy = fy;
  q.active = !done;
}

// Wavefront straggler compaction (RPT > 1).  After the common iterations most
// rays of the wave's 64 x RPT pool are done, but the slot-by-slot loop still pays
// a full wave pass for every slot that holds ONE unfinished ray.  Here the
// stragglers are densely re-packed onto lanes: ballots give per-slot masks,
// v_mbcnt prefix counts give every unfinished ray a dense id, the executing lane
// finds its source (slot, lane) as the rank-th set bit of that slot's mask and
// pulls the ray's state with ds_bpermute (__shfl); ceil(total/64) passes iterate
// the packed rays to completion and the results are shuffled back.  No LDS
// allocation, no barriers; only used when every lane of the wave is alive.
__device__ __forceinline__ int nth_set_bit(uint64_t mask, int rank) {
  int pos = 0;
#pragma unroll
  for (int w = 32; w >= 1; w >>= 1) {
    const uint64_t low = (w == 32) ? 0xffffffffull : ((1ull << w) - 1ull);
    const int cnt = __popcll((mask >> pos) & low);
    if (rank >= cnt) {
      rank -= cnt;
      pos += w;
    }
  }
  return pos;
}

template <typename T, int RPT>
__device__ __forceinline__ void newton_compacted(const DevSurf<T>& s, const T* __restrict__ c,
                                                 NewtonRay<T> (&q)[RPT], const Ray<T> (&r)[RPT],
                                                 const uint64_t (&ballots)[RPT], int total,
                                                 int it_start, uint32_t& status) {
  const int lane = (int)__lane_id();
  int base[RPT], acc = 0;
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    base[k] = acc;
    acc += __popcll(ballots[k]);
  }
  const int passes = (total + 63) >> 6;
  for (int p = 0; p < passes; ++p) {
    const int id = p * 64 + lane;
    const bool have = id < total;
    int slot = 0, rank = 0;
    uint64_t mask = ballots[0];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      if (id >= base[k]) {  // last slot whose base <= id
        slot = k;
        rank = id - base[k];
        mask = ballots[k];
      }
    }
    const int src = have ? nth_set_bit(mask, rank) : lane;
    NewtonRay<T> g;
    T L = T(0), M = T(0), N = T(1);
    g.xb = g.yb = g.zb = g.dt = g.fprev = g.gx = g.gy = T(0);
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const T xb = __shfl(q[k].xb, src), yb = __shfl(q[k].yb, src), zb = __shfl(q[k].zb, src);
      const T dt = __shfl(q[k].dt, src), fp = __shfl(q[k].fprev, src);
      const T l = __shfl(r[k].L, src), mm = __shfl(r[k].M, src), n = __shfl(r[k].N, src);
      if (slot == k) {
        g.xb = xb; g.yb = yb; g.zb = zb; g.dt = dt; g.fprev = fp;
        L = l; M = mm; N = n;
      }
    }
    g.active = have;
    for (int it = it_start; it < s.max_iter; ++it) {
      if (!__any(g.active)) break;
      if (g.active) newton_iterate(s, c, g, L, M, N, it, status);
    }
    // hand the results back to the owning (lane, slot)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const int myid = base[k] + (int)__builtin_amdgcn_mbcnt_hi(
                                     (uint32_t)(ballots[k] >> 32),
                                     __builtin_amdgcn_mbcnt_lo((uint32_t)ballots[k], 0u));
      const int from = myid & 63;
      const T dt = __shfl(g.dt, from), gx = __shfl(g.gx, from), gy = __shfl(g.gy, from);
      if (q[k].active && (myid >> 6) == p) {
        q[k].dt = dt;
        q[k].gx = gx;
        q[k].gy = gy;
        q[k].active = false;
      }
    }
  }
}

// --------------------------------------------------------------------------
// apertures: physical_apertures/{radial,offset_radial,rectangular,elliptical}.py
// --------------------------------------------------------------------------
// physical_apertures/polygon.py:54-71: matplotlib's crossings test over the implicitly
// closed polygon (see oracle/trace_oracle.c:polygon_contains, checked against
// matplotlib itself); vertices x0, y0, x1, y1, ... in the coefficient buffer.
template <typename T>
__device__ __forceinline__ bool polygon_contains(const T* __restrict__ v, int nv, T tx, T ty) {
  if (!(tx - tx == T(0) && ty - ty == T(0))) return false;  // non-finite points are outside
  bool inside = false;
  T x0 = v[0], y0 = v[1];
  bool yflag0 = y0 >= ty;
#pragma nounroll  // rare path: keep it out of the register budget of every kernel
  for (int k = 1; k <= nv; ++k) {
    const int j = k == nv ? 0 : k;  // k == nv closes the polygon
    const T x1 = v[2 * j], y1 = v[2 * j + 1];
    const bool yflag1 = y1 >= ty;
    if (yflag0 != yflag1 && (((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == yflag1))
      inside = !inside;
    yflag0 = yflag1;
    x0 = x1;
    y0 = y1;
  }
  return inside;
}

// FULL: the polygon test is compiled only into the "full" (NR != 0) kernel variants --
// the host routes systems with polygon apertures there; in the lean conic-only kernels
// it cost 12 VGPRs (46 -> 58) for a case that almost never occurs
template <typename T, bool FULL>
__device__ __forceinline__ bool leaf_contains(int kind, const T* __restrict__ ap,
                                              const T* __restrict__ coeffs, T x, T y) {
  using m = Math<T>;
  if constexpr (FULL) {
    if (kind == kApPolygon) return polygon_contains<T>(coeffs + (int)ap[0], (int)ap[1], x, y);
  }
  switch (kind) {
    case kApRadial: {
      T r2 = m::fma(x, x, y * y);
      return (r2 <= ap[1]) && (r2 >= ap[0]);
    }
    case kApOffsetRadial: {
      T dx = x - ap[2], dy = y - ap[3];
      T r2 = m::fma(dx, dx, dy * dy);
      return (r2 <= ap[1]) && (r2 >= ap[0]);
    }
    case kApRect:
      return (ap[0] <= x) && (x <= ap[1]) && (ap[2] <= y) && (y <= ap[3]);
    case kApElliptical: {
      T dx = x - ap[2], dy = y - ap[3];
      return m::fma(dx * dx, ap[0], dy * dy * ap[1]) <= T(1);
    }
    default:
      return true;
  }
}

// Boolean trees (physical_apertures/base.py:259-340) arrive as reverse-Polish
// tokens; the evaluation stack is one bit per entry in a 32-bit register
// (depth <= 16 checked on the host).  Token stream and op codes are wave-uniform.
template <typename T, bool FULL>
__device__ __forceinline__ bool aperture_contains(const DevSurf<T>& s,
                                                  const T* __restrict__ coeffs, T x, T y) {
  if (s.aperture_kind != kApComposite)
    return leaf_contains<T, FULL>(s.aperture_kind, s.cold->ap, coeffs, x, y);
  const T* tok = coeffs + s.cold->ap_off;
  uint32_t stack = 0;  // bit 0 = top of stack
  for (int i = 0; i < s.cold->ap_len; ++i, tok += kApTokenLen) {
    const int op = (int)tok[0];
    if (op < kApOpUnion) {
      stack = (stack << 1) | (leaf_contains<T, FULL>(op, tok + 1, coeffs, x, y) ? 1u : 0u);
    } else {
      const uint32_t b = stack & 1u, a = (stack >> 1) & 1u;
      const uint32_t v = op == kApOpUnion ? (a | b) : (op == kApOpIntersection ? (a & b) : (a & ~b & 1u));
      stack = ((stack >> 2) << 1) | v;
    }
  }
  return (stack & 1u) != 0;
}

// --------------------------------------------------------------------------
// polarisation: rays/polarized_rays.py:136-202 with J = diag(j0, j1, j2)
// --------------------------------------------------------------------------
// The s-vector (normal to the plane of incidence) is formed as k0 x n instead of
// the reference's k0 x k1: both are parallel (k1 = u k0 + w n for refraction,
// k0 - 2 dot n for reflection; the sign cancels in O_out J O_in), but k0 x k1
// degenerates to rounding noise whenever the surface barely deviates the ray
// (image plane with n1 == n2, near-vertex rays), where the reference only works
// because numpy's un-fused arithmetic happens to return exact zeros.  One
// Gram-Schmidt step keeps s orthogonal to k0 to rounding, so the residual noise
// in its azimuth only couples through the Jones anisotropy |ts - tp| ~ aoi^2.
template <typename T>
struct PolBasis {
  T sx, sy, sz, p0x, p0y, p0z, p1x, p1y, p1z;
};

template <typename T>
__device__ __forceinline__ PolBasis<T> pol_basis(T k0x, T k0y, T k0z, T k1x, T k1y, T k1z, T nx,
                                                 T ny, T nz) {
  using m = Math<T>;
  T sx = k0y * nz - k0z * ny, sy = k0z * nx - k0x * nz, sz = k0x * ny - k0y * nx;
  {
    T proj = m::fma(sx, k0x, m::fma(sy, k0y, sz * k0z));
    sx = m::fma(-proj, k0x, sx);
    sy = m::fma(-proj, k0y, sy);
    sz = m::fma(-proj, k0z, sz);
  }
  T mag2 = m::fma(sx, sx, m::fma(sy, sy, sz * sz));
  if (mag2 == T(0)) {
    // normal incidence: polarized_rays.py:153-166 fallback axes
    // p_f = k0 x x_hat = (0, k0z, -k0y); if zero, k0 x y_hat = (-k0z, 0, k0x)
    T px = T(0), py = k0z, pz = -k0y;
    if (py == T(0) && pz == T(0)) {
      px = -k0z;
      py = T(0);
      pz = k0x;
    }
    // s = p_f x k0
    sx = py * k0z - pz * k0y;
    sy = pz * k0x - px * k0z;
    sz = px * k0y - py * k0x;
    mag2 = m::fma(sx, sx, m::fma(sy, sy, sz * sz));
  }
  T im = m::rsqrt(mag2);
  PolBasis<T> b;
  b.sx = sx * im;
  b.sy = sy * im;
  b.sz = sz * im;
  // p0 = k0 x s, p1 = k1 x s
  b.p0x = k0y * b.sz - k0z * b.sy; b.p0y = k0z * b.sx - k0x * b.sz; b.p0z = k0x * b.sy - k0y * b.sx;
  b.p1x = k1y * b.sz - k1z * b.sy; b.p1y = k1z * b.sx - k1x * b.sz; b.p1z = k1x * b.sy - k1y * b.sx;
  return b;
}

// jones.py:120-181 (polarizer: J = u_out u_in^T) and jones.py:331-393 (retarder:
// J = cos(d/2) I - i sin(d/2) (2 u u^T - I)), u = the axis projected on (s, p).
template <typename T>
__device__ __forceinline__ Jones<T> axis_jones(const PolBasis<T>& b, const T* __restrict__ axis,
                                               bool retarder, T rc, T rs) {
  using m = Math<T>;
  const T ax = axis[0], ay = axis[1], az = axis[2];
  T ts = ax * b.sx + ay * b.sy + az * b.sz;
  T tpi = ax * b.p0x + ay * b.p0y + az * b.p0z;
  T ni = m::sqrt(m::fma(ts, ts, tpi * tpi));
  ni = ni == T(0) ? T(1) : ni;
  const T usi = m::div(ts, ni), upi = m::div(tpi, ni);
  Jones<T> J;
  J.j22 = T(1);
  if (retarder) {
    // e^{-id/2} us^2 + e^{id/2} up^2 = c (us^2+up^2) - i s (us^2 - up^2)
    const T q0 = usi * usi, q1 = upi * upi, q01 = usi * upi;
    J.a00 = rc * (q0 + q1); J.b00 = -rs * (q0 - q1);
    J.a11 = rc * (q0 + q1); J.b11 = rs * (q0 - q1);
    J.a01 = J.a10 = T(0);
    J.b01 = J.b10 = T(-2) * rs * q01;
  } else {
    T tpo = ax * b.p1x + ay * b.p1y + az * b.p1z;
    T no = m::sqrt(m::fma(ts, ts, tpo * tpo));
    no = no == T(0) ? T(1) : no;
    const T uso = m::div(ts, no), upo = m::div(tpo, no);
    J.a00 = uso * usi; J.a01 = uso * upi; J.a10 = upo * usi; J.a11 = upo * upi;
    J.b00 = J.b01 = J.b10 = J.b11 = T(0);
  }
  return J;
}

// P <- O_out J O_in P  (polarized_rays.py:180-202); O_in rows (s, p0, k0), O_out
// columns (s, p1, k1).  POLK == 2 carries the imaginary part too.
template <typename T, int POLK>
__device__ __forceinline__ void prt_apply(Prt<T, POLK>& P, const PolBasis<T>& b, T k0x, T k0y,
                                          T k0z, T k1x, T k1y, T k1z, const Jones<T>& J) {
  constexpr int NP = POLK == 2 ? 2 : 1;
  T v0[NP][3], v1[NP][3], v2[NP][3];
  T w0[NP][3], w1[NP][3], w2[NP][3];
#pragma unroll
  for (int c = 0; c < NP; ++c)
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const T* Q = P.m + 9 * c;
      w0[c][e] = b.sx * Q[e] + b.sy * Q[3 + e] + b.sz * Q[6 + e];
      w1[c][e] = b.p0x * Q[e] + b.p0y * Q[3 + e] + b.p0z * Q[6 + e];
      w2[c][e] = k0x * Q[e] + k0y * Q[3 + e] + k0z * Q[6 + e];
    }
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    v0[0][e] = J.a00 * w0[0][e] + J.a01 * w1[0][e];
    v1[0][e] = J.a10 * w0[0][e] + J.a11 * w1[0][e];
    v2[0][e] = J.j22 * w2[0][e];
    if constexpr (POLK == 2) {
      v0[0][e] -= J.b00 * w0[1][e] + J.b01 * w1[1][e];
      v1[0][e] -= J.b10 * w0[1][e] + J.b11 * w1[1][e];
      v0[1][e] = J.a00 * w0[1][e] + J.a01 * w1[1][e] + J.b00 * w0[0][e] + J.b01 * w1[0][e];
      v1[1][e] = J.a10 * w0[1][e] + J.a11 * w1[1][e] + J.b10 * w0[0][e] + J.b11 * w1[0][e];
      v2[1][e] = J.j22 * w2[1][e];
    }
  }
#pragma unroll
  for (int c = 0; c < NP; ++c)
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      T* Q = P.m + 9 * c;
      Q[e] = b.sx * v0[c][e] + b.p1x * v1[c][e] + k1x * v2[c][e];
      Q[3 + e] = b.sy * v0[c][e] + b.p1y * v1[c][e] + k1y * v2[c][e];
      Q[6 + e] = b.sz * v0[c][e] + b.p1z * v1[c][e] + k1z * v2[c][e];
    }
}

// The same update for a REAL DIAGONAL Jones matrix diag(j0, j1, j2) -- uncoated and
// Fresnel-coated surfaces (jones.py:71-117), i.e. every surface of a system without
// polarizers / retarders.  {s, p0, k0} is an orthonormal triad (s is Gram-Schmidt-ed
// against k0 in pol_basis), so s s^T = I - p0 p0^T - k0 k0^T and
//     O_out J O_in = j0 s s^T + j1 p1 p0^T + j2 k1 k0^T
//                  = j0 I + (j1 p1 - j0 p0) p0^T + (j2 k1 - j0 k0) k0^T :
// P' = j0 P + a (p0^T P) + b (k0^T P), two row-vector products and a rank-2 update
// (54 multiply-adds) instead of three products and a full recombination (75); s itself
// is only needed to build p0 and p1.  Equal to the general form up to the rounding of
// |k0|^2 - 1 (the reference never renormalises k either).
template <typename T, int POLK>
__device__ __forceinline__ void prt_apply_diag(Prt<T, POLK>& P, const PolBasis<T>& b, T k0x,
                                               T k0y, T k0z, T k1x, T k1y, T k1z, T j0, T j1,
                                               T j2) {
  using m = Math<T>;
  constexpr int NP = POLK == 2 ? 2 : 1;
  const T ax = m::fma(j1, b.p1x, -(j0 * b.p0x)), ay = m::fma(j1, b.p1y, -(j0 * b.p0y)),
          az = m::fma(j1, b.p1z, -(j0 * b.p0z));
  const T bx = m::fma(j2, k1x, -(j0 * k0x)), by = m::fma(j2, k1y, -(j0 * k0y)),
          bz = m::fma(j2, k1z, -(j0 * k0z));
#pragma unroll
  for (int c = 0; c < NP; ++c) {
    T* Q = P.m + 9 * c;
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const T r1 = m::fma(b.p0x, Q[e], m::fma(b.p0y, Q[3 + e], b.p0z * Q[6 + e]));
      const T r2 = m::fma(k0x, Q[e], m::fma(k0y, Q[3 + e], k0z * Q[6 + e]));
      Q[e] = m::fma(ax, r1, m::fma(bx, r2, j0 * Q[e]));
      Q[3 + e] = m::fma(ay, r1, m::fma(by, r2, j0 * Q[3 + e]));
      Q[6 + e] = m::fma(az, r1, m::fma(bz, r2, j0 * Q[6 + e]));
    }
  }
}

// First update of a FRESH matrix (P = I, OL_TRACE_PRT_IDENTITY): P' = O_out J O_in itself,
// 21 multiply-adds instead of 54.
template <typename T, int POLK>
__device__ __forceinline__ void prt_first_diag(Prt<T, POLK>& P, const PolBasis<T>& b, T k0x,
                                               T k0y, T k0z, T k1x, T k1y, T k1z, T j0, T j1,
                                               T j2) {
  using m = Math<T>;
  const T a[3] = {m::fma(j1, b.p1x, -(j0 * b.p0x)), m::fma(j1, b.p1y, -(j0 * b.p0y)),
                  m::fma(j1, b.p1z, -(j0 * b.p0z))};
  const T bb[3] = {m::fma(j2, k1x, -(j0 * k0x)), m::fma(j2, k1y, -(j0 * k0y)),
                   m::fma(j2, k1z, -(j0 * k0z))};
  const T p0[3] = {b.p0x, b.p0y, b.p0z}, k0[3] = {k0x, k0y, k0z};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 3; ++e)
      P.m[3 * i + e] = m::fma(a[i], p0[e], m::fma(bb[i], k0[e], i == e ? j0 : T(0)));
  if constexpr (POLK == 2) {
#pragma unroll
    for (int e = 9; e < 18; ++e) P.m[e] = P.m[e] * T(0) + (P.m[0] * T(0));  // 0, NaN kept
  }
}

// --------------------------------------------------------------------------
// one surface for the RPT rays of a thread: standard_surface.py:200-248 (minus
// record).  Phases run across the thread's rays so that independent chains
// interleave (ILP) and the Newton loop can look at all of them together.
// --------------------------------------------------------------------------
// Uniform (per-surface) branches are hoisted OUTSIDE the per-ray loops everywhere
// below: each branch body is then one basic block holding the arithmetic of all
// RPT rays, which is what lets their independent dependency chains interleave.
template <typename V, int RPT>
__device__ __forceinline__ void into_local_frame(const DevSurf<typename Math<V>::scalar>& s,
                                                 bool from_global, Ray<V> (&r)[RPT]) {
  using m = Math<V>;
  using T = typename m::scalar;
  // coordinate_system.py:73-89
  if (from_global) {
    const T ox = s.origin[0], oy = s.origin[1], oz = s.origin[2];
    if (s.flags & kSurfRotated) {
      const T* R = s.cold->rot;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        V x = r[k].x - ox, y = r[k].y - oy, z = r[k].z - oz;
        V L = r[k].L, M = r[k].M, N = r[k].N;
        r[k].x = R[0] * x + R[1] * y + R[2] * z;
        r[k].y = R[3] * x + R[4] * y + R[5] * z;
        r[k].z = R[6] * x + R[7] * y + R[8] * z;
        r[k].L = R[0] * L + R[1] * M + R[2] * N;
        r[k].M = R[3] * L + R[4] * M + R[5] * N;
        r[k].N = R[6] * L + R[7] * M + R[8] * N;
      }
    } else {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        r[k].x -= ox;
        r[k].y -= oy;
        r[k].z -= oz;
      }
    }
  } else if (s.flags & kSurfRelRotated) {
    const T* R = s.cold->rel_rot;
    const T ox = s.rel_off[0], oy = s.rel_off[1], oz = s.rel_off[2];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      V x = r[k].x, y = r[k].y, z = r[k].z, L = r[k].L, M = r[k].M, N = r[k].N;
      r[k].x = m::fma(m::splat(R[0]), x, m::fma(m::splat(R[1]), y, m::fma(m::splat(R[2]), z, m::splat(ox))));
      r[k].y = m::fma(m::splat(R[3]), x, m::fma(m::splat(R[4]), y, m::fma(m::splat(R[5]), z, m::splat(oy))));
      r[k].z = m::fma(m::splat(R[6]), x, m::fma(m::splat(R[7]), y, m::fma(m::splat(R[8]), z, m::splat(oz))));
      r[k].L = R[0] * L + R[1] * M + R[2] * N;
      r[k].M = R[3] * L + R[4] * M + R[5] * N;
      r[k].N = R[6] * L + R[7] * M + R[8] * N;
    }
  } else {
    const T ox = s.rel_off[0], oy = s.rel_off[1], oz = s.rel_off[2];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      r[k].x += ox;
      r[k].y += oy;
      r[k].z += oz;
    }
  }
}

// everything after the hit point is known: absorb, opd, clip, refract/reflect,
// coating, PRT.  (nx, ny, nz) is the unit surface normal at the hit.
// lane-wise aperture test: the scalar predicate per ray of the pack
template <typename V, bool FULL>
__device__ __forceinline__ typename Math<V>::mask aperture_mask(
    const DevSurf<typename Math<V>::scalar>& s, const typename Math<V>::scalar* __restrict__ coeffs,
    V x, V y) {
  if constexpr (Math<V>::lanes == 1) {
    return aperture_contains<typename Math<V>::scalar, FULL>(s, coeffs, x, y);
  } else {
    return {aperture_contains<typename Math<V>::scalar, FULL>(s, coeffs, x.x, y.x),
            aperture_contains<typename Math<V>::scalar, FULL>(s, coeffs, x.y, y.y)};
  }
}

template <typename V, int RPT, int POLK, bool FULL>
__device__ __forceinline__ void interact(const DevSurf<typename Math<V>::scalar>& s,
                                         const DevOptics<typename Math<V>::scalar>& o,
                                         const typename Math<V>::scalar* __restrict__ coeffs,
                                         const V (&t)[RPT], const V (&nx)[RPT], const V (&ny)[RPT],
                                         const V (&nz)[RPT], Ray<V> (&r)[RPT],
                                         Prt<typename Math<V>::scalar, POLK> (&P)[POLK ? RPT : 1],
                                         bool& prt_fresh) {
  // prt_fresh (wave-uniform): the matrices still hold the identity a fresh trace starts
  // from -- the first real update then writes O_out J O_in instead of multiplying by it
  using m = Math<V>;
  using T = typename m::scalar;
  static_assert(POLK == 0 || m::lanes == 1, "the polarised path is scalar");
  const V zero = m::splat(0), one = m::splat(1);
  // homogeneous.py:44-53, standard_surface.py:244
  if (o.absorb > T(0)) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      // i * exp(-alpha t).  A clipped ray (i = 0) that then runs a long NEGATIVE distance
      // through an absorbing medium has exp(+x) overflow fp32 (x > 88.7) long before it
      // overflows the reference's fp64 (x > 709.78): keep the reference's 0 * finite = 0
      // there instead of fp32's 0 * inf = NaN.
      const V arg = -o.absorb * t[k];
      const V prod = r[k].i * m::exp(arg);
      r[k].i = m::select(m::mand(m::eq(r[k].i, zero), m::lt(arg, m::splat(T(709.78)))), zero,
                         prod);
    }
  }
#pragma unroll
  for (int k = 0; k < RPT; ++k) r[k].opd = r[k].opd + m::abs(t[k] * o.n1);

  // clip (physical_apertures/base.py:71-82, real_rays.py:154-161)
  if (s.aperture_kind != kApNone) {
#pragma unroll
    for (int k = 0; k < RPT; ++k)
      r[k].i = m::select(aperture_mask<V, FULL>(s, coeffs, r[k].x, r[k].y), r[k].i, zero);
  }

  // refract / reflect (real_rays.py:163-205, 535-571)
  V L0[RPT], M0[RPT], N0[RPT], adot[RPT], ax[RPT], ay[RPT], az[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    L0[k] = r[k].L;
    M0[k] = r[k].M;
    N0[k] = r[k].N;
    V dot = m::fma(L0[k], nx[k], m::fma(M0[k], ny[k], N0[k] * nz[k]));
    // be.sign(dot): +-1, and 0 at 0.  (A NaN dot still poisons the new direction
    // through adot below, whatever sign it is given here.)
    const V sgn = m::select(m::ne(dot, zero), m::copysign(one, dot), zero);
    ax[k] = nx[k] * sgn;
    ay[k] = ny[k] * sgn;
    az[k] = nz[k] * sgn;
    adot[k] = m::abs(dot);
  }
  if (s.interaction == kReflect) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      V k2 = m::splat(-2) * adot[k];
      r[k].L = m::fma(k2, ax[k], L0[k]);
      r[k].M = m::fma(k2, ay[k], M0[k]);
      r[k].N = m::fma(k2, az[k], N0[k]);
    }
  } else {
    const T u = o.u;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      V root = m::sqrt(m::fma(m::splat(-u * u), m::fma(-adot[k], adot[k], one), one));  // NaN on TIR
      V w = m::fma(m::splat(-u), adot[k], root);
      r[k].L = m::fma(m::splat(u), L0[k], ax[k] * w);
      r[k].M = m::fma(m::splat(u), M0[k], ay[k] * w);
      r[k].N = m::fma(m::splat(u), N0[k], az[k] * w);
    }
  }

  // coating (interactions/base.py:111-128)
  if (s.coating_kind == kCoatSimple) {
    const T f = s.interaction == kReflect ? s.cold->coat[1] : s.cold->coat[0];
#pragma unroll
    for (int k = 0; k < RPT; ++k) r[k].i = r[k].i * f;
  }
  if constexpr (POLK != 0) {
    const int ck = s.coating_kind;
    const bool reflect = s.interaction == kReflect;
    const T nn = o.nn;
    // SimpleCoating.reflect / transmit only scale the intensity (coatings.py:199-237): they
    // never call rays.update(), so the PRT matrix passes through unchanged -- unlike an
    // UNCOATED surface, whose interaction model calls rays.update() with the identity
    // Jones matrix (interactions/base.py:124-125).
    if (ck == kCoatSimple) return;
    // An uncoated refracting surface between equal indices (every image plane, dummy
    // surfaces) leaves the direction unchanged (u = 1 => k1 = k0), its Jones matrix is
    // the identity and O_out O_in = I for ANY orthonormal basis: P' = P.  The
    // reference still multiplies it out (and its s = k0 x k1 there is rounding noise);
    // here the update is skipped and only the NaN state of a lost ray is carried into
    // the matrix, as the reference's product would.
    if (ck == kCoatNone && !reflect && o.u == T(1)) {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const T poison = (r[k].L + r[k].M + r[k].N) * T(0);  // 0, or NaN for a lost ray
#pragma unroll
        for (int e = 0; e < (POLK == 2 ? 18 : 9); ++e) P[k].m[e] += poison;
      }
      return;
    }
    if (ck == kCoatFresnel || ck == kCoatNone) {
      // real diagonal Jones matrix: rank-2 form of the update (prt_apply_diag)
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const PolBasis<T> b = pol_basis(L0[k], M0[k], N0[k], r[k].L, r[k].M, r[k].N, nx[k],
                                        ny[k], nz[k]);
        T j0 = T(1), j1 = T(1), j2 = T(1);
        if (ck == kCoatFresnel) {
          // coatings.py:72-92 + jones.py:71-117 with cos(aoi) = min(|n.k0|, 1):
          // root = sqrt(nn^2 - sin^2) is real unless TIR, where k1 is NaN already.
          T ci = adot[k] < T(1) ? adot[k] : (adot[k] >= T(1) ? T(1) : adot[k]);
          T root = m::sqrt(m::fma(nn, nn, m::fma(ci, ci, T(-1))));
          if (reflect) {
            j0 = m::div(ci - root, ci + root);
            j1 = -m::div(m::fma(nn * nn, ci, -root), m::fma(nn * nn, ci, root));
            j2 = T(-1);
          } else {
            j0 = m::div(T(2) * ci, ci + root);
            j1 = m::div(T(2) * nn * ci, m::fma(nn * nn, ci, root));
          }
        }
        if (prt_fresh)
          prt_first_diag<T, POLK>(P[k], b, L0[k], M0[k], N0[k], r[k].L, r[k].M, r[k].N, j0, j1,
                                  j2);
        else
          prt_apply_diag<T, POLK>(P[k], b, L0[k], M0[k], N0[k], r[k].L, r[k].M, r[k].N, j0, j1,
                                  j2);
      }
      prt_fresh = false;
      return;
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const PolBasis<T> b = pol_basis(L0[k], M0[k], N0[k], r[k].L, r[k].M, r[k].N, nx[k], ny[k],
                                      nz[k]);
      Jones<T> J;
      if (ck == kCoatPolarizer) {
        J = axis_jones(b, s.cold->axis, false, T(0), T(0));
      } else {  // kCoatRetarder
        J = axis_jones(b, s.cold->axis, true, s.cold->ret_cos, s.cold->ret_sin);
      }
      prt_apply<T, POLK>(P[k], b, L0[k], M0[k], N0[k], r[k].L, r[k].M, r[k].N, J);
    }
    prt_fresh = false;
  }
}

// NR: 0 = the surface range holds no Newton-Raphson geometry (lean kernel: none of
// that code, or its registers, is compiled in), 1 = Newton loop, 2 = Newton loop
// with wavefront straggler compaction.
template <typename V, int RPT, int POLK, int NR>
__device__ __forceinline__ void surface_step(const DevSurf<typename Math<V>::scalar>& s,
                                             const DevOptics<typename Math<V>::scalar>& o,
                                             const typename Math<V>::scalar* __restrict__ coeffs,
                                             bool from_global, Ray<V> (&r)[RPT],
                                             Prt<typename Math<V>::scalar, POLK> (&P)[POLK ? RPT : 1],
                                             uint32_t& status, bool& prt_fresh) {
  using m = Math<V>;
  using T = typename m::scalar;
  static_assert(NR == 0 || m::lanes == 1, "the Newton-Raphson path is scalar");
  into_local_frame<V, RPT>(s, from_global, r);

  const T* c = coeffs + s.coeff_off;
  V t[RPT], nx[RPT], ny[RPT], nz[RPT];  // distance, unit normal at the hit
  if (s.geom == kGeomPlane) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      t[k] = -m::div(r[k].z, r[k].N);  // plane.py:72-88
      nx[k] = ny[k] = m::splat(0);
      nz[k] = m::splat(1);  // plane.py:90-109
      r[k].x = m::fma(t[k], r[k].L, r[k].x);
      r[k].y = m::fma(t[k], r[k].M, r[k].y);
      r[k].z = m::fma(t[k], r[k].N, r[k].z);
    }
  } else if (s.geom == kGeomStandard) {
    if (s.flags & kSurfRadiusInf) {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        t[k] = flat_distance<V>(r[k].z, r[k].N);
        r[k].x = m::fma(t[k], r[k].L, r[k].x);
        r[k].y = m::fma(t[k], r[k].M, r[k].y);
        r[k].z = m::fma(t[k], r[k].N, r[k].z);
        nx[k] = ny[k] = m::splat(0);  // the conic normal with cv = 0
        nz[k] = m::splat(-1);
      }
    } else {
      const T cv = s.cv, kp1 = s.kp1;
#pragma unroll
      for (int k = 0; k < RPT; ++k)
        t[k] = curved_distance<V>(cv, kp1, r[k].x, r[k].y, r[k].z, r[k].L, r[k].M, r[k].N);
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        r[k].x = m::fma(t[k], r[k].L, r[k].x);
        r[k].y = m::fma(t[k], r[k].M, r[k].y);
        r[k].z = m::fma(t[k], r[k].N, r[k].z);
        conic_normal<V>(cv, kp1, r[k].x, r[k].y, r[k].z, nx[k], ny[k], nz[k]);
      }
    }
  } else if constexpr (NR != 0) {
    constexpr bool COMPACT = NR == 2;
    NewtonRay<T> q[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      t[k] = conic_distance(s, r[k].x, r[k].y, r[k].z, r[k].L, r[k].M, r[k].N);
      q[k].xb = m::fma(t[k], r[k].L, r[k].x);
      q[k].yb = m::fma(t[k], r[k].M, r[k].y);
      q[k].zb = m::fma(t[k], r[k].N, r[k].z);
      q[k].dt = T(0);
      q[k].fprev = T(0);
      q[k].gx = q[k].gy = T(0);
      q[k].active = true;
    }
    int it = 0;
    const bool can_compact = COMPACT && RPT > 1 && __popcll(__ballot(true)) == 64;
    for (; it < s.max_iter; ++it) {
      bool any = false;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        if (q[k].active) newton_iterate(s, c, q[k], r[k].L, r[k].M, r[k].N, it, status);
        any = any || q[k].active;
      }
      if constexpr (COMPACT && RPT > 1) {
        uint64_t ballots[RPT];
        int total = 0, nonempty = 0;
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
          ballots[k] = __ballot(q[k].active);
          total += __popcll(ballots[k]);
          nonempty += ballots[k] != 0;
        }
        if (total == 0) {
          ++it;
          break;
        }
        if (can_compact && ((total + 63) >> 6) < nonempty) {
          newton_compacted<T, RPT>(s, c, q, r, ballots, total, it + 1, status);
          ++it;
          break;
        }
      } else {
        if (!__any(any)) {
          ++it;
          break;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      r[k].x = m::fma(q[k].dt, r[k].L, q[k].xb);
      r[k].y = m::fma(q[k].dt, r[k].M, q[k].yb);
      r[k].z = m::fma(q[k].dt, r[k].N, q[k].zb);
      t[k] = t[k] + q[k].dt;
    }
    if (it == 0) {  // max_iter == 0: no evaluation happened, take the gradient here
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        T sag;
        uint32_t st = 0;
        nr_eval(s, c, r[k].x, r[k].y, sag, q[k].gx, q[k].gy, st);
      }
    }
    // n = (fx, fy, -1) / |.| from the sag gradient (newton_raphson.py:80-98)
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const T im = m::rsqrt(m::fma(q[k].gx, q[k].gx, m::fma(q[k].gy, q[k].gy, T(1))));
      nx[k] = q[k].gx * im;
      ny[k] = q[k].gy * im;
      nz[k] = -im;
    }
  } else {
    // unreachable: the host only selects NR == 0 for ranges without such surfaces
#pragma unroll
    for (int k = 0; k < RPT; ++k) t[k] = nx[k] = ny[k] = nz[k] = m::splat(0);
  }
  interact<V, RPT, POLK, NR != 0>(s, o, coeffs, t, nx, ny, nz, r, P, prt_fresh);
}

// local -> global for the recorded state (coordinate_system.py:91-107)
template <typename V>
__device__ __forceinline__ Ray<V> to_global(const DevSurf<typename Math<V>::scalar>& s,
                                            const Ray<V>& r) {
  using T = typename Math<V>::scalar;
  Ray<V> g = r;
  if (s.flags & kSurfRotated) {
    const T* R = s.cold->rot;  // inverse = transpose
    g.x = R[0] * r.x + R[3] * r.y + R[6] * r.z;
    g.y = R[1] * r.x + R[4] * r.y + R[7] * r.z;
    g.z = R[2] * r.x + R[5] * r.y + R[8] * r.z;
    g.L = R[0] * r.L + R[3] * r.M + R[6] * r.N;
    g.M = R[1] * r.L + R[4] * r.M + R[7] * r.N;
    g.N = R[2] * r.L + R[5] * r.M + R[8] * r.N;
  }
  g.x += s.origin[0];
  g.y += s.origin[1];
  g.z += s.origin[2];
  return g;
}

// The lean fp32 configuration -- four rays per lane, conic-only range, no
// polarisation -- runs on packed pairs: two f32x2 rays instead of four scalar ones.
// Everything else keeps V = T.
template <typename T, int RPT, int POLK, int NR>
struct LanePack {
#ifndef OL_PACKED_F32
#define OL_PACKED_F32 1
#endif
  static constexpr bool packed =
      OL_PACKED_F32 && sizeof(T) == 4 && (RPT == 4 || RPT == 2) && POLK == 0 && NR == 0;
  using V = typename std::conditional<packed, f32x2, T>::type;
  static constexpr int NV = packed ? RPT / 2 : RPT;
  // ray k of the thread: element (k % lanes) of pack (k / lanes)
  static __device__ __forceinline__ T get(const V& v, int e) {
    if constexpr (packed) return v[e]; else return v;
  }
  static __device__ __forceinline__ void set(V& v, int e, T x) {
    if constexpr (packed) v[e] = x; else v = x;
  }
  static __device__ __forceinline__ Ray<T> ray(const Ray<V> (&r)[NV], int k) {
    constexpr int L = packed ? 2 : 1;
    const Ray<V>& p = r[k / L];
    const int e = k % L;
    Ray<T> o;
    o.x = get(p.x, e); o.y = get(p.y, e); o.z = get(p.z, e);
    o.L = get(p.L, e); o.M = get(p.M, e); o.N = get(p.N, e);
    o.i = get(p.i, e); o.opd = get(p.opd, e);
    return o;
  }
  static __device__ __forceinline__ void put(Ray<V> (&r)[NV], int k, const Ray<T>& o) {
    constexpr int L = packed ? 2 : 1;
    Ray<V>& p = r[k / L];
    const int e = k % L;
    set(p.x, e, o.x); set(p.y, e, o.y); set(p.z, e, o.z);
    set(p.L, e, o.L); set(p.M, e, o.M); set(p.N, e, o.N);
    set(p.i, e, o.i); set(p.opd, e, o.opd);
  }
};

// --------------------------------------------------------------------------
// vector load / store of RPT consecutive rays of one plane
// --------------------------------------------------------------------------
// Store flavour, measured on MI355X with the 2 MiB-aligned record block
// (tools/microbench/rw_scope.hip, stream_write.hip): for ONE ray per lane (4/8-byte
// stores) non-temporal stores are ~2 % faster than plain ones (0.777 vs 0.792 ms on
// the record-all pattern; agent/system-scope write-through stores 0.795); for the
// 16-byte vector layout plain stores win by 1-2 %.
#ifndef OL_NT_SCALAR
#define OL_NT_SCALAR 1
#endif
#ifndef OL_NT_VECTOR
#define OL_NT_VECTOR 0
#endif

template <typename T, int RPT>
struct VecOf {
  typedef T type __attribute__((ext_vector_type(RPT)));
};

template <typename T, int RPT>
__device__ __forceinline__ T vec_get(const typename VecOf<T, RPT>::type& v, int k) {
  return v[k];  // ext_vector_type(1) is still a vector
}

// Index of a lane's first ray, split into the workgroup-uniform tile start and the 32-bit
// offset of the lane inside the tile.  With SADDR, `at(p)` = (p + tile) + lane lets the
// compiler address every plane access as SGPR base + 32-bit VGPR offset
// (global_store_dword v_off, v, s[b:b+1]) -- ONE lane offset shared by all planes --
// instead of forming a 64-bit per-lane address with a v_lshl_add_u64 per load / store.
// 33 fewer vector instructions per ray in the polarised Newton kernel (76 -> 73 VGPRs: it
// then reaches 7 waves without a spill), 46 -> 30 VGPRs in the lean record-all kernel.
// With the index split like this the compiler picks the SGPR-base form for both values
// of SADDR (checked in the ISA); the flag is kept as the A/B handle.
#ifndef OL_SADDR
#define OL_SADDR 1
#endif
template <bool SADDR>
struct RayIndexT {
  int64_t tile;   // wave-uniform
  uint32_t lane;  // < kTraceBlock * RPT (or the re-traced last ray of the SPOT variant)
  __device__ __forceinline__ int64_t full() const { return tile + (int64_t)lane; }
  template <typename P>
  __device__ __forceinline__ P* at(P* p) const {
    if constexpr (SADDR)
      return (p + tile) + lane;
    else
      return p + (tile + (int64_t)lane);
  }
};

template <typename T, int RPT, typename RayIndex>
__device__ __forceinline__ void store_plane(T* __restrict__ p, RayIndex base, int cnt,
                                            const T (&in)[RPT]) {
  T* q = base.at(p);
  if constexpr (RPT == 1) {
#if OL_NT_SCALAR
    __builtin_nontemporal_store(in[0], q);
#else
    *q = in[0];
#endif
  } else {
    using V = typename VecOf<T, RPT>::type;
    if (cnt == RPT) {
      V v;
#pragma unroll
      for (int k = 0; k < RPT; ++k) v[k] = in[k];
      // 8-byte lane vectors behave like the scalar fp64 stores (non-temporal wins);
      // 16-byte ones prefer plain stores
      if constexpr (OL_NT_VECTOR || sizeof(V) <= 8)
        __builtin_nontemporal_store(v, reinterpret_cast<V*>(q));
      else
        *reinterpret_cast<V*>(q) = v;
    } else {
#pragma unroll
      for (int k = 0; k < RPT; ++k)
        if (k < cnt) q[k] = in[k];
    }
  }
}

template <typename T, int RPT, typename RayIndex>
__device__ __forceinline__ void store_rays(T* __restrict__ row, int64_t stride, RayIndex base,
                                           int cnt, const Ray<T> (&g)[RPT]) {
  T tmp[RPT];
#define OL_STORE_FIELD(idx, fld)                         \
  _Pragma("unroll") for (int k = 0; k < RPT; ++k) tmp[k] = g[k].fld; \
  store_plane<T, RPT>(row + (int64_t)(idx) * stride, base, cnt, tmp);
  OL_STORE_FIELD(0, x)
  OL_STORE_FIELD(1, y)
  OL_STORE_FIELD(2, z)
  OL_STORE_FIELD(3, L)
  OL_STORE_FIELD(4, M)
  OL_STORE_FIELD(5, N)
  OL_STORE_FIELD(6, i)
  OL_STORE_FIELD(7, opd)
#undef OL_STORE_FIELD
}

// --------------------------------------------------------------------------
// image-plane spot moments (analysis/spot_diagram/core.py:329-372, mask :470-476):
// shared by the fused spot kernel and the optional epilogue of trace_kernel
// --------------------------------------------------------------------------
__device__ __forceinline__ double spot_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
__device__ __forceinline__ double spot_wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double o = __shfl_down(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

struct SpotAcc {
  double s[6] = {0, 0, 0, 0, 0, 0};
  double rmax = 0.0;
  template <typename T>
  __device__ __forceinline__ void add(T x, T y, T i, double cx, double cy) {
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
      rmax = r2 > rmax ? r2 : rmax;  // NaN hits compare false and are skipped
    }
  }
  // workgroup reduction -> 7 atomics into out[0..6]; EVERY thread of the workgroup
  // must call it (barrier inside)
  __device__ __forceinline__ void flush(double* __restrict__ out) {
    __shared__ double part[kTraceBlock / 64][7];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave reduction, STEP-major: the seven values move together, so each of the six
    // steps has 14 independent ds_bpermute in flight and one wait -- chain-major (one
    // value after the other) was 42 serialised LDS-crossbar round trips per wave and
    // cost the record-all kernel 20 % when this ran as its epilogue
    double v[7] = {s[0], s[1], s[2], s[3], s[4], s[5], rmax};
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      double o[7];
#pragma unroll
      for (int k = 0; k < 7; ++k) o[k] = __shfl_down(v[k], off, 64);
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] += o[k];
      v[6] = o[6] > v[6] ? o[6] : v[6];
    }
    if (lane == 0) {
#pragma unroll
      for (int k = 0; k < 7; ++k) part[wave][k] = v[k];
    }
    // Workgroup barrier that orders LDS only.  __syncthreads() is a release/acquire
    // fence over ALL address spaces: in the epilogue of the record-all kernel it made
    // every wave drain its ~100 outstanding record stores (s_waitcnt vmcnt(0)) before
    // the barrier instead of retiring with them in flight -- 0.78 -> 0.94 ms.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    if (threadIdx.x < 6) {
      double v = 0;
      for (int w = 0; w < kTraceBlock / 64; ++w) v += part[w][threadIdx.x];
      // hardware fp64 atomic add (the output lives in ordinary coarse-grained HBM)
      if (v != 0.0) unsafeAtomicAdd(&out[threadIdx.x], v);
    } else if (threadIdx.x == 6) {
      double v = part[0][6];
      for (int w = 1; w < kTraceBlock / 64; ++w) v = part[w][6] > v ? part[w][6] : v;
      // non-negative doubles order like their bit patterns
      if (v > 0.0)
        atomicMax(reinterpret_cast<unsigned long long*>(&out[6]),
                  (unsigned long long)__double_as_longlong(v));
    }
  }
};

// --------------------------------------------------------------------------
// the kernel
// --------------------------------------------------------------------------
// The three table pointers are separate `const __restrict__` kernel arguments on
// purpose: only then can the compiler prove that the record stores never clobber
// the table and fetch it with scalar loads (s_load_dwordx*, lgkmcnt).  As fields of
// the by-value argument struct they were fetched with per-lane global_load_dword +
// v_readfirstlane, and -- vmcnt being in-order on gfx9-family parts -- every
// surface's table read then waited for ALL outstanding record stores to retire,
// serialising compute behind HBM writes (measured: 1.00 ms -> see DESIGN.md).
// Minimum waves per SIMD asked of the register allocator.  Only the fp32 polarised
// Newton kernel (the Zernike + Fresnel configuration, VALU-issue bound with long SMEM /
// transcendental dependency chains) asks for more than the allocator gives by itself:
// it allocates 73 VGPRs unasked (6 waves); asked for 7 waves it fits 71 WITHOUT a spill
// and runs 1.5-2 % faster (profiles/r02_ab_zf_occupancy.txt, last block); 8 waves (64 VGPRs)
// spills 10 dwords and loses 30 %.  OL_POLNR_WAVES is the A/B knob
// (tools/build_variants.py), 0 = no request.
#ifndef OL_POLNR_WAVES
#define OL_POLNR_WAVES 7
#endif
template <typename T, int RPT, int POLK, int NR>
struct WavesPerEu {
  static constexpr int value =
      (OL_POLNR_WAVES > 0 && sizeof(T) == 4 && RPT == 1 && POLK == 1 && NR == 1) ? OL_POLNR_WAVES
                                                                                  : 1;
};

template <typename T, int RPT, bool RECORD, int POLK, int NR, bool SPOT>
__global__ __launch_bounds__(kTraceBlock)
__attribute__((amdgpu_waves_per_eu(WavesPerEu<T, RPT, POLK, NR>::value))) void trace_kernel(
    const DevSurfHot<T>* __restrict__ surf_tab, const DevSurfCold<T>* __restrict__ cold_tab,
    const DevOptics<T>* __restrict__ optics_tab, const T* __restrict__ coeff_tab,
    TraceArgs<T> a) {
#if OL_TABLE_IN_LDS
  // Experiment (BASELINE.json: "surface coefficients staged in LDS"): stage the hot
  // blocks and the optics of the traced range into LDS once per workgroup and read
  // them back with wave-uniform ds_reads.  Measured against the scalar-load path in
  // DESIGN.md 4.1; off by default.
  constexpr int kMaxLdsSurf = 64;
  __shared__ DevSurfHot<T> lds_hot[kMaxLdsSurf];
  __shared__ DevOptics<T> lds_opt[kMaxLdsSurf];
  {
    const int ns = a.last - a.first + 1;
    constexpr int HW = sizeof(DevSurfHot<T>) / 4, OW = sizeof(DevOptics<T>) / 4;
    const uint32_t* gh = reinterpret_cast<const uint32_t*>(surf_tab + a.first);
    uint32_t* lh = reinterpret_cast<uint32_t*>(lds_hot);
    for (int i = threadIdx.x; i < ns * HW; i += kTraceBlock) lh[i] = gh[i];
    uint32_t* lo = reinterpret_cast<uint32_t*>(lds_opt);
    for (int i = threadIdx.x; i < ns * OW; i += kTraceBlock) {
      const int sidx = i / OW, w = i % OW;
      lo[i] = reinterpret_cast<const uint32_t*>(optics_tab + (a.first + sidx) * a.n_wl + a.wl)[w];
    }
    __syncthreads();
  }
#endif
  // Lanes past the end leave at once.  With the spot epilogue (SPOT: a workgroup
  // reduction with a barrier) they have to stay: one-ray lanes then re-trace the LAST
  // ray (identical values to identical addresses, no predicate anywhere in the hot
  // loop -- a per-lane "live" guard around the stores cost 20 %), vector lanes take
  // the ragged-tail path with zero rays; either way they are masked out of the sums.
  using RayIndex = RayIndexT<(OL_SADDR != 0) && POLK != 0>;
  RayIndex base{(int64_t)blockIdx.x * (kTraceBlock * RPT), (uint32_t)threadIdx.x * RPT};
  const bool live = base.full() < a.n;
  if constexpr (!SPOT) {
    if (!live) return;
  } else if (RPT == 1 && !live) {
    base.lane = (uint32_t)(a.n - 1 - base.tile);
  }
  const int64_t left = a.n - base.full();
  // (one ray per lane: always exactly one ray -- said outright, so that the SPOT
  // variant's loads and stores stay as unpredicated as the plain kernel's)
  const int cnt = RPT == 1 ? 1 : (left >= RPT ? RPT : (left > 0 ? (int)left : 0));

  using LP = LanePack<T, RPT, POLK, NR>;  // fp32 lean kernel: packed pairs of rays
  using V = typename LP::V;
  constexpr int NV = LP::NV;
  Ray<V> r[NV];
  constexpr int NPRT = POLK == 2 ? 18 : 9;  // PRT planes (real, then imaginary)
  Prt<T, POLK> P[POLK ? RPT : 1];
  {
    // All plane loads are issued back to back under ONE branch (full vector vs
    // ragged tail): with the branch inside each plane's load the compiler placed
    // an s_waitcnt vmcnt(0) after every load, serialising 8 (+9) HBM round trips.
    T in[8][RPT];
    T pin[POLK ? NPRT : 1][RPT];
    if (RPT > 1 && cnt == RPT) {
      using V = typename VecOf<T, RPT>::type;
      V v[8];
#pragma unroll
      for (int f = 0; f < 8; ++f) v[f] = *reinterpret_cast<const V*>(base.at(a.rays[f]));
      if constexpr (POLK != 0) {
        if (!(a.flags & kTracePrtIdentity)) {
          V pv[NPRT];
#pragma unroll
          for (int e = 0; e < NPRT; ++e)
            pv[e] = *reinterpret_cast<const V*>(base.at(a.prt + (int64_t)e * a.n));
#pragma unroll
          for (int e = 0; e < NPRT; ++e)
#pragma unroll
            for (int k = 0; k < RPT; ++k) pin[e][k] = vec_get<T, RPT>(pv[e], k);
        }
      }
#pragma unroll
      for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int k = 0; k < RPT; ++k) in[f][k] = vec_get<T, RPT>(v[f], k);
    } else {
#pragma unroll
      for (int f = 0; f < 8; ++f)
#pragma unroll
        for (int k = 0; k < RPT; ++k) in[f][k] = k < cnt ? base.at(a.rays[f])[k] : T(0);
      if constexpr (POLK != 0) {
        if (!(a.flags & kTracePrtIdentity)) {
#pragma unroll
          for (int e = 0; e < NPRT; ++e)
#pragma unroll
            for (int k = 0; k < RPT; ++k)
              pin[e][k] = k < cnt ? base.at(a.prt + (int64_t)e * a.n)[k] : T(0);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      Ray<T> o;
      o.x = in[0][k]; o.y = in[1][k]; o.z = in[2][k];
      o.L = in[3][k]; o.M = in[4][k]; o.N = in[5][k];
      o.i = in[6][k]; o.opd = in[7][k];
      LP::put(r, k, o);
      if constexpr (POLK != 0) {
        const bool ident = (a.flags & kTracePrtIdentity) != 0;  // PRT starts as I
#pragma unroll
        for (int e = 0; e < NPRT; ++e)
          P[k].m[e] = ident ? ((e == 0 || e == 4 || e == 8) ? T(1) : T(0)) : pin[e][k];
      }
    }
  }

  uint32_t status = 0;
  bool is_global = true;  // frame of the state held in r[]
  bool prt_fresh = POLK != 0 && (a.flags & kTracePrtIdentity) != 0;
  DevSurf<T> last_traced;
  last_traced.cold = cold_tab;
  // hot block of the current surface by value (one s_load_dwordx16 for fp32); the
  // next surface's block is requested before the current one is worked on.
#if OL_TABLE_IN_LDS
  DevSurfHot<T> cur = lds_hot[0];
#else
  DevSurfHot<T> cur = surf_tab[a.first];
#endif
  for (int s = a.first; s <= a.last; ++s) {
    DevSurf<T> S;
    static_cast<DevSurfHot<T>&>(S) = cur;
    S.cold = cold_tab + s;
#if OL_TABLE_IN_LDS
    if (s < a.last) cur = lds_hot[s + 1 - a.first];
#else
    if (s < a.last) cur = surf_tab[s + 1];
#endif
    if (S.interaction != kRecordOnly) {
#if OL_TABLE_IN_LDS
      const DevOptics<T> O = lds_opt[s - a.first];
#else
      const DevOptics<T> O = optics_tab[s * a.n_wl + a.wl];
#endif
      surface_step<V, NV, POLK, NR>(S, O, coeff_tab, is_global, r, P, status, prt_fresh);
      is_global = false;
      last_traced = S;
    }
    if constexpr (RECORD) {
      T* row = a.record + (int64_t)(s - a.first) * 8 * a.record_stride;
      if (s == a.first && (a.flags & kTraceRow0IsInput)) {
        // the caller generated the rays straight into row 0 of the record block
        // (the object surface only records its input): nothing to write
      } else {
        Ray<V> gv[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) gv[j] = is_global ? r[j] : to_global<V>(last_traced, r[j]);
        Ray<T> g[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) g[k] = LP::ray(gv, k);
        store_rays<T, RPT>(row, a.record_stride, base, cnt, g);
      }
    }
  }

  if constexpr (SPOT) {
    // epilogue: masked moments of the final (global) state about (cx, cy),
    // into slot (workgroup % slots) of the caller's [slots][8] buffer -- the slots keep
    // the ~4e4 workgroups of a 1e7-ray launch off one address (7 x 39 k same-address
    // atomics were measured at 0.95 ms); the consumer adds the slots up
    Ray<V> gv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) gv[j] = is_global ? r[j] : to_global<V>(last_traced, r[j]);
    SpotAcc acc;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const Ray<T> g = LP::ray(gv, k);
      if (live && k < cnt) acc.add(g.x, g.y, g.i, a.cx, a.cy);
    }
    acc.flush(a.spot + 8 * (blockIdx.x % (unsigned)a.spot_slots));
  }
  // (SPOT: a lane that re-traced the last ray must not write it back -- its owner may
  // already have, and then this lane started from the final state)
  if ((a.flags & kTraceWriteRays) && (!SPOT || live)) {
    Ray<V> gv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) gv[j] = is_global ? r[j] : to_global<V>(last_traced, r[j]);
    Ray<T> g[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) g[k] = LP::ray(gv, k);
    T tmp[RPT];
#define OL_WB_FIELD(idx, fld)                                        \
  _Pragma("unroll") for (int k = 0; k < RPT; ++k) tmp[k] = g[k].fld; \
  store_plane<T, RPT>(a.rays[idx], base, cnt, tmp);
    OL_WB_FIELD(0, x)
    OL_WB_FIELD(1, y)
    OL_WB_FIELD(2, z)
    OL_WB_FIELD(3, L)
    OL_WB_FIELD(4, M)
    OL_WB_FIELD(5, N)
    OL_WB_FIELD(6, i)
    OL_WB_FIELD(7, opd)
#undef OL_WB_FIELD
  }
  if constexpr (POLK != 0) {
    T tmp[RPT];
#pragma unroll
    for (int e = 0; e < NPRT; ++e) {
#pragma unroll
      for (int k = 0; k < RPT; ++k) tmp[k] = P[k].m[e];
      store_plane<T, RPT>(a.prt + (int64_t)e * a.n, base, cnt, tmp);
    }
  }
  if (status && a.status) atomicOr(a.status, status);
}

// --------------------------------------------------------------------------
// host-side launcher
// --------------------------------------------------------------------------
template <typename T, int RPT, int NR>
static hipError_t launch_nr(const TraceArgs<T>& a, hipStream_t stream) {
  const int64_t threads = (a.n + RPT - 1) / RPT;
  const int64_t blocks = (threads + kTraceBlock - 1) / kTraceBlock;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  dim3 grid((unsigned)blocks), block(kTraceBlock);
  const bool rec = a.record != nullptr;
  const int polk = a.prt == nullptr ? 0 : ((a.flags & kTracePrtComplex) ? 2 : 1);
#define OL_LAUNCH_S(R, P, S)                                                                 \
  hipLaunchKernelGGL((trace_kernel<T, RPT, R, P, NR, S>), grid, block, 0, stream, a.surf,    \
                     a.cold, a.optics, a.coeffs, a)
#define OL_LAUNCH(R, P) OL_LAUNCH_S(R, P, false)
  if (a.spot != nullptr) {
    // the spot epilogue exists for unpolarised traces (the polarised intensity needs
    // the update_intensity epilogue first)
    if (polk != 0) return hipErrorInvalidValue;
    if (rec) OL_LAUNCH_S(true, 0, true); else OL_LAUNCH_S(false, 0, true);
    return hipGetLastError();
  }
  if (polk == 2) {
    // the complex-PRT variant exists for one ray per lane only (register budget)
    if constexpr (RPT == 1) {
      if (rec) OL_LAUNCH(true, 2); else OL_LAUNCH(false, 2);
    } else {
      return hipErrorInvalidValue;
    }
  } else if (rec && polk == 1) OL_LAUNCH(true, 1);
  else if (rec) OL_LAUNCH(true, 0);
  else if (polk == 1) OL_LAUNCH(false, 1);
  else OL_LAUNCH(false, 0);
#undef OL_LAUNCH_S
#undef OL_LAUNCH
  return hipGetLastError();
}

template <typename T, int RPT>
static hipError_t launch_rpt(const TraceArgs<T>& a, int nr, hipStream_t stream) {
  if (nr == 0) return launch_nr<T, RPT, 0>(a, stream);
  if constexpr (RPT > 1) {
    if (nr == 2) return launch_nr<T, RPT, 2>(a, stream);
  }
  return launch_nr<T, RPT, 1>(a, stream);
}

// fp32, two rays per lane = ONE packed pair, 8-byte loads / stores (lean unpolarised
// ranges only)
template <typename T>
static hipError_t launch_pair(const TraceArgs<T>& a, hipStream_t stream) {
  const int64_t threads = (a.n + 1) / 2;
  const int64_t blocks = (threads + kTraceBlock - 1) / kTraceBlock;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  dim3 grid((unsigned)blocks), block(kTraceBlock);
  if (a.record != nullptr)
    hipLaunchKernelGGL((trace_kernel<T, 2, true, 0, 0, false>), grid, block, 0, stream, a.surf,
                       a.cold, a.optics, a.coeffs, a);
  else
    hipLaunchKernelGGL((trace_kernel<T, 2, false, 0, 0, false>), grid, block, 0, stream, a.surf,
                       a.cold, a.optics, a.coeffs, a);
  return hipGetLastError();
}

#if OL_TRACE_TU != 2
Tuning& tuning() {
  static Tuning t = [] {
    Tuning v;
    if (const char* e = getenv("OL_TRACE_RPT")) v.rays_per_thread = atoi(e);
    return v;
  }();
  return t;
}
#endif

template <typename T>
hipError_t launch_trace(const TraceArgs<T>& a, bool vector_ok, bool has_newton,
                        hipStream_t stream) {
  constexpr int kVec = 16 / sizeof(T);  // rays per 16-byte lane vector
  const int nr = !has_newton ? 0 : ((a.flags & kTraceCompact) && tuning().compact ? 2 : 1);
  if (!vector_ok || (a.prt && (a.flags & kTracePrtComplex)))
    return launch_rpt<T, 1>(a, nr == 2 ? 1 : nr, stream);
  // Defaults from interleaved A/B runs on MI355X (tools/ab_bench.py):
  //  * record-all (HBM-write bound), fp32 and fp64: ONE ray per lane -- 46 / 103 VGPRs,
  //    8 / 4 waves per SIMD keep more stores in flight (fp32 +2 %, fp64 +6 % over the
  //    16-byte vector layout);
  //  * record-last on conic-only ranges (ALU bound): one 16-byte vector of rays per
  //    lane, whose independent chains interleave (fp32 0.205 vs 0.259 ms);
  //  * any range with Newton-Raphson surfaces: one ray per lane -- the iteration loop
  //    diverges per ray instead of per slot; 7-30 % faster on asphere / Zernike;
  //  * compaction only on request (measured slower on every surface tried: Newton
  //    iteration counts are nearly uniform across a wave once the stop rule is per ray).
  const int want = tuning().rays_per_thread;
  if constexpr (sizeof(T) == 4) {
    if (want == 3 && nr == 0 && a.prt == nullptr && a.spot == nullptr)
      return launch_pair<T>(a, stream);
  }
  const bool prefer_one = nr == 1 || a.record != nullptr;
  if (want == 1 || (want == 0 && prefer_one && nr != 2))
    return launch_rpt<T, 1>(a, nr == 2 ? 1 : nr, stream);
  return launch_rpt<T, kVec>(a, nr, stream);
}

// OL_TRACE_TU: 0 = everything in this translation unit; 1 / 2 = the fp32 / fp64
// instantiations only (trace_kernel_f32.hip / trace_kernel_f64.hip include this file so
// that the two halves compile in parallel)
#ifndef OL_TRACE_TU
#define OL_TRACE_TU 0
#endif
#if OL_TRACE_TU != 2
template hipError_t launch_trace<float>(const TraceArgs<float>&, bool, bool, hipStream_t);
#endif
#if OL_TRACE_TU != 1
template hipError_t launch_trace<double>(const TraceArgs<double>&, bool, bool, hipStream_t);
#endif

// --------------------------------------------------------------------------
// fused spot kernel: generate -> trace -> reduce, nothing but the pupil read
// --------------------------------------------------------------------------
// SURVEY.md 8 f1 + f2: the spot statistics of one (field, wavelength) need only
// sum / max reductions over the image-plane hits, so the rays never have to exist in
// HBM.  Each lane generates its rays from (Hx, Hy, Px, Py) (raygen_device.h, the same
// code as ol_generate_rays), walks the surface table exactly like trace_kernel and
// folds the hit into per-lane fp64 partial sums about the caller's centre (cx, cy):
//   out[0] += #{i > 0}        out[1] += sum dx      out[2] += sum dy
//   out[3] += sum dx^2        out[4] += sum dy^2    out[5] += sum i
//   out[6]  = max(out[6], max(dx^2 + dy^2))         (dx = x - cx, dy = y - cy)
// the mask i > 0 is analysis/spot_diagram/core.py:470-476.  HBM traffic: 2 planes
// read (+3 written when the caller asks for the hits); the kernel is ALU-bound.
// A workgroup walks `tiles_per_block` consecutive tiles (chosen by the launcher): few
// enough workgroups that the 7 same-address atomics per workgroup stay far below the
// L2 atomic rate, many enough (>= ~8 rounds of resident workgroups) that the last
// round's partial occupancy does not show.

template <typename T, int RPT, int NR, bool FIELDP, bool APOD>
__global__ __launch_bounds__(kTraceBlock) void spot_trace_kernel(
    const DevSurfHot<T>* __restrict__ surf_tab, const DevSurfCold<T>* __restrict__ cold_tab,
    const DevOptics<T>* __restrict__ optics_tab, const T* __restrict__ coeff_tab,
    SpotArgs<T> a) {
  const RaygenConsts<T> c(a.rg);
  // FIELDP: per-ray field planes.  A template parameter because the per-ray
  // double-precision tangent (tan_deg) would otherwise set the register budget of
  // the common launch-uniform-field case as well.
  constexpr bool field_planes = FIELDP;
  const RaygenIn<T>& in_ = a.in;
  const bool vig_planes = in_.vx != nullptr;
  // launch-uniform field: the two tangents come from the host (uniform_field_tangents)
  const T tx0 = in_.tx0, ty0 = in_.ty0;

  SpotAcc acc;
  uint32_t status = 0;
  bool prt_fresh = false;  // unpolarised kernel: unused
  constexpr int64_t kTileRays = (int64_t)kTraceBlock * RPT;
  const int64_t ntiles = (a.n + kTileRays - 1) / kTileRays;

  for (int tt = 0; tt < a.tiles_per_block; ++tt) {
    const int64_t tile = (int64_t)blockIdx.x * a.tiles_per_block + tt;
    if (tile >= ntiles) break;  // workgroup-uniform
    const int64_t base = (tile * kTraceBlock + threadIdx.x) * RPT;
    const int64_t left = a.n - base;
    const int cnt = left >= RPT ? RPT : (left > 0 ? (int)left : 0);

    // pupil (and optional per-ray field / vignetting) planes; lanes past the end
    // trace the on-axis pupil point and are masked out of the sums
    T in[6][RPT];
    if (RPT > 1 && cnt == RPT) {
      using V = typename VecOf<T, RPT>::type;
      V v[6];
      v[0] = *reinterpret_cast<const V*>(in_.px + base);
      v[1] = *reinterpret_cast<const V*>(in_.py + base);
      if constexpr (FIELDP) {
        v[2] = *reinterpret_cast<const V*>(in_.hx + base);
        v[3] = *reinterpret_cast<const V*>(in_.hy + base);
      }
      if (vig_planes) {
        v[4] = *reinterpret_cast<const V*>(in_.vx + base);
        v[5] = *reinterpret_cast<const V*>(in_.vy + base);
      }
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        in[0][k] = vec_get<T, RPT>(v[0], k);
        in[1][k] = vec_get<T, RPT>(v[1], k);
        in[2][k] = field_planes ? vec_get<T, RPT>(v[2], k) : T(0);
        in[3][k] = field_planes ? vec_get<T, RPT>(v[3], k) : T(0);
        in[4][k] = vig_planes ? vec_get<T, RPT>(v[4], k) : in_.vx0;
        in[5][k] = vig_planes ? vec_get<T, RPT>(v[5], k) : in_.vy0;
      }
    } else {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const bool ok = k < cnt;
        in[0][k] = ok ? in_.px[base + k] : T(0);
        in[1][k] = ok ? in_.py[base + k] : T(0);
        in[2][k] = (ok && field_planes) ? in_.hx[base + k] : T(0);
        in[3][k] = (ok && field_planes) ? in_.hy[base + k] : T(0);
        in[4][k] = (ok && vig_planes) ? in_.vx[base + k] : in_.vx0;
        in[5][k] = (ok && vig_planes) ? in_.vy[base + k] : in_.vy0;
      }
    }

    using LP = LanePack<T, RPT, 0, NR>;  // fp32 lean kernel: packed pairs of rays
    using V = typename LP::V;
    constexpr int NV = LP::NV;
    Ray<V> r[NV];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      T tx = tx0, ty = ty0, o[6];
      if constexpr (FIELDP) {
        if ((in_.flags & kRaygenCheckField) && (outside_unit(in[2][k]) || outside_unit(in[3][k])))
          status |= kStatusFieldRange;
        raygen_field<T>(c, in[2][k], in[3][k], tx, ty);
      }
      raygen_pupil<T>(in_.flags, in[4][k], in[5][k], in[0][k], in[1][k], status);
      raygen_one<T>(c, tx, ty, in[0][k], in[1][k], in[4][k], in[5][k], o);
      Ray<T> q;
      q.x = o[0]; q.y = o[1]; q.z = o[2];
      q.L = o[3]; q.M = o[4]; q.N = o[5];
      // APOD: a template parameter -- the exp / cos / pow code of the apodization
      // switch would otherwise set the register budget of every spot launch
      // (fp32 packed 74 -> 129 VGPRs, fp64 113 -> 186: 0.23 -> 0.26 / 0.57 -> 0.70 ms)
      if constexpr (APOD) q.i = raygen_apodize<T>(c, in[0][k], in[1][k]); else q.i = T(1);
      q.opd = T(0);
      LP::put(r, k, q);
    }

    bool is_global = true;
    DevSurf<T> last_traced;
    last_traced.cold = cold_tab;
    Prt<T, 0> P[1];
    DevSurfHot<T> cur = surf_tab[a.first];
    for (int s = a.first; s <= a.last; ++s) {
      DevSurf<T> S;
      static_cast<DevSurfHot<T>&>(S) = cur;
      S.cold = cold_tab + s;
      if (s < a.last) cur = surf_tab[s + 1];
      if (S.interaction != kRecordOnly) {
        const DevOptics<T> O = optics_tab[s * a.n_wl + a.wl];
        surface_step<V, NV, 0, NR>(S, O, coeff_tab, is_global, r, P, status, prt_fresh);
        is_global = false;
        last_traced = S;
      }
    }

    Ray<V> gv[NV];
#pragma unroll
    for (int j = 0; j < NV; ++j) gv[j] = is_global ? r[j] : to_global<V>(last_traced, r[j]);
    T hx_[RPT], hy_[RPT], hi_[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const Ray<T> g = LP::ray(gv, k);
      hx_[k] = g.x; hy_[k] = g.y; hi_[k] = g.i;
      if (k < cnt) acc.add(g.x, g.y, g.i, a.cx, a.cy);
    }
    if (a.hits[0] != nullptr && cnt > 0) {
      const RayIndexT<false> at{tile * (kTraceBlock * RPT), (uint32_t)threadIdx.x * RPT};
      store_plane<T, RPT>(a.hits[0], at, cnt, hx_);
      store_plane<T, RPT>(a.hits[1], at, cnt, hy_);
      store_plane<T, RPT>(a.hits[2], at, cnt, hi_);
    }
  }

  acc.flush(a.out);
  if (status && a.status) atomicOr(a.status, status);
}

template <typename T, int RPT, int NR>
static hipError_t launch_spot_nr(const SpotArgs<T>& a_in, hipStream_t stream) {
  SpotArgs<T> a = a_in;
  const int64_t tile_rays = (int64_t)kTraceBlock * RPT;
  const int64_t ntiles = (a.n + tile_rays - 1) / tile_rays;
  static const int forced = [] {
    const char* e = getenv("OL_SPOT_TILES");
    return e ? atoi(e) : 0;
  }();
  constexpr int64_t kTargetBlocks = 8192;  // ~8 rounds of 256 CUs x 4 resident workgroups
  int64_t tpb = forced > 0 ? forced : (ntiles + kTargetBlocks - 1) / kTargetBlocks;
  if (tpb < 1) tpb = 1;
  if (tpb > 1024) tpb = 1024;
  a.tiles_per_block = (int32_t)tpb;
  if (a.in.hx == nullptr) uniform_field_tangents<T>(a.rg, a.in);
  const int64_t blocks = (ntiles + tpb - 1) / tpb;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
#define OL_SPOT_LAUNCH(F, A)                                                               \
  hipLaunchKernelGGL((spot_trace_kernel<T, RPT, NR, F, A>), dim3((unsigned)blocks),        \
                     dim3(kTraceBlock), 0, stream, a.surf, a.cold, a.optics, a.coeffs, a)
  const bool fieldp = a.in.hx != nullptr, apod = a.rg.apod_kind != 0;
  if (fieldp && apod) OL_SPOT_LAUNCH(true, true);
  else if (fieldp) OL_SPOT_LAUNCH(true, false);
  else if (apod) OL_SPOT_LAUNCH(false, true);
  else OL_SPOT_LAUNCH(false, false);
#undef OL_SPOT_LAUNCH
  return hipGetLastError();
}

template <typename T>
hipError_t launch_spot_trace(const SpotArgs<T>& a, bool vector_ok, bool has_newton,
                             hipStream_t stream) {
  constexpr int kVec = 16 / sizeof(T);
  // same defaults as record-last traces (launch_trace): conic-only ranges are ALU
  // bound and want the 16-byte vector of rays per lane; Newton ranges one ray
  const int want = tuning().rays_per_thread;
  if (has_newton)
    return (vector_ok && want == 2) ? launch_spot_nr<T, kVec, 1>(a, stream)
                                    : launch_spot_nr<T, 1, 1>(a, stream);
  if (!vector_ok || want == 1) return launch_spot_nr<T, 1, 0>(a, stream);
  return launch_spot_nr<T, kVec, 0>(a, stream);
}

#if OL_TRACE_TU != 2
template hipError_t launch_spot_trace<float>(const SpotArgs<float>&, bool, bool, hipStream_t);
#endif
#if OL_TRACE_TU != 1
template hipError_t launch_spot_trace<double>(const SpotArgs<double>&, bool, bool, hipStream_t);
#endif

// --------------------------------------------------------------------------
// fused generate -> trace -> OPD: the wavefront analogue of the spot
// kernel.  One launch takes the normalised pupil coordinates of ONE field point to the
// per-ray OPD in waves against the reference sphere / plane (wavefront_device.h), the
// image-plane intensity and -- as device-side sums -- everything the consumers reduce
// the map to: the nine weighted moments of the tilt fit (wavefront/wavefront.py:103-148),
// count / sum / sum of squares of the OPD over rays with i > 0 (piston, RMS:
// wavefront/opd.py:145-159).  The rays never exist in HBM: 2 planes in, 2 (+3) out,
// where the un-fused chain (ol_generate_rays + record-all ol_trace + ol_wavefront_opd +
// torch reductions) moves 8 (S + 2) + 13 planes.  One ray per lane: wavefront work is
// fp64 (an OPD good to lambda/1000 over a 200 mm path) and the interesting systems carry
// aspheres.  out[] of the moments:
//   0 sum w   1 sum w X   2 sum w Y   3 sum w XX   4 sum w XY   5 sum w YY
//   6 sum w o 7 sum w o X 8 sum w o Y      (w = intensity, o = OPD, X/Y = pupil point)
//   9 #{i > 0}   10 sum o [i > 0]   11 sum o^2 [i > 0]
// --------------------------------------------------------------------------
template <typename T, int NR, bool APOD>
__global__ __launch_bounds__(kTraceBlock) void opd_trace_kernel(
    const DevSurfHot<T>* __restrict__ surf_tab, const DevSurfCold<T>* __restrict__ cold_tab,
    const DevOptics<T>* __restrict__ optics_tab, const T* __restrict__ coeff_tab,
    OpdArgs<T> a) {
  const RaygenConsts<T> c(a.rg);
  const WavefrontConsts<T> w(a.wf);
  const RaygenIn<T>& in_ = a.in;
  uint32_t status = 0;
  bool prt_fresh = false;
  double s[kOpdMoments];
#pragma unroll
  for (int k = 0; k < kOpdMoments; ++k) s[k] = 0.0;

  for (int64_t j = (int64_t)blockIdx.x * kTraceBlock + threadIdx.x; j < a.n;
       j += (int64_t)gridDim.x * kTraceBlock) {
    T px = in_.px[j], py = in_.py[j];
    T vx = in_.vx0, vy = in_.vy0, o[6];
    raygen_pupil<T>(in_.flags, vx, vy, px, py, status);
    raygen_one<T>(c, in_.tx0, in_.ty0, px, py, vx, vy, o);
    Ray<T> r[1];
    r[0].x = o[0]; r[0].y = o[1]; r[0].z = o[2];
    r[0].L = o[3]; r[0].M = o[4]; r[0].N = o[5];
    if constexpr (APOD) r[0].i = raygen_apodize<T>(c, px, py); else r[0].i = T(1);
    r[0].opd = T(0);

    bool is_global = true;
    DevSurf<T> last_traced;
    last_traced.cold = cold_tab;
    Prt<T, 0> P[1];
    DevSurfHot<T> cur = surf_tab[a.first];
    for (int sidx = a.first; sidx <= a.last; ++sidx) {
      DevSurf<T> S;
      static_cast<DevSurfHot<T>&>(S) = cur;
      S.cold = cold_tab + sidx;
      if (sidx < a.last) cur = surf_tab[sidx + 1];
      if (S.interaction != kRecordOnly) {
        const DevOptics<T> O = optics_tab[sidx * a.n_wl + a.wl];
        surface_step<T, 1, 0, NR>(S, O, coeff_tab, is_global, r, P, status, prt_fresh);
        is_global = false;
        last_traced = S;
      }
    }
    const Ray<T> g = is_global ? r[0] : to_global<T>(last_traced, r[0]);
    T pu[3];
    // the pupil coordinates of the tilt term are the ones the CALLER passed (the
    // reference corrects with the distribution's points, strategy.py:88-139)
    const T ov = wavefront_one<T>(w, g.x, g.y, g.z, g.L, g.M, g.N, g.opd, in_.px[j], in_.py[j], pu);
    a.opd[j] = ov;
    a.inten[j] = g.i;
    if (a.pupil[0]) {
      a.pupil[0][j] = pu[0];
      a.pupil[1][j] = pu[1];
      a.pupil[2][j] = pu[2];
    }
    const double wi = (double)g.i, od = (double)ov, X = (double)pu[0], Y = (double)pu[1];
    s[0] += wi; s[1] += wi * X; s[2] += wi * Y;
    s[3] += wi * X * X; s[4] += wi * X * Y; s[5] += wi * Y * Y;
    s[6] += wi * od; s[7] += wi * od * X; s[8] += wi * od * Y;
    if (g.i > T(0)) {
      s[9] += 1.0;
      s[10] += od;
      s[11] += od * od;
    }
  }

  // workgroup reduction -> kOpdMoments atomics (every thread reaches this point)
  __shared__ double part[kTraceBlock / 64][kOpdMoments];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double t[kOpdMoments];
#pragma unroll
    for (int k = 0; k < kOpdMoments; ++k) t[k] = __shfl_down(s[k], off, 64);
#pragma unroll
    for (int k = 0; k < kOpdMoments; ++k) s[k] += t[k];
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < kOpdMoments; ++k) part[wave][k] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < kOpdMoments) {
    double v = 0;
    for (int q = 0; q < kTraceBlock / 64; ++q) v += part[q][threadIdx.x];
    // (a NaN partial sum must reach the output too: v != 0.0 is true for NaN)
    if (v != 0.0) unsafeAtomicAdd(&a.mom[threadIdx.x], v);
  }
  if (status && a.status) atomicOr(a.status, status);
}

template <typename T>
hipError_t launch_opd_trace(const OpdArgs<T>& a_in, bool has_newton, hipStream_t stream) {
  OpdArgs<T> a = a_in;
  uniform_field_tangents<T>(a.rg, a.in);
  int64_t blocks = (a.n + kTraceBlock - 1) / kTraceBlock;
  if (blocks == 0) return hipSuccess;
  if (blocks > 8192) blocks = 8192;  // grid-stride beyond: keeps the atomics few
  const bool apod = a.rg.apod_kind != 0;
#define OL_OPD_LAUNCH(N, A)                                                                  \
  hipLaunchKernelGGL((opd_trace_kernel<T, N, A>), dim3((unsigned)blocks), dim3(kTraceBlock), \
                     0, stream, a.surf, a.cold, a.optics, a.coeffs, a)
  if (has_newton) {
    if (apod) OL_OPD_LAUNCH(1, true); else OL_OPD_LAUNCH(1, false);
  } else {
    if (apod) OL_OPD_LAUNCH(0, true); else OL_OPD_LAUNCH(0, false);
  }
#undef OL_OPD_LAUNCH
  return hipGetLastError();
}

#if OL_TRACE_TU != 1
template hipError_t launch_opd_trace<double>(const OpdArgs<double>&, bool, hipStream_t);
#endif

}  // namespace ol
