// surface_math.h -- the per-ray, per-surface arithmetic of the fused trace: everything
// between "ray state in the previous surface's frame" and "ray state after the
// interaction", as function templates over the working type.  ONE definition, used by
// every kernel in trace_kernel.hip (device code).
//
// The functions are declared with OL_DEV (device_table.h).  In the product build that is
// `OL_DEV`: device code only, nothing in liboptiland_hip.so can run
// this arithmetic on the host.  tests/hostmath/harness.hip defines OL_HOST_MATH before
// including this header, which makes them `__host__ __device__` so that the SAME source
// can be executed ray by ray on the CPU of a box without a GPU and held against the
// oracle (tests/test_hostmath.py): a second check of the kernel arithmetic that does not
// need GPU minutes.  Test infrastructure only -- see tests/hostmath/README.md.
//
// Per-surface arithmetic follows SURVEY.md Appendix A; each function cites the
// reference lines it implements.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include <type_traits>

#include "device_table.h"

namespace ol {
// --------------------------------------------------------------------------
// hardware primitives.  Device code: the gfx950 instructions.  Host code exists only
// under OL_HOST_MATH (tests): the IEEE operation the instruction approximates to 1 ulp.
// --------------------------------------------------------------------------
namespace hw {
#if defined(OL_HOST_MATH) && !defined(__HIP_DEVICE_COMPILE__)
OL_DEV float rcp(float x) { return 1.0f / x; }
OL_DEV float sqrt(float x) { return ::sqrtf(x); }
OL_DEV float rsq(float x) { return 1.0f / ::sqrtf(x); }
OL_DEV float exp(float x) { return ::expf(x); }
OL_DEV int float_bits(float x) { int i; __builtin_memcpy(&i, &x, 4); return i; }
OL_DEV long long double_bits(double x) { long long i; __builtin_memcpy(&i, &x, 8); return i; }
OL_DEV bool wave_any(bool v) { return v; }  // a "wave" of one ray
OL_DEV int wave_max(int v) { return v; }
OL_DEV bool wave_leader() { return true; }
OL_DEV void atomic_max_i32(int32_t* p, int32_t v) { if (v > *p) *p = v; }
OL_DEV void atomic_or_i32(int32_t* p, int32_t v) { *p |= v; }
OL_DEV int32_t load_i32(const int32_t* p) { return *p; }
#else
OL_DEV float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
OL_DEV float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
OL_DEV float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
// fp64 division and square root from the hardware seeds (v_rcp_f64 / v_rsq_f64, ~2^-27
// relative) refined by two Newton / Goldschmidt steps in FMA arithmetic: within ~1 ulp of the
// correctly rounded result (not always THE correctly rounded one), zeros, infinities and
// NaNs as IEEE has them.  The compiler's own sequences are correctly rounded and carry the
// scaling that makes them so for operands near the ends of the exponent range
// (v_div_scale / v_div_fmas / v_div_fixup: 10 vector instructions for a quotient; 20 for a
// square root) -- lengths, cosines and optical paths are nowhere near those ends, and the
// fused fp64 kernels (spot, OPD: the wavefront path) are bound by vector issue, with two
// square roots and two quotients per conic surface.
// v_cmp_class masks: bit 2 -inf, 5 -0, 6 +0, 9 +inf
constexpr int kClassZeroOrInf = 0x264;  // +-0 | +-inf
constexpr int kClassDenormal = 0x090;   // +-denormal
OL_DEV double rcp_f64(double x) {
  const double y0 = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, y0, 1.0);
  double y = __builtin_fma(y0, e, y0);
  e = __builtin_fma(-x, y, 1.0);
  y = __builtin_fma(y, e, y);
  // x = +-0 / +-inf: the seed IS the answer (+-inf / +-0), the refinement would make it NaN
  return __builtin_amdgcn_class(x, kClassZeroOrInf) ? y0 : y;
}
// Not IEEE in two corners, both outside anything a trace forms (lengths, cosines, optical
// paths): an INFINITE numerator with a finite divisor gives NaN (the residual is inf - inf),
// and a DENORMAL divisor whose seed overflows gives NaN instead of inf.  Catching them costs
// two more v_cmp_class per quotient in kernels that are bound by vector issue; the
// hostmath / GPU edge-case tests pin the behaviour (tests/test_gpu_hostmath.py).
// OL_DIV_F64_STEPS (1 | 2): Newton steps on the reciprocal before the quotient is formed.  ONE
// is enough: the residual correction q + (a - b q) y squares the error once more, so the
// quotient is within 2^-52 x 2^-52 of exact before its final rounding whether y carries 2^-52
// or 2^-100 -- measured on 8e6 operands per range (tests/test_gpu_math_probe.py): 0.5 ulp
// maximum, i.e. correctly rounded, with either; two fewer FMAs per quotient, two quotients per
// conic surface.
#ifndef OL_DIV_F64_STEPS
#define OL_DIV_F64_STEPS 1
#endif
OL_DEV double div_f64(double a, double b) {
  const double y0 = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, y0, 1.0);
  double y = __builtin_fma(y0, e, y0);
#if OL_DIV_F64_STEPS > 1
  e = __builtin_fma(-b, y, 1.0);
  y = __builtin_fma(y, e, y);
#endif
  const double q = a * y;
  const double r = __builtin_fma(-b, q, a);
  const double refined = __builtin_fma(r, y, q);
  return __builtin_amdgcn_class(b, kClassZeroOrInf) ? a * y0 : refined;
}
// TWO quotients n1 / d1 and n2 / d2 from ONE reciprocal seed: y ~ 1 / (d1 d2) (seed + one
// Newton step), 1 / d1 ~ d2 y, 1 / d2 ~ d1 y, each quotient then corrected once with its own
// residual like div_f64 -- 12 vector instructions and one quarter-rate seed where two div_f64
// take 20 and two.  The conic intersection forms C / q and q / A for every ray and surface
// (standard.py:112-146): with the sign product of round 5 an fp64 surface goes 124 -> ~113
// in the kernels that take it (curved_distance<V, SHARE>: the vector-issue-bound ones).
// Lanes whose product d1 d2 is zero or infinite (a paraboloid met by an axis-parallel ray has
// A = 0 exactly) take div_f64 -- per LANE, under a wave-uniform branch that such waves alone
// enter: a ray's result never depends on its neighbours in the wave.
OL_DEV void div2_f64(double n1, double d1, double n2, double d2, double& q1, double& q2) {
  const double dd = d1 * d2;
  const double y0 = __builtin_amdgcn_rcp(dd);
  const double e = __builtin_fma(-dd, y0, 1.0);
  const double y = __builtin_fma(y0, e, y0);
  const double i1 = d2 * y, i2 = d1 * y;
  double a = n1 * i1;
  a = __builtin_fma(__builtin_fma(-d1, a, n1), i1, a);
  double b = n2 * i2;
  b = __builtin_fma(__builtin_fma(-d2, b, n2), i2, b);
  // (... or a denormal: its reciprocal seed overflows and both quotients would be NaN where
  // n1 / d1 and n2 / d2 are finite -- the product of two modest divisors can land there)
  const bool odd = __builtin_amdgcn_class(dd, kClassZeroOrInf | kClassDenormal);
  if (__any(odd)) {
    const double sa = div_f64(n1, d1), sb = div_f64(n2, d2);
    a = odd ? sa : a;
    b = odd ? sb : b;
  }
  q1 = a;
  q2 = b;
}
OL_DEV double sqrt_f64(double x) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  // sqrt(+-0) = +-0, sqrt(+inf) = +inf (0 * inf above); negative / NaN arguments are NaN
  // through the seed already
  return __builtin_amdgcn_class(x, 0x260) ? x : g;   // +-0 | +inf (-inf: NaN via the seed)
}
OL_DEV double rsq_f64(double x) {
  const double y0 = __builtin_amdgcn_rsq(x);
  // y <- y (1.5 - 0.5 x y^2), twice
  double hx = 0.5 * x;
  double e = __builtin_fma(-hx * y0, y0, 0.5);
  double y = __builtin_fma(y0, e, y0);
  e = __builtin_fma(-hx * y, y, 0.5);
  y = __builtin_fma(y, e, y);
  // +-0 (seed +-inf) and +inf (seed 0): the seed is the answer; -inf: NaN through the seed
  return __builtin_amdgcn_class(x, 0x260) ? y0 : y;
}
OL_DEV float exp(float x) { return __expf(x); }
OL_DEV int float_bits(float x) { return __float_as_int(x); }
OL_DEV long long double_bits(double x) { return __double_as_longlong(x); }
OL_DEV bool wave_any(bool v) { return __any(v) != 0; }
// (the reference-Newton launches only: maximum over the ACTIVE lanes of the wave, one lane of
// them, and the two atomics their per-surface words take)
OL_DEV int wave_max(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const int o = __shfl_xor(v, off, 64);  // (a disabled source lane reads as 0; v >= 0)
    v = o > v ? o : v;
  }
  return v;
}
OL_DEV bool wave_leader() {
  const uint64_t m = __ballot(true);
  return (int)__lane_id() == __builtin_ctzll(m);
}
OL_DEV void atomic_max_i32(int32_t* p, int32_t v) {
  if (v > __atomic_load_n(p, __ATOMIC_RELAXED)) atomicMax(p, v);
}
OL_DEV void atomic_or_i32(int32_t* p, int32_t v) { atomicOr(p, v); }
OL_DEV int32_t load_i32(const int32_t* p) { return __atomic_load_n(p, __ATOMIC_RELAXED); }
#endif
}  // namespace hw

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
  static OL_DEV T splat(T v) { return v; }                            \
  static OL_DEV bool lt(T a, T b) { return a < b; }                   \
  static OL_DEV bool le(T a, T b) { return a <= b; }                  \
  static OL_DEV bool gt(T a, T b) { return a > b; }                   \
  static OL_DEV bool ge(T a, T b) { return a >= b; }                  \
  static OL_DEV bool eq(T a, T b) { return a == b; }                  \
  static OL_DEV bool ne(T a, T b) { return a != b; }                  \
  static OL_DEV bool all(bool v) { return v; }                        \
  static OL_DEV bool mnot(bool m) { return !m; }                      \
  static OL_DEV bool mand(bool p, bool q) { return p && q; }          \
  static OL_DEV bool mor(bool p, bool q) { return p || q; }           \
  static OL_DEV bool any(bool p) { return p; }                        \
  static OL_DEV bool same(bool p, bool q) { return p == q; }          \
  static OL_DEV T select(bool m, T a, T b) { return m ? a : b; }      \
  static OL_DEV bool mselect(bool m, bool a, bool b) { return m ? a : b; }

template <>
struct Math<float> {
  // v_rcp_f32 / v_sqrt_f32 / v_rsq_f32: 1 ulp, quarter rate, no denormal
  // fix-up sequences -- well inside the 1e-4 fp32 parity budget.
  static OL_DEV float rcp(float x) { return hw::rcp(x); }
  static OL_DEV float sqrt(float x) { return hw::sqrt(x); }
  static OL_DEV float rsqrt(float x) { return hw::rsq(x); }
  static OL_DEV float div(float a, float b) { return a * rcp(b); }
  static OL_DEV void div2(float n1, float d1, float n2, float d2, float& q1, float& q2) {
    q1 = div(n1, d1);
    q2 = div(n2, d2);
  }
  static OL_DEV float exp(float x) { return hw::exp(x); }
  static OL_DEV float abs(float x) { return __builtin_fabsf(x); }
  static OL_DEV float copysign(float a, float b) {
    return __builtin_copysignf(a, b);
  }
  static OL_DEV float fma(float a, float b, float c) {
    return __builtin_fmaf(a, b, c);
  }
  static OL_DEV float eps() { return 1.1920929e-7f; }
  static OL_DEV float guard() { return 1e-14f; }
  OL_SCALAR_LANE_OPS(float)
};

// OL_FAST_F64 (default on, device code only): fp64 quotients and square roots from the
// hardware seeds + two refinement steps (hw::div_f64 ...) instead of the compiler's correctly
// rounded sequences.  A/B knob, tools/build_variants.py.
#ifndef OL_FAST_F64
#define OL_FAST_F64 1
#endif
#if OL_FAST_F64 && !(defined(OL_HOST_MATH) && !defined(__HIP_DEVICE_COMPILE__))
#define OL_F64_FAST_PATH 1
#else
#define OL_F64_FAST_PATH 0
#endif

template <>
struct Math<double> {
#if OL_F64_FAST_PATH
  static OL_DEV double rcp(double x) { return hw::rcp_f64(x); }
  static OL_DEV double sqrt(double x) { return hw::sqrt_f64(x); }
  static OL_DEV double rsqrt(double x) { return hw::rsq_f64(x); }
  static OL_DEV double div(double a, double b) { return hw::div_f64(a, b); }
#ifndef OL_DIV2_F64
#define OL_DIV2_F64 1
#endif
  static OL_DEV void div2(double n1, double d1, double n2, double d2, double& q1, double& q2) {
    if (OL_DIV2_F64) {
      hw::div2_f64(n1, d1, n2, d2, q1, q2);
    } else {
      q1 = div(n1, d1);
      q2 = div(n2, d2);
    }
  }
#else
  static OL_DEV double rcp(double x) { return 1.0 / x; }
  static OL_DEV double sqrt(double x) { return __builtin_sqrt(x); }
  static OL_DEV double rsqrt(double x) { return 1.0 / __builtin_sqrt(x); }
  static OL_DEV double div(double a, double b) { return a / b; }
  static OL_DEV void div2(double n1, double d1, double n2, double d2, double& q1, double& q2) {
    q1 = n1 / d1;
    q2 = n2 / d2;
  }
#endif
  static OL_DEV double exp(double x) { return ::exp(x); }
  static OL_DEV double abs(double x) { return __builtin_fabs(x); }
  static OL_DEV double copysign(double a, double b) {
    return __builtin_copysign(a, b);
  }
  static OL_DEV double fma(double a, double b, double c) {
    return __builtin_fma(a, b, c);
  }
  static OL_DEV double eps() { return 2.220446049250313e-16; }
  static OL_DEV double guard() { return 1e-14; }
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
#if !defined(__HIP_DEVICE_COMPILE__)
// (host build of the kernel source: the ordering handle of refresh_after, device_table.h, on one
// element -- x86 has no register constraint for an 8-byte float vector)
template <typename P>
OL_DEV P refresh_after(P p, f32x2& acc) {
  float lo = acc.x;
  asm volatile("" : "+r"(p), "+x"(lo));
  acc.x = lo;
  return p;
}
#endif

template <>
struct Math<f32x2> {
  using scalar = float;
  using mask = Mask2;
  static constexpr int lanes = 2;
  using V = f32x2;
  static OL_DEV V rcp(V x) {
    return V{hw::rcp(x.x), hw::rcp(x.y)};
  }
  static OL_DEV V sqrt(V x) {
    return V{hw::sqrt(x.x), hw::sqrt(x.y)};
  }
  static OL_DEV V rsqrt(V x) {
    return V{hw::rsq(x.x), hw::rsq(x.y)};
  }
  static OL_DEV V div(V a, V b) { return a * rcp(b); }
  static OL_DEV void div2(V n1, V d1, V n2, V d2, V& q1, V& q2) {
    q1 = div(n1, d1);
    q2 = div(n2, d2);
  }
  static OL_DEV V exp(V x) { return V{hw::exp(x.x), hw::exp(x.y)}; }
  static OL_DEV V abs(V x) {
    return V{__builtin_fabsf(x.x), __builtin_fabsf(x.y)};
  }
  static OL_DEV V copysign(V a, V b) {
    return V{__builtin_copysignf(a.x, b.x), __builtin_copysignf(a.y, b.y)};
  }
  static OL_DEV V fma(V a, V b, V c) {
    return __builtin_elementwise_fma(a, b, c);
  }
  static OL_DEV float eps() { return 1.1920929e-7f; }
  static OL_DEV float guard() { return 1e-14f; }
  static OL_DEV V splat(float v) { return V{v, v}; }
  static OL_DEV mask lt(V a, V b) { return {a.x < b.x, a.y < b.y}; }
  static OL_DEV mask le(V a, V b) { return {a.x <= b.x, a.y <= b.y}; }
  static OL_DEV mask gt(V a, V b) { return {a.x > b.x, a.y > b.y}; }
  static OL_DEV mask ge(V a, V b) { return {a.x >= b.x, a.y >= b.y}; }
  static OL_DEV mask eq(V a, V b) { return {a.x == b.x, a.y == b.y}; }
  static OL_DEV mask ne(V a, V b) { return {a.x != b.x, a.y != b.y}; }
  static OL_DEV mask all(bool v) { return {v, v}; }
  static OL_DEV mask mnot(mask m) { return {!m.a, !m.b}; }
  static OL_DEV mask mand(mask p, mask q) { return {p.a && q.a, p.b && q.b}; }
  static OL_DEV mask mor(mask p, mask q) { return {p.a || q.a, p.b || q.b}; }
  static OL_DEV bool any(mask p) { return p.a || p.b; }
  static OL_DEV mask same(mask p, mask q) { return {p.a == q.a, p.b == q.b}; }
  static OL_DEV V select(mask m, V a, V b) {
    return V{m.a ? a.x : b.x, m.b ? a.y : b.y};
  }
  static OL_DEV mask mselect(mask m, mask a, mask b) {
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
// element type of the matrices a surface_step<V, ...> instantiation carries: V itself in a
// polarised launch (one matrix per lane element), the scalar type in the unpolarised ones (their
// callers hand over a stand-in that nothing reads)
template <typename V, int POLK>
struct PrtLane {
  using type = typename std::conditional<POLK != 0, V, typename Math<V>::scalar>::type;
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
OL_DEV V flat_distance(V z, V N) {  // standard.py:108-111
  using m = Math<V>;
  const V g = m::splat(m::guard());
  V Ns = m::select(m::gt(m::abs(N), g), N, g);
  return -m::div(z, Ns);
}

// SHARE: the two quotients of the intersection from ONE reciprocal (Math<V>::div2; fp64 only
// makes a difference).  Measured arm against arm in one process (profiles/r05_ab_arith.txt):
// ON where vector issue binds and a lane carries two rays whose chains interleave -- the fused
// fp64 spot kernel -4 ... -5 %, on the asphere system -1 ... -3 %; OFF in the record-all kernels,
// which are bound by their stores and ran 2.5-3.9 % SLOWER with it on a placed block (stores
// that issue faster collide more: the arithmetic-free fill GAINS 5 % when 8 FMAs are put in
// front of every store), and OFF with one ray per lane (fused OPD kernel: the one
// long dependent chain cost 7-9 % more than the eight instructions saved).
template <typename V, bool SHARE = false>
OL_DEV V curved_distance(typename Math<V>::scalar cv,
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
  V ta, tb;
  if constexpr (SHARE) {
    m::div2(C, q, q, A, ta, tb);
  } else {
    ta = m::div(C, q);
    tb = m::div(q, A);
  }
  // t_b is the reference's t1 iff -sgn(E) == sgn(R); the reference keeps t1 when
  // |z + t1 N| <= |z + t2 N| and t2 otherwise (also when the comparison is NaN)
  const auto b_is_t1 = m::same(m::lt(E, zero), m::all(cv > 0));
  const V t1 = m::select(b_is_t1, tb, ta), t2 = m::select(b_is_t1, ta, tb);
  const V z1 = m::abs(m::fma(t1, N, z)), z2 = m::abs(m::fma(t2, N, z));
  V t = m::select(m::le(z1, z2), t1, t2);
  t = m::select(m::eq(A, zero), ta, t);
  return t;
}

// OL_SURF_REFERENCE_ROOT (opt-in, per surface): the intersection in the reference's OWN form
// (standard.py:112-146) -- the quadratic in R-scaled coefficients, both roots from
// (-b +- sqrt(d)) / (2 a), nothing contracted into FMAs -- instead of the cancellation-free
// form above.  The two agree to rounding wherever the reference's formula is well conditioned;
// where it is not (|a| = |1 + k| << 1 with near-axial rays: a nearly parabolic mirror) the
// reference carries a SYSTEMATIC error of eps |b| / |2a| in t, and goldens generated with it
// encode that error: the Hubble on-axis `OPD_difference` of the reference's test_operand.py is
// 0.00132951 waves with this form and 0.00132994 with the stable one, against a tolerance of
// 1.1e-7.  (One-ulp changes of the inputs move it by 7e-9: the form matters, not the bits.)
// (R and k as given: with |a| << 1 even one ulp of R = 1 / cv moves t by eps |R| / |a|.)
template <typename V>
OL_DEV V reference_distance(typename Math<V>::scalar R, typename Math<V>::scalar k, V x, V y,
                            V z, V L, V M, V N) {
#pragma clang fp contract(off)
  using m = Math<V>;
  using T = typename m::scalar;
  const V NN = N * N, zz = z * z;
  const V a = ((m::splat(k) * NN + L * L) + M * M) + NN;
  const V b = ((((m::splat(T(2) * k) * N) * z + (m::splat(2) * L) * x) + (m::splat(2) * M) * y) -
               (m::splat(2) * N) * m::splat(R)) + (m::splat(2) * N) * z;
  const V c = (((m::splat(k) * zz - m::splat(T(2) * R) * z) + x * x) + y * y) + zz;
  const V d = b * b - (m::splat(4) * a) * c;
  const V sq = m::sqrt(d);  // NaN when the ray misses (standard.py:132-137)
  const V two_a = m::splat(2) * a;
  const V t1 = m::div(-b + sq, two_a), t2 = m::div(-b - sq, two_a);
  const V z1 = m::abs(z + t1 * N), z2 = m::abs(z + t2 * N);
  V t = m::select(m::le(z1, z2), t1, t2);
  t = m::select(m::eq(a, m::splat(0)), m::div(-c, b), t);
  return t;
}

template <typename T, bool SHARE = false>
OL_DEV T conic_distance(const DevSurf<T>& s, T x, T y, T z, T L, T M, T N) {
  if (s.flags & kSurfRadiusInf) return flat_distance(z, N);
  return curved_distance<T, SHARE>(s.cv, s.kp1, x, y, z, L, M, N);
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
OL_DEV void conic_normal(typename Math<V>::scalar cv,
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
OL_DEV void even_asphere_eval(const DevSurf<T>& s, cptr<T> c,
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
OL_DEV void odd_asphere_eval(const DevSurf<T>& s, cptr<T> c,
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
OL_DEV void polynomial_eval(const DevSurf<T>& s, cptr<T> c,
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

OL_DEV int slot_int(cptr<float> p) { return hw::float_bits(*p); }
OL_DEV int slot_int(cptr<double> p) { return (int)hw::double_bits(*p); }

// conic base, normalised coordinates and the range check shared by both series forms
template <typename T>
OL_DEV void zernike_begin(const DevSurf<T>& s, T x, T y, T& sag, T& fx, T& fy, T& xn, T& yn,
                          T& u, uint32_t& status) {
  using m = Math<T>;
  T r2 = m::fma(x, x, y * y);
  T g = m::sqrt(m::fma(-s.kp1 * s.cv * s.cv, r2, T(1)));
  sag = m::div(s.cv * r2, T(1) + g);
  T f = m::div(s.cv, g);
  fx = x * f;
  fy = y * f;
  const T inv = s.cold->inv_norm;
  xn = x * inv;
  yn = y * inv;
  if (m::abs(xn) > T(1) || m::abs(yn) > T(1)) status |= 0x1u;  // OL_STATUS_ZERNIKE_RANGE
  u = m::fma(xn, xn, yn * yn);
}

// (zsum, gx, gy): the series and its gradient w.r.t. (x_n, y_n) -> sag and sag gradient
template <typename T>
OL_DEV void zernike_finish(const DevSurf<T>& s, T xn, T yn, T u, T zsum, T gx, T gy, T& sag,
                           T& fx, T& fy) {
  using m = Math<T>;
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
  const T inv = s.cold->inv_norm;
  sag += zsum;
  fx = m::fma(gx, inv, fx);
  fy = m::fma(gy, inv, fy);
}

template <typename T>
OL_DEV void zernike_eval(const DevSurf<T>& s, cptr<T> c, T x,
                                             T y, T& sag, T& fx, T& fy, uint32_t& status) {
  using m = Math<T>;
  using V2 = vec2<T>;
  T xn, yn, u;
  zernike_begin(s, x, y, sag, fx, fy, xn, yn, u, status);
  const V2 uu = {u, u};

  T zsum = T(0), gx = T(0), gy = T(0);  // gradient w.r.t. (x_n, y_n)
  // H = (A_m, B_m) of the current order, Hp of the one below (weighted by m = 0 at order 0)
  V2 H = {T(1), T(0)}, Hp = {T(0), T(0)};
  int mcur = 0;
  cptr<T> p = c;
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
    cptr<T> e = p + kZernLevelStride * (K - 1);
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
  zernike_finish(s, xn, yn, u, zsum, gx, gy, sag, fx, fy);
}

// Low-order Zernike surface as one bivariate polynomial (kGeomZernikeMono, block layout in
// device_table.h, expansion in capi.hip:build_zernike_mono_block): the sag polynomial S and
// the gradient (dQ/dx_n, dQ/dy_n) of the normal's polynomial by nested Horner -- outer in
// x_n, inner in y_n -- straight from the coefficient stream.  One FMA per coefficient for S
// and one PACKED FMA per coefficient pair for the gradient (v_pk_fma_f32: both partial
// derivatives in one issue slot); no harmonic recurrence, no per-order epilogue.  For the
// 12 fringe terms of configuration C5 (degree 4): 15 + 10 FMAs against ~130 vector
// instructions of the level form.  Degrees 2..6 are compiled with the degree known (fully
// unrolled: the scalar loads of the whole block issue up front), the rest run the loops.
// OL_ZERN_MONO_SPLIT (default on): keeps the compiler from hoisting ALL of the gradient
// block's scalar loads above the sag chain -- with 35 coefficient SGPRs live at once at
// degree 4 the allocator spilled ~17 SGPRs to VGPR lanes (v_writelane / v_readlane, vector
// instructions) around every evaluation; split, the degree-4 block is 36 vector
// instructions and no spill.  A/B knob, tools/build_variants.py.
// OL_ZERN_MONO_SPLIT (default on): the coefficient stream of an unrolled instance is read
// through a WINDOW (CoeffWindow): chunks of 16 dwords, chunk k + OL_ZERN_MONO_AHEAD requested
// when the Horner chain starts to consume chunk k (the request is ordered after the chain's
// accumulator, device_table.h: refresh_after).  Without it the scheduler hoists every scalar
// load of the block to the top of the evaluation -- 35 (degree 4) to 70 (degree 6)
// coefficient values, twice as many SGPRs in fp64, on top of what the kernel holds anyway --
// and the allocator spills scalars to VGPR lanes (v_writelane / v_readlane: vector
// instructions) around every evaluation.  A/B knob, tools/build_variants.py.
#ifndef OL_ZERN_MONO_SPLIT
#define OL_ZERN_MONO_SPLIT 1
#endif
#ifndef OL_ZERN_MONO_AHEAD
#define OL_ZERN_MONO_AHEAD 1
#endif
#ifndef OL_ZERN_MONO_CHUNK
#define OL_ZERN_MONO_CHUNK 8  // dwords (SGPRs) per chunk
#endif
// OL_ZERN_MONO_FIXED = 0: no unrolled instances, every degree runs the loops (A/B knob)
#ifndef OL_ZERN_MONO_FIXED
#define OL_ZERN_MONO_FIXED 1
#endif

// Sequential reader of TOTAL coefficients for FULLY UNROLLED code: the element index `e`
// of every call is a constant after unrolling, so the chunk table is scalar-replaced and the
// request branches fold away.
template <typename T, int TOTAL, bool WINDOWED>
struct CoeffWindow {
  static constexpr int CH = 4 * OL_ZERN_MONO_CHUNK / sizeof(T);  // elements per chunk
  static constexpr int NCH = (TOTAL + CH - 1) / CH;
  cptr<T> base;
  cptr<T> chunk[NCH];
  OL_DEV explicit CoeffWindow(cptr<T> c) : base(c) {
    if constexpr (WINDOWED) {
#pragma unroll
      for (int j = 0; j < NCH; ++j) chunk[j] = c + j * CH;
#pragma unroll
      for (int j = 0; j < NCH && j < OL_ZERN_MONO_AHEAD; ++j) chunk[j] = refresh(c + j * CH);
    }
  }
  // element e; `acc` = the value the consuming chain has reached (orders the next request)
  template <typename A>
  OL_DEV T get(int e, A& acc) {
    if constexpr (WINDOWED) {
      const int j = e / CH, k = e % CH;
      if (k == 0 && j + OL_ZERN_MONO_AHEAD < NCH)
        chunk[j + OL_ZERN_MONO_AHEAD] =
            refresh_after(base + (j + OL_ZERN_MONO_AHEAD) * CH, acc);
      return chunk[j][k];
    } else {
      return base[e];
    }
  }
};

#define OL_ZERN_MONO_BODY(UNROLL, STREAM)                                      \
  using m = Math<T>;                                                           \
  using V2 = vec2<T>;                                                          \
  int e = 0;                                                                   \
  T P = T(0);                                                                  \
  UNROLL for (int i = N; i >= 0; --i) {                                        \
    T q = STREAM.get(e, P);                                                    \
    ++e;                                                                       \
    UNROLL for (int j = N - i - 1; j >= 0; --j) {                              \
      const T cj = STREAM.get(e, q);                                           \
      ++e;                                                                     \
      q = m::fma(q, yn, cj);                                                   \
    }                                                                          \
    P = i == N ? q : m::fma(P, xn, q);                                         \
  }                                                                            \
  const V2 xx = {xn, xn}, yy = {yn, yn};                                       \
  V2 G = {T(0), T(0)};                                                         \
  T tie = P;                                                                   \
  UNROLL for (int i = N - 1; i >= 0; --i) {                                    \
    V2 q;                                                                      \
    q.x = STREAM.get(e, tie);                                                  \
    q.y = STREAM.get(e + 1, tie);                                              \
    e += 2;                                                                    \
    UNROLL for (int j = N - 2 - i; j >= 0; --j) {                              \
      tie = q.x;                                                               \
      V2 cj;                                                                   \
      cj.x = STREAM.get(e, tie);                                               \
      cj.y = STREAM.get(e + 1, tie);                                           \
      e += 2;                                                                  \
      q.x = tie;                                                               \
      q = q * yy + cj;                                                         \
    }                                                                          \
    G = i == N - 1 ? q : G * xx + q;                                           \
    tie = G.x;                                                                 \
  }                                                                            \
  P = tie == tie ? P : P; /* (tie is only an ordering handle) */               \
  zsum = P;                                                                    \
  gx = G.x;                                                                    \
  gy = G.y;

// OL_ZERN_MONO_F32_TWO_BLOCKS (default on): the fp32 unrolled instances read the stream as
// TWO blocks -- the sag polynomial's coefficients, then, requested when the sag chain has
// finished, the gradient's -- instead of through the 8-dword window.  In fp32 one block is at
// most 15 + 20 SGPRs at degree 4 and the window's ordering handles cost vector moves in a
// kernel (C5) that is bound by vector issue: 732 -> 672 vector instructions per ray.  The fp64
// kernels, where a block is twice as wide, keep the window.
#ifndef OL_ZERN_MONO_F32_TWO_BLOCKS
#define OL_ZERN_MONO_F32_TWO_BLOCKS 1
#endif
template <typename T, int N>
OL_DEV void zernike_mono_two_blocks(cptr<T> c, T xn, T yn, T& zsum, T& gx, T& gy) {
  using m = Math<T>;
  using V2 = vec2<T>;
  constexpr int NS = (N + 1) * (N + 2) / 2;
  cptr<T> p = refresh(c);
  int e = 0;
  T P = T(0);
#pragma unroll
  for (int i = N; i >= 0; --i) {
    T q = p[e++];
#pragma unroll
    for (int j = N - i - 1; j >= 0; --j) q = m::fma(q, yn, p[e++]);
    P = i == N ? q : m::fma(P, xn, q);
  }
  cptr<T> g = refresh_after(c + NS, P);  // the gradient block: not before the sag chain ends
  e = 0;
  const V2 xx = {xn, xn}, yy = {yn, yn};
  V2 G = {T(0), T(0)};
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    V2 q = {g[e], g[e + 1]};
    e += 2;
#pragma unroll
    for (int j = N - 2 - i; j >= 0; --j) {
      const V2 cj = {g[e], g[e + 1]};
      e += 2;
      q = q * yy + cj;
    }
    G = i == N - 1 ? q : G * xx + q;
  }
  zsum = P;
  gx = G.x;
  gy = G.y;
}

template <typename T, int N>
OL_DEV void zernike_mono_fixed(cptr<T> c, T xn, T yn, T& zsum, T& gx, T& gy) {
  if constexpr (OL_ZERN_MONO_F32_TWO_BLOCKS && OL_ZERN_MONO_SPLIT && sizeof(T) == 4) {
    zernike_mono_two_blocks<T, N>(c, xn, yn, zsum, gx, gy);
  } else {
    constexpr int TOTAL = (N + 1) * (N + 2) / 2 + N * (N + 1);
    CoeffWindow<T, TOTAL, OL_ZERN_MONO_SPLIT != 0> stream(c);
    OL_ZERN_MONO_BODY(_Pragma("unroll"), stream)
  }
}
#undef OL_ZERN_MONO_BODY

// Degrees without an unrolled instance: the same nested Horner, one ROW per loop trip.  A
// row's coefficients (at most OL_ZERN_MONO_MAX_DEG + 1 values, or that many pairs) are
// consecutive in the stream and are requested together -- the row length is wave-uniform, so
// each length has its own unrolled chain -- and at most one row occupies SGPRs.
constexpr int kZernMonoMaxRow = 9;  // radial order <= 8 (capi.hip: kZernMonoMaxOrder)

template <typename T, int LEN>
OL_DEV T mono_row(cptr<T> p, T yn) {
  using m = Math<T>;
  T q = p[0];
#pragma unroll
  for (int j = 1; j < LEN; ++j) q = m::fma(q, yn, p[j]);
  return q;
}
template <typename T, int LEN>
OL_DEV vec2<T> mono_row2(cptr<T> p, vec2<T> yy) {
  vec2<T> q = {p[0], p[1]};
#pragma unroll
  for (int j = 1; j < LEN; ++j) {
    const vec2<T> cj = {p[2 * j], p[2 * j + 1]};
    q = q * yy + cj;
  }
  return q;
}

template <typename T>
OL_DEV void zernike_mono_loop(cptr<T> c, int N, T xn, T yn, T& zsum, T& gx, T& gy) {
  using m = Math<T>;
  using V2 = vec2<T>;
  cptr<T> p = c;
  T P = T(0);
  for (int len = 1; len <= N + 1; ++len) {  // rows by descending power of x
    T q;
    switch (len) {
#define OL_ROW(L) case L: q = mono_row<T, L>(p, yn); break;
      OL_ROW(1) OL_ROW(2) OL_ROW(3) OL_ROW(4) OL_ROW(5) OL_ROW(6) OL_ROW(7) OL_ROW(8)
#undef OL_ROW
      default: q = mono_row<T, kZernMonoMaxRow>(p, yn); break;
    }
    p = refresh(p + len);
    P = len == 1 ? q : m::fma(P, xn, q);
  }
  const V2 xx = {xn, xn}, yy = {yn, yn};
  V2 G = {T(0), T(0)};
  for (int len = 1; len <= N; ++len) {
    V2 q;
    switch (len) {
#define OL_ROW(L) case L: q = mono_row2<T, L>(p, yy); break;
      OL_ROW(1) OL_ROW(2) OL_ROW(3) OL_ROW(4) OL_ROW(5) OL_ROW(6) OL_ROW(7)
#undef OL_ROW
      default: q = mono_row2<T, kZernMonoMaxRow - 1>(p, yy); break;
    }
    p = refresh(p + 2 * len);
    G = len == 1 ? q : G * xx + q;
  }
  zsum = P;
  gx = G.x;
  gy = G.y;
}

template <typename T>
OL_DEV void zernike_mono_eval(const DevSurf<T>& s, cptr<T> c, T x, T y, T& sag,
                              T& fx, T& fy, uint32_t& status) {
  T xn, yn, u, zsum, gx, gy;
  zernike_begin(s, x, y, sag, fx, fy, xn, yn, u, status);
  switch (OL_ZERN_MONO_FIXED ? s.n_coeff : 0) {  // wave-uniform
    case 2: zernike_mono_fixed<T, 2>(c, xn, yn, zsum, gx, gy); break;
    case 3: zernike_mono_fixed<T, 3>(c, xn, yn, zsum, gx, gy); break;
    case 4: zernike_mono_fixed<T, 4>(c, xn, yn, zsum, gx, gy); break;
    default: zernike_mono_loop<T>(c, s.n_coeff, xn, yn, zsum, gx, gy); break;
  }
  zernike_finish(s, xn, yn, u, zsum, gx, gy, sag, fx, fy);
}

// ---- the polynomial form on a PAIR of fp32 rays (Math<f32x2>) ------------------------------
// Configuration C5 on two rays per lane (trace_kernel.hip: OL_POLZ_PAIR).  Element for element
// the operations of the scalar functions above, in the same order -- the same bits -- with the
// multiply-adds of the two rays in ONE v_pk_fma_f32 / v_pk_mul_f32 each (square roots,
// reciprocals, compares and selects stay one instruction per ray: there are no packed forms).
// A branch on a ray's own values becomes a select; wave-uniform branches stay branches.  The
// launcher sends only ranges whose Zernike surfaces all have the polynomial form here
// (capi.hip: pair_polz_ok).
OL_DEV void zernike_begin(const DevSurf<float>& s, f32x2 x, f32x2 y, f32x2& sag, f32x2& fx,
                          f32x2& fy, f32x2& xn, f32x2& yn, f32x2& u, Mask2& out_of_range) {
  using m = Math<f32x2>;
  f32x2 r2 = m::fma(x, x, y * y);
  f32x2 g = m::sqrt(m::fma(m::splat(-s.kp1 * s.cv * s.cv), r2, m::splat(1)));
  sag = m::div(s.cv * r2, 1.0f + g);
  f32x2 f = m::div(m::splat(s.cv), g);
  fx = x * f;
  fy = y * f;
  const float inv = s.cold->inv_norm;
  xn = x * inv;
  yn = y * inv;
  out_of_range = m::mor(m::gt(m::abs(xn), m::splat(1)), m::gt(m::abs(yn), m::splat(1)));
  u = m::fma(xn, xn, yn * yn);
}

OL_DEV void zernike_finish(const DevSurf<float>& s, f32x2 xn, f32x2 yn, f32x2 u, f32x2 zsum,
                           f32x2 gx, f32x2 gy, f32x2& sag, f32x2& fx, f32x2& fy) {
  using m = Math<f32x2>;
  const Mask2 near = m::lt(u, m::splat(1e-8f));
  if (m::any(near)) {  // the reference's eps-regularised chain rule near / at the vertex
    const float eps = m::guard();
    const f32x2 Rr = m::fma(xn, gx, yn * gy);
    const f32x2 Az = m::fma(xn, gy, -(yn * gx));
    const f32x2 rho = m::sqrt(u);
    const f32x2 d1 = m::select(m::gt(u, m::splat(0)), m::rcp(m::fma(m::splat(eps), rho, u)),
                               m::splat(0));
    const f32x2 d2 = m::rcp(u + eps);
    const f32x2 hx = m::fma(Rr * d1, xn, -(Az * d2 * yn));
    const f32x2 hy = m::fma(Rr * d1, yn, Az * d2 * xn);
    gx = m::select(near, hx, gx);
    gy = m::select(near, hy, gy);
  }
  const float inv = s.cold->inv_norm;
  sag += zsum;
  fx = m::fma(gx, m::splat(inv), fx);
  fy = m::fma(gy, m::splat(inv), fy);
}

// (the gradient chains run per partial derivative ACROSS the two rays -- coefficient j of
// d/dx_n, then of d/dy_n, each a broadcast scalar -- where the scalar form packs the two
// partial derivatives of ONE ray: the same multiply-adds element for element)
template <int N>
OL_DEV void zernike_mono_two_blocks(cptr<float> c, f32x2 xn, f32x2 yn, f32x2& zsum, f32x2& gx,
                                    f32x2& gy) {
  using m = Math<f32x2>;
  constexpr int NS = (N + 1) * (N + 2) / 2;
  cptr<float> p = refresh(c);
  int e = 0;
  f32x2 P = m::splat(0);
#pragma unroll
  for (int i = N; i >= 0; --i) {
    f32x2 q = m::splat(p[e++]);
#pragma unroll
    for (int j = N - i - 1; j >= 0; --j) q = m::fma(q, yn, m::splat(p[e++]));
    P = i == N ? q : m::fma(P, xn, q);
  }
  cptr<float> g = refresh_after(c + NS, P);  // the gradient block: not before the sag chain ends
  e = 0;
  f32x2 Gx = m::splat(0), Gy = m::splat(0);
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    f32x2 qx = m::splat(g[e]), qy = m::splat(g[e + 1]);
    e += 2;
#pragma unroll
    for (int j = N - 2 - i; j >= 0; --j) {
      qx = qx * yn + m::splat(g[e]);
      qy = qy * yn + m::splat(g[e + 1]);
      e += 2;
    }
    Gx = i == N - 1 ? qx : Gx * xn + qx;
    Gy = i == N - 1 ? qy : Gy * xn + qy;
  }
  zsum = P;
  gx = Gx;
  gy = Gy;
}

template <int LEN>
OL_DEV f32x2 mono_row(cptr<float> p, f32x2 yn) {
  using m = Math<f32x2>;
  f32x2 q = m::splat(p[0]);
#pragma unroll
  for (int j = 1; j < LEN; ++j) q = m::fma(q, yn, m::splat(p[j]));
  return q;
}
template <int LEN>
OL_DEV void mono_row2(cptr<float> p, f32x2 yn, f32x2& qx, f32x2& qy) {
  using m = Math<f32x2>;
  qx = m::splat(p[0]);
  qy = m::splat(p[1]);
#pragma unroll
  for (int j = 1; j < LEN; ++j) {
    qx = qx * yn + m::splat(p[2 * j]);
    qy = qy * yn + m::splat(p[2 * j + 1]);
  }
}

OL_DEV void zernike_mono_loop(cptr<float> c, int N, f32x2 xn, f32x2 yn, f32x2& zsum, f32x2& gx,
                              f32x2& gy) {
  using m = Math<f32x2>;
  cptr<float> p = c;
  f32x2 P = m::splat(0);
  for (int len = 1; len <= N + 1; ++len) {  // rows by descending power of x
    f32x2 q;
    switch (len) {
#define OL_ROW(L) case L: q = mono_row<L>(p, yn); break;
      OL_ROW(1) OL_ROW(2) OL_ROW(3) OL_ROW(4) OL_ROW(5) OL_ROW(6) OL_ROW(7) OL_ROW(8)
#undef OL_ROW
      default: q = mono_row<kZernMonoMaxRow>(p, yn); break;
    }
    p = refresh(p + len);
    P = len == 1 ? q : m::fma(P, xn, q);
  }
  f32x2 Gx = m::splat(0), Gy = m::splat(0);
  for (int len = 1; len <= N; ++len) {
    f32x2 qx, qy;
    switch (len) {
#define OL_ROW(L) case L: mono_row2<L>(p, yn, qx, qy); break;
      OL_ROW(1) OL_ROW(2) OL_ROW(3) OL_ROW(4) OL_ROW(5) OL_ROW(6) OL_ROW(7)
#undef OL_ROW
      default: mono_row2<kZernMonoMaxRow - 1>(p, yn, qx, qy); break;
    }
    p = refresh(p + 2 * len);
    Gx = len == 1 ? qx : Gx * xn + qx;
    Gy = len == 1 ? qy : Gy * xn + qy;
  }
  zsum = P;
  gx = Gx;
  gy = Gy;
}

// `report`: which rays of the pair may raise OL_STATUS_ZERNIKE_RANGE (the active ones)
OL_DEV void zernike_mono_eval(const DevSurf<float>& s, cptr<float> c, f32x2 x, f32x2 y, f32x2& sag,
                              f32x2& fx, f32x2& fy, Mask2 report, uint32_t& status) {
  using m = Math<f32x2>;
  f32x2 xn, yn, u, zsum, gx, gy;
  Mask2 out;
  zernike_begin(s, x, y, sag, fx, fy, xn, yn, u, out);
  if (m::any(m::mand(out, report))) status |= 0x1u;  // OL_STATUS_ZERNIKE_RANGE
  // (the unrolled instances exist in the two-block form only; A/B builds that switch that form
  // off run the loops -- the same multiply-adds in the same order)
  constexpr bool kFixed = OL_ZERN_MONO_FIXED && OL_ZERN_MONO_F32_TWO_BLOCKS && OL_ZERN_MONO_SPLIT;
  switch (kFixed ? s.n_coeff : 0) {  // wave-uniform
    case 2: zernike_mono_two_blocks<2>(c, xn, yn, zsum, gx, gy); break;
    case 3: zernike_mono_two_blocks<3>(c, xn, yn, zsum, gx, gy); break;
    case 4: zernike_mono_two_blocks<4>(c, xn, yn, zsum, gx, gy); break;
    default: zernike_mono_loop(c, s.n_coeff, xn, yn, zsum, gx, gy); break;
  }
  zernike_finish(s, xn, yn, u, zsum, gx, gy, sag, fx, fy);
}

// chebyshev.py:126-225.  T_n by the three-term recurrence instead of
// cos(n acos x); T_n'(x) = n U_{n-1}(x) instead of n sin(n acos x)/sqrt(1-x^2)
// (identical for |x| < 1; at |x| == 1 the reference divides by zero).  As in the
// reference the derivative is taken w.r.t. the NORMALISED coordinate and is not
// divided by norm_x / norm_y (chebyshev.py:176-186).
template <typename T>
OL_DEV void chebyshev_eval(const DevSurf<T>& s, cptr<T> c, T x,
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
  cptr<T> grid = c + 2;
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
OL_DEV void biconic_eval(const DevSurf<T>& s, cptr<T> c, T x,
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
OL_DEV void toroidal_eval(const DevSurf<T>& s, cptr<T> c, T x,
                                              T y, T& sag, T& fx, T& fy) {
  using m = Math<T>;
  const T R = c[0], invR = c[1], k1 = c[2], cyz = c[3];
  cptr<T> a = c + 4;
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

// Which sag functors a kernel instantiation carries (template parameter NR of surface_step
// and of the kernels).  The generic Newton kernel compiles all nine in: it is the kernel
// that is shortest of scalar registers -- on MI355X ~250 of its vector instructions per
// ray were SGPR spills to VGPR lanes (v_writelane / v_readlane) and the branch flags of
// the nine-way dispatch, BEFORE any Newton arithmetic (profiles/r02_nr_kernel_overhead.txt).
// A traced range whose Newton surfaces all belong to one family runs an instantiation with
// that family's functors only (capi.hip:newton_family picks it per launch).
// (the constants kNrNone ... kNrReference: device_table.h -- the launchers' callers name them)

template <int NR = kNrGeneric, typename T>
OL_DEV void nr_eval(const DevSurf<T>& s, cptr<T> c, T x, T y,
                                        T& sag, T& fx, T& fy, uint32_t& status) {
  if constexpr (NR == kNrZernike) {
    if (s.geom == kGeomZernikeMono) zernike_mono_eval(s, c, x, y, sag, fx, fy, status);
    else zernike_eval(s, c, x, y, sag, fx, fy, status);
    return;
  }
  if constexpr (NR == kNrEvenAsphere) {
    even_asphere_eval(s, c, x, y, sag, fx, fy);
    return;
  }
  switch (s.geom) {
    case kGeomEvenAsphere: even_asphere_eval(s, c, x, y, sag, fx, fy); break;
    case kGeomOddAsphere: odd_asphere_eval(s, c, x, y, sag, fx, fy); break;
    case kGeomPolynomial: polynomial_eval(s, c, x, y, sag, fx, fy); break;
    case kGeomChebyshev: chebyshev_eval(s, c, x, y, sag, fx, fy, status); break;
    case kGeomBiconic: biconic_eval(s, c, x, y, sag, fx, fy); break;
    case kGeomToroidal: toroidal_eval(s, c, x, y, sag, fx, fy); break;
    case kGeomZernikeMono: zernike_mono_eval(s, c, x, y, sag, fx, fy, status); break;
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
//    halves AT THE ROUNDING FLOOR of sag - z (fp32 with tol below that noise: within
//    OL_NR_STALL_ULPS eps of |sag| + |z|) leaves too instead of spinning to
//    max_iter.  (Until round 5 the rule had no floor: a ray that starts badly -- the rim
//    of an oblate biconic mirror, residuals 4.0, 2.3, 0.97, 0.13, 2e-3, 7e-7 mm -- was cut
//    off after its second step with 1 mm of residual where the reference converges;
//    fuzz_7016 of profiles/r05_gpu_fuzz_tables.txt.)  A ray that neither converges nor
//    reaches the floor runs to max_iter, as the reference's does;
//  * the gradient of the LAST evaluation is returned and reused for the surface
//    normal: the hit point moved by |f|/|f'| < tol since, which changes the
//    normal by < curvature * tol.
// Returns t = t0 + dt and leaves the hit point in (x, y, z).
template <typename T>
struct NewtonRay {
  T xb, yb, zb, dt, fprev, gx, gy;
  typename Math<T>::mask active;
};

template <int NR = kNrGeneric, typename T>
OL_DEV void newton_iterate(const DevSurf<T>& s, cptr<T> c,
                                               NewtonRay<T>& q, T L, T M, T N, int it,
                                               uint32_t& status) {
  using m = Math<T>;
  T xi = m::fma(q.dt, L, q.xb), yi = m::fma(q.dt, M, q.yb), zi = m::fma(q.dt, N, q.zb);
  T sag, fx, fy;
  nr_eval<NR>(s, c, xi, yi, sag, fx, fy, status);
  T f = sag - zi;
  T af = m::abs(f);
  bool done = !(af >= s.cold->tol);                       // converged, or NaN
#ifndef OL_NR_STALL_ULPS
#define OL_NR_STALL_ULPS 1024  // (0: the rule of rounds 1-4, no floor -- A/B only)
#endif
  // residual stopped halving, and that close to the noise of the subtraction
  const bool at_floor = OL_NR_STALL_ULPS == 0 ||
      !(af > T(OL_NR_STALL_ULPS) * m::eps() * (m::abs(sag) + m::abs(zi)));
  done = done || (it > 0 && at_floor && !(af < T(0.5) * q.fprev));
  T df = m::fma(fx, L, m::fma(fy, M, -N));
  T dfs = m::abs(df) > m::guard() ? df : m::guard();
  q.dt = q.dt - m::div(f, dfs);
  q.fprev = af;
  q.gx = fx;
  q.gy = fy;
  q.active = !done;
}

// One Newton step of a PAIR of fp32 rays on a polynomial-form Zernike surface (see the pair forms
// of the evaluation above): both rays are evaluated, and what the scalar form skips for a ray
// that has left the iteration is a select on `active` here -- per ray the same values.
OL_DEV void newton_iterate(const DevSurf<float>& s, cptr<float> c, NewtonRay<f32x2>& q, f32x2 L,
                           f32x2 M, f32x2 N, int it, uint32_t& status) {
  using m = Math<f32x2>;
  f32x2 xi = m::fma(q.dt, L, q.xb), yi = m::fma(q.dt, M, q.yb), zi = m::fma(q.dt, N, q.zb);
  f32x2 sag, fx, fy;
  zernike_mono_eval(s, c, xi, yi, sag, fx, fy, q.active, status);
  f32x2 f = sag - zi;
  f32x2 af = m::abs(f);
  Mask2 done = m::mnot(m::ge(af, m::splat(s.cold->tol)));  // converged, or NaN
  const Mask2 at_floor =
      OL_NR_STALL_ULPS == 0
          ? m::all(true)
          : m::mnot(m::gt(af, (float(OL_NR_STALL_ULPS) * m::eps()) * (m::abs(sag) + m::abs(zi))));
  done = m::mor(done, m::mand(m::all(it > 0), m::mand(at_floor, m::mnot(m::lt(af, 0.5f * q.fprev)))));
  f32x2 df = m::fma(fx, L, m::fma(fy, M, -N));
  f32x2 dfs = m::select(m::gt(m::abs(df), m::splat(m::guard())), df, m::splat(m::guard()));
  const f32x2 dt = q.dt - m::div(f, dfs);
  q.dt = m::select(q.active, dt, q.dt);
  q.fprev = m::select(q.active, af, q.fprev);
  q.gx = m::select(q.active, fx, q.gx);
  q.gy = m::select(q.active, fy, q.gy);
  q.active = m::mand(q.active, m::mnot(done));
}

// OL_SURF_REFERENCE_NEWTON (opt-in, ABI 11; kernel family kNrReference): the reference's OWN
// stop rule, for users who need its NUMBERS rather than the converged intersection.
// newton_raphson.py:137-166 iterates the WHOLE batch in lockstep,
//     for _ in range(max_iter):  f = sag(P + t D) - z;  if max_j |f_j| < tol: break;  t -= f / f'
// so every ray of a trace call takes the SAME number K of updates: the first k at which all rays
// are below tol, or max_iter -- also when any ray of the batch is NaN (be.max of an array with a
// NaN is NaN, `NaN < tol` is False: a reference quirk, kept).  The surface normal is then
// evaluated at the end point (standard_surface.py: `surface_normal(rays)` after the propagation),
// not at the point of the last evaluation.  With the factory tolerance (1e-6) the per-ray rule
// above lands within 1e-7 of this; with a user-set loose tolerance, or in what OPD / PSF
// consumers make of 1e-7, the difference shows.
// K is a property of the batch, so it is found by launches of its own (ol_newton_count): the
// launch that has `count_at == surface` runs every ray to ITS first k with |f_k| < tol (NaN or
// never: max_iter) and takes the maximum over the batch into iters[surface] -- a lower bound of
// K that is K itself unless a ray that was below tol is above it again later (rounding noise of
// the order of tol); every other launch takes exactly iters[surface] updates and, when that is
// less than max_iter, checks the rule at the end point: a ray that is NOT below tol there raises
// iters[n_surf + surface], and the host moves K up by one and asks again (engine.py).
struct NrRefCtl {
  int32_t* iters;    // [2 * n_surf]: K per surface, then the "rule violated at K" words
  int32_t surface;   // the surface being traced
  int32_t n_surf;
  int32_t count_at;  // the surface whose K this launch determines, or -1
};

template <typename T>
OL_DEV void newton_reference(const DevSurf<T>& s, cptr<T> c, const NrRefCtl& ctl, T t0,
                             Ray<T>& r, T& t, T& gx, T& gy, uint32_t& status) {
  using m = Math<T>;
  const T L = r.L, M = r.M, N = r.N;
  const T xb = m::fma(t0, L, r.x), yb = m::fma(t0, M, r.y), zb = m::fma(t0, N, r.z);
  const T tol = s.cold->tol;
  const int max_iter = s.max_iter;
  const bool counting = ctl.surface == ctl.count_at;
  int K = max_iter;
  if (!counting) {
    K = hw::load_i32(ctl.iters + ctl.surface);
    K = K < 0 ? 0 : (K > max_iter ? max_iter : K);
  }
  T dt = T(0), sag, fx, fy;
  int mine = max_iter;  // counting: this ray's first k with |f_k| < tol
  bool active = true;
  for (int it = 0; it < K; ++it) {
    if (counting && !hw::wave_any(active)) break;
    if (active) {
      const T xi = m::fma(dt, L, xb), yi = m::fma(dt, M, yb), zi = m::fma(dt, N, zb);
      nr_eval<kNrReference>(s, c, xi, yi, sag, fx, fy, status);
      const T f = sag - zi;
      if (counting) {
        if (m::abs(f) < tol) {
          mine = it;       // the batch may stop here as far as this ray is concerned
          active = false;
        } else if (f != f) {
          active = false;  // NaN now, NaN for good: the batch runs to max_iter
        }
      }
      if (active) {
        const T df = m::fma(fx, L, m::fma(fy, M, -N));
        const T dfs = m::abs(df) > m::guard() ? df : m::guard();
        dt = dt - m::div(f, dfs);
      }
    }
  }
  if (counting) {
    const int all = hw::wave_max(mine);
    if (hw::wave_leader()) hw::atomic_max_i32(ctl.iters + ctl.surface, all);
  }
  r.x = m::fma(dt, L, xb);
  r.y = m::fma(dt, M, yb);
  r.z = m::fma(dt, N, zb);
  t = t0 + dt;
  // the normal at the END point -- and, for a K below max_iter, the rule that stopped the batch
  nr_eval<kNrReference>(s, c, r.x, r.y, sag, gx, gy, status);
  if (!counting && K < max_iter) {
    const bool bad = !(m::abs(sag - r.z) < tol);
    if (hw::wave_any(bad) && hw::wave_leader())
      hw::atomic_or_i32(ctl.iters + ctl.n_surf + ctl.surface, 1);
  }
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
__device__ __forceinline__ void newton_compacted(const DevSurf<T>& s, cptr<T> c,
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
OL_DEV bool polygon_contains(cptr<T> v, int nv, T tx, T ty) {
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
OL_DEV bool leaf_contains(int kind, cptr<T> ap,
                                              cptr<T> coeffs, T x, T y) {
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
OL_DEV bool aperture_contains(const DevSurf<T>& s,
                                                  cptr<T> coeffs, T x, T y) {
  if (s.aperture_kind != kApComposite)
    return leaf_contains<T, FULL>(s.aperture_kind, s.cold->ap, coeffs, x, y);
  cptr<T> tok = coeffs + s.cold->ap_off;
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
OL_DEV PolBasis<T> pol_basis(T k0x, T k0y, T k0z, T k1x, T k1y, T k1z, T nx,
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
OL_DEV Jones<T> axis_jones(const PolBasis<T>& b, cptr<T> axis,
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
OL_DEV void prt_apply(Prt<T, POLK>& P, const PolBasis<T>& b, T k0x, T k0y,
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
// OL_PRT_PACKED (default on, fp32): the two updates below on PAIRS of matrix elements --
// columns 0 and 1 of every row, and rows 0 and 1 of column 2 -- so that two of every three
// multiply-adds issue as one v_pk_fma_f32 (v_pk_mul_f32): 57 -> 35 vector instructions for the
// rank-2 update, 26 -> 16 for the first one (tools/phase_costs.py), in kernels that are bound by
// vector issue.  Element for element the same operations in the same order as the scalar form
// (which fp64, with no packed instructions, keeps): the same bits.  A/B knob.
#ifndef OL_PRT_PACKED
#define OL_PRT_PACKED 1
#endif
template <typename T>
OL_DEV vec2<T> fma2(vec2<T> a, vec2<T> b, vec2<T> c) {
  return __builtin_elementwise_fma(a, b, c);
}
template <typename T>
OL_DEV vec2<T> splat2(T v) {
  return vec2<T>{v, v};
}

template <typename T, int POLK>
OL_DEV void prt_apply_diag(Prt<T, POLK>& P, const PolBasis<T>& b, T k0x,
                                               T k0y, T k0z, T k1x, T k1y, T k1z, T j0, T j1,
                                               T j2) {
  using m = Math<T>;
  constexpr int NP = POLK == 2 ? 2 : 1;
  if constexpr (OL_PRT_PACKED && sizeof(T) == 4) {
    using V2 = vec2<T>;
    const V2 p0xy = {b.p0x, b.p0y}, p1xy = {b.p1x, b.p1y}, k0xy = {k0x, k0y}, k1xy = {k1x, k1y};
    const V2 J0 = splat2(j0);
    const V2 axy = fma2(splat2(j1), p1xy, -(J0 * p0xy));
    const V2 bxy = fma2(splat2(j2), k1xy, -(J0 * k0xy));
    const T az = m::fma(j1, b.p1z, -(j0 * b.p0z)), bz = m::fma(j2, k1z, -(j0 * k0z));
    const V2 P0x = splat2(b.p0x), P0y = splat2(b.p0y), P0z = splat2(b.p0z);
    const V2 K0x = splat2(k0x), K0y = splat2(k0y), K0z = splat2(k0z);
#pragma unroll
    for (int c = 0; c < NP; ++c) {
      T* Q = P.m + 9 * c;
      const V2 q0 = {Q[0], Q[1]}, q1 = {Q[3], Q[4]}, q2 = {Q[6], Q[7]}, c2 = {Q[2], Q[5]};
      const V2 r1 = fma2(P0x, q0, fma2(P0y, q1, P0z * q2));
      const V2 r2 = fma2(K0x, q0, fma2(K0y, q1, K0z * q2));
      const T r1z = m::fma(b.p0x, Q[2], m::fma(b.p0y, Q[5], b.p0z * Q[8]));
      const T r2z = m::fma(k0x, Q[2], m::fma(k0y, Q[5], k0z * Q[8]));
      const V2 n0 = fma2(splat2(axy.x), r1, fma2(splat2(bxy.x), r2, J0 * q0));
      const V2 n1 = fma2(splat2(axy.y), r1, fma2(splat2(bxy.y), r2, J0 * q1));
      const V2 n2 = fma2(splat2(az), r1, fma2(splat2(bz), r2, J0 * q2));
      const V2 nc = fma2(axy, splat2(r1z), fma2(bxy, splat2(r2z), J0 * c2));
      Q[8] = m::fma(az, r1z, m::fma(bz, r2z, j0 * Q[8]));
      Q[0] = n0.x; Q[1] = n0.y; Q[3] = n1.x; Q[4] = n1.y; Q[6] = n2.x; Q[7] = n2.y;
      Q[2] = nc.x; Q[5] = nc.y;
    }
    return;
  }
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
OL_DEV void prt_first_diag(Prt<T, POLK>& P, const PolBasis<T>& b, T k0x,
                                               T k0y, T k0z, T k1x, T k1y, T k1z, T j0, T j1,
                                               T j2) {
  using m = Math<T>;
  if constexpr (OL_PRT_PACKED && sizeof(T) == 4) {
    using V2 = vec2<T>;
    const V2 p0xy = {b.p0x, b.p0y}, p1xy = {b.p1x, b.p1y}, k0xy = {k0x, k0y}, k1xy = {k1x, k1y};
    const V2 J0 = splat2(j0);
    const V2 axy = fma2(splat2(j1), p1xy, -(J0 * p0xy));
    const V2 bxy = fma2(splat2(j2), k1xy, -(J0 * k0xy));
    const T az = m::fma(j1, b.p1z, -(j0 * b.p0z)), bz = m::fma(j2, k1z, -(j0 * k0z));
    // rows 0..2, columns (0, 1): a_i p0_xy + b_i k0_xy + (j0 on the diagonal)
    const V2 n0 = fma2(splat2(axy.x), p0xy, fma2(splat2(bxy.x), k0xy, V2{j0, T(0)}));
    const V2 n1 = fma2(splat2(axy.y), p0xy, fma2(splat2(bxy.y), k0xy, V2{T(0), j0}));
    const V2 n2 = fma2(splat2(az), p0xy, fma2(splat2(bz), k0xy, V2{T(0), T(0)}));
    // column 2, rows (0, 1): a_xy p0_z + b_xy k0_z
    const V2 nc = fma2(axy, splat2(b.p0z), fma2(bxy, splat2(k0z), V2{T(0), T(0)}));
    P.m[8] = m::fma(az, b.p0z, m::fma(bz, k0z, j0));
    P.m[0] = n0.x; P.m[1] = n0.y; P.m[3] = n1.x; P.m[4] = n1.y; P.m[6] = n2.x; P.m[7] = n2.y;
    P.m[2] = nc.x; P.m[5] = nc.y;
  } else {
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
  }
  if constexpr (POLK == 2) {
#pragma unroll
    for (int e = 9; e < 18; ++e) P.m[e] = P.m[e] * T(0) + (P.m[0] * T(0));  // 0, NaN kept
  }
}

// ---- the polarised update of a PAIR of fp32 rays (Math<f32x2>; real diagonal Jones matrices) --
// pol_basis / the Fresnel amplitudes / prt_first_diag / prt_apply_diag as above, element for
// element in the same order (the rank-2 updates in the form the scalar code states for fp64,
// which its packed fp32 form reproduces bit for bit): every multiply-add of the two rays is one
// packed instruction, 27 + 6 per update and pair instead of 2 x 35.
OL_DEV PolBasis<f32x2> pol_basis(f32x2 k0x, f32x2 k0y, f32x2 k0z, f32x2 k1x, f32x2 k1y, f32x2 k1z,
                                 f32x2 nx, f32x2 ny, f32x2 nz) {
  using m = Math<f32x2>;
  const f32x2 zero = m::splat(0);
  f32x2 sx = k0y * nz - k0z * ny, sy = k0z * nx - k0x * nz, sz = k0x * ny - k0y * nx;
  {
    f32x2 proj = m::fma(sx, k0x, m::fma(sy, k0y, sz * k0z));
    sx = m::fma(-proj, k0x, sx);
    sy = m::fma(-proj, k0y, sy);
    sz = m::fma(-proj, k0z, sz);
  }
  f32x2 mag2 = m::fma(sx, sx, m::fma(sy, sy, sz * sz));
  const Mask2 normal = m::eq(mag2, zero);
  if (m::any(normal)) {
    // normal incidence: polarized_rays.py:153-166 fallback axes
    f32x2 px = zero, py = k0z, pz = -k0y;
    const Mask2 second = m::mand(m::eq(py, zero), m::eq(pz, zero));
    px = m::select(second, -k0z, px);
    py = m::select(second, zero, py);
    pz = m::select(second, k0x, pz);
    const f32x2 fx = py * k0z - pz * k0y, fy = pz * k0x - px * k0z, fz = px * k0y - py * k0x;
    sx = m::select(normal, fx, sx);
    sy = m::select(normal, fy, sy);
    sz = m::select(normal, fz, sz);
    mag2 = m::select(normal, m::fma(fx, fx, m::fma(fy, fy, fz * fz)), mag2);
  }
  f32x2 im = m::rsqrt(mag2);
  PolBasis<f32x2> b;
  b.sx = sx * im;
  b.sy = sy * im;
  b.sz = sz * im;
  b.p0x = k0y * b.sz - k0z * b.sy; b.p0y = k0z * b.sx - k0x * b.sz; b.p0z = k0x * b.sy - k0y * b.sx;
  b.p1x = k1y * b.sz - k1z * b.sy; b.p1y = k1z * b.sx - k1x * b.sz; b.p1z = k1x * b.sy - k1y * b.sx;
  return b;
}

OL_DEV void prt_apply_diag(Prt<f32x2, 1>& P, const PolBasis<f32x2>& b, f32x2 k0x, f32x2 k0y,
                           f32x2 k0z, f32x2 k1x, f32x2 k1y, f32x2 k1z, f32x2 j0, f32x2 j1,
                           f32x2 j2) {
  using m = Math<f32x2>;
  const f32x2 ax = m::fma(j1, b.p1x, -(j0 * b.p0x)), ay = m::fma(j1, b.p1y, -(j0 * b.p0y)),
              az = m::fma(j1, b.p1z, -(j0 * b.p0z));
  const f32x2 bx = m::fma(j2, k1x, -(j0 * k0x)), by = m::fma(j2, k1y, -(j0 * k0y)),
              bz = m::fma(j2, k1z, -(j0 * k0z));
  f32x2* Q = P.m;
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const f32x2 r1 = m::fma(b.p0x, Q[e], m::fma(b.p0y, Q[3 + e], b.p0z * Q[6 + e]));
    const f32x2 r2 = m::fma(k0x, Q[e], m::fma(k0y, Q[3 + e], k0z * Q[6 + e]));
    Q[e] = m::fma(ax, r1, m::fma(bx, r2, j0 * Q[e]));
    Q[3 + e] = m::fma(ay, r1, m::fma(by, r2, j0 * Q[3 + e]));
    Q[6 + e] = m::fma(az, r1, m::fma(bz, r2, j0 * Q[6 + e]));
  }
}

OL_DEV void prt_first_diag(Prt<f32x2, 1>& P, const PolBasis<f32x2>& b, f32x2 k0x, f32x2 k0y,
                           f32x2 k0z, f32x2 k1x, f32x2 k1y, f32x2 k1z, f32x2 j0, f32x2 j1,
                           f32x2 j2) {
  using m = Math<f32x2>;
  const f32x2 a[3] = {m::fma(j1, b.p1x, -(j0 * b.p0x)), m::fma(j1, b.p1y, -(j0 * b.p0y)),
                      m::fma(j1, b.p1z, -(j0 * b.p0z))};
  const f32x2 bb[3] = {m::fma(j2, k1x, -(j0 * k0x)), m::fma(j2, k1y, -(j0 * k0y)),
                       m::fma(j2, k1z, -(j0 * k0z))};
  const f32x2 p0[3] = {b.p0x, b.p0y, b.p0z}, k0[3] = {k0x, k0y, k0z};
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int e = 0; e < 3; ++e)
      P.m[3 * i + e] = m::fma(a[i], p0[e], m::fma(bb[i], k0[e], i == e ? j0 : m::splat(0)));
}

// interact()'s polarised tail for the pair: uncoated / Fresnel surfaces (the launcher keeps
// ranges with polarizer or retarder coatings, and bundles with OL_TRACE_NONUNIT_K, on the
// one-ray form)
OL_DEV void polarise_pair(const DevSurf<float>& s, const DevOptics<float>& o, f32x2 L0, f32x2 M0,
                          f32x2 N0, f32x2 adot, f32x2 nx, f32x2 ny, f32x2 nz, const Ray<f32x2>& r,
                          Prt<f32x2, 1>& P, bool& prt_fresh) {
  using m = Math<f32x2>;
  const int ck = s.coating_kind;
  const bool reflect = s.interaction == kReflect;
  const float nn = o.nn;
  if (ck == kCoatSimple) return;  // (never calls rays.update(): the matrix passes through)
  if (ck == kCoatNone && !reflect && o.u == 1.0f) {  // identity update: only a lost ray's NaN
    const f32x2 poison = (r.L + r.M + r.N) * 0.0f;
    const Mask2 lost = m::ne(poison, poison);
    if (m::any(lost)) {
#pragma unroll
      for (int e = 0; e < 9; ++e) P.m[e] = m::select(lost, P.m[e] + poison, P.m[e]);
    }
    return;
  }
  const f32x2 k0x = L0, k0y = M0, k0z = N0, k1x = r.L, k1y = r.M, k1z = r.N;
  const PolBasis<f32x2> b = pol_basis(k0x, k0y, k0z, k1x, k1y, k1z, nx, ny, nz);
  f32x2 j0 = m::splat(1), j1 = m::splat(1), j2 = m::splat(1);
  if (ck == kCoatFresnel) {
    const f32x2 one = m::splat(1);
    f32x2 ci = m::select(m::lt(adot, one), adot, m::select(m::ge(adot, one), one, adot));
    f32x2 root = m::sqrt(m::fma(m::splat(nn), m::splat(nn), m::fma(ci, ci, m::splat(-1))));
    if (reflect) {
      j0 = m::div(ci - root, ci + root);
      j1 = -m::div(m::fma(m::splat(nn * nn), ci, -root), m::fma(m::splat(nn * nn), ci, root));
      j2 = m::splat(-1);
    } else {
      j0 = m::div(2.0f * ci, ci + root);
      j1 = m::div(2.0f * nn * ci, m::fma(m::splat(nn * nn), ci, root));
    }
  }
  if (prt_fresh) prt_first_diag(P, b, k0x, k0y, k0z, k1x, k1y, k1z, j0, j1, j2);
  else prt_apply_diag(P, b, k0x, k0y, k0z, k1x, k1y, k1z, j0, j1, j2);
  prt_fresh = false;
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
OL_DEV void into_local_frame(const DevSurf<typename Math<V>::scalar>& s,
                                                 bool from_global, Ray<V> (&r)[RPT]) {
  using m = Math<V>;
  using T = typename m::scalar;
  // coordinate_system.py:73-89
  if (from_global) {
    const T ox = s.origin[0], oy = s.origin[1], oz = s.origin[2];
    if (s.flags & kSurfRotated) {
      cptr<T> R = s.cold->rot;
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
    cptr<T> R = s.cold->rel_rot;
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
OL_DEV typename Math<V>::mask aperture_mask(
    const DevSurf<typename Math<V>::scalar>& s, cptr<typename Math<V>::scalar> coeffs,
    V x, V y) {
  if constexpr (Math<V>::lanes == 1) {
    return aperture_contains<typename Math<V>::scalar, FULL>(s, coeffs, x, y);
  } else {
    return {aperture_contains<typename Math<V>::scalar, FULL>(s, coeffs, x.x, y.x),
            aperture_contains<typename Math<V>::scalar, FULL>(s, coeffs, x.y, y.y)};
  }
}

template <typename V, int RPT, int POLK, bool FULL>
OL_DEV void interact(const DevSurf<typename Math<V>::scalar>& s,
                                         const DevOptics<typename Math<V>::scalar>& o,
                                         cptr<typename Math<V>::scalar> coeffs,
                                         const V (&t)[RPT], const V (&nx)[RPT], const V (&ny)[RPT],
                                         const V (&nz)[RPT], Ray<V> (&r)[RPT],
                                         Prt<typename PrtLane<V, POLK>::type, POLK> (&P)[POLK ? RPT : 1],
                                         bool& prt_fresh, uint32_t pol_flags = 0) {
  // prt_fresh (wave-uniform): the matrices still hold the identity a fresh trace starts
  // from -- the first real update then writes O_out J O_in instead of multiplying by it
  // pol_flags (launch-uniform): kPolNonUnitK, see below
  using m = Math<V>;
  using T = typename m::scalar;
  static_assert(POLK == 0 || m::lanes == 1 || POLK == 1,
                "the polarised path: scalar, or the real-matrix fp32 pair");
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

  // refract / reflect (real_rays.py:163-205, 535-571).  The reference aligns the normal with
  // the ray first (n <- sign(n.k) n, dot <- |n.k|; sign(0) = 0) and then forms
  //   reflect:  k' = k - 2 |dot| n_aligned          refract:  k' = u k + n_aligned (root - u |dot|)
  // Multiplying by sign(dot) = +-1 (or 0) is exact and round-to-nearest is symmetric, so the
  // same bits come out of
  //   reflect:  k' = k - 2 dot n                    refract:  k' = u k + n (sign(dot) root - u dot)
  // with ONE product sign(dot) root instead of three products n sign(dot): 2 vector
  // instructions fewer per ray and refracting surface, 4 per reflecting one (round 5).
#ifndef OL_SNELL_SIGN_ON_ROOT
#define OL_SNELL_SIGN_ON_ROOT 1   // 0: rounds 1-4, the aligned normal formed explicitly (A/B knob)
#endif
  V L0[RPT], M0[RPT], N0[RPT], adot[RPT];
  V sdot[RPT];
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    L0[k] = r[k].L;
    M0[k] = r[k].M;
    N0[k] = r[k].N;
    sdot[k] = m::fma(L0[k], nx[k], m::fma(M0[k], ny[k], N0[k] * nz[k]));
    adot[k] = m::abs(sdot[k]);  // (read by the polarised coatings below only)
  }
#if OL_SNELL_SIGN_ON_ROOT
  if (s.interaction == kReflect) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      V k2 = m::splat(-2) * sdot[k];
      r[k].L = m::fma(k2, nx[k], L0[k]);
      r[k].M = m::fma(k2, ny[k], M0[k]);
      r[k].N = m::fma(k2, nz[k], N0[k]);
    }
  } else {
    const T u = o.u;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      V root = m::sqrt(m::fma(m::splat(-u * u), m::fma(-sdot[k], sdot[k], one), one));  // NaN on TIR
      // be.sign(dot): +-1, and 0 at 0 -- as a FACTOR of root, so that a NaN root (total
      // internal reflection) at exactly grazing incidence still poisons the direction as the
      // reference's 0 * NaN does.  (A NaN dot poisons it through the product u dot below,
      // whatever sign it is given here.)
      const V sgn = m::select(m::ne(sdot[k], zero), m::copysign(one, sdot[k]), zero);
      V w = m::fma(m::splat(-u), sdot[k], sgn * root);
      r[k].L = m::fma(m::splat(u), L0[k], nx[k] * w);
      r[k].M = m::fma(m::splat(u), M0[k], ny[k] * w);
      r[k].N = m::fma(m::splat(u), N0[k], nz[k] * w);
    }
  }
#else
  {
    V ax[RPT], ay[RPT], az[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const V sgn = m::select(m::ne(sdot[k], zero), m::copysign(one, sdot[k]), zero);
      ax[k] = nx[k] * sgn;
      ay[k] = ny[k] * sgn;
      az[k] = nz[k] * sgn;
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
        V root = m::sqrt(m::fma(m::splat(-u * u), m::fma(-adot[k], adot[k], one), one));
        V w = m::fma(m::splat(-u), adot[k], root);
        r[k].L = m::fma(m::splat(u), L0[k], ax[k] * w);
        r[k].M = m::fma(m::splat(u), M0[k], ay[k] * w);
        r[k].N = m::fma(m::splat(u), N0[k], az[k] * w);
      }
    }
  }
#endif

  // coating (interactions/base.py:111-128)
  if (s.coating_kind == kCoatSimple) {
    const T f = s.interaction == kReflect ? s.cold->coat[1] : s.cold->coat[0];
#pragma unroll
    for (int k = 0; k < RPT; ++k) r[k].i = r[k].i * f;
  }
  if constexpr (POLK != 0 && m::lanes == 2) {
#pragma unroll
    for (int k = 0; k < RPT; ++k)
      polarise_pair(s, o, L0[k], M0[k], N0[k], adot[k], nx[k], ny[k], nz[k], r[k], P[k], prt_fresh);
  } else if constexpr (POLK != 0) {
    const int ck = s.coating_kind;
    const bool reflect = s.interaction == kReflect;
    const T nn = o.nn;
    // SimpleCoating.reflect / transmit only scale the intensity (coatings.py:199-237): they
    // never call rays.update(), so the PRT matrix passes through unchanged -- unlike an
    // UNCOATED surface, whose interaction model calls rays.update() with the identity
    // Jones matrix (interactions/base.py:124-125).
    if (ck == kCoatSimple) return;
    // kPolNonUnitK (OL_TRACE_NONUNIT_K, ABI 11; launch-uniform): the bundle's direction cosines
    // are NOT unit vectors -- what the reference's iterative / robust ray aimers hand out
    // (|k|^2 - 1 ~ 1e-3, rays/ray_aiming/iterative.py:339-366) and nothing renormalises.  Its
    // PRT algebra takes k as it comes (polarized_rays.py:136-202): s = (k0 x k1) / |.| is a
    // unit vector, p0 = k0 x s and p1 = k1 x s have the lengths |k0| and |k1|, and
    //     O_out J O_in = j0 s s^T + j1 p1 p0^T + j2 k1 k0^T
    //                  = j0 s s^T + |k0||k1| (j1 p1^ p0^^T + j2 k1^ k0^^T)     (^: normalised)
    // -- the rank-2 form below on the NORMALISED directions with j1 and j2 scaled by |k0||k1|
    // (exact; without the flag it is the same matrix only for |k| = 1 to rounding).  The
    // Fresnel amplitudes read cos(aoi) = |n . k0| from k0 as it is, like the reference.
    const bool nonunit = (pol_flags & kPolNonUnitK) != 0;
    // An uncoated refracting surface between equal indices (every image plane, dummy
    // surfaces) leaves the direction unchanged (u = 1 => k1 = k0), its Jones matrix is
    // the identity and O_out O_in = I for ANY orthonormal basis: P' = P.  The
    // reference still multiplies it out (and its s = k0 x k1 there is rounding noise);
    // here the update is skipped and only the NaN state of a lost ray is carried into
    // the matrix, as the reference's product would.
    // (Not with kPolNonUnitK: the product is then s s^T + |k0|^2 (I - s s^T), which depends on
    // a vector s the reference takes from the rounding noise of k0 x k1 -- the update below
    // runs with s = k0 x n; callers that need the reference's numbers there hand such a
    // surface to the reference itself, as integration.py does.)
    if (ck == kCoatNone && !reflect && o.u == T(1) && !nonunit) {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const T poison = (r[k].L + r[k].M + r[k].N) * T(0);  // 0, or NaN for a lost ray
        // (a branch: the nine additions only run in waves that hold a lost ray)
        if (poison != poison) {
#pragma unroll
          for (int e = 0; e < (POLK == 2 ? 18 : 9); ++e) P[k].m[e] += poison;
        }
      }
      return;
    }
    if (ck == kCoatFresnel || ck == kCoatNone) {
      // real diagonal Jones matrix: rank-2 form of the update (prt_apply_diag)
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        T k0x = L0[k], k0y = M0[k], k0z = N0[k], k1x = r[k].L, k1y = r[k].M, k1z = r[k].N;
        T jscale = T(1);
        if (nonunit) {
          const T q0 = m::fma(k0x, k0x, m::fma(k0y, k0y, k0z * k0z));
          const T q1 = m::fma(k1x, k1x, m::fma(k1y, k1y, k1z * k1z));
          const T i0 = m::rsqrt(q0), i1 = m::rsqrt(q1);
          k0x *= i0; k0y *= i0; k0z *= i0;
          k1x *= i1; k1y *= i1; k1z *= i1;
          jscale = (q0 * i0) * (q1 * i1);  // |k0| |k1|
        }
        const PolBasis<T> b = pol_basis(k0x, k0y, k0z, k1x, k1y, k1z, nx[k], ny[k], nz[k]);
        T j0 = T(1), j1 = T(1), j2 = T(1);
        if (ck == kCoatFresnel) {
          // coatings.py:72-92 + jones.py:71-117 with cos(aoi) = min(|n.k0|, 1):
          // root = sqrt(nn^2 - sin^2) is real unless TIR, where k1 is NaN already.
          T ci = adot[k] < T(1) ? adot[k] : (adot[k] >= T(1) ? T(1) : adot[k]);
          // (root = nn cos(theta_t) could be taken from Snell's square root above, but keeping
          // that value live to here costs the fp32 polarised Newton kernel its 7th wave: two
          // VGPR spills to scratch, tools/kernel_probe.py)
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
        if (nonunit) {
          j1 *= jscale;
          j2 *= jscale;
        }
        if (prt_fresh)
          prt_first_diag<T, POLK>(P[k], b, k0x, k0y, k0z, k1x, k1y, k1z, j0, j1, j2);
        else
          prt_apply_diag<T, POLK>(P[k], b, k0x, k0y, k0z, k1x, k1y, k1z, j0, j1, j2);
      }
      prt_fresh = false;
      return;
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      PolBasis<T> b;
      if (nonunit) {
        // polarizer / retarder: the reference's own (non-orthonormal) triads -- s from the
        // normalised directions, p0 and p1 with the lengths |k0| and |k1| they have there;
        // the axis projections of jones.py:157-170 and the product below then are its own
        const T q0 = m::fma(L0[k], L0[k], m::fma(M0[k], M0[k], N0[k] * N0[k]));
        const T q1 = m::fma(r[k].L, r[k].L, m::fma(r[k].M, r[k].M, r[k].N * r[k].N));
        const T i0 = m::rsqrt(q0), i1 = m::rsqrt(q1);
        b = pol_basis(L0[k] * i0, M0[k] * i0, N0[k] * i0, r[k].L * i1, r[k].M * i1, r[k].N * i1,
                      nx[k], ny[k], nz[k]);
        const T m0 = q0 * i0, m1 = q1 * i1;
        b.p0x *= m0; b.p0y *= m0; b.p0z *= m0;
        b.p1x *= m1; b.p1y *= m1; b.p1z *= m1;
      } else {
        b = pol_basis(L0[k], M0[k], N0[k], r[k].L, r[k].M, r[k].N, nx[k], ny[k], nz[k]);
      }
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
// H: SurfLoaded<T> or SurfFetched<T> (device_table.h) -- `h.surf()` / `h.optics()` are asked
// for again at every phase; with SurfFetched each call re-reads the table, so no table
// field is live from one phase into the next.
// SHARE: see curved_distance.
template <typename V, int RPT, int POLK, int NR, bool SHARE = false, typename H>
OL_DEV void surface_step(const H& h, cptr<typename Math<V>::scalar> coeffs, bool from_global,
                         Ray<V> (&r)[RPT],
                         Prt<typename PrtLane<V, POLK>::type, POLK> (&P)[POLK ? RPT : 1],
                         uint32_t& status, bool& prt_fresh, const NrRefCtl* ref = nullptr,
                         uint32_t pol_flags = 0) {
  using m = Math<V>;
  using T = typename m::scalar;
  static_assert(NR == 0 || m::lanes == 1 || NR == kNrZernike,
                "the Newton-Raphson path: scalar, or the polynomial-Zernike fp32 pair");
  static_assert(NR != kNrReference || RPT == 1, "reference-Newton launches: one ray per lane");
  {
    const DevSurf<T> s = h.surf();
    into_local_frame<V, RPT>(s, from_global, r);
  }

  V t[RPT], nx[RPT], ny[RPT], nz[RPT];  // distance, unit normal at the hit
  const int geom = h.surf().geom;
  if (geom == kGeomPlane) {
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      t[k] = -m::div(r[k].z, r[k].N);  // plane.py:72-88
      nx[k] = ny[k] = m::splat(0);
      nz[k] = m::splat(1);  // plane.py:90-109
      r[k].x = m::fma(t[k], r[k].L, r[k].x);
      r[k].y = m::fma(t[k], r[k].M, r[k].y);
      r[k].z = m::fma(t[k], r[k].N, r[k].z);
    }
  } else if (geom == kGeomStandard) {
    const DevSurf<T> s = h.surf();
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
#ifndef OL_REFERENCE_ROOT
#define OL_REFERENCE_ROOT 1   // 0: the opt-in reference-formula root compiled out (A/B knob)
#endif
      if (OL_REFERENCE_ROOT && (s.flags & kSurfReferenceRoot)) {  // surface-uniform, opt-in
        const T Rr = s.cold->radius, kr = s.cold->conic;
#pragma unroll
        for (int k = 0; k < RPT; ++k)
          t[k] = reference_distance<V>(Rr, kr, r[k].x, r[k].y, r[k].z, r[k].L, r[k].M, r[k].N);
      } else {
#pragma unroll
        for (int k = 0; k < RPT; ++k)
          t[k] = curved_distance<V, SHARE>(cv, kp1, r[k].x, r[k].y, r[k].z, r[k].L, r[k].M,
                                           r[k].N);
      }
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        r[k].x = m::fma(t[k], r[k].L, r[k].x);
        r[k].y = m::fma(t[k], r[k].M, r[k].y);
        r[k].z = m::fma(t[k], r[k].N, r[k].z);
        conic_normal<V>(cv, kp1, r[k].x, r[k].y, r[k].z, nx[k], ny[k], nz[k]);
      }
    }
  } else if constexpr (NR == kNrReference) {
    const DevSurf<T> s = h.surf();
    cptr<T> c = coeffs + s.coeff_off;
    T gx, gy;
    const T t0 = conic_distance<T, SHARE>(s, r[0].x, r[0].y, r[0].z, r[0].L, r[0].M, r[0].N);
    newton_reference<T>(s, c, *ref, t0, r[0], t[0], gx, gy, status);
    const T im = m::rsqrt(m::fma(gx, gx, m::fma(gy, gy, T(1))));
    nx[0] = gx * im;
    ny[0] = gy * im;
    nz[0] = -im;
  } else if constexpr (NR != 0 && m::lanes == 2) {
    // the pair form (polynomial-form Zernike surfaces only: capi.hip, pair_polz_ok): the loop of
    // the scalar branch below with both rays of a lane in one NewtonRay<f32x2>
    NewtonRay<V> q[RPT];
    int max_iter;
    {
      const DevSurf<T> s = h.surf();
      max_iter = s.max_iter;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        t[k] = (s.flags & kSurfRadiusInf)
                   ? flat_distance<V>(r[k].z, r[k].N)
                   : curved_distance<V, SHARE>(s.cv, s.kp1, r[k].x, r[k].y, r[k].z, r[k].L, r[k].M,
                                               r[k].N);
        q[k].xb = m::fma(t[k], r[k].L, r[k].x);
        q[k].yb = m::fma(t[k], r[k].M, r[k].y);
        q[k].zb = m::fma(t[k], r[k].N, r[k].z);
        q[k].dt = m::splat(0);
        q[k].fprev = m::splat(0);
        q[k].gx = q[k].gy = m::splat(0);
        q[k].active = m::all(true);
      }
    }
    int it = 0;
    for (; it < max_iter; ++it) {
      bool any = false;
      const DevSurf<T> s = h.surf();
      cptr<T> c = coeffs + s.coeff_off;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        if (m::any(q[k].active)) newton_iterate(s, c, q[k], r[k].L, r[k].M, r[k].N, it, status);
        any = any || m::any(q[k].active);
      }
      if (!hw::wave_any(any)) {
        ++it;
        break;
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
      const DevSurf<T> s = h.surf();
      cptr<T> c = coeffs + s.coeff_off;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        V sag;
        uint32_t st = 0;
        zernike_mono_eval(s, c, r[k].x, r[k].y, sag, q[k].gx, q[k].gy, m::all(false), st);
      }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const V im = m::rsqrt(m::fma(q[k].gx, q[k].gx, m::fma(q[k].gy, q[k].gy, m::splat(1))));
      nx[k] = q[k].gx * im;
      ny[k] = q[k].gy * im;
      nz[k] = -im;
    }
  } else if constexpr (NR != 0) {
    constexpr bool COMPACT = NR == kNrCompact;
    NewtonRay<T> q[RPT];
    int max_iter;
    {
      const DevSurf<T> s = h.surf();
      max_iter = s.max_iter;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        t[k] = conic_distance<T, SHARE>(s, r[k].x, r[k].y, r[k].z, r[k].L, r[k].M, r[k].N);
        q[k].xb = m::fma(t[k], r[k].L, r[k].x);
        q[k].yb = m::fma(t[k], r[k].M, r[k].y);
        q[k].zb = m::fma(t[k], r[k].N, r[k].z);
        q[k].dt = T(0);
        q[k].fprev = T(0);
        q[k].gx = q[k].gy = T(0);
        q[k].active = true;
      }
    }
    int it = 0;
    bool can_compact = false;  // (only when every lane of the wave is alive)
    if constexpr (COMPACT && RPT > 1) can_compact = __popcll(__ballot(true)) == 64;
    for (; it < max_iter; ++it) {
      bool any = false;
      // one copy of the table fields per ITERATION: with SurfFetched the few scalars an
      // evaluation needs (curvature, tolerance, block offset) are re-read here and the
      // coefficient loads below them start from a fresh pointer -- nothing of the surface
      // is carried around the loop in SGPRs
      const DevSurf<T> s = h.surf();
      cptr<T> c = coeffs + s.coeff_off;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        if (q[k].active) newton_iterate<NR>(s, c, q[k], r[k].L, r[k].M, r[k].N, it, status);
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
        if (!hw::wave_any(any)) {
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
      const DevSurf<T> s = h.surf();
      cptr<T> c = coeffs + s.coeff_off;
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        T sag;
        uint32_t st = 0;
        nr_eval<NR>(s, c, r[k].x, r[k].y, sag, q[k].gx, q[k].gy, st);
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
  {
    const DevSurf<T> s = h.surf();
    const DevOptics<T> o = h.optics();
    interact<V, RPT, POLK, NR != 0>(s, o, coeffs, t, nx, ny, nz, r, P, prt_fresh, pol_flags);
  }
}

// the rows loaded by the caller (lean kernels; tests)
template <typename V, int RPT, int POLK, int NR, bool SHARE = false>
OL_DEV void surface_step(const DevSurf<typename Math<V>::scalar>& s,
                         const DevOptics<typename Math<V>::scalar>& o,
                         cptr<typename Math<V>::scalar> coeffs, bool from_global,
                         Ray<V> (&r)[RPT],
                         Prt<typename PrtLane<V, POLK>::type, POLK> (&P)[POLK ? RPT : 1],
                         uint32_t& status, bool& prt_fresh, const NrRefCtl* ref = nullptr,
                         uint32_t pol_flags = 0) {
  const SurfLoaded<typename Math<V>::scalar> h{s, o};
  surface_step<V, RPT, POLK, NR, SHARE>(h, coeffs, from_global, r, P, status, prt_fresh, ref,
                                        pol_flags);
}

// local -> global for the recorded state (coordinate_system.py:91-107)
template <typename V>
OL_DEV Ray<V> to_global(const DevSurf<typename Math<V>::scalar>& s,
                                            const Ray<V>& r) {
  using T = typename Math<V>::scalar;
  Ray<V> g = r;
  if (s.flags & kSurfRotated) {
    cptr<T> R = s.cold->rot;  // inverse = transpose
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

}  // namespace ol
