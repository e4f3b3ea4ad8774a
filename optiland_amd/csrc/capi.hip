// capi.hip -- the C ABI declared in include/optiland_hip.h.
//
// Host-side only: validates arguments, re-expresses the public surface table as
// DevSurf<T>/DevOptics<T> for T in {float, double} (device_table.h) and launches
// the kernels on the caller's stream.  No ray memory is ever allocated here.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <type_traits>
#include <vector>

#include "../../include/optiland_hip.h"
#include "device_table.h"
#include "trace_launch.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define OL_HIP_CHECK(expr)                                                        \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess)                                                         \
      return fail(OL_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_));        \
  } while (0)

void mat3_mul_abt(const double* A, const double* B, double* out) {  // A * B^T
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0;
      for (int k = 0; k < 3; ++k) acc += A[3 * i + k] * B[3 * j + k];
      out[3 * i + j] = acc;
    }
}

bool is_identity(const double* R) {
  static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
  return std::memcmp(R, I, sizeof(I)) == 0 ||
         std::equal(R, R + 9, I);  // -0.0 == 0.0
}

// device form of a leaf aperture: squared radii, inverse squared semi-axes
void convert_aperture(int kind, const double* a, double* out) {
  out[0] = out[1] = out[2] = out[3] = 0.0;
  switch (kind) {
    case OL_AP_RADIAL:
    case OL_AP_OFFSET_RADIAL:
      out[0] = a[0] * a[0];
      out[1] = a[1] * a[1];
      out[2] = a[2];
      out[3] = a[3];
      break;
    case OL_AP_RECTANGULAR:
      for (int k = 0; k < 4; ++k) out[k] = a[k];
      break;
    case OL_AP_ELLIPTICAL:
      out[0] = 1.0 / (a[0] * a[0]);
      out[1] = 1.0 / (a[1] * a[1]);
      out[2] = a[2];
      out[3] = a[3];
      break;
    default:
      break;
  }
}

double factorial(int n) {
  double f = 1.0;
  for (int i = 2; i <= n; ++i) f *= i;
  return f;
}

// Regroup Zernike terms (c_j, n_j, m_j, N_j) into one LEVEL per azimuthal order |m| with
// the (cos, sin) radial polynomials in u = rho^2, the factor rho^|m| removed:
//   R_n^m(rho) = sum_k (-1)^k (n-k)! / (k! ((n+m)/2-k)! ((n-m)/2-k)!) rho^(n-2k)
// (zernike/base.py:216-239), rho^(n-2k) = rho^m u^((n-m)/2-k).
// Layout per level (device_table.h): [m, K] as integers (their slots are recorded in
// `int_slots` and uploaded as bit patterns), then K x {a_c, a_s, b_c, b_s, d_c, d_s}.
int build_zernike_block(const double* terms, int n_terms, std::vector<double>& out,
                        std::vector<size_t>& int_slots, int* n_levels) {
  struct Level { std::vector<double> a[2], b[2]; };
  std::map<int, Level> levels;
  for (int j = 0; j < n_terms; ++j) {
    const double c = terms[4 * j];
    const int n = (int)terms[4 * j + 1];
    const int m = (int)terms[4 * j + 2];
    const double N = terms[4 * j + 3];
    if (c == 0.0) continue;  // zernike.py:234 (and contributes 0 to the sag)
    const int ma = std::abs(m);
    if (n < ma || ((n - ma) & 1) || n > 60) return -1;
    const int K = (n - ma) / 2 + 1;
    Level& lv = levels[ma];
    const int kind = m < 0 ? 1 : 0;
    for (int q = 0; q < 2; ++q)
      if ((int)lv.a[q].size() < K) {
        lv.a[q].resize(K, 0.0);
        lv.b[q].resize(K, 0.0);
      }
    for (int k = 0; k < K; ++k) {
      double coef = ((k & 1) ? -1.0 : 1.0) * factorial(n - k) /
                    (factorial(k) * factorial((n + ma) / 2 - k) * factorial((n - ma) / 2 - k));
      const int p = (n - ma) / 2 - k;
      lv.a[kind][p] += c * N * coef;
      lv.b[kind][p] += c * coef;
    }
  }
  *n_levels = (int)levels.size();
  for (auto& kv : levels) {  // std::map iterates in ascending m
    const Level& lv = kv.second;
    const int K = (int)lv.a[0].size();
    int_slots.push_back(out.size());
    out.push_back((double)kv.first);
    int_slots.push_back(out.size());
    out.push_back((double)K);
    for (int k = 0; k < K; ++k) {
      out.push_back(lv.a[0][k]);
      out.push_back(lv.a[1][k]);
      out.push_back(lv.b[0][k]);
      out.push_back(lv.b[1][k]);
      out.push_back(k + 1 < K ? (k + 1) * lv.b[0][k + 1] : 0.0);
      out.push_back(k + 1 < K ? (k + 1) * lv.b[1][k + 1] : 0.0);
    }
  }
  return 0;
}

// Low-order Zernike surfaces as ONE bivariate polynomial in the normalised Cartesian
// coordinates (device_table.h: kGeomZernikeMono).  Every term is expanded,
//   rho^m cos(m phi) = Re (x + i y)^m,   rho^m sin(m phi) = Im (x + i y)^m,
//   R_n^m(rho) rho^-m = sum_k r_k (x^2 + y^2)^((n-m)/2-k)          (zernike/base.py:216-239)
// into the monomials x^i y^j, i + j <= n: S (with the normalisation constants N_j, the
// sag, zernike.py:153-180) and Q (without them: what the reference differentiates for the
// normal, zernike.py:234-240); the block holds S and the two derivative triangles of Q in
// Horner order.  The expansion is exact up to rounding in double.  Conditioning: on the unit
// circle sum |s_ij| |x|^i |y|^j of one term is sum_k |r_k| (|x| + |y|)^m -- the radial
// polynomial's own growth, which the per-|m| level form has too, times at most 2^(m/2)
// from the binomial expansion of the harmonic (the level form builds the harmonics by a
// stable recurrence instead).  The degree cap keeps that extra factor at <= 16; higher
// orders stay on the level form.  Returns false when the surface has to stay there.
constexpr int kZernMonoMaxDegree = 8;

// OPTILAND_HIP_ZERNIKE_MONO=0 keeps every Zernike surface on the level form (A/B runs,
// parity tests of the level evaluator on low-order systems)
bool zernike_mono_enabled() {
  const char* e = getenv("OPTILAND_HIP_ZERNIKE_MONO");
  return !(e && e[0] == '0');
}

double binomial(int n, int k) { return factorial(n) / (factorial(k) * factorial(n - k)); }

bool build_zernike_mono_block(const double* terms, int n_terms, std::vector<double>& out,
                              int* degree) {
  int nmax = -1;
  for (int j = 0; j < n_terms; ++j) {
    if (terms[4 * j] == 0.0) continue;
    const int n = (int)terms[4 * j + 1], ma = std::abs((int)terms[4 * j + 2]);
    if (n < ma || ((n - ma) & 1) || n > kZernMonoMaxDegree) return false;
    nmax = std::max(nmax, n);
  }
  if (nmax < 0) return false;  // no term at all: the level form handles "nothing" already
  const int W = nmax + 1;
  std::vector<double> S(W * W, 0.0), Q(W * W, 0.0);  // [i * W + j] = coefficient of x^i y^j
  for (int t = 0; t < n_terms; ++t) {
    const double c = terms[4 * t];
    if (c == 0.0) continue;
    const int n = (int)terms[4 * t + 1], m = (int)terms[4 * t + 2], ma = std::abs(m);
    const double N = terms[4 * t + 3];
    for (int k = 0; k <= (n - ma) / 2; ++k) {
      const double rk = ((k & 1) ? -1.0 : 1.0) * factorial(n - k) /
                        (factorial(k) * factorial((n + ma) / 2 - k) * factorial((n - ma) / 2 - k));
      const int pw = (n - ma) / 2 - k;  // power of u = x^2 + y^2
      for (int q = 0; q <= pw; ++q) {   // u^pw = sum_q C(pw, q) x^(2 (pw - q)) y^(2 q)
        const double uq = binomial(pw, q);
        // harmonic part: cos -> even powers of y, sin -> odd powers of y
        for (int h = (m < 0 ? 1 : 0); h <= ma; h += 2) {
          const double sign = ((h / 2) & 1) ? -1.0 : 1.0;  // i^h: +1, (+i), -1, (-i), ...
          const double hc = binomial(ma, h) * sign;
          const int i = 2 * (pw - q) + (ma - h), jj = 2 * q + h;
          S[i * W + jj] += c * N * rk * uq * hc;
          Q[i * W + jj] += c * rk * uq * hc;
        }
      }
    }
  }
  // Horner order: outer in x from the highest power, inner in y from the highest power
  for (int i = nmax; i >= 0; --i)
    for (int j = nmax - i; j >= 0; --j) out.push_back(S[i * W + j]);
  for (int i = nmax - 1; i >= 0; --i)
    for (int j = nmax - 1 - i; j >= 0; --j) {
      out.push_back((i + 1) * Q[(i + 1) * W + j]);  // dQ/dx
      out.push_back((j + 1) * Q[i * W + j + 1]);    // dQ/dy
    }
  *degree = nmax;
  return true;
}

// host-side staging record: every field of both device blocks, in double
struct HostSurf {
  int32_t geom, interaction, aperture_kind, coating_kind, coeff_off, n_coeff, max_iter;
  uint32_t flags;
  int32_t poly_cols, coeff_len, ap_off, ap_len;
  double cv, kp1, tol, inv_norm;
  double origin[3], rot[9], rel_off[3], rel_rot[9], ap[4], coat[2], axis[3], ret_cos, ret_sin;
  double radius, conic;
};

template <typename T>
struct DeviceTable {
  // ONE allocation: [hot rows | cold rows | optics rows | coefficients], each block on a
  // 256-byte boundary -- created by one copy and re-written in place by one copy
  // (ol_system_update: an optimiser edits the prescription between every two traces)
  char* blob = nullptr;
  ol::DevSurfHot<T>* surf = nullptr;
  ol::DevSurfCold<T>* cold = nullptr;
  ol::DevOptics<T>* optics = nullptr;
  T* coeffs = nullptr;
  size_t n_surf = 0, n_opt = 0, coef_capacity = 0;  // allocation sizes (ol_system_update)
  size_t off_cold = 0, off_opt = 0, off_coef = 0;
};

inline size_t round256(size_t v) { return (v + 255u) & ~size_t(255u); }

// everything ol_system_create derives from its arguments, in double, before any device call
struct Staged {
  std::vector<struct HostSurf> surf;
  std::vector<ol::DevOptics<double>> optics;
  std::vector<double> coeffs;
  std::vector<size_t> int_slots;  // coeffs entries uploaded as integer bit patterns
  std::vector<int32_t> interaction, coating, geom;
  std::vector<uint8_t> polygon, ref_newton, no_pair;
};

}  // namespace

struct ol_system {
  int32_t n_surf = 0;
  int32_t n_wl = 0;
  int device = 0;
  DeviceTable<float> f32;
  DeviceTable<double> f64;
  std::vector<int32_t> interaction;  // host copy for validation
  std::vector<int32_t> coating;
  std::vector<int32_t> geom;
  std::vector<uint8_t> polygon;  // surface uses a polygon aperture (top level or in a tree)
  std::vector<uint8_t> ref_newton;  // OL_SURF_REFERENCE_NEWTON on a traced Newton-Raphson surface
  // the surface keeps a polarised Zernike launch off the two-rays-per-lane form
  // (trace_kernel.hip: OL_POLZ_PAIR): a Zernike surface in the level form, a polarizer / retarder
  std::vector<uint8_t> no_pair;
  // false between the two in-place uploads of ol_system_update and for good if the second
  // one fails: the fp32 and fp64 tables (and the host copies) then describe different
  // prescriptions -- every entry point refuses such a system instead of tracing through it
  bool consistent = true;
};

#define OL_CHECK_CONSISTENT(sys, who)                                                          \
  if (!(sys)->consistent)                                                                      \
    return fail(OL_EINVAL, who ": the system's tables are inconsistent after a failed "         \
                               "ol_system_update (destroy it and create a new one)")

namespace {

// in_place: the allocation of `dst` is reused (it must fit: checked by the caller) and the
// copy is queued on `stream`, i.e. ordered after every launch already queued there that
// still reads the old table; otherwise a fresh allocation and a blocking copy.
template <typename T>
int upload(const std::vector<HostSurf>& surf64,
           const std::vector<ol::DevOptics<double>>& opt64, const std::vector<double>& coef64,
           const std::vector<size_t>& int_slots, DeviceTable<T>& dst, bool in_place = false,
           hipStream_t stream = nullptr) {
  const size_t n_surf = surf64.size(), n_opt = opt64.size();
  const size_t n_coef = coef64.size() ? coef64.size() : 1;
  const size_t off_cold = round256(n_surf * sizeof(ol::DevSurfHot<T>));
  const size_t off_opt = off_cold + round256(n_surf * sizeof(ol::DevSurfCold<T>));
  const size_t off_coef = off_opt + round256(n_opt * sizeof(ol::DevOptics<T>));
  const size_t used = off_coef + n_coef * sizeof(T);
  std::vector<char> host(used, 0);
  auto* surf = reinterpret_cast<ol::DevSurfHot<T>*>(host.data());
  auto* cold = reinterpret_cast<ol::DevSurfCold<T>*>(host.data() + off_cold);
  auto* opt = reinterpret_cast<ol::DevOptics<T>*>(host.data() + off_opt);
  T* coef = reinterpret_cast<T*>(host.data() + off_coef);
  for (size_t i = 0; i < n_surf; ++i) {
    const auto& a = surf64[i];
    auto& b = surf[i];
    auto& c = cold[i];
    b.geom = a.geom; b.interaction = a.interaction; b.aperture_kind = a.aperture_kind;
    b.coating_kind = a.coating_kind; b.coeff_off = a.coeff_off; b.n_coeff = a.n_coeff;
    b.max_iter = a.max_iter; b.flags = a.flags;
    b.cv = (T)a.cv; b.kp1 = (T)a.kp1;
    c.poly_cols = a.poly_cols; c.coeff_len = a.coeff_len; c.ap_off = a.ap_off;
    c.ap_len = a.ap_len; c.tol = (T)a.tol; c.inv_norm = (T)a.inv_norm;
    for (int k = 0; k < 3; ++k) { b.origin[k] = (T)a.origin[k]; b.rel_off[k] = (T)a.rel_off[k]; }
    for (int k = 0; k < 9; ++k) { c.rot[k] = (T)a.rot[k]; c.rel_rot[k] = (T)a.rel_rot[k]; }
    for (int k = 0; k < 4; ++k) c.ap[k] = (T)a.ap[k];
    for (int k = 0; k < 2; ++k) c.coat[k] = (T)a.coat[k];
    for (int k = 0; k < 3; ++k) c.axis[k] = (T)a.axis[k];
    c.ret_cos = (T)a.ret_cos; c.ret_sin = (T)a.ret_sin;
    c.radius = (T)a.radius; c.conic = (T)a.conic;
  }
  for (size_t i = 0; i < n_opt; ++i) {
    opt[i].n1 = (T)opt64[i].n1; opt[i].n2 = (T)opt64[i].n2; opt[i].u = (T)opt64[i].u;
    opt[i].nn = (T)opt64[i].nn; opt[i].absorb = (T)opt64[i].absorb;
  }
  for (size_t i = 0; i < coef64.size(); ++i) coef[i] = (T)coef64[i];
  for (size_t i : int_slots) {  // loop headers: integer bit patterns (scalar-unit operands)
    using I = typename std::conditional<sizeof(T) == 4, int32_t, int64_t>::type;
    const I v = (I)coef64[i];
    std::memcpy(&coef[i], &v, sizeof(T));
  }

  if (in_place) {
    // same surface / wavelength counts (checked by the caller): the block offsets are the
    // ones of the allocation.  (Pageable source: the runtime stages it before it returns,
    // the vector may go.)
    OL_HIP_CHECK(hipMemcpyAsync(dst.blob, host.data(), used, hipMemcpyHostToDevice, stream));
    return OL_OK;
  }
  dst.n_surf = n_surf;
  dst.n_opt = n_opt;
  // head room: a surface whose coefficient block grows a little (an asphere gaining a term)
  // still updates in place
  dst.coef_capacity = n_coef + 64;
  dst.off_cold = off_cold; dst.off_opt = off_opt; dst.off_coef = off_coef;
  OL_HIP_CHECK(hipMalloc((void**)&dst.blob, off_coef + dst.coef_capacity * sizeof(T)));
  dst.surf = reinterpret_cast<ol::DevSurfHot<T>*>(dst.blob);
  dst.cold = reinterpret_cast<ol::DevSurfCold<T>*>(dst.blob + off_cold);
  dst.optics = reinterpret_cast<ol::DevOptics<T>*>(dst.blob + off_opt);
  dst.coeffs = reinterpret_cast<T*>(dst.blob + off_coef);
  OL_HIP_CHECK(hipMemcpy(dst.blob, host.data(), used, hipMemcpyHostToDevice));
  return OL_OK;
}

template <typename T>
void release(DeviceTable<T>& t) {
  if (t.blob) (void)hipFree(t.blob);
  t = DeviceTable<T>();
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Which Newton kernel a traced range needs (trace_launch.h: launch_trace): 0 = none,
// 3 / 4 = all of its Newton-Raphson surfaces are Zernike surfaces / even aspheres, 5 = the
// reference's batch-global stop rule was asked for (OL_SURF_REFERENCE_NEWTON), else 1.
// (A polygon aperture on a conic-only range also selects the generic kernel: polygons live
// in the full kernels only.)
int newton_family(const ol_system* sys, int32_t first, int32_t last) {
  bool any = false, polygon = false, all_zernike = true, all_even = true;
  // OL_SURF_REFERENCE_NEWTON (opt-in, ABI 11) on any Newton surface of the range: the kernel
  // family that iterates the batch in lockstep (surface_math.h: newton_reference)
  for (int32_t s = first; s <= last; ++s)
    if (sys->ref_newton[s]) return ol::kNrReference;
  for (int32_t s = first; s <= last; ++s) {
    polygon = polygon || sys->polygon[s];
    const int g = sys->geom[s];
    if (g == OL_GEOM_PLANE || g == OL_GEOM_STANDARD) continue;
    any = true;
    all_zernike = all_zernike && g == OL_GEOM_ZERNIKE;
    all_even = all_even && g == OL_GEOM_EVEN_ASPHERE;
  }
  if (!any) return polygon ? 1 : 0;
  // OPTILAND_HIP_NR_FAMILY=0: always the generic Newton kernel (A/B runs; parity tests of the
  // generic instantiation on single-family systems)
  static const bool families = [] {
    const char* e = getenv("OPTILAND_HIP_NR_FAMILY");
    return !(e && e[0] == '0');
  }();
  if (families && all_zernike) return 3;
  if (families && all_even) return 4;
  return 1;
}

template <typename T>
int do_trace(const ol_system* sys, const DeviceTable<T>& tab, int64_t n, void* const rays[8],
             int32_t wl, void* record, int64_t record_stride, void* prt, int32_t first,
             int32_t last, uint32_t flags, uint32_t* status, const ol_trace_extras* extras,
             hipStream_t stream) {
  ol::TraceArgs<T> a{};
  a.spot = extras ? extras->spot_slots : nullptr;
  a.cx = extras ? extras->cx : 0.0;
  a.cy = extras ? extras->cy : 0.0;
  a.spot_slots = OL_SPOT_SLOTS;
  a.surf = tab.surf;
  a.cold = tab.cold;
  a.optics = tab.optics;
  a.coeffs = tab.coeffs;
  bool vec = true;
  constexpr int64_t kVec = 16 / sizeof(T);
  for (int k = 0; k < 8; ++k) {
    a.rays[k] = static_cast<T*>(rays[k]);
    vec = vec && aligned16(rays[k]);
  }
  a.record = static_cast<T*>(record);
  a.prt = static_cast<T*>(prt);
  if (record) vec = vec && aligned16(record) && (record_stride % kVec == 0);
  if (prt) vec = vec && aligned16(prt) && (n % kVec == 0);
  a.status = status;
  a.n = n;
  a.record_stride = record_stride;
  a.first = first;
  a.last = last;
  a.n_wl = sys->n_wl;
  a.wl = wl;
  a.flags = flags & 0xffu;  // public flags only
  a.record_from = (extras && extras->record_first_surface > first) ? extras->record_first_surface
                                                                     : first;
  if (!prt) a.flags &= ~ol::kTracePrtIdentity;
  // Zero-copy object row: when the caller's ray planes ARE row 0 of the record
  // block and the first surface only records (ObjectSurface.trace,
  // surfaces/object_surface.py:56-69), the kernel skips that row's stores.
  if (record && a.record_from == first && sys->interaction[first] == OL_INTERACT_RECORD_ONLY) {
    bool alias = true;
    for (int k = 0; k < 8; ++k)
      alias = alias && (static_cast<T*>(rays[k]) == static_cast<T*>(record) + k * record_stride);
    if (alias) a.flags |= ol::kTraceRow0IsInput;
  }
  const int family = newton_family(sys, first, last);
  a.nr_iters = extras ? extras->newton_iterations : nullptr;
  a.nr_count_at = extras && extras->newton_iterations ? extras->newton_count_surface : -1;
  a.n_surf = sys->n_surf;
  if (family == ol::kNrReference) {
    if (!a.nr_iters)
      return fail(OL_EINVAL, "ol_trace: a surface of the range carries OL_SURF_REFERENCE_NEWTON: "
                             "pass ol_trace_extras.newton_iterations (see ol_newton_count)");
    if (a.spot)
      return fail(OL_EUNSUPPORTED, "ol_trace_ex: no spot epilogue on a reference-Newton range");
  }
  hipError_t e = ol::launch_trace<T>(a, vec, family, stream);
  if (e != hipSuccess) return fail(OL_EHIP, "trace launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

// the fused entry points (one launch: generate -> trace [-> reduce]) do not serve ranges whose
// Newton iteration count is a property of the batch
int refuse_reference_newton(const char* who, const ol_system* sys) {
  if (newton_family(sys, 0, sys->n_surf - 1) != ol::kNrReference) return OL_OK;
  return fail(OL_EUNSUPPORTED, "%s: the system carries OL_SURF_REFERENCE_NEWTON surfaces "
                               "(reference stop rule): generate with ol_generate_rays, count "
                               "with ol_newton_count, trace with ol_trace_ex", who);
}

// ol_raygen_inputs -> working precision; host-side part of the range validation
// (launch-uniform scalars are checked here, planes in the kernel)
template <typename T>
int convert_inputs(const char* who, const ol_raygen_inputs* in, uint32_t* status,
                   ol::RaygenIn<T>& o, bool& aligned) {
  if (!in || !in->px || !in->py) return fail(OL_EINVAL, "%s: NULL argument", who);
  if ((in->hx == nullptr) != (in->hy == nullptr) || (in->vx == nullptr) != (in->vy == nullptr))
    return fail(OL_EINVAL, "%s: hx/hy (and vx/vy) must be given together", who);
  if ((in->flags & (OL_RAYGEN_CHECK_FIELD | OL_RAYGEN_CHECK_PUPIL)) && !status)
    return fail(OL_EINVAL, "%s: a CHECK flag needs a status word", who);
  if ((in->flags & OL_RAYGEN_CHECK_FIELD) && !in->hx) {
    auto bad = [](double v) { return !(v >= -1.0 && v <= 1.0); };
    if (bad(in->hx0) || bad(in->hy0))  // real_ray_tracer.py:156-173, same text
      return fail(OL_EINVAL, "Normalized field coordinates must be within (-1, 1)");
  }
  o.hx = static_cast<const T*>(in->hx);
  o.hy = static_cast<const T*>(in->hy);
  o.px = static_cast<const T*>(in->px);
  o.py = static_cast<const T*>(in->py);
  o.vx = static_cast<const T*>(in->vx);
  o.vy = static_cast<const T*>(in->vy);
  o.hx0 = (T)in->hx0;
  o.hy0 = (T)in->hy0;
  o.vx0 = (T)in->vx0;
  o.vy0 = (T)in->vy0;
  o.tx0 = o.ty0 = T(0);
  o.flags = in->flags;
  aligned = aligned16(in->px) && aligned16(in->py) && aligned16(in->hx) && aligned16(in->hy) &&
            aligned16(in->vx) && aligned16(in->vy);  // NULL counts as aligned
  return OL_OK;
}

ol::RaygenDev raygen_dev(const ol_raygen_params* g) {
  return ol::RaygenDev{g->object_infinite, g->field_kind, g->EPL,     g->EPD,    g->max_field,
                       g->offset,          g->z_first,    g->tele_dz, g->apod_a, g->apod_b,
                       g->apod_kind};
}

template <typename T>
int do_generate_rays(const ol_raygen_params* p, int64_t n, const ol_raygen_inputs* in,
                     void* const out[8], uint32_t* status, hipStream_t stream) {
  ol::RaygenIn<T> ri;
  bool al = true;
  if (int rc = convert_inputs<T>("ol_generate_rays", in, status, ri, al)) return rc;
  if (n == 0) return OL_OK;
  T* o[8];
  for (int k = 0; k < 8; ++k) o[k] = static_cast<T*>(out[k]);
  hipError_t e = ol::launch_raygen<T>(raygen_dev(p), ri, n, o, status, stream);
  if (e != hipSuccess) return fail(OL_EHIP, "raygen launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

template <typename T>
int do_trace_generate(const ol_system* sys, const DeviceTable<T>& tab, int64_t n,
                      const ol_raygen_params* p, const ol_raygen_inputs* in, int32_t wl,
                      void* record, int64_t record_stride, void* const rays_out[8], void* prt,
                      uint32_t flags, uint32_t* status, const ol_trace_extras* extras,
                      hipStream_t stream) {
  ol::TraceArgs<T> a{};
  bool al = true;
  if (int rc = convert_inputs<T>("ol_trace_generate", in, status, a.in, al)) return rc;
  if (n == 0) return OL_OK;
  a.spot = extras ? extras->spot_slots : nullptr;
  a.cx = extras ? extras->cx : 0.0;
  a.cy = extras ? extras->cy : 0.0;
  a.spot_slots = OL_SPOT_SLOTS;
  a.surf = tab.surf;
  a.cold = tab.cold;
  a.optics = tab.optics;
  a.coeffs = tab.coeffs;
  const ol::RaygenDev rg = raygen_dev(p);
  a.rgc = ol::RaygenConsts<T>(rg);
  if (a.in.hx == nullptr) ol::uniform_field_tangents<T>(rg, a.in);
  for (int k = 0; k < 8; ++k) a.rays[k] = rays_out ? static_cast<T*>(rays_out[k]) : nullptr;
  a.record = static_cast<T*>(record);
  a.prt = static_cast<T*>(prt);
  a.status = status;
  a.n = n;
  a.record_stride = record_stride;
  a.first = 0;
  a.last = sys->n_surf - 1;
  a.record_from = (extras && extras->record_first_surface > 0) ? extras->record_first_surface : 0;
  a.n_wl = sys->n_wl;
  a.wl = wl;
  a.flags = (flags & (ol::kTracePrtComplex | ol::kTraceFewWaves)) |
            (prt ? ol::kTracePrtIdentity : 0u) | (rays_out ? ol::kTraceWriteRays : 0u);
  if (extras && extras->updated_intensity && extras->update_intensity_state) {
    if (!prt)
      return fail(OL_EINVAL, "ol_trace_generate: the update_intensity epilogue needs a "
                             "polarised launch (prt)");
    const ol_polarization_state* ps = extras->update_intensity_state;
    ol::PolStateDev st{ps->is_polarized, ps->Ex, ps->Ey, ps->phase_x, ps->phase_y};
    a.pf = ol::PolFields<T>(st);
    a.i_updated = static_cast<T*>(extras->updated_intensity);
  }
  // packed pairs (fp32 lean form): 8-byte accesses to px / py, the record rows, the final state
  bool pair_ok = al && (reinterpret_cast<uintptr_t>(record) % 8 == 0) && record_stride % 2 == 0;
  if (rays_out)
    for (int k = 0; k < 8; ++k)
      pair_ok = pair_ok && reinterpret_cast<uintptr_t>(rays_out[k]) % 8 == 0;
  const int family = newton_family(sys, 0, sys->n_surf - 1);
  if (family == ol::kNrZernike) {
    // the polarised Zernike pair (fp32): polynomial-form Zernike surfaces, real diagonal Jones
    // matrices, an even number of rays (the PRT planes are written with 8-byte lane accesses)
    for (int32_t s = 0; s < sys->n_surf; ++s) pair_ok = pair_ok && !sys->no_pair[s];
    pair_ok = pair_ok && prt && n % 2 == 0 && reinterpret_cast<uintptr_t>(prt) % 8 == 0 &&
              reinterpret_cast<uintptr_t>(a.i_updated) % 8 == 0;
  }
  hipError_t e = ol::launch_trace_generate<T>(a, family, pair_ok, stream);
  if (e != hipSuccess) return fail(OL_EHIP, "trace launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

template <typename T>
int do_trace_spot(const ol_system* sys, const DeviceTable<T>& tab, int64_t n,
                  const ol_raygen_params* p, const ol_raygen_inputs* in, double cx, double cy,
                  int32_t wl, void* const hits[3], double* out7, uint32_t* status,
                  hipStream_t stream) {
  ol::SpotArgs<T> a{};
  bool vec = true;
  if (int rc = convert_inputs<T>("ol_trace_spot", in, status, a.in, vec)) return rc;
  if (n == 0) return OL_OK;
  a.surf = tab.surf;
  a.cold = tab.cold;
  a.optics = tab.optics;
  a.coeffs = tab.coeffs;
  a.rg = raygen_dev(p);
  a.cx = cx;
  a.cy = cy;
  for (int k = 0; k < 3; ++k) {
    a.hits[k] = hits ? static_cast<T*>(hits[k]) : nullptr;
    vec = vec && aligned16(a.hits[k]);
  }
  a.out = out7;
  a.status = status;
  a.n = n;
  a.first = 0;
  a.last = sys->n_surf - 1;
  a.n_wl = sys->n_wl;
  a.wl = wl;
  a.tiles_per_block = 1;
  hipError_t e = ol::launch_spot_trace<T>(a, vec, newton_family(sys, 0, sys->n_surf - 1), stream);
  if (e != hipSuccess) return fail(OL_EHIP, "spot launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

template <typename T>
const DeviceTable<T>& table_of(const ol_system* sys);
template <>
const DeviceTable<float>& table_of<float>(const ol_system* sys) { return sys->f32; }
template <>
const DeviceTable<double>& table_of<double>(const ol_system* sys) { return sys->f64; }

template <typename T>
int do_trace_spot_batch(const ol_system* sys, const DeviceTable<T>& tab, int64_t n,
                        const ol_raygen_params* p, const ol_raygen_inputs* in, int32_t n_cells,
                        const ol_spot_cell* cells, void* hits, int64_t hits_stride, double* out8,
                        uint32_t* status, hipStream_t stream) {
  ol::SpotArgs<T> a{};
  bool vec = true;
  if (int rc = convert_inputs<T>("ol_trace_spot_batch", in, status, a.in, vec)) return rc;
  if (n == 0 || n_cells == 0) return OL_OK;
  a.surf = tab.surf;
  a.cold = tab.cold;
  a.optics = tab.optics;
  a.coeffs = tab.coeffs;
  a.rg = raygen_dev(p);
  T* hb = static_cast<T*>(hits);
  for (int k = 0; k < 3; ++k) a.hits[k] = hb ? hb + k * hits_stride : nullptr;
  // every cell's planes must keep the 16-byte alignment the vector stores need
  vec = vec && aligned16(hb) && (hits_stride * (int64_t)sizeof(T)) % 16 == 0;
  a.out = out8;
  a.status = status;
  a.n = n;
  a.first = 0;
  a.last = sys->n_surf - 1;
  a.n_wl = sys->n_wl;
  a.wl = 0;
  a.tiles_per_block = 1;
  ol::SpotBatch<T> b{};
  b.n_cells = n_cells;
  b.hits_stride = hits_stride;
  for (int c = 0; c < n_cells; ++c) {
    ol::RaygenIn<T> f = a.in;   // the launch-uniform field of this cell -> its tangents
    f.hx0 = (T)cells[c].hx;
    f.hy0 = (T)cells[c].hy;
    ol::uniform_field_tangents<T>(a.rg, f);
    b.c[c].tx = f.tx0;
    b.c[c].ty = f.ty0;
    b.c[c].vx = (T)cells[c].vx;
    b.c[c].vy = (T)cells[c].vy;
    b.c[c].cx = cells[c].cx;
    b.c[c].cy = cells[c].cy;
    b.c[c].wl = cells[c].wavelength_index;
    const ol_system* from = cells[c].optics_of ? cells[c].optics_of : sys;
    b.c[c].optics = table_of<T>(from).optics;
    b.c[c].n_wl = from->n_wl;
  }
  hipError_t e = ol::launch_spot_batch<T>(a, b, vec, newton_family(sys, 0, sys->n_surf - 1), stream);
  if (e != hipSuccess) return fail(OL_EHIP, "spot batch launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

template <typename T>
int do_wavefront_reference(const ol_system* sys, const DeviceTable<T>& tab,
                           const ol_raygen_params* p, const ol_raygen_inputs* in,
                           const ol_wavefront_params* w, double pupil_z, int32_t planar,
                           int32_t wl, void* reference_dev, void* chief8, uint32_t* status,
                           hipStream_t stream) {
  ol::ChiefArgs<T> a{};
  // launch-uniform field / vignetting only; the pupil point is (0, 0)
  a.in.hx = a.in.hy = a.in.px = a.in.py = a.in.vx = a.in.vy = nullptr;
  a.in.hx0 = (T)in->hx0; a.in.hy0 = (T)in->hy0; a.in.vx0 = (T)in->vx0; a.in.vy0 = (T)in->vy0;
  a.in.tx0 = a.in.ty0 = T(0);
  a.in.flags = in->flags & OL_RAYGEN_PRESCALE_PUPIL;
  a.surf = tab.surf;
  a.cold = tab.cold;
  a.optics = tab.optics;
  a.coeffs = tab.coeffs;
  a.rg = raygen_dev(p);
  ol::WavefrontDev wd{0, 0, 0, 0, w->n_image, 0, w->ux, w->uy, w->half_epd, w->wavelength_um,
                      0, 0, planar ? 1.0 : 0.0, w->last_thickness, w->last_absorb};
  a.wfc = ol::WavefrontConsts<T>(wd);   // planar flag from nz != 0; centre / R filled on device
  a.pupil_z = (T)pupil_z;
  a.out = static_cast<ol::WavefrontConsts<T>*>(reference_dev);
  a.chief = static_cast<T*>(chief8);
  a.status = status;
  a.first = 0;
  a.last = sys->n_surf - 1;
  a.n_wl = sys->n_wl;
  a.wl = wl;
  hipError_t e = ol::launch_chief_reference<T>(a, newton_family(sys, 0, sys->n_surf - 1), stream);
  if (e != hipSuccess)
    return fail(OL_EHIP, "chief-ray launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

template <typename T>
int do_trace_opd(const ol_system* sys, const DeviceTable<T>& tab, int64_t n,
                 const ol_raygen_params* p, const ol_raygen_inputs* in,
                 const ol_wavefront_params* w, int32_t wl, void* opd, void* inten,
                 void* const pupil[3], double* mom, uint32_t* status, hipStream_t stream,
                 const void* reference_dev = nullptr) {
  ol::OpdArgs<T> a{};
  a.wf_dev = static_cast<const ol::WavefrontConsts<T>*>(reference_dev);
  bool vec = true;
  if (int rc = convert_inputs<T>("ol_trace_opd", in, status, a.in, vec)) return rc;
  if (a.in.hx != nullptr || a.in.vx != nullptr)
    return fail(OL_EINVAL, "ol_trace_opd: one field point per launch (launch-uniform field and "
                           "vignetting, no hx / hy / vx / vy planes)");
  if (n == 0) return OL_OK;
  a.surf = tab.surf;
  a.cold = tab.cold;
  a.optics = tab.optics;
  a.coeffs = tab.coeffs;
  a.rg = raygen_dev(p);
  if (w)
    a.wf = ol::WavefrontDev{w->xc, w->yc, w->zc, w->R, w->n_image, w->opd_ref, w->ux, w->uy,
                            w->half_epd, w->wavelength_um, w->nx, w->ny, w->nz,
                            w->last_thickness, w->last_absorb};
  else
    a.wf = ol::WavefrontDev{0, 0, 0, 1, 1, 0, 0, 0, 1, 1, 0, 0, 0};  // (unused: reference_dev)
  a.opd = static_cast<T*>(opd);
  a.inten = static_cast<T*>(inten);
  for (int k = 0; k < 3; ++k) a.pupil[k] = pupil ? static_cast<T*>(pupil[k]) : nullptr;
  a.mom = mom;
  a.status = status;
  a.n = n;
  a.first = 0;
  a.last = sys->n_surf - 1;
  a.n_wl = sys->n_wl;
  a.wl = wl;
  vec = vec && aligned16(opd) && aligned16(inten);
  for (int k = 0; k < 3; ++k) vec = vec && aligned16(a.pupil[k]);
  hipError_t e =
      ol::launch_opd_trace<T>(a, vec, newton_family(sys, 0, sys->n_surf - 1), stream);
  if (e != hipSuccess) return fail(OL_EHIP, "opd launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

}  // namespace

extern "C" {

const char* ol_last_error(void) { return g_err.c_str(); }
int32_t ol_abi_version(void) { return OL_ABI_VERSION; }

int ol_set_tuning(int32_t knob, int32_t value) {
  switch (knob) {
    case OL_TUNE_RAYS_PER_THREAD:
      if (value < 0 || value > 3)
        return fail(OL_EINVAL, "ol_set_tuning: rays per thread must be 0 (auto), 1, 2 (vector) "
                               "or 3 (fp32 pair)");
      ol::tuning().rays_per_thread = value;
      return OL_OK;
    case OL_TUNE_COMPACT:
      ol::tuning().compact = value ? 1 : 0;
      return OL_OK;
    case OL_TUNE_RECORD_WG_CAP:
      if (value < 0 || value > 8)
        return fail(OL_EINVAL, "ol_set_tuning: record workgroup cap 0 (default policy), "
                               "1 (never) or 2 ... 8");
      ol::tuning().record_wg_cap = value;
      return OL_OK;
    case OL_TUNE_FIT_GRID:
      if (value < 0 || value > ol::kFitMaxBlocks)
        return fail(OL_EINVAL, "ol_set_tuning: fit grid 0 (default) ... %d", ol::kFitMaxBlocks);
      ol::tuning().fit_grid = value;
      return OL_OK;
    default:
      return fail(OL_EINVAL, "ol_set_tuning: unknown knob %d", knob);
  }
}

namespace {
// argument validation + every host-side conversion of ol_system_create / ol_system_update
int stage_system(const char* who, const ol_surface_desc* surf, int32_t n_surf,
                 const double* coeffs, int32_t n_coeffs, const ol_surface_optics* optics,
                 int32_t n_wavelengths, Staged& st) {
  if (!surf || n_surf <= 0) return fail(OL_EINVAL, "%s: no surfaces", who);
  if (!optics || n_wavelengths <= 0) return fail(OL_EINVAL, "%s: no optics table", who);
  if (n_coeffs < 0 || (n_coeffs > 0 && !coeffs))
    return fail(OL_EINVAL, "%s: bad coefficient buffer", who);

  st.surf.assign(n_surf, HostSurf());
  std::vector<HostSurf>& dev = st.surf;
  std::vector<double>& dcoef = st.coeffs;
  std::vector<size_t>& int_slots = st.int_slots;
  int32_t prev_traced = -1;
  for (int32_t i = 0; i < n_surf; ++i) {
    const ol_surface_desc& s = surf[i];
    HostSurf& d = dev[i];
    std::memset(&d, 0, sizeof(d));
    if (s.geom_kind < OL_GEOM_PLANE || s.geom_kind > OL_GEOM_TOROIDAL)
      return fail(OL_EUNSUPPORTED, "surface %d: geometry kind %d", i, s.geom_kind);
    if (s.interaction < OL_INTERACT_RECORD_ONLY || s.interaction > OL_INTERACT_REFLECT)
      return fail(OL_EUNSUPPORTED, "surface %d: interaction %d", i, s.interaction);
    if (s.aperture_kind < OL_AP_NONE || s.aperture_kind > OL_AP_POLYGON)
      return fail(OL_EUNSUPPORTED, "surface %d: aperture kind %d", i, s.aperture_kind);
    if (s.coating_kind < OL_COAT_NONE || s.coating_kind > OL_COAT_RETARDER)
      return fail(OL_EUNSUPPORTED, "surface %d: coating kind %d", i, s.coating_kind);
    if (s.n_coeff < 0 || s.coeff_offset < 0)
      return fail(OL_EINVAL, "surface %d: negative coefficient range", i);
    const int per = s.geom_kind == OL_GEOM_ZERNIKE ? 4 : 1;
    const int extra = s.geom_kind == OL_GEOM_CHEBYSHEV ? 2 : 0;
    if ((int64_t)s.coeff_offset + (int64_t)s.n_coeff * per + extra > n_coeffs)
      return fail(OL_EINVAL, "surface %d: coefficient block exceeds the buffer", i);

    d.geom = s.geom_kind;
    d.interaction = s.interaction;
    d.aperture_kind = s.aperture_kind;
    d.coating_kind = s.coating_kind;
    d.max_iter = s.max_iter;
    d.poly_cols = s.poly_cols;
    d.flags = (s.flags & OL_SURF_ROTATED) ? ol::kSurfRotated : 0u;
    const bool inf_r = std::isinf(s.radius) || s.geom_kind == OL_GEOM_PLANE;
    if (inf_r) d.flags |= ol::kSurfRadiusInf;
    if ((s.flags & OL_SURF_REFERENCE_ROOT) && s.geom_kind == OL_GEOM_STANDARD && !inf_r)
      d.flags |= ol::kSurfReferenceRoot;
    d.cv = inf_r ? 0.0 : 1.0 / s.radius;
    d.kp1 = 1.0 + s.conic;
    d.radius = s.radius;
    d.conic = s.conic;
    d.tol = s.tol;
    d.inv_norm = s.norm_radius != 0.0 ? 1.0 / s.norm_radius : 0.0;
    for (int k = 0; k < 3; ++k) d.origin[k] = s.origin[k];
    for (int k = 0; k < 9; ++k) d.rot[k] = s.rot[k];
    if (!(d.flags & ol::kSurfRotated)) {
      static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
      std::memcpy(d.rot, I, sizeof(I));
    }
    // relative transform from the local frame of the previous TRACED surface (the
    // kernel keeps its state in that frame; record-only surfaces do not move it)
    static const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::memcpy(d.rel_rot, I, sizeof(I));
    if (prev_traced >= 0) {
      const HostSurf& p = dev[prev_traced];
      mat3_mul_abt(d.rot, p.rot, d.rel_rot);
      double dv[3] = {p.origin[0] - d.origin[0], p.origin[1] - d.origin[1],
                      p.origin[2] - d.origin[2]};
      for (int r = 0; r < 3; ++r) {
        double v = d.rot[3 * r] * dv[0] + d.rot[3 * r + 1] * dv[1] + d.rot[3 * r + 2] * dv[2];
        d.rel_off[r] = std::isfinite(v) ? v : 0.0;
      }
      if (!is_identity(d.rel_rot)) d.flags |= ol::kSurfRelRotated;
    }
    if (s.interaction != OL_INTERACT_RECORD_ONLY) prev_traced = i;
    // apertures: pre-square / pre-invert in double; polygon vertex blocks move into the
    // device coefficient buffer and their token / descriptor offsets are rewritten
    auto stage_polygon = [&](const double* a, double* q) -> int {
      const int64_t voff = (int64_t)a[0], nv = (int64_t)a[1];
      if (voff < 0 || nv < 1 || voff + 2 * nv > n_coeffs)
        return fail(OL_EINVAL, "surface %d: polygon vertex block outside the buffer", i);
      q[0] = (double)dcoef.size();
      q[1] = (double)nv;
      q[2] = q[3] = 0.0;
      dcoef.insert(dcoef.end(), coeffs + voff, coeffs + voff + 2 * nv);
      return OL_OK;
    };
    if (s.aperture_kind == OL_AP_COMPOSITE) {
      const int64_t off = (int64_t)s.aperture[0], cnt = (int64_t)s.aperture[1];
      if (off < 0 || cnt <= 0 || off + cnt * OL_AP_TOKEN_DOUBLES > n_coeffs)
        return fail(OL_EINVAL, "surface %d: aperture token list outside the buffer", i);
      // pass 1: converted tokens (polygon vertices are appended to dcoef as they come)
      std::vector<double> toks;
      int depth = 0;
      for (int64_t t = 0; t < cnt; ++t) {
        const double* tok = coeffs + off + t * OL_AP_TOKEN_DOUBLES;
        const int op = (int)tok[0];
        double q[4];
        if (op >= OL_AP_RADIAL && op <= OL_AP_ELLIPTICAL) {
          convert_aperture(op, tok + 1, q);
          if (++depth > OL_AP_MAX_DEPTH)
            return fail(OL_EUNSUPPORTED, "surface %d: aperture tree too deep", i);
        } else if (op == OL_AP_POLYGON) {
          if (int rc = stage_polygon(tok + 1, q)) return rc;
          if (++depth > OL_AP_MAX_DEPTH)
            return fail(OL_EUNSUPPORTED, "surface %d: aperture tree too deep", i);
        } else if (op >= OL_AP_OP_UNION && op <= OL_AP_OP_DIFFERENCE) {
          q[0] = q[1] = q[2] = q[3] = 0.0;
          if (--depth < 1) return fail(OL_EINVAL, "surface %d: malformed aperture tokens", i);
        } else {
          return fail(OL_EUNSUPPORTED, "surface %d: aperture token op %d", i, op);
        }
        toks.push_back((double)op);
        toks.insert(toks.end(), q, q + 4);
      }
      if (depth != 1) return fail(OL_EINVAL, "surface %d: malformed aperture tokens", i);
      // pass 2: the token list itself, contiguous
      d.ap_off = (int32_t)dcoef.size();
      d.ap_len = (int32_t)cnt;
      dcoef.insert(dcoef.end(), toks.begin(), toks.end());
    } else if (s.aperture_kind == OL_AP_POLYGON) {
      if (int rc = stage_polygon(s.aperture, d.ap)) return rc;
    } else {
      convert_aperture(s.aperture_kind, s.aperture, d.ap);
    }
    d.coat[0] = s.coat[0];
    d.coat[1] = s.coat[1];
    if (s.coating_kind == OL_COAT_POLARIZER || s.coating_kind == OL_COAT_RETARDER) {
      const int need = s.coating_kind == OL_COAT_RETARDER ? 4 : 3;
      const int64_t off = (int64_t)s.coat[0];
      if (off < 0 || off + need > n_coeffs)
        return fail(OL_EINVAL, "surface %d: coating axis block outside the buffer", i);
      for (int k = 0; k < 3; ++k) d.axis[k] = coeffs[off + k];
      if (s.coating_kind == OL_COAT_RETARDER) {
        d.ret_cos = std::cos(coeffs[off + 3] / 2);
        d.ret_sin = std::sin(coeffs[off + 3] / 2);
      }
    }

    // coefficient block
    d.coeff_off = (int32_t)dcoef.size();
    const double* src = coeffs ? coeffs + s.coeff_offset : nullptr;
    if (s.geom_kind == OL_GEOM_ZERNIKE) {
      int ng = 0;
      if (zernike_mono_enabled() && build_zernike_mono_block(src, s.n_coeff, dcoef, &ng)) {
        d.geom = ol::kGeomZernikeMono;  // low order, well conditioned: one polynomial
        d.n_coeff = ng;
      } else {
        if (build_zernike_block(src, s.n_coeff, dcoef, int_slots, &ng) != 0)
          return fail(OL_EINVAL, "surface %d: invalid Zernike (n, m) index", i);
        d.n_coeff = ng;
      }
    } else if (s.geom_kind == OL_GEOM_EVEN_ASPHERE || s.geom_kind == OL_GEOM_ODD_ASPHERE ||
               s.geom_kind == OL_GEOM_POLYNOMIAL) {
      if (s.geom_kind == OL_GEOM_POLYNOMIAL &&
          (s.poly_cols <= 0 || s.n_coeff % s.poly_cols != 0))
        return fail(OL_EINVAL, "surface %d: polynomial grid %d x ? cols %d", i, s.n_coeff,
                    s.poly_cols);
      d.n_coeff = s.n_coeff;
      dcoef.insert(dcoef.end(), src, src + s.n_coeff);
    } else if (s.geom_kind == OL_GEOM_CHEBYSHEV) {
      if (s.poly_cols <= 0 || s.n_coeff % s.poly_cols != 0)
        return fail(OL_EINVAL, "surface %d: chebyshev grid %d, cols %d", i, s.n_coeff,
                    s.poly_cols);
      d.n_coeff = s.n_coeff;
      dcoef.push_back(1.0 / src[0]);
      dcoef.push_back(1.0 / src[1]);
      dcoef.insert(dcoef.end(), src + 2, src + 2 + s.n_coeff);
    } else if (s.geom_kind == OL_GEOM_BICONIC) {
      // biconic.py:57-66: cx/cy = 0 for infinite or zero radii
      if (s.n_coeff != 2) return fail(OL_EINVAL, "surface %d: biconic needs {Ry, ky}", i);
      const double Rx = s.radius, Ry = src[0];
      d.cv = (std::isinf(Rx) || Rx == 0.0) ? 0.0 : 1.0 / Rx;
      d.n_coeff = 2;
      dcoef.push_back((std::isinf(Ry) || Ry == 0.0) ? 0.0 : 1.0 / Ry);
      dcoef.push_back(1.0 + src[1]);
    } else if (s.geom_kind == OL_GEOM_TOROIDAL) {
      if (s.n_coeff < 2) return fail(OL_EINVAL, "surface %d: toroidal needs {R_rot, k_yz}", i);
      const double Rr = src[0], Ryz = s.radius;
      d.n_coeff = s.n_coeff - 2;  // number of y^(2i) terms
      dcoef.push_back(Rr);
      dcoef.push_back(std::isinf(Rr) ? 0.0 : 1.0 / Rr);
      dcoef.push_back(1.0 + src[1]);
      dcoef.push_back((std::isfinite(Ryz) && Ryz != 0.0) ? 1.0 / Ryz : 0.0);
      dcoef.insert(dcoef.end(), src + 2, src + s.n_coeff);
    } else {
      d.n_coeff = 0;
    }
    d.coeff_len = (int32_t)dcoef.size() - d.coeff_off;
  }

  std::vector<ol::DevOptics<double>>& dopt = st.optics;
  dopt.assign((size_t)n_surf * n_wavelengths, ol::DevOptics<double>());
  for (size_t i = 0; i < dopt.size(); ++i) {
    const ol_surface_optics& o = optics[i];
    std::memset(&dopt[i], 0, sizeof(dopt[i]));
    dopt[i].n1 = o.n1;
    dopt[i].n2 = o.n2;
    dopt[i].u = o.n1 / o.n2;
    dopt[i].nn = o.n2 / o.n1;
    dopt[i].absorb = o.absorb > 0.0 ? o.absorb : 0.0;
  }

  for (int32_t i = 0; i < n_surf; ++i) {
    st.interaction.push_back(surf[i].interaction);
    st.coating.push_back(surf[i].coating_kind);
    st.geom.push_back(surf[i].geom_kind);
    {
      bool poly = surf[i].aperture_kind == OL_AP_POLYGON;
      if (surf[i].aperture_kind == OL_AP_COMPOSITE) {
        const int64_t off = (int64_t)surf[i].aperture[0], cnt = (int64_t)surf[i].aperture[1];
        for (int64_t t = 0; t < cnt; ++t)
          poly = poly || (int)coeffs[off + t * OL_AP_TOKEN_DOUBLES] == OL_AP_POLYGON;
      }
      st.polygon.push_back(poly ? 1 : 0);
    }
    st.no_pair.push_back((surf[i].geom_kind == OL_GEOM_ZERNIKE && dev[i].geom != ol::kGeomZernikeMono) ||
                                 surf[i].coating_kind == OL_COAT_POLARIZER ||
                                 surf[i].coating_kind == OL_COAT_RETARDER
                             ? 1 : 0);
    st.ref_newton.push_back((surf[i].flags & OL_SURF_REFERENCE_NEWTON) &&
                                    surf[i].geom_kind != OL_GEOM_PLANE &&
                                    surf[i].geom_kind != OL_GEOM_STANDARD &&
                                    surf[i].interaction != OL_INTERACT_RECORD_ONLY
                                ? 1 : 0);
  }
  return OL_OK;
}

void adopt_host_copies(ol_system* sys, Staged& st) {
  sys->interaction.swap(st.interaction);
  sys->coating.swap(st.coating);
  sys->geom.swap(st.geom);
  sys->polygon.swap(st.polygon);
  sys->ref_newton.swap(st.ref_newton);
  sys->no_pair.swap(st.no_pair);
}
}  // namespace

int ol_system_create(const ol_surface_desc* surf, int32_t n_surf, const double* coeffs,
                     int32_t n_coeffs, const ol_surface_optics* optics, int32_t n_wavelengths,
                     ol_system** out) {
  if (!out) return fail(OL_EINVAL, "ol_system_create: out is NULL");
  *out = nullptr;
  Staged st;
  if (int rc = stage_system("ol_system_create", surf, n_surf, coeffs, n_coeffs, optics,
                            n_wavelengths, st))
    return rc;
  ol_system* sys = new (std::nothrow) ol_system();
  if (!sys) return fail(OL_ENOMEM, "ol_system_create: out of host memory");
  sys->n_surf = n_surf;
  sys->n_wl = n_wavelengths;
  adopt_host_copies(sys, st);
  if (hipGetDevice(&sys->device) != hipSuccess) {
    delete sys;
    return fail(OL_EHIP, "ol_system_create: no HIP device (hipGetDevice failed)");
  }
  int rc = upload<float>(st.surf, st.optics, st.coeffs, st.int_slots, sys->f32);
  if (rc == OL_OK) rc = upload<double>(st.surf, st.optics, st.coeffs, st.int_slots, sys->f64);
  if (rc != OL_OK) {
    release(sys->f32);
    release(sys->f64);
    delete sys;
    return rc;
  }
  *out = sys;
  return OL_OK;
}

int ol_system_update(ol_system* sys, const ol_surface_desc* surf, int32_t n_surf,
                     const double* coeffs, int32_t n_coeffs, const ol_surface_optics* optics,
                     int32_t n_wavelengths, void* stream) {
  if (!sys) return fail(OL_EINVAL, "ol_system_update: system is NULL");
  if (n_surf != sys->n_surf || n_wavelengths != sys->n_wl)
    return fail(OL_EUNSUPPORTED, "ol_system_update: %d surfaces x %d wavelengths do not fit the "
                                 "system's %d x %d tables", n_surf, n_wavelengths, sys->n_surf,
                sys->n_wl);
  Staged st;
  if (int rc = stage_system("ol_system_update", surf, n_surf, coeffs, n_coeffs, optics,
                            n_wavelengths, st))
    return rc;
  OL_CHECK_CONSISTENT(sys, "ol_system_update");
  const size_t need = st.coeffs.size() ? st.coeffs.size() : 1;
  if (need > sys->f32.coef_capacity || need > sys->f64.coef_capacity)
    return fail(OL_EUNSUPPORTED, "ol_system_update: coefficient block of %zu values exceeds the "
                                 "allocated %zu", need, sys->f32.coef_capacity);
  {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != sys->device)
      return fail(OL_EINVAL, "ol_system_update: current HIP device %d is not the system's "
                             "device %d", cur, sys->device);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  // the fp64 table first: if its copy fails nothing has changed yet and the system stays
  // usable; only a failure of the SECOND copy leaves the two precisions apart
  int rc = upload<double>(st.surf, st.optics, st.coeffs, st.int_slots, sys->f64, true, s);
  if (rc != OL_OK) return rc;
  sys->consistent = false;
  rc = upload<float>(st.surf, st.optics, st.coeffs, st.int_slots, sys->f32, true, s);
  if (rc != OL_OK) return rc;
  adopt_host_copies(sys, st);
  sys->consistent = true;
  return OL_OK;
}

void ol_system_destroy(ol_system* sys) {
  if (!sys) return;
  release(sys->f32);
  release(sys->f64);
  delete sys;
}

int32_t ol_system_num_surfaces(const ol_system* sys) { return sys ? sys->n_surf : 0; }

int ol_trace(const ol_system* sys, ol_dtype dt, int64_t n_rays, void* const rays[8],
             int32_t wavelength_index, void* record, int64_t record_stride, void* prt,
             int32_t first_surface, int32_t last_surface, uint32_t flags, uint32_t* status,
             void* stream) {
  return ol_trace_ex(sys, dt, n_rays, rays, wavelength_index, record, record_stride, prt,
                     first_surface, last_surface, flags, status, nullptr, stream);
}

int ol_trace_ex(const ol_system* sys, ol_dtype dt, int64_t n_rays, void* const rays[8],
                int32_t wavelength_index, void* record, int64_t record_stride, void* prt,
                int32_t first_surface, int32_t last_surface, uint32_t flags, uint32_t* status,
                const ol_trace_extras* extras, void* stream) {
  if (!sys) return fail(OL_EINVAL, "ol_trace: system is NULL");
  OL_CHECK_CONSISTENT(sys, "ol_trace");
  if (dt != OL_F32 && dt != OL_F64) return fail(OL_EINVAL, "ol_trace: bad dtype %d", (int)dt);
  if (n_rays < 0) return fail(OL_EINVAL, "ol_trace: negative ray count");
  if (first_surface < 0 || last_surface >= sys->n_surf || first_surface > last_surface)
    return fail(OL_EINVAL, "ol_trace: surface range [%d, %d] outside [0, %d)", first_surface,
                last_surface, sys->n_surf);
  if (wavelength_index < 0 || wavelength_index >= sys->n_wl)
    return fail(OL_EINVAL, "ol_trace: wavelength index %d outside [0, %d)", wavelength_index,
                sys->n_wl);
  if (n_rays == 0) return OL_OK;
  {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != sys->device)
      return fail(OL_EINVAL, "ol_trace: current HIP device %d is not the system's device %d",
                  cur, sys->device);
  }
  if (!rays) return fail(OL_EINVAL, "ol_trace: rays is NULL");
  for (int k = 0; k < 8; ++k)
    if (!rays[k]) return fail(OL_EINVAL, "ol_trace: rays[%d] is NULL", k);
  if (record && record_stride < n_rays)
    return fail(OL_EINVAL, "ol_trace: record_stride %lld < n_rays %lld", (long long)record_stride,
                (long long)n_rays);
  if (!record && !(flags & OL_TRACE_WRITE_RAYS) && !prt && !(extras && extras->spot_slots))
    return fail(OL_EINVAL, "ol_trace: nothing to write (no record, no OL_TRACE_WRITE_RAYS)");
  if (extras && extras->record_first_surface > last_surface)
    return fail(OL_EINVAL, "ol_trace_ex: record_first_surface %d beyond last_surface %d",
                extras->record_first_surface, last_surface);
  if (prt && extras && extras->spot_slots)
    return fail(OL_EINVAL, "ol_trace_ex: the spot epilogue is for unpolarised traces (the "
                           "polarised intensity needs ol_polarized_intensity first)");
  if (!prt) {
    // rays/ray_generator.py:89-94: polarization-dependent coatings need polarized rays
    for (int32_t s = first_surface; s <= last_surface; ++s)
      if (sys->coating[s] >= OL_COAT_FRESNEL)
        return fail(OL_EINVAL,
                    "Polarization must be set when surfaces have polarization-dependent "
                    "coatings.");
  }
  if (prt && !(flags & OL_TRACE_PRT_COMPLEX)) {
    for (int32_t s = first_surface; s <= last_surface; ++s)
      if (sys->coating[s] == OL_COAT_RETARDER)
        return fail(OL_EINVAL, "ol_trace: surface %d is a retarder (complex Jones matrix): "
                               "pass an 18-plane prt with OL_TRACE_PRT_COMPLEX", s);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dt == OL_F32)
    return do_trace<float>(sys, sys->f32, n_rays, rays, wavelength_index, record, record_stride,
                           prt, first_surface, last_surface, flags, status, extras, st);
  return do_trace<double>(sys, sys->f64, n_rays, rays, wavelength_index, record, record_stride,
                          prt, first_surface, last_surface, flags, status, extras, st);
}

int ol_newton_count(const ol_system* sys, ol_dtype dt, int64_t n_rays, void* const rays[8],
                    int32_t wavelength_index, int32_t first_surface, int32_t surface,
                    int32_t* iterations, int32_t verify, void* stream) {
  if (!sys) return fail(OL_EINVAL, "ol_newton_count: system is NULL");
  OL_CHECK_CONSISTENT(sys, "ol_newton_count");
  if (dt != OL_F32 && dt != OL_F64)
    return fail(OL_EINVAL, "ol_newton_count: bad dtype %d", (int)dt);
  if (n_rays < 0) return fail(OL_EINVAL, "ol_newton_count: negative ray count");
  if (first_surface < 0 || surface >= sys->n_surf || first_surface > surface)
    return fail(OL_EINVAL, "ol_newton_count: surface range [%d, %d] outside [0, %d)",
                first_surface, surface, sys->n_surf);
  if (wavelength_index < 0 || wavelength_index >= sys->n_wl)
    return fail(OL_EINVAL, "ol_newton_count: wavelength index %d outside [0, %d)",
                wavelength_index, sys->n_wl);
  if (!iterations) return fail(OL_EINVAL, "ol_newton_count: iterations is NULL");
  if (!sys->ref_newton[surface])
    return fail(OL_EINVAL, "ol_newton_count: surface %d is not a traced Newton-Raphson surface "
                           "with OL_SURF_REFERENCE_NEWTON", surface);
  if (n_rays == 0) return OL_OK;
  {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != sys->device)
      return fail(OL_EINVAL, "ol_newton_count: current HIP device %d is not the system's "
                             "device %d", cur, sys->device);
  }
  if (!rays) return fail(OL_EINVAL, "ol_newton_count: rays is NULL");
  for (int k = 0; k < 8; ++k)
    if (!rays[k]) return fail(OL_EINVAL, "ol_newton_count: rays[%d] is NULL", k);
  // the geometry of a ray does not depend on its polarisation: the unpolarised kernel over
  // [first_surface, surface], nothing recorded, nothing written back
  ol_trace_extras ex{};
  ex.newton_iterations = iterations;
  ex.newton_count_surface = verify ? -1 : surface;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dt == OL_F32)
    return do_trace<float>(sys, sys->f32, n_rays, rays, wavelength_index, nullptr, 0, nullptr,
                           first_surface, surface, 0u, nullptr, &ex, st);
  return do_trace<double>(sys, sys->f64, n_rays, rays, wavelength_index, nullptr, 0, nullptr,
                          first_surface, surface, 0u, nullptr, &ex, st);
}

int ol_trace_generate(const ol_system* sys, ol_dtype dt, int64_t n_rays,
                      const ol_raygen_params* p, const ol_raygen_inputs* in,
                      int32_t wavelength_index, void* record, int64_t record_stride,
                      void* const rays_out[8], void* prt, uint32_t flags, uint32_t* status,
                      const ol_trace_extras* extras, void* stream) {
  if (!sys) return fail(OL_EINVAL, "ol_trace_generate: system is NULL");
  OL_CHECK_CONSISTENT(sys, "ol_trace_generate");
  if (int rc = refuse_reference_newton("ol_trace_generate", sys)) return rc;
  if (dt != OL_F32 && dt != OL_F64)
    return fail(OL_EINVAL, "ol_trace_generate: bad dtype %d", (int)dt);
  if (!p || !in) return fail(OL_EINVAL, "ol_trace_generate: NULL argument");
  if (n_rays < 0) return fail(OL_EINVAL, "ol_trace_generate: negative ray count");
  if (wavelength_index < 0 || wavelength_index >= sys->n_wl)
    return fail(OL_EINVAL, "ol_trace_generate: wavelength index %d outside [0, %d)",
                wavelength_index, sys->n_wl);
  if ((in->vx || in->vy) && !(in->hx && in->hy))
    return fail(OL_EUNSUPPORTED, "ol_trace_generate: per-ray vignetting planes come with "
                                 "per-ray field planes (else ol_generate_rays + ol_trace)");
  // (ABI 10: polarised launches take per-ray field planes and apodized pupils too)
  if (extras && extras->spot_slots) {
    if (prt)
      return fail(OL_EINVAL, "ol_trace_generate: the spot epilogue is for unpolarised traces "
                             "(the polarised intensity needs update_intensity first)");
    if (in->hx || in->hy || p->apod_kind != OL_APOD_NONE)
      return fail(OL_EUNSUPPORTED, "ol_trace_generate: the spot epilogue is one field point "
                                   "without apodization (else ol_trace_spot)");
    if (extras->record_first_surface > 0)
      return fail(OL_EINVAL, "ol_trace_generate: spot epilogue and record_first_surface "
                             "cannot be combined");
  }
  if (!record) return fail(OL_EINVAL, "ol_trace_generate: record is NULL");
  if (record_stride < n_rays)
    return fail(OL_EINVAL, "ol_trace_generate: record_stride %lld < n_rays %lld",
                (long long)record_stride, (long long)n_rays);
  if (extras && extras->record_first_surface >= sys->n_surf)
    return fail(OL_EINVAL, "ol_trace_generate: record_first_surface %d outside [0, %d)",
                extras->record_first_surface, sys->n_surf);
  if (rays_out)
    for (int k = 0; k < 8; ++k)
      if (!rays_out[k]) return fail(OL_EINVAL, "ol_trace_generate: rays_out[%d] is NULL", k);
  if (!prt) {
    for (int32_t s = 0; s < sys->n_surf; ++s)
      if (sys->coating[s] >= OL_COAT_FRESNEL)
        return fail(OL_EINVAL,
                    "Polarization must be set when surfaces have polarization-dependent "
                    "coatings.");
  }
  if (prt && !(flags & OL_TRACE_PRT_COMPLEX)) {
    for (int32_t s = 0; s < sys->n_surf; ++s)
      if (sys->coating[s] == OL_COAT_RETARDER)
        return fail(OL_EINVAL, "ol_trace_generate: surface %d is a retarder (complex Jones "
                               "matrix): pass an 18-plane prt with OL_TRACE_PRT_COMPLEX", s);
  }
  if (n_rays > 0) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != sys->device)
      return fail(OL_EINVAL, "ol_trace_generate: current HIP device %d is not the system's "
                             "device %d", cur, sys->device);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dt == OL_F32)
    return do_trace_generate<float>(sys, sys->f32, n_rays, p, in, wavelength_index, record,
                                    record_stride, rays_out, prt, flags, status, extras, st);
  return do_trace_generate<double>(sys, sys->f64, n_rays, p, in, wavelength_index, record,
                                   record_stride, rays_out, prt, flags, status, extras, st);
}

int ol_generate_rays(const ol_raygen_params* p, ol_dtype dt, int64_t n,
                     const ol_raygen_inputs* in, void* const out[8], uint32_t* status,
                     void* stream) {
  if (!p || !in || !out) return fail(OL_EINVAL, "ol_generate_rays: NULL argument");
  if (n < 0) return fail(OL_EINVAL, "ol_generate_rays: negative count");
  for (int k = 0; k < 7; ++k)
    if (!out[k] && n > 0) return fail(OL_EINVAL, "ol_generate_rays: out[%d] is NULL", k);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dt == OL_F32) return do_generate_rays<float>(p, n, in, out, status, st);
  if (dt == OL_F64) return do_generate_rays<double>(p, n, in, out, status, st);
  return fail(OL_EINVAL, "ol_generate_rays: bad dtype %d", (int)dt);
}

int ol_trace_spot(const ol_system* sys, ol_dtype dt, int64_t n_rays, const ol_raygen_params* p,
                  const ol_raygen_inputs* in, double cx, double cy, int32_t wavelength_index,
                  void* const hits[3], double* out7, uint32_t* status, void* stream) {
  if (!sys) return fail(OL_EINVAL, "ol_trace_spot: system is NULL");
  OL_CHECK_CONSISTENT(sys, "ol_trace_spot");
  if (int rc = refuse_reference_newton("ol_trace_spot", sys)) return rc;
  if (dt != OL_F32 && dt != OL_F64) return fail(OL_EINVAL, "ol_trace_spot: bad dtype %d", (int)dt);
  if (!p || !in || !out7) return fail(OL_EINVAL, "ol_trace_spot: NULL argument");
  if (hits && (!hits[0] || !hits[1] || !hits[2]))
    return fail(OL_EINVAL, "ol_trace_spot: hits needs three planes");
  if (n_rays < 0) return fail(OL_EINVAL, "ol_trace_spot: negative ray count");
  if (wavelength_index < 0 || wavelength_index >= sys->n_wl)
    return fail(OL_EINVAL, "ol_trace_spot: wavelength index %d outside [0, %d)", wavelength_index,
                sys->n_wl);
  if (!(in->flags & OL_SPOT_POLARIZED_OK))
    for (int32_t s = 0; s < sys->n_surf; ++s)
      if (sys->coating[s] >= OL_COAT_FRESNEL)
        return fail(OL_EINVAL,
                    "Polarization must be set when surfaces have polarization-dependent "
                    "coatings.");
  if (n_rays > 0) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != sys->device)
      return fail(OL_EINVAL, "ol_trace_spot: current HIP device %d is not the system's device %d",
                  cur, sys->device);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dt == OL_F32)
    return do_trace_spot<float>(sys, sys->f32, n_rays, p, in, cx, cy, wavelength_index, hits,
                                out7, status, st);
  return do_trace_spot<double>(sys, sys->f64, n_rays, p, in, cx, cy, wavelength_index, hits,
                               out7, status, st);
}

int ol_trace_spot_batch(const ol_system* sys, ol_dtype dt, int64_t n_rays, const ol_raygen_params* p,
                        const ol_raygen_inputs* in, int32_t n_cells, const ol_spot_cell* cells,
                        void* hits, int64_t hits_stride, double* out8, uint32_t* status,
                        void* stream) {
  if (!sys) return fail(OL_EINVAL, "ol_trace_spot_batch: system is NULL");
  OL_CHECK_CONSISTENT(sys, "ol_trace_spot_batch");
  if (int rc = refuse_reference_newton("ol_trace_spot_batch", sys)) return rc;
  if (dt != OL_F32 && dt != OL_F64)
    return fail(OL_EINVAL, "ol_trace_spot_batch: bad dtype %d", (int)dt);
  if (!p || !in || !out8 || (n_cells > 0 && !cells))
    return fail(OL_EINVAL, "ol_trace_spot_batch: NULL argument");
  if (n_rays < 0) return fail(OL_EINVAL, "ol_trace_spot_batch: negative ray count");
  if (n_cells < 0 || n_cells > OL_SPOT_BATCH_MAX_CELLS)
    return fail(OL_EINVAL, "ol_trace_spot_batch: %d cells outside [0, %d]", n_cells,
                OL_SPOT_BATCH_MAX_CELLS);
  if (in->hx || in->hy || in->vx || in->vy)
    return fail(OL_EINVAL, "ol_trace_spot_batch: per-ray field / vignetting planes with cells "
                           "(each cell has ONE field point)");
  if (hits && hits_stride < n_rays)
    return fail(OL_EINVAL, "ol_trace_spot_batch: hits_stride %lld < %lld rays",
                (long long)hits_stride, (long long)n_rays);
  for (int32_t c = 0; c < n_cells; ++c) {
    const ol_system* from = cells[c].optics_of ? cells[c].optics_of : sys;
    if (from != sys) {
      OL_CHECK_CONSISTENT(from, "ol_trace_spot_batch (optics_of)");
      if (from->n_surf != sys->n_surf || from->device != sys->device)
        return fail(OL_EINVAL, "ol_trace_spot_batch: cell %d: optics_of has %d surfaces on device "
                               "%d, the system %d on device %d", c, from->n_surf, from->device,
                    sys->n_surf, sys->device);
    }
    if (cells[c].wavelength_index < 0 || cells[c].wavelength_index >= from->n_wl)
      return fail(OL_EINVAL, "ol_trace_spot_batch: cell %d: wavelength index %d outside [0, %d)",
                  c, cells[c].wavelength_index, from->n_wl);
    if (in->flags & OL_RAYGEN_CHECK_FIELD) {
      auto bad = [](double v) { return !(v >= -1.0 && v <= 1.0); };
      if (bad(cells[c].hx) || bad(cells[c].hy))  // real_ray_tracer.py:156-173, same text
        return fail(OL_EINVAL, "Normalized field coordinates must be within (-1, 1)");
    }
  }
  if (!(in->flags & OL_SPOT_POLARIZED_OK))
    for (int32_t s = 0; s < sys->n_surf; ++s)
      if (sys->coating[s] >= OL_COAT_FRESNEL)
        return fail(OL_EINVAL,
                    "Polarization must be set when surfaces have polarization-dependent "
                    "coatings.");
  if (n_rays > 0 && n_cells > 0) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != sys->device)
      return fail(OL_EINVAL,
                  "ol_trace_spot_batch: current HIP device %d is not the system's device %d", cur,
                  sys->device);
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (dt == OL_F32)
    return do_trace_spot_batch<float>(sys, sys->f32, n_rays, p, in, n_cells, cells, hits,
                                      hits_stride, out8, status, st);
  return do_trace_spot_batch<double>(sys, sys->f64, n_rays, p, in, n_cells, cells, hits,
                                     hits_stride, out8, status, st);
}

int ol_irradiance(ol_dtype dt, int64_t n_rays, const void* x, const void* y, const void* power,
                  const double* x_edges, int32_t nx, const double* y_edges, int32_t ny,
                  double* hist, void* stream) {
  if (!x || !y || !power || !x_edges || !y_edges || !hist)
    return fail(OL_EINVAL, "ol_irradiance: NULL argument");
  if (n_rays < 0) return fail(OL_EINVAL, "ol_irradiance: negative count");
  if (nx < 1 || ny < 1 || (int64_t)nx * ny > (int64_t)1 << 28)
    return fail(OL_EINVAL, "ol_irradiance: %d x %d bins", nx, ny);
  if (n_rays == 0) return OL_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e;
  if (dt == OL_F32)
    e = ol::launch_irradiance<float>(n_rays, (const float*)x, (const float*)y,
                                     (const float*)power, x_edges, nx, y_edges, ny, hist, st);
  else if (dt == OL_F64)
    e = ol::launch_irradiance<double>(n_rays, (const double*)x, (const double*)y,
                                      (const double*)power, x_edges, nx, y_edges, ny, hist, st);
  else
    return fail(OL_EINVAL, "ol_irradiance: bad dtype %d", (int)dt);
  if (e != hipSuccess) return fail(OL_EHIP, "launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_radial_energy(ol_dtype dt, int64_t n_rays, const void* x, const void* y,
                     const void* intensity, double cx, double cy, const double* r_step,
                     int32_t n_steps, double* bins, void* stream) {
  if (!x || !y || !intensity || !r_step || !bins)
    return fail(OL_EINVAL, "ol_radial_energy: NULL argument");
  if (n_rays < 0) return fail(OL_EINVAL, "ol_radial_energy: negative count");
  if (n_steps < 1 || n_steps > 1024)
    return fail(OL_EINVAL, "ol_radial_energy: n_steps %d outside [1, 1024]", n_steps);
  if (n_rays == 0) return OL_OK;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e;
  if (dt == OL_F32)
    e = ol::launch_radial_energy<float>(n_rays, (const float*)x, (const float*)y,
                                        (const float*)intensity, cx, cy, r_step, n_steps, bins, st);
  else if (dt == OL_F64)
    e = ol::launch_radial_energy<double>(n_rays, (const double*)x, (const double*)y,
                                         (const double*)intensity, cx, cy, r_step, n_steps, bins,
                                         st);
  else
    return fail(OL_EINVAL, "ol_radial_energy: bad dtype %d", (int)dt);
  if (e != hipSuccess) return fail(OL_EHIP, "launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_polarized_intensity(ol_dtype dt, int64_t n_rays, const void* prt, int32_t prt_complex,
                           const void* const k0[3], const void* i0,
                           const ol_polarization_state* state, void* intensity,
                           uint32_t* status, void* stream) {
  if (!prt || !k0 || !k0[0] || !k0[1] || !k0[2] || !i0 || !state || !intensity)
    return fail(OL_EINVAL, "ol_polarized_intensity: NULL argument");
  if (n_rays < 0) return fail(OL_EINVAL, "ol_polarized_intensity: negative count");
  if (n_rays == 0) return OL_OK;
  ol::PolStateDev s{state->is_polarized, state->Ex, state->Ey, state->phase_x, state->phase_y};
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e;
  if (dt == OL_F32) {
    const float* k[3] = {(const float*)k0[0], (const float*)k0[1], (const float*)k0[2]};
    e = ol::launch_pol_intensity<float>(n_rays, (const float*)prt, prt_complex != 0, k,
                                        (const float*)i0, s,
                                        (float*)intensity, status, st);
  } else if (dt == OL_F64) {
    const double* k[3] = {(const double*)k0[0], (const double*)k0[1], (const double*)k0[2]};
    e = ol::launch_pol_intensity<double>(n_rays, (const double*)prt, prt_complex != 0, k,
                                         (const double*)i0, s,
                                         (double*)intensity, status, st);
  } else {
    return fail(OL_EINVAL, "ol_polarized_intensity: bad dtype %d", (int)dt);
  }
  if (e != hipSuccess) return fail(OL_EHIP, "launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_wavefront_opd(const ol_wavefront_params* p, ol_dtype dt, int64_t n_rays,
                     const void* const rays[7], const void* px, const void* py,
                     void* opd_waves, void* const pupil[3], void* stream) {
  if (!p || !rays || !px || !py || !opd_waves)
    return fail(OL_EINVAL, "ol_wavefront_opd: NULL argument");
  if (n_rays < 0) return fail(OL_EINVAL, "ol_wavefront_opd: negative count");
  if (n_rays == 0) return OL_OK;
  for (int k = 0; k < 7; ++k)
    if (!rays[k]) return fail(OL_EINVAL, "ol_wavefront_opd: rays[%d] is NULL", k);
  if (pupil && (!pupil[0] || !pupil[1] || !pupil[2]))
    return fail(OL_EINVAL, "ol_wavefront_opd: pupil planes must all be given or pupil = NULL");
  ol::WavefrontDev d{p->xc, p->yc, p->zc, p->R, p->n_image, p->opd_ref,
                     p->ux, p->uy, p->half_epd, p->wavelength_um, p->nx, p->ny, p->nz};
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e;
  if (dt == OL_F32) {
    const float* r[7];
    for (int k = 0; k < 7; ++k) r[k] = static_cast<const float*>(rays[k]);
    float* pu[3] = {pupil ? (float*)pupil[0] : nullptr, pupil ? (float*)pupil[1] : nullptr,
                    pupil ? (float*)pupil[2] : nullptr};
    e = ol::launch_wavefront<float>(d, n_rays, r, (const float*)px, (const float*)py,
                                    (float*)opd_waves, pupil ? pu : nullptr, st);
  } else if (dt == OL_F64) {
    const double* r[7];
    for (int k = 0; k < 7; ++k) r[k] = static_cast<const double*>(rays[k]);
    double* pu[3] = {pupil ? (double*)pupil[0] : nullptr, pupil ? (double*)pupil[1] : nullptr,
                     pupil ? (double*)pupil[2] : nullptr};
    e = ol::launch_wavefront<double>(d, n_rays, r, (const double*)px, (const double*)py,
                                     (double*)opd_waves, pupil ? pu : nullptr, st);
  } else {
    return fail(OL_EINVAL, "ol_wavefront_opd: bad dtype %d", (int)dt);
  }
  if (e != hipSuccess) return fail(OL_EHIP, "launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_trace_opd(const ol_system* sys, ol_dtype dt, int64_t n_rays, const ol_raygen_params* p,
                 const ol_raygen_inputs* in, const ol_wavefront_params* w,
                 int32_t wavelength_index, void* opd_waves, void* intensity,
                 void* const pupil[3], double* moments12, uint32_t* status, void* stream) {
  if (!sys) return fail(OL_EINVAL, "ol_trace_opd: system is NULL");
  OL_CHECK_CONSISTENT(sys, "ol_trace_opd");
  if (int rc = refuse_reference_newton("ol_trace_opd", sys)) return rc;
  if (dt != OL_F64)
    return fail(dt == OL_F32 ? OL_EUNSUPPORTED : OL_EINVAL,
                "ol_trace_opd: wavefront work is fp64 only (dtype %d)", (int)dt);
  if (!p || !in || !w || !opd_waves || !intensity || !moments12)
    return fail(OL_EINVAL, "ol_trace_opd: NULL argument");
  if (pupil && (!pupil[0] || !pupil[1] || !pupil[2]))
    return fail(OL_EINVAL, "ol_trace_opd: pupil planes must all be given or pupil = NULL");
  if (n_rays < 0) return fail(OL_EINVAL, "ol_trace_opd: negative ray count");
  if (wavelength_index < 0 || wavelength_index >= sys->n_wl)
    return fail(OL_EINVAL, "ol_trace_opd: wavelength index %d outside [0, %d)", wavelength_index,
                sys->n_wl);
  for (int32_t s = 0; s < sys->n_surf; ++s)
    if (sys->coating[s] >= OL_COAT_FRESNEL)
      return fail(OL_EINVAL,
                  "Polarization must be set when surfaces have polarization-dependent "
                  "coatings.");
  if (n_rays > 0) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != sys->device)
      return fail(OL_EINVAL, "ol_trace_opd: current HIP device %d is not the system's device %d",
                  cur, sys->device);
  }
  return do_trace_opd<double>(sys, sys->f64, n_rays, p, in, w, wavelength_index, opd_waves,
                              intensity, pupil, moments12, status,
                              static_cast<hipStream_t>(stream));
}

static int opd_common_checks(const char* who, const ol_system* sys, ol_dtype dt,
                             int32_t wavelength_index) {
  if (!sys) return fail(OL_EINVAL, "%s: system is NULL", who);
  if (!sys->consistent)
    return fail(OL_EINVAL, "%s: the system's tables are inconsistent after a failed "
                           "ol_system_update (destroy it and create a new one)", who);
  if (dt == OL_F32)
    return fail(OL_EUNSUPPORTED, "%s: wavefront work is fp64 only (an OPD in waves needs 1e-9 "
                                 "of the path length)", who);
  if (dt != OL_F64) return fail(OL_EINVAL, "%s: bad dtype %d", who, (int)dt);
  if (wavelength_index < 0 || wavelength_index >= sys->n_wl)
    return fail(OL_EINVAL, "%s: wavelength index %d outside [0, %d)", who, wavelength_index,
                sys->n_wl);
  if (int rc = refuse_reference_newton(who, sys)) return rc;
  for (int32_t s = 0; s < sys->n_surf; ++s)
    if (sys->coating[s] >= OL_COAT_FRESNEL)
      return fail(OL_EINVAL,
                  "Polarization must be set when surfaces have polarization-dependent "
                  "coatings.");
  int cur = -1;
  if (hipGetDevice(&cur) != hipSuccess || cur != sys->device)
    return fail(OL_EINVAL, "%s: current HIP device %d is not the system's device %d", who, cur,
                sys->device);
  return OL_OK;
}

int ol_wavefront_reference(const ol_system* sys, ol_dtype dt, const ol_raygen_params* p,
                           const ol_raygen_inputs* in, const ol_wavefront_params* w,
                           double pupil_z, int32_t planar, int32_t wavelength_index,
                           void* reference_dev, void* chief8, uint32_t* status, void* stream) {
  if (int rc = opd_common_checks("ol_wavefront_reference", sys, dt, wavelength_index)) return rc;
  if (!p || !in || !w || !reference_dev)
    return fail(OL_EINVAL, "ol_wavefront_reference: NULL argument");
  if (in->hx || in->hy || in->vx || in->vy)
    return fail(OL_EINVAL, "ol_wavefront_reference: one field point (launch-uniform field and "
                           "vignetting)");
  return do_wavefront_reference<double>(sys, sys->f64, p, in, w, pupil_z, planar,
                                        wavelength_index, reference_dev, chief8, status,
                                        static_cast<hipStream_t>(stream));
}

int ol_trace_opd_dev(const ol_system* sys, ol_dtype dt, int64_t n_rays,
                     const ol_raygen_params* p, const ol_raygen_inputs* in,
                     const void* reference_dev, int32_t wavelength_index, void* opd_waves,
                     void* intensity, void* const pupil[3], double* moments12, uint32_t* status,
                     void* stream) {
  if (int rc = opd_common_checks("ol_trace_opd_dev", sys, dt, wavelength_index)) return rc;
  if (!p || !in || !reference_dev || !opd_waves || !intensity || !moments12)
    return fail(OL_EINVAL, "ol_trace_opd_dev: NULL argument");
  if (pupil && (!pupil[0] || !pupil[1] || !pupil[2]))
    return fail(OL_EINVAL, "ol_trace_opd_dev: pupil needs three planes");
  if (n_rays < 0) return fail(OL_EINVAL, "ol_trace_opd_dev: negative ray count");
  return do_trace_opd<double>(sys, sys->f64, n_rays, p, in, nullptr, wavelength_index, opd_waves,
                              intensity, pupil, moments12, status,
                              static_cast<hipStream_t>(stream), reference_dev);
}

int ol_wavefront_fit(int32_t kind, const ol_wavefront_params* w, double trim_std,
                     uint32_t flags, int32_t planar, int64_t n_rays,
                     const double* const rays[8], const double* px, const double* py,
                     double* workspace, void* reference_dev, uint32_t* fit_status,
                     void* stream) {
  static_assert(OL_WAVEFRONT_FIT_WORKSPACE_DOUBLES == ol::kFitWorkspaceDoubles, "header");
  static_assert(sizeof(ol::WavefrontConsts<double>) <= OL_WAVEFRONT_REFERENCE_DOUBLES * 8,
                "header: the device reference structure");
  static_assert(OL_FIT_NO_VALID == ol::kFitNoValid && OL_FIT_TOO_FEW == ol::kFitTooFew &&
                OL_FIT_NO_ALIVE == ol::kFitNoAlive && OL_FIT_SINGULAR == ol::kFitSingular &&
                OL_FIT_CENTROID == ol::kFitCentroid && OL_FIT_BEST_FIT == ol::kFitBestFit,
                "header");
  if (!w || !rays || !px || !py || !workspace || !reference_dev || !fit_status)
    return fail(OL_EINVAL, "ol_wavefront_fit: NULL argument");
  if (kind != OL_FIT_CENTROID && kind != OL_FIT_BEST_FIT)
    return fail(OL_EINVAL, "ol_wavefront_fit: unknown kind %d", (int)kind);
  if (n_rays < 0) return fail(OL_EINVAL, "ol_wavefront_fit: negative count");
  if (flags & ~(uint32_t)(OL_FIT_STD_DDOF1 | OL_FIT_PISTON_SKIPS_NAN))
    return fail(OL_EINVAL, "ol_wavefront_fit: unknown flags 0x%x", (unsigned)flags);
  if (!(w->n_image > 0.0) || !(w->wavelength_um > 0.0))
    return fail(OL_EINVAL, "ol_wavefront_fit: n_image %g, wavelength %g", w->n_image,
                w->wavelength_um);
  for (int k = 0; k < 8; ++k)
    if (!rays[k]) return fail(OL_EINVAL, "ol_wavefront_fit: rays[%d] is NULL", k);
  ol::FitArgs a{};
  a.p.ni = w->n_image;
  a.p.inv_w = 1.0 / (w->wavelength_um * 1e-3);
  a.p.ux = w->ux;
  a.p.uy = w->uy;
  a.p.half_epd = w->half_epd;
  a.p.trim_std = trim_std > 0.0 ? trim_std : 0.0;  // (NaN: no trimming either)
  a.p.kind = kind;
  a.p.planar = planar ? 1 : 0;
  a.p.ddof = (flags & OL_FIT_STD_DDOF1) ? 1 : 0;
  a.p.skip_nan = (flags & OL_FIT_PISTON_SKIPS_NAN) ? 1 : 0;
  for (int k = 0; k < 8; ++k) a.ray[k] = rays[k];
  a.px = px;
  a.py = py;
  a.n = n_rays;
  a.workspace = workspace;
  a.out = static_cast<ol::WavefrontConsts<double>*>(reference_dev);
  a.status = fit_status;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = ol::launch_wavefront_fit(a, st);  // (its first pass writes fit_status)
  if (e != hipSuccess) return fail(OL_EHIP, "ol_wavefront_fit: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_wavefront_opd_fitted(int64_t n_rays, const double* const rays[7], const double* px,
                            const double* py, const void* reference_dev, double* opd_waves,
                            double* const pupil[3], void* stream) {
  if (!rays || !px || !py || !reference_dev || !opd_waves)
    return fail(OL_EINVAL, "ol_wavefront_opd_fitted: NULL argument");
  if (n_rays < 0) return fail(OL_EINVAL, "ol_wavefront_opd_fitted: negative count");
  for (int k = 0; k < 7; ++k)
    if (!rays[k]) return fail(OL_EINVAL, "ol_wavefront_opd_fitted: rays[%d] is NULL", k);
  if (pupil && (!pupil[0] || !pupil[1] || !pupil[2]))
    return fail(OL_EINVAL, "ol_wavefront_opd_fitted: pupil needs three planes");
  if (n_rays == 0) return OL_OK;
  hipError_t e = ol::launch_wavefront_fitted(
      static_cast<const ol::WavefrontConsts<double>*>(reference_dev), n_rays, rays, px, py,
      opd_waves, pupil, static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(OL_EHIP, "ol_wavefront_opd_fitted: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_pupil_fill(ol_dtype dt, int64_t n_rays, const void* opd_waves, const void* intensity,
                  const void* pupil_x, const void* pupil_y, const double plane[3],
                  const int32_t* cell, int32_t n_side, int32_t grid_size, double* grid,
                  void* stream) {
  if (!opd_waves || !intensity || !cell || !grid)
    return fail(OL_EINVAL, "ol_pupil_fill: NULL argument");
  if ((pupil_x == nullptr) != (pupil_y == nullptr) || (pupil_x && !plane))
    return fail(OL_EINVAL, "ol_pupil_fill: pupil_x, pupil_y and plane go together");
  if (n_rays < 0 || n_side < 1 || grid_size < n_side || grid_size > (1 << 15))
    return fail(OL_EINVAL, "ol_pupil_fill: n_rays %lld, %d samples per side, grid %d",
                (long long)n_rays, n_side, grid_size);
  if (n_rays > (int64_t)n_side * n_side)
    return fail(OL_EINVAL, "ol_pupil_fill: more samples than cells");
  if (n_rays == 0) return OL_OK;
  const double zero[3] = {0.0, 0.0, 0.0};
  const double* co = plane ? plane : zero;
  const int32_t pad = (grid_size - n_side) / 2;  // psf/fft.py:139-160
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e;
  if (dt == OL_F32)
    e = ol::launch_pupil_fill<float>(n_rays, (const float*)opd_waves, (const float*)intensity,
                                     (const float*)pupil_x, (const float*)pupil_y, co, cell,
                                     n_side, grid_size, pad, grid, st);
  else if (dt == OL_F64)
    e = ol::launch_pupil_fill<double>(n_rays, (const double*)opd_waves, (const double*)intensity,
                                      (const double*)pupil_x, (const double*)pupil_y, co, cell,
                                      n_side, grid_size, pad, grid, st);
  else
    return fail(OL_EINVAL, "ol_pupil_fill: bad dtype %d", (int)dt);
  if (e != hipSuccess) return fail(OL_EHIP, "launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_pupil_points(int32_t kind, int32_t param, ol_dtype dt, int64_t n_points,
                    const int32_t* row_first, const int64_t* row_offset, void* x, void* y,
                    void* stream) {
  if (kind != OL_PUPIL_HEXAPOLAR && kind != OL_PUPIL_UNIFORM)
    return fail(OL_EINVAL, "ol_pupil_points: unknown sampler %d", kind);
  if (n_points < 0 || (n_points > 0 && (!x || !y)))
    return fail(OL_EINVAL, "ol_pupil_points: bad output planes");
  if (kind == OL_PUPIL_HEXAPOLAR) {
    if (param < 0 || n_points != 1 + 3 * (int64_t)param * ((int64_t)param + 1))
      return fail(OL_EINVAL, "ol_pupil_points: %d rings are %lld points, not %lld", param,
                  (long long)(1 + 3 * (int64_t)param * ((int64_t)param + 1)),
                  (long long)n_points);
  } else {
    if (param < 2) return fail(OL_EINVAL, "ol_pupil_points: uniform grid side %d < 2", param);
    if (n_points > 0 && (!row_first || !row_offset))
      return fail(OL_EINVAL, "ol_pupil_points: the uniform sampler needs its row tables");
  }
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e;
  if (dt == OL_F32)
    e = ol::launch_pupil_points<float>(kind, param, n_points, row_first, row_offset,
                                       static_cast<float*>(x), static_cast<float*>(y), st);
  else if (dt == OL_F64)
    e = ol::launch_pupil_points<double>(kind, param, n_points, row_first, row_offset,
                                        static_cast<double*>(x), static_cast<double*>(y), st);
  else
    return fail(OL_EINVAL, "ol_pupil_points: bad dtype %d", (int)dt);
  if (e != hipSuccess) return fail(OL_EHIP, "pupil launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_math_probe(int32_t op, ol_dtype dt, int64_t n, const void* a, const void* b, void* out,
                  void* stream) {
  if (op < 0 || op > 3) return fail(OL_EINVAL, "ol_math_probe: op must be 0..3");
  if (n < 0 || (n > 0 && (!a || !out || (op == 1 && !b))))
    return fail(OL_EINVAL, "ol_math_probe: NULL argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e;
  if (dt == OL_F32)
    e = ol::launch_math_probe<float>(op, n, static_cast<const float*>(a),
                                     static_cast<const float*>(b), static_cast<float*>(out), st);
  else if (dt == OL_F64)
    e = ol::launch_math_probe<double>(op, n, static_cast<const double*>(a),
                                      static_cast<const double*>(b), static_cast<double*>(out),
                                      st);
  else
    return fail(OL_EINVAL, "ol_math_probe: bad dtype %d", (int)dt);
  if (e != hipSuccess) return fail(OL_EHIP, "probe launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_stream_fill(void* dst, int64_t bytes, int32_t store_bytes, int32_t planes,
                   uint32_t pattern, void* stream) {
  if (bytes < 0 || (bytes > 0 && !dst)) return fail(OL_EINVAL, "ol_stream_fill: bad buffer");
  if (planes < 1) return fail(OL_EINVAL, "ol_stream_fill: planes must be >= 1");
  if (store_bytes != 4 && store_bytes != 8 && store_bytes != 16)
    return fail(OL_EINVAL, "ol_stream_fill: store_bytes must be 4, 8 or 16");
  if (bytes % ((int64_t)store_bytes * planes) != 0 ||
      (reinterpret_cast<uintptr_t>(dst) % store_bytes) != 0)
    return fail(OL_EINVAL, "ol_stream_fill: buffer not a multiple of planes x store_bytes, or "
                           "not aligned to store_bytes");
  hipError_t e = ol::launch_stream_fill(dst, bytes, store_bytes, planes, pattern,
                                        static_cast<hipStream_t>(stream));
  if (e != hipSuccess) return fail(OL_EHIP, "fill launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_arena_alloc(int64_t bytes, void** out) {
  if (!out) return fail(OL_EINVAL, "ol_arena_alloc: out is NULL");
  *out = nullptr;
  if (bytes <= 0) return fail(OL_EINVAL, "ol_arena_alloc: %lld bytes", (long long)bytes);
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, (size_t)bytes);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // (an out-of-memory answer is not a sticky error)
    return fail(OL_EHIP, "ol_arena_alloc: %lld bytes: %s", (long long)bytes,
                hipGetErrorString(e));
  }
  *out = p;
  return OL_OK;
}

int ol_arena_free(void* arena) {
  if (!arena) return OL_OK;
  hipError_t e = hipFree(arena);
  if (e != hipSuccess) return fail(OL_EHIP, "ol_arena_free: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_spot_moments(ol_dtype dt, int64_t n_rays, const void* x, const void* y,
                    const void* intensity, double* out6, void* stream) {
  if (!x || !y || !intensity || !out6) return fail(OL_EINVAL, "ol_spot_moments: NULL argument");
  if (n_rays <= 0) return n_rays == 0 ? OL_OK : fail(OL_EINVAL, "ol_spot_moments: negative count");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = dt == OL_F32
                     ? ol::launch_spot_moments<float>(n_rays, (const float*)x, (const float*)y,
                                                      (const float*)intensity, out6, st)
                     : ol::launch_spot_moments<double>(n_rays, (const double*)x, (const double*)y,
                                                       (const double*)intensity, out6, st);
  if (e != hipSuccess) return fail(OL_EHIP, "launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

int ol_spot_max_r2(ol_dtype dt, int64_t n_rays, const void* x, const void* y,
                   const void* intensity, double cx, double cy, double* out1, void* stream) {
  if (!x || !y || !intensity || !out1) return fail(OL_EINVAL, "ol_spot_max_r2: NULL argument");
  if (n_rays <= 0) return n_rays == 0 ? OL_OK : fail(OL_EINVAL, "ol_spot_max_r2: negative count");
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipError_t e = dt == OL_F32
                     ? ol::launch_spot_max_r2<float>(n_rays, (const float*)x, (const float*)y,
                                                     (const float*)intensity, cx, cy, out1, st)
                     : ol::launch_spot_max_r2<double>(n_rays, (const double*)x, (const double*)y,
                                                      (const double*)intensity, cx, cy, out1, st);
  if (e != hipSuccess) return fail(OL_EHIP, "launch failed: %s", hipGetErrorString(e));
  return OL_OK;
}

}  // extern "C"
