// harness.hip -- TEST INFRASTRUCTURE: the kernel's per-surface arithmetic, executed ray by
// ray on the host, behind the same C ABI.
//
// What this is.  optiland_amd/csrc/surface_math.h holds every arithmetic function of the
// fused trace kernel (intersection, Newton-Raphson on the sag functors, apertures,
// refraction / reflection, coatings, Fresnel + PRT update, frame changes).  In the product
// library those are `__device__` functions: device code only.  This translation unit
// defines OL_HOST_MATH, which turns them into `__host__ __device__`, and drives them with a
// plain loop over rays that mirrors trace_kernel<T, 1, RECORD, POLK, NR, false> line by
// line.  Linked with an unmodified, host-only compile of csrc/capi.hip and a five-function
// stand-in for the HIP runtime (hipMalloc = malloc: the surface table stays in host
// memory), it gives `libol_hostmath.so`: `ol_system_create` / `ol_trace_ex` with HOST
// pointers.  tests/test_hostmath.py holds it against the golden vectors and the oracle --
// the same checks tests/test_gpu_parity.py makes on the MI355X -- so that a change to the
// kernel arithmetic can be verified on a box without a GPU.
//
// What this is not.  It is not a CPU fallback: nothing under optiland_amd/ builds, loads
// or knows about this library, `optiland_amd._capi` only ever opens liboptiland_hip.so
// and raises when that is missing, and the entry points other than ol_system_* / ol_trace*
// / ol_generate_rays return "not supported" here.  It exists under tests/ and is built by
// tests/hostmath/build.py only.
//
// Round 4: the packed-pair form of the lean fp32 generating launch (trace_kernel<float, 2, ...,
// kGenUniform>) is mirrored as well -- surface_step<f32x2, ...> compiles for the host, two rays
// in one ext_vector -- and chosen by the same rule as the device's launcher.
//
// Differences from the device: v_rcp_f32 / v_sqrt_f32 / v_rsq_f32 / v_exp_f32 (1 ulp) are
// the correctly rounded host operations; the order of operations, the FMA contractions
// written out as fma() and every branch are the kernel's own.
#define OL_HOST_MATH 1
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "../../optiland_amd/csrc/surface_math.h"
#include "../../optiland_amd/csrc/raygen_device.h"
#include "../../optiland_amd/csrc/wavefront_device.h"
#include "../../optiland_amd/csrc/wavefront_fit_device.h"
#include "../../optiland_amd/csrc/epilogue_device.h"
#include "../../optiland_amd/csrc/trace_launch.h"

// ---------------------------------------------------------------------------
// stand-in for the five HIP runtime calls capi.hip makes for its surface tables
// ---------------------------------------------------------------------------
extern "C" {
hipError_t hipMalloc(void** p, size_t n) {
  *p = std::malloc(n ? n : 1);
  return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void* p) {
  std::free(p);
  return hipSuccess;
}
hipError_t hipGetLastError(void) { return hipSuccess; }  // (ol_arena_alloc clears a failed malloc)
hipError_t hipMemcpy(void* dst, const void* src, size_t n, hipMemcpyKind) {
  std::memcpy(dst, src, n);
  return hipSuccess;
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t n, hipMemcpyKind, hipStream_t) {
  std::memcpy(dst, src, n);
  return hipSuccess;
}
hipError_t hipMemsetAsync(void* dst, int value, size_t n, hipStream_t) {
  std::memset(dst, value, n);
  return hipSuccess;
}
hipError_t hipGetDevice(int* d) {
  *d = 0;
  return hipSuccess;
}
const char* hipGetErrorString(hipError_t e) {
  return e == hipErrorNotSupported ? "entry point not available in the host-math harness"
                                   : "host-math harness: stand-in HIP runtime error";
}
// marker symbol: lets a test assert which library it has loaded
int ol_hostmath_harness(void) { return 1; }
// how many generating launches took the packed-pair form (tests/test_generate_fused.py)
static long g_pair_launches = 0;
long ol_hostmath_pair_launches(void) { return g_pair_launches; }
}

namespace ol {

Tuning& tuning() {
  static Tuning t;
  return t;
}

namespace {

// trace_kernel<T, 1, RECORD, POLK, NR, false> for ray i (trace_kernel.hip), statement by
// statement; the loads / stores are plain indexed accesses.
template <typename T, int POLK, int NR, bool GEN = false>
void trace_one(const TraceArgs<T>& a, int64_t i, uint32_t& status) {
  constexpr int NPRT = POLK == 2 ? 18 : 9;
  Ray<T> r[1];
  if constexpr (GEN) {  // trace_kernel<..., GEN = true>: the generating prologue
    // (GEN's kGenFieldPlanes / kGenApod forms are template variants on the device; one
    // run-time switch here)
    const RaygenIn<T>& in_ = a.in;
    T tx = in_.tx0, ty = in_.ty0, vx = in_.vx0, vy = in_.vy0, o[6];
    if (in_.hx != nullptr) {
      const T hx = in_.hx[i], hy = in_.hy[i];
      if (in_.vx != nullptr) { vx = in_.vx[i]; vy = in_.vy[i]; }
      if ((in_.flags & kRaygenCheckField) && (outside_unit(hx) || outside_unit(hy)))
        status |= kStatusFieldRange;
      raygen_field<T>(a.rgc, hx, hy, tx, ty);
    }
    T px = in_.px[i], py = in_.py[i];
    raygen_pupil<T>(in_.flags, vx, vy, px, py, status);
    raygen_one<T>(a.rgc, tx, ty, px, py, vx, vy, o);
    r[0].x = o[0]; r[0].y = o[1]; r[0].z = o[2];
    r[0].L = o[3]; r[0].M = o[4]; r[0].N = o[5];
    r[0].i = a.rgc.apod_kind != 0 ? raygen_apodize<T>(a.rgc, px, py) : T(1);
    r[0].opd = T(0);
  } else {
    r[0].x = a.rays[0][i]; r[0].y = a.rays[1][i]; r[0].z = a.rays[2][i];
    r[0].L = a.rays[3][i]; r[0].M = a.rays[4][i]; r[0].N = a.rays[5][i];
    r[0].i = a.rays[6][i]; r[0].opd = a.rays[7][i];
  }
  Prt<T, POLK> P[1];
  if constexpr (POLK != 0) {
    const bool ident = GEN || (a.flags & kTracePrtIdentity) != 0;
    for (int e = 0; e < NPRT; ++e)
      P[0].m[e] = ident ? ((e == 0 || e == 4 || e == 8) ? T(1) : T(0)) : a.prt[(int64_t)e * a.n + i];
  }
  bool is_global = true;
  bool prt_fresh = POLK != 0 && (GEN || (a.flags & kTracePrtIdentity) != 0);
  DevSurf<T> last_traced;
  std::memset(static_cast<void*>(&last_traced), 0, sizeof(last_traced));
  last_traced.cold = a.cold;
  const int rec_from = a.record_from > a.first ? a.record_from : a.first;
  for (int s = a.first; s <= a.last; ++s) {
    DevSurf<T> S;
    static_cast<DevSurfHot<T>&>(S) = a.surf[s];
    S.cold = a.cold + s;
    if (S.interaction != kRecordOnly) {
      // (trace_kernel: OL_TRACE_NONUNIT_K on polarised launches over the caller's rays)
      const uint32_t polf =
          (POLK != 0 && !GEN && (a.flags & kTraceNonUnitK)) ? kPolNonUnitK : 0u;
      if constexpr (NR != 0) {
        // the Newton kernels hand surface_step table POINTERS and every phase re-reads its
        // fields (SurfFetched, device_table.h) -- same handle here
        const SurfFetched<T> h{a.surf + s, a.cold + s, a.optics + (s * a.n_wl + a.wl)};
        if constexpr (NR == kNrReference) {  // (trace_kernel: the batch's iteration counts)
          const NrRefCtl ctl{a.nr_iters, s, a.n_surf, a.nr_count_at};
          surface_step<T, 1, POLK, NR>(h, a.coeffs, is_global, r, P, status, prt_fresh, &ctl,
                                       polf);
        } else {
          surface_step<T, 1, POLK, NR>(h, a.coeffs, is_global, r, P, status, prt_fresh, nullptr,
                                       polf);
        }
      } else {
        const DevOptics<T> O = a.optics[s * a.n_wl + a.wl];
        surface_step<T, 1, POLK, NR>(S, O, a.coeffs, is_global, r, P, status, prt_fresh, nullptr,
                                     polf);
      }
      is_global = false;
      last_traced = S;
    }
    if (a.record && s >= rec_from) {
      T* row = a.record + (int64_t)(s - rec_from) * 8 * a.record_stride;
      if (!(s == a.first && (a.flags & kTraceRow0IsInput))) {
        const Ray<T> g = is_global ? r[0] : to_global<T>(last_traced, r[0]);
        row[0 * a.record_stride + i] = g.x; row[1 * a.record_stride + i] = g.y;
        row[2 * a.record_stride + i] = g.z; row[3 * a.record_stride + i] = g.L;
        row[4 * a.record_stride + i] = g.M; row[5 * a.record_stride + i] = g.N;
        row[6 * a.record_stride + i] = g.i; row[7 * a.record_stride + i] = g.opd;
      }
    }
  }
  if (r[0].L != r[0].L && r[0].x == r[0].x) status |= kStatusNanDirection;  // (trace_kernel)
  if (a.flags & kTraceWriteRays) {
    const Ray<T> g = is_global ? r[0] : to_global<T>(last_traced, r[0]);
    a.rays[0][i] = g.x; a.rays[1][i] = g.y; a.rays[2][i] = g.z;
    a.rays[3][i] = g.L; a.rays[4][i] = g.M; a.rays[5][i] = g.N;
    a.rays[6][i] = g.i; a.rays[7][i] = g.opd;
  }
  if constexpr (POLK != 0) {
    for (int e = 0; e < NPRT; ++e) a.prt[(int64_t)e * a.n + i] = P[0].m[e];
  }
}

// trace_kernel<float, 2, true, 0, 0, SPOT, kGenUniform> (the packed PAIR form of the lean fp32
// generating launch): two rays generated one after the other, traced as ONE f32x2 through the
// same surface_step<f32x2, ...> the device instantiates; a ragged tail's second ray is a
// padding ray from pupil point (0, 0) that is neither reported nor stored.
inline void trace_pair_gen(const TraceArgs<float>& a, int64_t i0, uint32_t& status) {
  using V = f32x2;
  const int cnt = (a.n - i0) >= 2 ? 2 : 1;
  Ray<V> r[1];
  const RaygenIn<float>& in_ = a.in;
  for (int k = 0; k < 2; ++k) {
    float px = k < cnt ? in_.px[i0 + k] : 0.0f, py = k < cnt ? in_.py[i0 + k] : 0.0f, o[6];
    uint32_t st_k = 0;
    raygen_pupil<float>(in_.flags, in_.vx0, in_.vy0, px, py, st_k);
    if (k < cnt) status |= st_k;
    raygen_one<float>(a.rgc, in_.tx0, in_.ty0, px, py, in_.vx0, in_.vy0, o);
    r[0].x[k] = o[0]; r[0].y[k] = o[1]; r[0].z[k] = o[2];
    r[0].L[k] = o[3]; r[0].M[k] = o[4]; r[0].N[k] = o[5];
    r[0].i[k] = 1.0f; r[0].opd[k] = 0.0f;
  }
  Prt<float, 0> P[1];
  bool is_global = true, prt_fresh = false;
  DevSurf<float> last_traced;
  std::memset(static_cast<void*>(&last_traced), 0, sizeof(last_traced));
  last_traced.cold = a.cold;
  const int rec_from = a.record_from > a.first ? a.record_from : a.first;
  auto put = [&](float* const plane[8], int64_t stride_or_zero, float* row) {
    const Ray<V> g = is_global ? r[0] : to_global<V>(last_traced, r[0]);
    const V f[8] = {g.x, g.y, g.z, g.L, g.M, g.N, g.i, g.opd};
    for (int q = 0; q < 8; ++q)
      for (int k = 0; k < cnt; ++k) {
        if (row) row[q * stride_or_zero + i0 + k] = f[q][k];
        else plane[q][i0 + k] = f[q][k];
      }
  };
  for (int s = a.first; s <= a.last; ++s) {
    DevSurf<float> S;
    static_cast<DevSurfHot<float>&>(S) = a.surf[s];
    S.cold = a.cold + s;
    if (S.interaction != kRecordOnly) {
      const DevOptics<float> O = a.optics[s * a.n_wl + a.wl];
      surface_step<V, 1, 0, 0>(S, O, a.coeffs, is_global, r, P, status, prt_fresh);
      is_global = false;
      last_traced = S;
    }
    if (a.record && s >= rec_from && !(s == a.first && (a.flags & kTraceRow0IsInput)))
      put(nullptr, a.record_stride, a.record + (int64_t)(s - rec_from) * 8 * a.record_stride);
  }
  for (int k = 0; k < cnt; ++k)
    if (r[0].L[k] != r[0].L[k] && r[0].x[k] == r[0].x[k]) status |= kStatusNanDirection;
  if (a.flags & kTraceWriteRays) put(a.rays, 0, nullptr);
}

// trace_kernel<float, 2, true, 1, kNrZernike, false, kGenUniform, EPI> (OL_POLZ_PAIR: the polarised
// Zernike fp32 launch on packed pairs): the general generating prologue per ray, then ONE f32x2
// through surface_step<f32x2, 1, 1, kNrZernike> with a pair of PRT matrices -- what the device
// instantiates.  (The launcher sends even ray counts only.)
inline void trace_pair_gen_polz(const TraceArgs<float>& a, int64_t i0, uint32_t& status) {
  using V = f32x2;
  const int cnt = (a.n - i0) >= 2 ? 2 : 1;
  Ray<V> r[1];
  const RaygenIn<float>& in_ = a.in;
  for (int k = 0; k < 2; ++k) {
    const int64_t i = i0 + (k < cnt ? k : 0);
    float tx = in_.tx0, ty = in_.ty0, vx = in_.vx0, vy = in_.vy0, o[6];
    uint32_t st_k = 0;
    if (in_.hx != nullptr) {
      const float hx = in_.hx[i], hy = in_.hy[i];
      if (in_.vx != nullptr) { vx = in_.vx[i]; vy = in_.vy[i]; }
      if ((in_.flags & kRaygenCheckField) && (outside_unit(hx) || outside_unit(hy)))
        st_k |= kStatusFieldRange;
      raygen_field<float>(a.rgc, hx, hy, tx, ty);
    }
    float px = k < cnt ? in_.px[i] : 0.0f, py = k < cnt ? in_.py[i] : 0.0f;
    raygen_pupil<float>(in_.flags, vx, vy, px, py, st_k);
    if (k < cnt) status |= st_k;
    raygen_one<float>(a.rgc, tx, ty, px, py, vx, vy, o);
    r[0].x[k] = o[0]; r[0].y[k] = o[1]; r[0].z[k] = o[2];
    r[0].L[k] = o[3]; r[0].M[k] = o[4]; r[0].N[k] = o[5];
    r[0].i[k] = a.rgc.apod_kind != 0 ? raygen_apodize<float>(a.rgc, px, py) : 1.0f;
    r[0].opd[k] = 0.0f;
  }
  Prt<V, 1> P[1];
  for (int e = 0; e < 9; ++e) P[0].m[e] = V{(e % 4 == 0) ? 1.0f : 0.0f, (e % 4 == 0) ? 1.0f : 0.0f};
  bool is_global = true, prt_fresh = true;
  DevSurf<float> last_traced;
  std::memset(static_cast<void*>(&last_traced), 0, sizeof(last_traced));
  last_traced.cold = a.cold;
  const int rec_from = a.record_from > a.first ? a.record_from : a.first;
  auto put = [&](float* const plane[8], int64_t stride_or_zero, float* row) {
    const Ray<V> g = is_global ? r[0] : to_global<V>(last_traced, r[0]);
    const V f[8] = {g.x, g.y, g.z, g.L, g.M, g.N, g.i, g.opd};
    for (int q = 0; q < 8; ++q)
      for (int k = 0; k < cnt; ++k) {
        if (row) row[q * stride_or_zero + i0 + k] = f[q][k];
        else plane[q][i0 + k] = f[q][k];
      }
  };
  for (int s = a.first; s <= a.last; ++s) {
    DevSurf<float> S;
    static_cast<DevSurfHot<float>&>(S) = a.surf[s];
    S.cold = a.cold + s;
    if (S.interaction != kRecordOnly) {
      const SurfFetched<float> h{a.surf + s, a.cold + s, a.optics + (s * a.n_wl + a.wl)};
      surface_step<V, 1, 1, kNrZernike>(h, a.coeffs, is_global, r, P, status, prt_fresh, nullptr, 0u);
      is_global = false;
      last_traced = S;
    }
    if (a.record && s >= rec_from && !(s == a.first && (a.flags & kTraceRow0IsInput)))
      put(nullptr, a.record_stride, a.record + (int64_t)(s - rec_from) * 8 * a.record_stride);
  }
  for (int k = 0; k < cnt; ++k)
    if (r[0].L[k] != r[0].L[k] && r[0].x[k] == r[0].x[k]) status |= kStatusNanDirection;
  if (a.flags & kTraceWriteRays) put(a.rays, 0, nullptr);
  for (int e = 0; e < 9; ++e)
    for (int k = 0; k < cnt; ++k) a.prt[(int64_t)e * a.n + i0 + k] = P[0].m[e][k];
}

template <typename T, int POLK, int NR, bool GEN = false>
void trace_all(const TraceArgs<T>& a) {
  uint32_t status = 0;
  for (int64_t i = 0; i < a.n; ++i) trace_one<T, POLK, NR, GEN>(a, i, status);
  if (status && a.status) *a.status |= status;
}

template <typename T, int NR>
hipError_t trace_nr(const TraceArgs<T>& a) {
  if (a.spot != nullptr) return hipErrorNotSupported;  // the spot epilogue is a wave reduction
  const int polk = a.prt == nullptr ? 0 : ((a.flags & kTracePrtComplex) ? 2 : 1);
  if (polk == 2) trace_all<T, 2, NR>(a);
  else if (polk == 1) trace_all<T, 1, NR>(a);
  else trace_all<T, 0, NR>(a);
  return hipSuccess;
}

}  // namespace

// same choice of instantiation as trace_kernel.hip:launch_trace / launch_rpt (RPT = 1)
template <typename T>
hipError_t launch_trace(const TraceArgs<T>& a, bool, int nr_family, hipStream_t) {
  switch (nr_family) {
    case kNrNone: return trace_nr<T, kNrNone>(a);
    case kNrZernike: return trace_nr<T, kNrZernike>(a);
    case kNrEvenAsphere: return trace_nr<T, kNrEvenAsphere>(a);
    case kNrReference:
      if (a.nr_iters == nullptr) return hipErrorInvalidValue;
      return trace_nr<T, kNrReference>(a);
    default: return trace_nr<T, kNrGeneric>(a);
  }
}
template hipError_t launch_trace<float>(const TraceArgs<float>&, bool, int, hipStream_t);
template hipError_t launch_trace<double>(const TraceArgs<double>&, bool, int, hipStream_t);

// launch_trace_generate (trace_kernel.hip): the generating variant, same instance choice
template <typename T, int NR>
static hipError_t gen_nr(const TraceArgs<T>& a, bool pair_ok) {
  const int polk = a.prt == nullptr ? 0 : ((a.flags & kTracePrtComplex) ? 2 : 1);
  if (polk != 0 && a.spot != nullptr) return hipErrorInvalidValue;  // as launch_gen_nr
  if (a.spot != nullptr && (a.in.hx != nullptr || a.rgc.apod_kind != 0))
    return hipErrorInvalidValue;
  bool paired = false;
  if constexpr (sizeof(T) == 4 && NR == 0) {
    // the device's choice (trace_kernel.hip: launch_gen_nr): the lean fp32 form on packed pairs
    const int want = tuning().rays_per_thread;
    if (pair_ok && polk == 0 && a.in.hx == nullptr && a.rgc.apod_kind == 0 &&
        (want == 3 || want == 0)) {
      uint32_t status = 0;
      ++g_pair_launches;
      for (int64_t i = 0; i < a.n; i += 2) trace_pair_gen(a, i, status);
      if (status && a.status) *a.status |= status;
      paired = true;
    }
  }
  if constexpr (sizeof(T) == 4 && NR == kNrZernike) {
    // the device's choice (trace_kernel.hip: launch_gen_nr): the polarised Zernike pair
    const int want = tuning().rays_per_thread;
    if (pair_ok && polk == 1 && a.spot == nullptr && a.n % 2 == 0 &&
        (want == 3 || (want == 0 && OL_POLZ_PAIR))) {
      uint32_t status = 0;
      ++g_pair_launches;
      for (int64_t i = 0; i < a.n; i += 2) trace_pair_gen_polz(a, i, status);
      if (status && a.status) *a.status |= status;
      paired = true;
    }
  }
  if (paired) {
  } else if (polk == 2) trace_all<T, 2, NR, true>(a);
  else if (polk == 1) trace_all<T, 1, NR, true>(a);
  else trace_all<T, 0, NR, true>(a);
  if (a.spot != nullptr) {
    // the spot epilogue of the generating launch (ABI 8): on the device a workgroup
    // reduction of the final state in registers; here the same per-ray accumulation
    // (epilogue_device.h) over the last recorded row, ray by ray into slot 0
    const int rec_from = a.record_from > a.first ? a.record_from : a.first;
    const T* row = a.record + (int64_t)(a.last - rec_from) * 8 * a.record_stride;
    double m[6] = {0, 0, 0, 0, 0, 0}, rmax = 0.0;
    for (int64_t j = 0; j < a.n; ++j)
      spot_accumulate<T>(m, rmax, row[j], row[a.record_stride + j], row[6 * a.record_stride + j],
                         a.cx, a.cy);
    for (int k = 0; k < 6; ++k) a.spot[k] += m[k];
    a.spot[6] = rmax > a.spot[6] ? rmax : a.spot[6];
  }
  if (polk != 0 && a.i_updated != nullptr) {
    // the update_intensity epilogue of the generating launch (trace_kernel.hip, ABI 7): the
    // device takes the matrix from its registers and regenerates the launch direction; here
    // both are read back from what the launch wrote (PRT planes, row 0 of the record)
    uint32_t flag = 0;
    for (int64_t j = 0; j < a.n; ++j) {
      T P[9], Q[9];
      for (int e = 0; e < 9; ++e) {
        P[e] = a.prt[(int64_t)e * a.n + j];
        Q[e] = polk == 2 ? a.prt[(int64_t)(9 + e) * a.n + j] : T(0);
      }
      const T* row0 = a.record;
      const T kx = row0[3 * a.record_stride + j], ky = row0[4 * a.record_stride + j],
              kz = row0[5 * a.record_stride + j];
      // `_i0`: the intensity the ray was generated with (1, or the pupil apodization)
      const T i0 = row0[6 * a.record_stride + j];
      a.i_updated[j] = polk == 2
                           ? pol_intensity_one<T, true>(a.pf, kx, ky, kz, P, Q, i0, flag)
                           : pol_intensity_one<T, false>(a.pf, kx, ky, kz, P, Q, i0, flag);
    }
    if (flag && a.status) *a.status |= flag;
  }
  return hipSuccess;
}
template <typename T>
hipError_t launch_trace_generate(const TraceArgs<T>& a, int nr_family, bool pair_ok,
                                 hipStream_t) {
  switch (nr_family) {
    case kNrNone: return gen_nr<T, kNrNone>(a, pair_ok);
    case kNrZernike: return gen_nr<T, kNrZernike>(a, pair_ok);
    case kNrEvenAsphere: return gen_nr<T, kNrEvenAsphere>(a, false);
    case kNrReference: return hipErrorInvalidValue;  // (as the device launcher)
    default: return gen_nr<T, kNrGeneric>(a, false);
  }
}
template hipError_t launch_trace_generate<float>(const TraceArgs<float>&, int, bool, hipStream_t);
template hipError_t launch_trace_generate<double>(const TraceArgs<double>&, int, bool,
                                                  hipStream_t);

// launch_pupil_points (aux_kernels.hip), point by point, the same fp64 expressions
template <typename T>
hipError_t launch_pupil_points(int kind, int32_t param, int64_t n, const int32_t* first,
                               const int64_t* offset, T* x, T* y, hipStream_t) {
#pragma clang fp contract(off)
  auto lin = [](int64_t i, int64_t last, double start, double stop, double step) {
    const double v = (double)i * step;
    return i == last ? stop : v + start;
  };
  if (kind == 0) {
    const double step = 1.0 / (double)param;
    for (int64_t p = 0; p < n; ++p) {
      if (p == 0) { x[0] = T(0); y[0] = T(0); continue; }
      const int64_t q = p - 1;
      int64_t i = (int64_t)((3.0 + std::sqrt(9.0 + 12.0 * (double)q)) / 6.0);
      while (3 * i * (i - 1) > q) --i;
      while (3 * (i + 1) * i <= q) ++i;
      const int64_t j = q - 3 * i * (i - 1);
      const double theta = (double)j * (6.283185307179586 / (double)(6 * i));
      const double r = lin(i, param, 0.0, 1.0, step);
      x[p] = (T)(r * std::cos(theta));
      y[p] = (T)(r * std::sin(theta));
    }
  } else {
    const double step = 2.0 / (double)(param - 1);
    for (int64_t p = 0; p < n; ++p) {
      int lo = 0, hi = param;
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offset[mid] <= p) lo = mid; else hi = mid;
      }
      const int64_t col = (int64_t)first[lo] + (p - offset[lo]);
      x[p] = (T)lin(col, param - 1, -1.0, 1.0, step);
      y[p] = (T)lin(lo, param - 1, -1.0, 1.0, step);
    }
  }
  return hipSuccess;
}
template hipError_t launch_pupil_points<float>(int, int32_t, int64_t, const int32_t*,
                                               const int64_t*, float*, float*, hipStream_t);
template hipError_t launch_pupil_points<double>(int, int32_t, int64_t, const int32_t*,
                                                const int64_t*, double*, double*, hipStream_t);

// launch_math_probe (aux_kernels.hip): the host run of Math<T> (IEEE operations)
template <typename T>
hipError_t launch_math_probe(int op, int64_t n, const T* a, const T* b, T* out, hipStream_t) {
  using m = Math<T>;
  for (int64_t j = 0; j < n; ++j) {
    const T x = a[j], y = b ? b[j] : T(0);
    out[j] = op == 0 ? m::rcp(x) : (op == 1 ? m::div(x, y) : (op == 2 ? m::sqrt(x) : m::rsqrt(x)));
  }
  return hipSuccess;
}
template hipError_t launch_math_probe<float>(int, int64_t, const float*, const float*, float*,
                                             hipStream_t);
template hipError_t launch_math_probe<double>(int, int64_t, const double*, const double*, double*,
                                              hipStream_t);

// launch_stream_fill (aux_kernels.hip): the device kernel is a bandwidth yardstick; the
// host stand-in just writes the pattern
hipError_t launch_stream_fill(void* dst, int64_t bytes, int, int, uint32_t pattern, hipStream_t) {
  uint32_t* p = static_cast<uint32_t*>(dst);
  for (int64_t j = 0; j < bytes / 4; ++j) p[j] = pattern;
  return hipSuccess;
}

// raygen_kernel + launch_raygen (aux_kernels.hip), ray by ray
template <typename T>
hipError_t launch_raygen(const RaygenDev& p, const RaygenIn<T>& in_, int64_t n, T* const out[8],
                         uint32_t* status, hipStream_t) {
  RaygenIn<T> in = in_;
  if (in.hx == nullptr) uniform_field_tangents<T>(p, in);
  const RaygenConsts<T> c(p);
  const bool field_planes = in.hx != nullptr, vig_planes = in.vx != nullptr;
  uint32_t st = 0;
  for (int64_t j = 0; j < n; ++j) {
    T tx = in.tx0, ty = in.ty0, o[6];
    if (field_planes) {
      const T hx = in.hx[j], hy = in.hy[j];
      if ((in.flags & kRaygenCheckField) && (outside_unit(hx) || outside_unit(hy)))
        st |= kStatusFieldRange;
      raygen_field<T>(c, hx, hy, tx, ty);
    }
    T px = in.px[j], py = in.py[j];
    const T vx = vig_planes ? in.vx[j] : in.vx0, vy = vig_planes ? in.vy[j] : in.vy0;
    raygen_pupil<T>(in.flags, vx, vy, px, py, st);
    raygen_one<T>(c, tx, ty, px, py, vx, vy, o);
    for (int k = 0; k < 6; ++k) out[k][j] = o[k];
    out[6][j] = raygen_apodize<T>(c, px, py);
    if (out[7]) out[7][j] = T(0);
  }
  if (st && status) *status |= st;
  return hipSuccess;
}
template hipError_t launch_raygen<float>(const RaygenDev&, const RaygenIn<float>&, int64_t,
                                         float* const[8], uint32_t*, hipStream_t);
template hipError_t launch_raygen<double>(const RaygenDev&, const RaygenIn<double>&, int64_t,
                                          double* const[8], uint32_t*, hipStream_t);

// pol_intensity_kernel (aux_kernels.hip), ray by ray
template <typename T>
hipError_t launch_pol_intensity(int64_t n, const T* prt, bool prt_complex, const T* const k0[3],
                                const T* i0, const PolStateDev& st, T* intensity,
                                uint32_t* status, hipStream_t) {
  const PolFields<T> fld(st);
  uint32_t flag = 0;
  for (int64_t j = 0; j < n; ++j) {
    T P[9], Q[9];
    for (int e = 0; e < 9; ++e) {
      P[e] = prt[(int64_t)e * n + j];
      Q[e] = prt_complex ? prt[(int64_t)(9 + e) * n + j] : T(0);
    }
    intensity[j] = prt_complex
                       ? pol_intensity_one<T, true>(fld, k0[0][j], k0[1][j], k0[2][j], P, Q, i0[j], flag)
                       : pol_intensity_one<T, false>(fld, k0[0][j], k0[1][j], k0[2][j], P, Q, i0[j], flag);
  }
  if (flag && status) *status |= flag;
  return hipSuccess;
}
template hipError_t launch_pol_intensity<float>(int64_t, const float*, bool, const float* const[3],
                                                const float*, const PolStateDev&, float*,
                                                uint32_t*, hipStream_t);
template hipError_t launch_pol_intensity<double>(int64_t, const double*, bool,
                                                 const double* const[3], const double*,
                                                 const PolStateDev&, double*, uint32_t*,
                                                 hipStream_t);

// wavefront_kernel (aux_kernels.hip), ray by ray
template <typename T>
hipError_t launch_wavefront(const WavefrontDev& p, int64_t n, const T* const rays[7], const T* px,
                            const T* py, T* opd_waves, T* const pupil[3], hipStream_t) {
  const WavefrontConsts<T> w(p);
  for (int64_t j = 0; j < n; ++j) {
    T pu[3];
    opd_waves[j] = wavefront_one<T>(w, rays[0][j], rays[1][j], rays[2][j], rays[3][j], rays[4][j],
                                    rays[5][j], rays[6][j], px[j], py[j], pu);
    if (pupil[0]) {
      pupil[0][j] = pu[0];
      pupil[1][j] = pu[1];
      pupil[2][j] = pu[2];
    }
  }
  return hipSuccess;
}
template hipError_t launch_wavefront<float>(const WavefrontDev&, int64_t, const float* const[7],
                                            const float*, const float*, float*, float* const[3],
                                            hipStream_t);
template hipError_t launch_wavefront<double>(const WavefrontDev&, int64_t, const double* const[7],
                                             const double*, const double*, double*,
                                             double* const[3], hipStream_t);

// pupil_fill_kernel (aux_kernels.hip), sample by sample
template <typename T>
hipError_t launch_pupil_fill(int64_t n, const T* opd, const T* inten, const T* pupil_x,
                             const T* pupil_y, const double coef[3], const int32_t* cell,
                             int32_t n_side, int32_t grid, int32_t pad, double* out, hipStream_t) {
  for (int64_t j = 0; j < n; ++j) {
    double o = (double)opd[j];
    if (pupil_x) o -= coef[0] + coef[1] * (double)pupil_x[j] + coef[2] * (double)pupil_y[j];
    double re, im;
    pupil_sample(o, (double)inten[j], re, im);
    const int64_t at = pupil_cell_offset(cell[j], n_side, grid, pad);
    out[at] = re;
    out[at + 1] = im;
  }
  return hipSuccess;
}
template hipError_t launch_pupil_fill<float>(int64_t, const float*, const float*, const float*,
                                             const float*, const double[3], const int32_t*,
                                             int32_t, int32_t, int32_t, double*, hipStream_t);
template hipError_t launch_pupil_fill<double>(int64_t, const double*, const double*,
                                              const double*, const double*, const double[3],
                                              const int32_t*, int32_t, int32_t, int32_t, double*,
                                              hipStream_t);

// The surface loop of the fused kernels for one ray (spot_trace_kernel / opd_trace_kernel:
// generate -> trace -> final global state), RPT = 1
template <typename T, int NR>
Ray<T> fused_trace_one(const DevSurfHot<T>* surf, const DevSurfCold<T>* cold,
                       const DevOptics<T>* optics, const T* coeffs, int first, int last,
                       int n_wl, int wl, Ray<T> q, uint32_t& status, bool keep_local = false) {
  Ray<T> r[1] = {q};
  Prt<T, 0> P[1];
  bool is_global = true, prt_fresh = false;
  DevSurf<T> last_traced;
  std::memset(static_cast<void*>(&last_traced), 0, sizeof(last_traced));
  last_traced.cold = cold;
  for (int s = first; s <= last; ++s) {
    DevSurf<T> S;
    static_cast<DevSurfHot<T>&>(S) = surf[s];
    S.cold = cold + s;
    if (S.interaction != kRecordOnly) {
      const DevOptics<T> O = optics[s * n_wl + wl];
      surface_step<T, 1, 0, NR>(S, O, coeffs, is_global, r, P, status, prt_fresh);
      is_global = false;
      last_traced = S;
    }
  }
  return (is_global || keep_local) ? r[0] : to_global<T>(last_traced, r[0]);
}

template <typename T>
Ray<T> fused_trace_family(int family, const DevSurfHot<T>* surf, const DevSurfCold<T>* cold,
                          const DevOptics<T>* optics, const T* coeffs, int first, int last,
                          int n_wl, int wl, const Ray<T>& q, uint32_t& status,
                          bool keep_local = false) {
  switch (family) {
    case kNrNone: return fused_trace_one<T, kNrNone>(surf, cold, optics, coeffs, first, last, n_wl, wl, q, status, keep_local);
    case kNrZernike: return fused_trace_one<T, kNrZernike>(surf, cold, optics, coeffs, first, last, n_wl, wl, q, status, keep_local);
    case kNrEvenAsphere: return fused_trace_one<T, kNrEvenAsphere>(surf, cold, optics, coeffs, first, last, n_wl, wl, q, status, keep_local);
    default: return fused_trace_one<T, kNrGeneric>(surf, cold, optics, coeffs, first, last, n_wl, wl, q, status, keep_local);
  }
}

// spot_trace_kernel + launch_spot_trace (trace_kernel.hip), ray by ray; the sums are
// formed in ray order (the device adds wave / workgroup partials: equal to rounding)
template <typename T>
static hipError_t launch_spot_cell(const SpotArgs<T>& a, int nr_family);
template <typename T>
hipError_t launch_spot_trace(const SpotArgs<T>& a_in, bool, int nr_family, hipStream_t) {
  SpotArgs<T> a = a_in;
  if (a.in.hx == nullptr) uniform_field_tangents<T>(a.rg, a.in);
  return launch_spot_cell<T>(a, nr_family);
}
// one (field, wavelength) cell; the launch-uniform tangents are in a.in.tx0 / ty0 already
template <typename T>
static hipError_t launch_spot_cell(const SpotArgs<T>& a, int nr_family) {
  const RaygenConsts<T> c(a.rg);
  const RaygenIn<T>& in_ = a.in;
  const bool field_planes = in_.hx != nullptr, vig_planes = in_.vx != nullptr;
  double s[6] = {0, 0, 0, 0, 0, 0}, rmax = 0.0;
  uint32_t status = 0;
  for (int64_t j = 0; j < a.n; ++j) {
    T px = in_.px[j], py = in_.py[j];
    const T vx = vig_planes ? in_.vx[j] : in_.vx0, vy = vig_planes ? in_.vy[j] : in_.vy0;
    T tx = in_.tx0, ty = in_.ty0, o[6];
    if (field_planes) {
      if ((in_.flags & kRaygenCheckField) && (outside_unit(in_.hx[j]) || outside_unit(in_.hy[j])))
        status |= kStatusFieldRange;
      raygen_field<T>(c, in_.hx[j], in_.hy[j], tx, ty);
    }
    raygen_pupil<T>(in_.flags, vx, vy, px, py, status);
    raygen_one<T>(c, tx, ty, px, py, vx, vy, o);
    Ray<T> q;
    q.x = o[0]; q.y = o[1]; q.z = o[2];
    q.L = o[3]; q.M = o[4]; q.N = o[5];
    q.i = a.rg.apod_kind != 0 ? raygen_apodize<T>(c, px, py) : T(1);
    q.opd = T(0);
    const Ray<T> g = fused_trace_family<T>(nr_family, a.surf, a.cold, a.optics, a.coeffs, a.first,
                                           a.last, a.n_wl, a.wl, q, status,
                                           (in_.flags & kSpotHitsLocal) != 0);
    spot_accumulate<T>(s, rmax, g.x, g.y, g.i, a.cx, a.cy);
    if (a.hits[0] != nullptr) {
      a.hits[0][j] = g.x;
      a.hits[1][j] = g.y;
      a.hits[2][j] = g.i;
    }
  }
  for (int k = 0; k < 6; ++k) a.out[k] += s[k];
  if (rmax > a.out[6]) a.out[6] = rmax;
  if (status && a.status) *a.status |= status;
  return hipSuccess;
}
template hipError_t launch_spot_trace<float>(const SpotArgs<float>&, bool, int, hipStream_t);
template hipError_t launch_spot_trace<double>(const SpotArgs<double>&, bool, int, hipStream_t);

// launch_spot_batch (trace_kernel.hip: blockIdx.y = cell): cell by cell through the single-cell
// launch above, with what the cell changes put where that launch reads it
template <typename T>
hipError_t launch_spot_batch(const SpotArgs<T>& a_in, const SpotBatch<T>& b, bool vec, int nr_family,
                             hipStream_t st) {
  for (int c = 0; c < b.n_cells; ++c) {
    SpotArgs<T> a = a_in;
    a.in.vx0 = b.c[c].vx;
    a.in.vy0 = b.c[c].vy;
    a.in.tx0 = b.c[c].tx;
    a.in.ty0 = b.c[c].ty;
    a.cx = b.c[c].cx;
    a.cy = b.c[c].cy;
    a.wl = b.c[c].wl;
    a.optics = b.c[c].optics;
    a.n_wl = b.c[c].n_wl;
    for (int k = 0; k < 3; ++k)
      if (a.hits[k]) a.hits[k] += (int64_t)c * 3 * b.hits_stride;
    a.out = a_in.out + 8 * c;
    hipError_t e = launch_spot_cell<T>(a, nr_family);
    if (e != hipSuccess) return e;
  }
  (void)vec; (void)st;
  return hipSuccess;
}
template hipError_t launch_spot_batch<float>(const SpotArgs<float>&, const SpotBatch<float>&, bool,
                                             int, hipStream_t);
template hipError_t launch_spot_batch<double>(const SpotArgs<double>&, const SpotBatch<double>&,
                                              bool, int, hipStream_t);

// opd_trace_kernel + launch_opd_trace (trace_kernel.hip), ray by ray
template <typename T>
hipError_t launch_opd_trace(const OpdArgs<T>& a_in, bool /*vector_ok*/, int nr_family,
                            hipStream_t) {
  OpdArgs<T> a = a_in;
  uniform_field_tangents<T>(a.rg, a.in);
  const RaygenConsts<T> c(a.rg);
  // (ol_trace_opd_dev: the reference left in "device" memory by launch_chief_reference)
  const WavefrontConsts<T> w = a.wf_dev ? *a.wf_dev : WavefrontConsts<T>(a.wf);
  const RaygenIn<T>& in_ = a.in;
  uint32_t status = 0;
  double s[kOpdMoments];
  for (int k = 0; k < kOpdMoments; ++k) s[k] = 0.0;
  for (int64_t j = 0; j < a.n; ++j) {
    T px = in_.px[j], py = in_.py[j];
    T vx = in_.vx0, vy = in_.vy0, o[6];
    raygen_pupil<T>(in_.flags, vx, vy, px, py, status);
    raygen_one<T>(c, in_.tx0, in_.ty0, px, py, vx, vy, o);
    Ray<T> q;
    q.x = o[0]; q.y = o[1]; q.z = o[2];
    q.L = o[3]; q.M = o[4]; q.N = o[5];
    q.i = a.rg.apod_kind != 0 ? raygen_apodize<T>(c, px, py) : T(1);
    q.opd = T(0);
    const Ray<T> g = fused_trace_family<T>(nr_family, a.surf, a.cold, a.optics, a.coeffs, a.first,
                                           a.last, a.n_wl, a.wl, q, status);
    T pu[3];
    Ray<T> gp = g;
    final_propagate<T, false>(w, gp);
    const T ov = wavefront_one<T>(w, gp.x, gp.y, gp.z, gp.L, gp.M, gp.N, gp.opd, in_.px[j],
                                  in_.py[j], pu);
    a.opd[j] = ov;
    a.inten[j] = gp.i;
    if (a.pupil[0]) {
      a.pupil[0][j] = pu[0];
      a.pupil[1][j] = pu[1];
      a.pupil[2][j] = pu[2];
    }
    opd_accumulate(s, (double)gp.i, (double)ov, (double)pu[0], (double)pu[1], gp.i > T(0));
  }
  for (int k = 0; k < kOpdMoments; ++k) a.mom[k] += s[k];
  if (status && a.status) *a.status |= status;
  return hipSuccess;
}
template hipError_t launch_opd_trace<double>(const OpdArgs<double>&, bool, int, hipStream_t);

// chief_ref_kernel + launch_chief_reference (trace_kernel.hip), statement by statement
template <typename T>
hipError_t launch_chief_reference(const ChiefArgs<T>& a_in, int nr_family, hipStream_t) {
  using m = Math<T>;
  ChiefArgs<T> a = a_in;
  uniform_field_tangents<T>(a.rg, a.in);
  const RaygenConsts<T> c(a.rg);
  uint32_t status = 0;
  T px = T(0), py = T(0), vx = a.in.vx0, vy = a.in.vy0, o[6];
  raygen_pupil<T>(a.in.flags, vx, vy, px, py, status);
  raygen_one<T>(c, a.in.tx0, a.in.ty0, px, py, vx, vy, o);
  Ray<T> q;
  q.x = o[0]; q.y = o[1]; q.z = o[2];
  q.L = o[3]; q.M = o[4]; q.N = o[5];
  q.i = T(1);
  q.opd = T(0);
  Ray<T> g = fused_trace_family<T>(nr_family == kNrNone ? kNrNone : kNrGeneric, a.surf,
                                   a.cold, a.optics, a.coeffs, a.first, a.last, a.n_wl, a.wl,
                                   q, status);
  WavefrontConsts<T> w = a.wfc;
  final_propagate<T, true>(w, g);
  w.xc = g.x; w.yc = g.y; w.zc = g.z;
  T t_back;
  if (w.planar) {
    w.R = T(0);
    w.nx = g.L; w.ny = g.M; w.nz = g.N;
    t_back = T(0);
  } else {
    const T dz = g.z - a.pupil_z;
    w.R = m::sqrt(g.x * g.x + g.y * g.y + dz * dz);
    w.nx = w.ny = w.nz = T(0);
    const T aa = g.L * g.L + g.M * g.M + g.N * g.N;
    const T d = T(4) * aa * w.R * w.R;
    const T sq = m::sqrt(d < T(0) ? T(0) : d);
    const T t1 = m::div(-sq, T(2) * aa), t2 = m::div(sq, T(2) * aa);
    t_back = t1 < T(0) ? t2 : t1;
  }
  w.opd_ref = g.opd - w.ni * t_back;
  *a.out = w;
  if (a.chief != nullptr) {
    a.chief[0] = g.x; a.chief[1] = g.y; a.chief[2] = g.z; a.chief[3] = g.L;
    a.chief[4] = g.M; a.chief[5] = g.N; a.chief[6] = g.i; a.chief[7] = g.opd;
  }
  if (status && a.status) *a.status |= status;
  return hipSuccess;
}
template hipError_t launch_chief_reference<double>(const ChiefArgs<double>&, int, hipStream_t);

// fit_pass_kernel + launch_wavefront_fit (aux_kernels.hip): the same passes in the same order,
// each pass's sums formed ray by ray (the device adds per-block rows in block order)
template <int PASS>
void fit_pass(const FitArgs& a) {
  FitState* st = reinterpret_cast<FitState*>(a.workspace);
  const FitState seen = *st;
  WavefrontConsts<double> ref{};
  if (PASS == kPassMean) ref = *a.out;
  double s[kFitSums] = {0};
  for (int64_t j = 0; j < a.n; ++j) {
    const FitRay r{a.ray[0][j], a.ray[1][j], a.ray[2][j], a.ray[3][j], a.ray[4][j],
                   a.ray[5][j], a.ray[6][j], a.ray[7][j], a.px[j],     a.py[j]};
    fit_accumulate<PASS>(a.p, seen, ref, r, s);
  }
  uint32_t bits = 0;
  fit_finish<PASS>(a.p, *st, s, a.out, &bits);
  if (PASS == kPassC1 || PASS == kPassB1) *a.status = bits;
  else *a.status |= bits;
}

hipError_t launch_wavefront_fit(const FitArgs& a, hipStream_t) {
  std::memset(a.workspace, 0, kFitStateDoubles * sizeof(double));
  if (a.p.kind == kFitBestFit) {
    fit_pass<kPassB1>(a);
    fit_pass<kPassB2>(a);
  } else {
    fit_pass<kPassC1>(a);
    if (a.p.trim_std > 0.0) {
      fit_pass<kPassC2>(a);
      fit_pass<kPassC3>(a);
      fit_pass<kPassC4>(a);
    }
    if (!a.p.planar) fit_pass<kPassC5>(a);
  }
  fit_pass<kPassMean>(a);
  return hipSuccess;
}

hipError_t launch_wavefront_fitted(const WavefrontConsts<double>* ref, int64_t n,
                                   const double* const rays[7], const double* px,
                                   const double* py, double* opd_waves, double* const pupil[3],
                                   hipStream_t) {
  const WavefrontConsts<double> w = *ref;
  for (int64_t j = 0; j < n; ++j) {
    double pu[3];
    opd_waves[j] = wavefront_one<double, true>(w, rays[0][j], rays[1][j], rays[2][j], rays[3][j],
                                               rays[4][j], rays[5][j], rays[6][j], px[j], py[j],
                                               pu);
    if (pupil) {
      pupil[0][j] = pu[0];
      pupil[1][j] = pu[1];
      pupil[2][j] = pu[2];
    }
  }
  return hipSuccess;
}

// the four reduction kernels of aux_kernels.hip, element by element (sums in element order)
template <typename T>
hipError_t launch_spot_moments(int64_t n, const T* x, const T* y, const T* inten, double* out6,
                               hipStream_t) {
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t j = 0; j < n; ++j) {
    const double xv = (double)x[j], yv = (double)y[j];
    if (inten[j] > T(0)) {
      s[0] += 1.0; s[1] += xv; s[2] += yv; s[3] += xv * xv; s[4] += yv * yv; s[5] += 1.0;
    }
  }
  for (int k = 0; k < 6; ++k) out6[k] += s[k];
  return hipSuccess;
}

template <typename T>
hipError_t launch_spot_max_r2(int64_t n, const T* x, const T* y, const T* inten, double cx,
                              double cy, double* out1, hipStream_t) {
  double best = 0.0;
  for (int64_t j = 0; j < n; ++j) {
    if (inten[j] > T(0)) {
      const double dx = (double)x[j] - cx, dy = (double)y[j] - cy;
      const double r2 = dx * dx + dy * dy;
      best = r2 > best ? r2 : best;
    }
  }
  if (best > *out1) *out1 = best;
  return hipSuccess;
}

template <typename T>
hipError_t launch_radial_energy(int64_t n, const T* x, const T* y, const T* inten, double cx,
                                double cy, const double* r_step, int n_steps, double* bins,
                                hipStream_t) {
  if (n_steps < 1 || n_steps > 1024) return hipErrorInvalidValue;  // kMaxEeSteps
  for (int64_t j = 0; j < n; ++j) {
    const double e = (double)inten[j];
    const double dx = (double)x[j] - cx, dy = (double)y[j] - cy;
    const int lo = radial_step_index(r_step, n_steps, sqrt(dx * dx + dy * dy), e);
    if (lo >= 0 && e != 0.0) bins[lo] += e;
  }
  return hipSuccess;
}

template <typename T>
hipError_t launch_irradiance(int64_t n, const T* x, const T* y, const T* power,
                             const double* xe, int nx, const double* ye, int ny, double* hist,
                             hipStream_t) {
  if (nx < 1 || ny < 1) return hipErrorInvalidValue;
  for (int64_t j = 0; j < n; ++j) {
    const double p = (double)power[j];
    if (!(p > 0.0)) continue;
    const int ix = edge_bin(xe, nx, (double)x[j]);
    if (ix < 0) continue;
    const int iy = edge_bin(ye, ny, (double)y[j]);
    if (iy < 0) continue;
    hist[(int64_t)ix * ny + iy] += p;
  }
  return hipSuccess;
}

#define OL_INST(T)                                                                             \
  template hipError_t launch_spot_moments<T>(int64_t, const T*, const T*, const T*, double*,   \
                                             hipStream_t);                                     \
  template hipError_t launch_irradiance<T>(int64_t, const T*, const T*, const T*,              \
                                           const double*, int, const double*, int, double*,    \
                                           hipStream_t);                                       \
  template hipError_t launch_radial_energy<T>(int64_t, const T*, const T*, const T*, double,   \
                                              double, const double*, int, double*,             \
                                              hipStream_t);                                    \
  template hipError_t launch_spot_max_r2<T>(int64_t, const T*, const T*, const T*, double,     \
                                            double, double*, hipStream_t);
OL_INST(float)
OL_INST(double)
#undef OL_INST

}  // namespace ol
