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
// The per-surface arithmetic lives in surface_math.h (one definition for every kernel
// here, compiled a second time for the host by tests/hostmath as a checker); it follows
// SURVEY.md Appendix A and each function cites the reference lines it implements.
// Differences that are deliberate:
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
#include "epilogue_device.h"
#include "trace_launch.h"

#ifndef OL_TABLE_IN_LDS
#define OL_TABLE_IN_LDS 0  // 1: stage the surface table in LDS (measured slower)
#endif
// Hot-block prefetch policy (A/B knobs, tools/build_variants.py).  The next surface's
// DevSurfHot (16 elements: 16 SGPRs in fp32, 32 in fp64) is requested before the current
// surface is worked on.  Measured good for the fp32 lean trace kernel (round 1) and bad for
// the fp32 Newton trace kernels (SGPR spills); neutral (within 1 %) on
// the fp64 lean record-all and fused-spot kernels although it holds 32 more SGPRs there and
// the static spill counts drop without it (profiles/r02_ab_prefetch_f64.txt).  Not yet
// measured: the fused spot / OPD Newton kernels and opd_trace_kernel<double,0> (223 lane
// operations of 1072 vector instructions) -- OL_PREFETCH_F64 = 0 and
// OL_FUSED_NR_PREFETCH = 0 switch those off (tools/gpu_ab_prefetch.sh).
#ifndef OL_NR_PREFETCH
#define OL_NR_PREFETCH 0  // 1: prefetch the next surface's hot block in the Newton kernels too
#endif
#ifndef OL_PREFETCH_F64
#define OL_PREFETCH_F64 1
#endif
#ifndef OL_FUSED_NR_PREFETCH
#define OL_FUSED_NR_PREFETCH 1
#endif
// How much a kernel re-reads at the point of use instead of holding it in SGPRs from its
// prologue ("fetch level"):
//   0  nothing: argument block and the current surface's table rows by value (round-2 form);
//   1  the ARGUMENT BLOCK (plane pointers, strides, counts, generator constants) is read from
//      the kernarg segment by the phase that needs it (kernargs(), arg_view<true>);
//   2  ... and the TABLE ROWS too: every phase of the surface body re-reads the few fields it
//      uses (SurfFetched), nothing of a surface is live across the Newton loop.
// Level 2 has 0-6 static SGPR spills where level 0 has 40-320 in the Newton / fp64 kernels,
// but every re-read is a scalar-load round trip the by-value form does not make, and that
// costs where the kernel was not short of SGPRs to begin with.  Interleaved A/Bs against the
// round-2 library on one MI355X, 1e7 rays (profiles/r03_ab_fetch_levels.txt; ratio of kernel
// times, < 1 = faster than round 2):
//   fp32 Newton kernels   level 0  1.00   level 1  1.00-1.26   level 2  1.02-1.18
//                         (C5 fp32 record-all: 0.322 / 0.404 / 0.380 ms -- 38 spills cost
//                          less than the extra round trips; fp32 has 8 waves to hide neither)
//   fp64 Zernike kernels  level 1  0.80-0.94   level 2  0.77-0.94   (fused OPD 0.637 -> 0.488,
//                         fused spot 0.449 -> 0.358, record-all 0.530 -> 0.497 ms)
//   fp64 even-asphere     level 1  0.90-0.98   level 2  0.96-1.00
//   fp64 conic fused      level 0  0.99-1.00   level 1  0.99   level 2  1.02-1.06
//   fp32 conic fused spot level 0  1.00        level 1  0.97   level 2  1.07
// Hence: the level per kernel class below.  Overridable for A/B builds
// (tools/build_variants.py).
#ifndef OL_FETCH_NR_F32           // polarised fp32 Newton trace kernels, fused fp32 Newton
#define OL_FETCH_NR_F32 0         // kernels (spot)
#endif
#ifndef OL_FETCH_NR_F32_PLAIN     // unpolarised fp32 Newton trace kernels: level 1 is as fast
#define OL_FETCH_NR_F32_PLAIN 1   // as level 0 (1.00) and spills 0-1 scalars instead of 23
#endif
#ifndef OL_FETCH_NR_F64           // Zernike-family and generic Newton kernels, fp64
#define OL_FETCH_NR_F64 2
#endif
#ifndef OL_FETCH_NR_F64_ASPHERE   // even-asphere family, fp64
#define OL_FETCH_NR_F64_ASPHERE 1
#endif
#ifndef OL_FETCH_LEAN_F64   // fused spot / OPD kernels without a Newton surface, fp64
#define OL_FETCH_LEAN_F64 1
#endif
#ifndef OL_FETCH_LEAN_F32   // fused spot kernels without a Newton surface, fp32
#define OL_FETCH_LEAN_F32 1
#endif

#include "surface_math.h"

namespace ol {

// fetch level of a kernel instance (see the OL_FETCH_* knobs above); `fused` = the spot / OPD
// kernels (the plain trace kernel without a Newton surface is always level 0: its hot block
// is one prefetched s_load_dwordx16 and it spills nothing)
template <typename T, int NR, bool FUSED, int POLK = 0>
constexpr int fetch_level() {
  if (OL_TABLE_IN_LDS) return 0;
  if (NR != 0) {
    if (sizeof(T) == 4) return (!FUSED && POLK == 0) ? OL_FETCH_NR_F32_PLAIN : OL_FETCH_NR_F32;
    return NR == kNrEvenAsphere ? OL_FETCH_NR_F64_ASPHERE : OL_FETCH_NR_F64;
  }
  if (!FUSED) return 0;
  return sizeof(T) == 4 ? OL_FETCH_LEAN_F32 : OL_FETCH_LEAN_F64;
}

// The lean fp32 configuration -- four rays per lane, conic-only range, no
// polarisation -- runs on packed pairs: two f32x2 rays instead of four scalar ones.
// Everything else keeps V = T.
template <typename T, int RPT, int POLK, int NR>
struct LanePack {
#ifndef OL_PACKED_F32
#define OL_PACKED_F32 1
#endif
#ifndef OL_POLZ_PACKED
#define OL_POLZ_PACKED 1  // 0: the polarised Zernike pair on two SCALAR rays per lane (A/B knob)
#endif
  static constexpr bool packed =
      sizeof(T) == 4 && ((OL_PACKED_F32 && (RPT == 4 || RPT == 2) && POLK == 0 && NR == 0) ||
                         (OL_POLZ_PACKED && RPT == 2 && POLK == 1 && NR == kNrZernike));
  using V = typename std::conditional<packed, f32x2, T>::type;
  static constexpr int NV = packed ? RPT / 2 : RPT;
  // the PRT matrices of a polarised launch: one per lane ELEMENT (Prt<V>: a pair of matrices in
  // the packed form); element e of ray k's matrix
  using PT = typename PrtLane<V, POLK>::type;
  using PrtArr = Prt<PT, POLK>[POLK ? NV : 1];
  static __device__ __forceinline__ T pget(const PrtArr& P, int k, int e) {
    if constexpr (packed && POLK != 0) return P[k / 2].m[e][k % 2]; else return P[k].m[e];
  }
  static __device__ __forceinline__ void pset(PrtArr& P, int k, int e, T x) {
    if constexpr (packed && POLK != 0) P[k / 2].m[e][k % 2] = x; else P[k].m[e] = x;
  }
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
// Round 6, three experiments on the one-ray-per-lane Newton / polarised kernels, judged in ENGINE
// CYCLES per launch (GRBM_GUI_ACTIVE; profiles/r06_cycles.txt -- their times are at the mercy of
// the clock the part's power management grants, profiles/r06_clock_transient_f64.txt):
//   OL_RECORD_DIRECT      record rows written without a copy of the ray (three additions with a
//                         wave-uniform offset, the other five planes out of the ray's own
//                         registers): -35 vector instructions per ray and +3 % (C5 fp32), +5 %
//                         (C4 fp32), +13 % (C5 fp64) CYCLES.  OFF.  The count is not the cost.
//   OL_PRT_SCALAR_BASE    PRT plane bases pinned in scalar registers: -25 instructions, cycles
//                         within +-1 %.  OFF (every pin is a scheduling barrier).
//   OL_RECORD_ARGS_FRESH  record block address / stride re-read from the kernarg segment per row
//                         instead of held (and SGPR-spilled) across the surface body: 10 fewer
//                         spills, cycles within +-1 % (C4 -1.2 %).  ON.
#ifndef OL_RECORD_DIRECT
#define OL_RECORD_DIRECT 0
#endif
#ifndef OL_RECORD_ARGS_FRESH
#define OL_RECORD_ARGS_FRESH 1
#endif
#ifndef OL_PRT_SCALAR_BASE
#define OL_PRT_SCALAR_BASE 0
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
    spot_accumulate<T>(s, rmax, x, y, i, cx, cy);  // epilogue_device.h
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

// trace_kernel's arguments are four table pointers (32 bytes) and then the TraceArgs block
// (8-byte aligned): its offset in the kernarg segment.
constexpr int kTraceArgsKernargOffset = 32;

// A kernel argument block read from the kernarg segment at the point of use (scalar loads
// from the constant address space) instead of being kept live from the prologue.  The
// empty asm hides the pointer's provenance, so the loads cannot be hoisted back.
template <typename A>
__device__ __forceinline__ cptr<A> kernarg_again(int offset) {
#if defined(__HIP_DEVICE_COMPILE__)
  using CP = const __attribute__((address_space(4))) char*;
  CP p = (CP)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return reinterpret_cast<cptr<A>>(p + offset);
#else
  return nullptr;  // (host pass of the single-source compile: never executed)
#endif
}

// The kernarg segment of the three trace kernels as a struct: four table pointers, then the
// argument block (8-byte aligned, so at offset 32).  `kernargs<T, A>()` is a fresh view of it
// (kernarg_again): a kernel that is short of SGPRs reads a pointer or a stride from it where
// it needs one instead of carrying it from the prologue.
template <typename T, typename A>
struct KernArgs {
  const DevSurfHot<T>* surf;
  const DevSurfCold<T>* cold;
  const DevOptics<T>* optics;
  const T* coeffs;
  A a;
};
template <typename T, typename A>
__device__ __forceinline__ cptr<KernArgs<T, A>> kernargs() {
  using KA = KernArgs<T, A>;
  static_assert(__builtin_offsetof(KA, a) == kTraceArgsKernargOffset, "kernarg layout");
  return kernarg_again<KernArgs<T, A>>(0);
}
// The argument block as a kernel phase reads it: FETCH = a fresh view of the kernarg segment
// (constant address space; fields are scalar-loaded where they are used and die with the
// phase), otherwise the by-value parameter (loaded once in the prologue).
template <bool FETCH, typename T, typename A>
__device__ __forceinline__ auto arg_view(const A& a) {
  if constexpr (FETCH) return &kernargs<T, A>()->a;
  else return &a;
}
template <typename C>
__device__ __forceinline__ C consts_of(const C* p) { return *p; }
#if defined(__HIP_DEVICE_COMPILE__)
template <typename C>
__device__ __forceinline__ C consts_of(cptr<C> p) { return load_consts(p); }
#endif
// the rows of surface `s` for wavelength slot (n_wl, wl), addressed from the kernarg segment
template <typename T, typename A>
__device__ __forceinline__ SurfFetched<T> fetched_surface(int s) {
  const auto ka = kernargs<T, A>();
  return SurfFetched<T>{as_const(ka->surf) + s, as_const(ka->cold) + s,
                        as_const(ka->optics) + (s * ka->a.n_wl + ka->a.wl)};
}
// ... for an explicit optics table and wavelength slot (the batched spot kernel: per cell)
template <typename T, typename A>
__device__ __forceinline__ SurfFetched<T> fetched_surface_of(int s, const DevOptics<T>* optics,
                                                             int n_wl, int wl) {
  const auto ka = kernargs<T, A>();
  return SurfFetched<T>{as_const(ka->surf) + s, as_const(ka->cold) + s,
                        as_const(optics) + (s * n_wl + wl)};
}
#ifndef OL_POLNR_WAVES_F64
#define OL_POLNR_WAVES_F64 0
#endif
// occupancy request for the lean (conic-only, unpolarised) fp64 kernels: the generating
// record-all form allocates 82 VGPRs = 5 waves, two registers over the 6-wave budget
#ifndef OL_LEAN_F64_WAVES
#define OL_LEAN_F64_WAVES 0
#endif
// (The generic-family kernel WITH the generator prologue does not fit 7 waves without two
// dwords of scratch: it keeps the allocator's own choice.)
// the polarised Zernike fp32 PAIR (OL_POLZ_PAIR): 108 VGPRs = 4 waves left to itself
#ifndef OL_POLZ_PAIR_WAVES
#define OL_POLZ_PAIR_WAVES 0
#endif
template <typename T, int RPT, int POLK, int NR, bool GEN = false>  // GEN: any generating form
struct WavesPerEu {
  static constexpr int value =
      (OL_POLZ_PAIR_WAVES > 0 && sizeof(T) == 4 && RPT == 2 && POLK == 1 && NR == kNrZernike)
          ? OL_POLZ_PAIR_WAVES
      : (OL_POLNR_WAVES > 0 && sizeof(T) == 4 && RPT == 1 && POLK == 1 && NR != 0 &&
       NR != kNrReference && !(GEN && NR == 1))
          ? OL_POLNR_WAVES
          : ((OL_POLNR_WAVES_F64 > 0 && sizeof(T) == 8 && RPT == 1 && POLK == 1 && NR != 0)
                 ? OL_POLNR_WAVES_F64
                 : ((OL_LEAN_F64_WAVES > 0 && sizeof(T) == 8 && RPT == 1 && POLK == 0 && NR == 0)
                        ? OL_LEAN_F64_WAVES
                        : 1));
};

// GEN != 0: the rays are GENERATED in the prologue from normalised pupil planes and a
// launch-uniform field (raygen_device.h, the arithmetic of ol_generate_rays) instead of being
// read from eight planes -- ol_trace_generate, the record-all path of Optic.trace() in one
// launch: no separate generator launch, and the object row is written once instead of
// written by one kernel and read back by the next.  GEN is a bit set (trace_launch.h):
// kGenUniform alone is that form; | kGenFieldPlanes reads per-ray field planes hx, hy (and
// vx, vy when the caller has them) -- trace_generic(Hx[], Hy[], Px[], Py[]), the expanded
// fields x pupil of a multi-field trace (real_ray_tracer.py:88-98, 120-154); | kGenApod
// takes the initial intensity from the pupil apodization (ray_generator.py:81-85).  Template
// parameters because the per-ray fp64 tangent and the exp / cos / pow of the apodization
// switch would otherwise set the register budget of the plain form (the same split as
// spot_trace_kernel's FIELDP / APOD).
// SPOT with GEN: the masked image-plane moments as an epilogue of the GENERATING launch --
// the per-step form of the sharded trace (distributed.py): one launch per step at N > 1.
// EPI (GEN == kGenUniform && POLK != 0 only): PolarizedRays.update_intensity as an epilogue
// of the launch -- an instantiation of its own, so that launches without it keep their
// register budget
// LDS slots of the update_intensity epilogue (EPI): launch direction and `_i0` per lane, from
// the generating prologue to the epilogue.  One object for both ends (a function-local
// __shared__ array is a static of THIS function), allocated only in kernels that call it.
template <typename T, int RPT = 1>
__device__ __forceinline__ T (*epi_launch_slots())[kTraceBlock * RPT] {
  __shared__ T slots[4][kTraceBlock * RPT];
  return slots;
}

template <typename T, int RPT, bool RECORD, int POLK, int NR, bool SPOT, int GEN = 0,
          bool EPI = false>
__global__ __launch_bounds__(kTraceBlock)
__attribute__((amdgpu_waves_per_eu(WavesPerEu<T, RPT, POLK, NR, GEN != 0 || EPI>::value))) void trace_kernel(
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
  typename LP::PrtArr P;  // (one matrix per ray; per PAIR of rays in the packed polarised form)
  uint32_t status = 0;
  if constexpr (GEN != 0) {
    // one ray per lane -- or, for the lean fp32 launch-uniform form, the packed PAIR of the
    // record-mode kernel (two rays generated one after the other, then traced as one f32x2)
    static_assert(RPT == 1 || (RPT == 2 && GEN == kGenUniform && sizeof(T) == 4 &&
                               ((POLK == 0 && NR == 0) || (POLK == 1 && NR == kNrZernike))),
                  "the generating prologue: one ray per lane, the lean fp32 pair, or the "
                  "polarised Zernike fp32 pair (OL_POLZ_PAIR)");
    // POLARISED generating launches (round 5): ONE form, GEN == kGenUniform, that takes per-ray
    // field planes and an apodized pupil at RUN time (launch-uniform branches) -- the
    // prologue is nowhere near the register peak of a kernel that carries a PRT matrix through
    // the surface loop, so the split that protects the lean kernels' budget buys nothing here
    // and would double the polarised instantiations.
    static_assert(GEN == kGenUniform || POLK == 0,
                  "polarised launches: the run-time general form (GEN == kGenUniform)");
    constexpr bool kGeneral = POLK != 0;
    const auto A0 = arg_view<(fetch_level<T, NR, false, POLK>() >= 1), T>(a);
    const auto& in_ = A0->in;
    // (same order of operations as raygen_kernel, aux_kernels.hip: the two-launch path and
    // this prologue produce the same bits)
    T tx = in_.tx0, ty = in_.ty0, vx = in_.vx0, vy = in_.vy0, o[6];
    const RaygenConsts<T> c = consts_of(&A0->rgc);
    if constexpr ((GEN & kGenFieldPlanes) != 0 || kGeneral) {
      if (!kGeneral || in_.hx != nullptr) {  // (kGeneral: launch-uniform)
        const T hx = base.at(in_.hx)[0], hy = base.at(in_.hy)[0];
        if (in_.vx != nullptr) {  // launch-uniform
          vx = base.at(in_.vx)[0];
          vy = base.at(in_.vy)[0];
        }
        if ((in_.flags & kRaygenCheckField) && (outside_unit(hx) || outside_unit(hy)))
          status |= kStatusFieldRange;
        raygen_field<T>(c, hx, hy, tx, ty);
      }
    }
    // (two rays per lane: each ray its own field point / vignetting factors)
    T txs[RPT], tys[RPT], vxs[RPT], vys[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      txs[k] = tx; tys[k] = ty; vxs[k] = vx; vys[k] = vy;
    }
    if constexpr (RPT > 1 && kGeneral) {
      if (in_.hx != nullptr) {  // launch-uniform
#pragma unroll
        for (int k = 1; k < RPT; ++k) {
          const int j = k < cnt ? k : 0;  // (a padding ray of the ragged tail repeats ray 0)
          const T hx = base.at(in_.hx)[j], hy = base.at(in_.hy)[j];
          vxs[k] = in_.vx0;
          vys[k] = in_.vy0;
          if (in_.vx != nullptr) {
            vxs[k] = base.at(in_.vx)[j];
            vys[k] = base.at(in_.vy)[j];
          }
          if (k < cnt && (in_.flags & kRaygenCheckField) && (outside_unit(hx) || outside_unit(hy)))
            status |= kStatusFieldRange;
          txs[k] = in_.tx0;
          tys[k] = in_.ty0;
          raygen_field<T>(c, hx, hy, txs[k], tys[k]);
        }
      }
    }
    T pxs[RPT], pys[RPT];
    if constexpr (RPT == 1) {
      pxs[0] = base.at(in_.px)[0];
      pys[0] = base.at(in_.py)[0];
    } else if (cnt == RPT) {
      using PV = typename VecOf<T, RPT>::type;
      const PV vx_ = *reinterpret_cast<const PV*>(base.at(in_.px));
      const PV vy_ = *reinterpret_cast<const PV*>(base.at(in_.py));
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        pxs[k] = vec_get<T, RPT>(vx_, k);
        pys[k] = vec_get<T, RPT>(vy_, k);
      }
    } else {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        pxs[k] = k < cnt ? base.at(in_.px)[k] : T(0);
        pys[k] = k < cnt ? base.at(in_.py)[k] : T(0);
      }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      T px = pxs[k], py = pys[k];
      uint32_t st_k = 0;
      raygen_pupil<T>(in_.flags, vxs[k], vys[k], px, py, st_k);
      if (k < cnt) status |= st_k;  // (a padding ray of the ragged tail reports nothing)
      raygen_one<T>(c, txs[k], tys[k], px, py, vxs[k], vys[k], o);
      Ray<T> q;
      q.x = o[0]; q.y = o[1]; q.z = o[2];
      q.L = o[3]; q.M = o[4]; q.N = o[5];
      // (raygen_apodize returns 1 for apod_kind == 0: a launch-uniform branch)
      if constexpr ((GEN & kGenApod) != 0 || kGeneral) q.i = raygen_apodize<T>(c, px, py);
      else q.i = T(1);
      q.opd = T(0);
      LP::put(r, k, q);
      if constexpr (EPI) {
        // what the update_intensity epilogue needs of the LAUNCH state -- the direction and
        // `_i0` (polarized_rays.py:51-54) -- waits in LDS, one slot per lane (written and read
        // by the same lane: no barrier).  Until round 6 the epilogue generated the ray a second
        // time (32 vector instructions per ray in fp32 and the pupil / field planes read
        // again, in a kernel bound by vector issue; profiles/r06_phase_costs_before.txt);
        // LDS traffic goes down its own pipe and 4 KB of the CU's 160 KB were idle anyway.
        T (*epi_launch)[kTraceBlock * RPT] = epi_launch_slots<T, RPT>();
        epi_launch[0][threadIdx.x * RPT + k] = q.L;
        epi_launch[1][threadIdx.x * RPT + k] = q.M;
        epi_launch[2][threadIdx.x * RPT + k] = q.N;
        epi_launch[3][threadIdx.x * RPT + k] = q.i;
      }
    }
    if constexpr (POLK != 0) {
#pragma unroll
      for (int k = 0; k < RPT; ++k)
#pragma unroll
        for (int e = 0; e < NPRT; ++e) LP::pset(P, k, e, (e == 0 || e == 4 || e == 8) ? T(1) : T(0));
    }
  } else
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
          LP::pset(P, k, e, ident ? ((e == 0 || e == 4 || e == 8) ? T(1) : T(0)) : pin[e][k]);
      }
    }
  }

  bool is_global = true;  // frame of the state held in r[]
  bool prt_fresh = POLK != 0 && (GEN != 0 || (a.flags & kTracePrtIdentity) != 0);
  // Newton-Raphson ranges: table rows re-read phase by phase (SurfFetched), kernel arguments
  // re-read from the kernarg segment where they are used -- these kernels are short of SGPRs
  constexpr bool kFetch = fetch_level<T, NR, false, POLK>() >= 2;      // table rows
  constexpr bool kFetchArgs = fetch_level<T, NR, false, POLK>() >= 1;  // argument block
  const cptr<T> coeffs_c = as_const(coeff_tab);
  // one reciprocal for both quotients of the conic intersection (curved_distance<V, SHARE>):
  // only where vector issue binds AND a lane carries more than one ray -- record-last traces
  // on vectors of rays.  Record-all kernels are bound by their stores (-2.5 ... -3.9 % with it);
  // with one ray per lane the longer dependent chain costs more than the 8 instructions save
  // (fused OPD kernel: -7 ... -9 %, profiles/r05_ab_arith.txt).
  constexpr bool kShareRcp = !RECORD && RPT > 1;
  DevSurf<T> last_traced;
  last_traced.cold = as_const(cold_tab);
  int last_idx = a.first;  // kFetch: the surface whose frame r[] is held in
  auto to_global_now = [&](Ray<V> (&gv)[NV]) {
    if constexpr (kFetch) {
      const DevSurf<T> lt = fetched_surface<T, TraceArgs<T>>(last_idx).surf();
#pragma unroll
      for (int j = 0; j < NV; ++j) gv[j] = is_global ? r[j] : to_global<V>(lt, r[j]);
    } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) gv[j] = is_global ? r[j] : to_global<V>(last_traced, r[j]);
    }
  };
  // hot block of the current surface by value (one s_load_dwordx16 for fp32); the
  // next surface's block is requested before the current one is worked on.
#if OL_TABLE_IN_LDS
  DevSurfHot<T> cur = lds_hot[0];
#else
  // (the Newton kernels are short of SGPRs, not of latency hiding: no prefetch there --
  // 16 fewer live scalars across the whole surface body)
  constexpr bool kPrefetch = (OL_NR_PREFETCH || NR == 0) && (OL_PREFETCH_F64 || sizeof(T) == 4);
  DevSurfHot<T> cur;
  if constexpr (!kFetch) cur = surf_tab[a.first];
#endif
  const int first = a.first, last = a.last;
  const int rec_from = a.record_from > first ? a.record_from : first;
  for (int s = first; s <= last; ++s) {
    // OL_TRACE_NONUNIT_K: polarised launches over the CALLER's rays only -- the generating
    // prologue normalises its directions, so those kernels do not carry the branch at all
    uint32_t polf = 0;
    if constexpr (POLK != 0 && GEN == 0) {
      uint32_t fl;
      if constexpr (kFetchArgs) fl = kernargs<T, TraceArgs<T>>()->a.flags;
      else fl = a.flags;
      polf = (fl & kTraceNonUnitK) ? kPolNonUnitK : 0u;
    }
    if constexpr (kFetch) {
      const SurfFetched<T> h = fetched_surface<T, TraceArgs<T>>(s);
      if (refresh(h.hot)->interaction != kRecordOnly) {
        if constexpr (NR == kNrReference) {
          const auto ka = kernargs<T, TraceArgs<T>>();
          const NrRefCtl ctl{ka->a.nr_iters, s, ka->a.n_surf, ka->a.nr_count_at};
          surface_step<V, NV, POLK, NR, kShareRcp>(h, refresh(coeffs_c), is_global, r, P, status,
                                                 prt_fresh, &ctl, polf);
        } else {
          surface_step<V, NV, POLK, NR, kShareRcp>(h, refresh(coeffs_c), is_global, r, P, status,
                                                 prt_fresh, nullptr, polf);
        }
        is_global = false;
        last_idx = s;
      }
    } else {
      DevSurf<T> S;
#if OL_TABLE_IN_LDS
      static_cast<DevSurfHot<T>&>(S) = cur;
      if (s < a.last) cur = lds_hot[s + 1 - a.first];
#else
      if constexpr (kPrefetch) {
        static_cast<DevSurfHot<T>&>(S) = cur;
        if (s < a.last) cur = surf_tab[s + 1];
      } else {
        static_cast<DevSurfHot<T>&>(S) = surf_tab[s];
      }
#endif
      S.cold = as_const(cold_tab) + s;
      if (S.interaction != kRecordOnly) {
#if OL_TABLE_IN_LDS
        const DevOptics<T> O = lds_opt[s - a.first];
#else
        const DevOptics<T> O = optics_tab[s * a.n_wl + a.wl];
#endif
        if constexpr (NR == kNrReference) {
          const NrRefCtl ctl{a.nr_iters, s, a.n_surf, a.nr_count_at};
          surface_step<V, NV, POLK, NR, kShareRcp>(S, O, coeffs_c, is_global, r, P, status,
                                                 prt_fresh, &ctl, polf);
        } else {
          surface_step<V, NV, POLK, NR, kShareRcp>(S, O, coeffs_c, is_global, r, P, status,
                                                 prt_fresh, nullptr, polf);
        }
        is_global = false;
        last_traced = S;
      }
    }
    if (RECORD && s >= rec_from) {
      T* record;
      int64_t stride;
      uint32_t flags;
      // (OL_RECORD_ARGS_FRESH: also in the polarised / Newton kernels that otherwise keep
      // their arguments by value -- held across the surface body, the block's address and
      // stride were SGPR-spilled and came back through 6-8 v_readlane per recorded row)
      if constexpr (kFetchArgs || (OL_RECORD_ARGS_FRESH && (NR != 0 || POLK != 0) &&
                                   (RPT == 1 || GEN != 0))) {
        const auto ka = kernargs<T, TraceArgs<T>>();
        record = ka->a.record;
        stride = ka->a.record_stride;
        flags = ka->a.flags;
      } else {
        record = a.record;
        stride = a.record_stride;
        flags = a.flags;
      }
      T* row = record + (int64_t)(s - rec_from) * 8 * stride;
      if (s == first && (flags & kTraceRow0IsInput)) {
        // the caller generated the rays straight into row 0 of the record block
        // (the object surface only records its input): nothing to write
      } else if constexpr (OL_RECORD_DIRECT && RPT == 1 && NV == 1 && (NR != 0 || POLK != 0)) {
        // Round 6, the one-ray-per-lane Newton / polarised kernels (bound by vector issue): the
        // row without a copy of the ray.  The state is global (before the first traced surface)
        // or in the last traced surface's frame; unless that frame is rotated, global = local +
        // origin -- three additions with a WAVE-UNIFORM offset that is -0 for a global state
        // (x + -0 = x bit for bit, signed zeros and NaN included), and the other five planes go
        // out of the ray's own registers.  Before: the selected frame's copy of all eight
        // values was made first, ~10 moves per row on top of the 8 stores
        // (profiles/r06_c5_valu.txt: 201 vector instructions per ray with NO surface traced).
        bool rotated = false;
        T ox = T(-0.0), oy = T(-0.0), oz = T(-0.0);
        DevSurf<T> lt = last_traced;
        if (!is_global) {
          if constexpr (kFetch) lt = fetched_surface<T, TraceArgs<T>>(last_idx).surf();
          rotated = (lt.flags & kSurfRotated) != 0;
          if (!rotated) {
            ox = lt.origin[0];
            oy = lt.origin[1];
            oz = lt.origin[2];
          }
        }
        if (rotated) {
          // (the marker keeps this rare tail from being MERGED with the one below: merged, the
          // common path pays the copies into the registers the two tails would share)
          Ray<T> g[1] = {to_global<T>(lt, r[0])};
          asm volatile("s_nop 0 ; record row of a rotated frame");
          store_rays<T, 1>(row, stride, base, cnt, g);
          asm volatile("s_nop 0");
        } else {
          const T gx[1] = {r[0].x + ox}, gy[1] = {r[0].y + oy}, gz[1] = {r[0].z + oz};
          const T dL[1] = {r[0].L}, dM[1] = {r[0].M}, dN[1] = {r[0].N}, di[1] = {r[0].i},
                  dp[1] = {r[0].opd};
          store_plane<T, 1>(row, base, cnt, gx);
          store_plane<T, 1>(row + stride, base, cnt, gy);
          store_plane<T, 1>(row + 2 * stride, base, cnt, gz);
          store_plane<T, 1>(row + 3 * stride, base, cnt, dL);
          store_plane<T, 1>(row + 4 * stride, base, cnt, dM);
          store_plane<T, 1>(row + 5 * stride, base, cnt, dN);
          store_plane<T, 1>(row + 6 * stride, base, cnt, di);
          store_plane<T, 1>(row + 7 * stride, base, cnt, dp);
        }
      } else {
        Ray<V> gv[NV];
        to_global_now(gv);
        Ray<T> g[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) g[k] = LP::ray(gv, k);
        store_rays<T, RPT>(row, stride, base, cnt, g);
      }
    }
  }

  // Round 5: total internal reflection at the LAST traced surface leaves a ray with a position
  // and no direction.  The reference's trace then ends with `x += t L` (t = the last
  // thickness, 0 in every sample), which makes that position NaN too; the drop-in applies that
  // to the rays it RETURNS -- the recorded row keeps the position -- and needs to know when:
  // an informational status bit instead of three extra elementwise passes per call.
#ifndef OL_NAN_DIRECTION_BIT
#define OL_NAN_DIRECTION_BIT 1  // (0: A/B only -- the drop-in then misses such rays)
#endif
#if OL_NAN_DIRECTION_BIT
#pragma unroll
  for (int k = 0; k < RPT; ++k) {
    const Ray<T> q = LP::ray(r, k);
    if (k < cnt && q.L != q.L && q.x == q.x) status |= kStatusNanDirection;
  }
#endif

  if constexpr (SPOT) {
    // epilogue: masked moments of the final (global) state about (cx, cy),
    // into slot (workgroup % slots) of the caller's [slots][8] buffer -- the slots keep
    // the ~4e4 workgroups of a 1e7-ray launch off one address (7 x 39 k same-address
    // atomics were measured at 0.95 ms); the consumer adds the slots up
    Ray<V> gv[NV];
    to_global_now(gv);
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
  // What the epilogue needs of the argument block -- the eight plane pointers, the PRT
  // pointer, n, the status pointer -- is read AGAIN from the kernarg segment here, through
  // a pointer the compiler cannot see through.  Held live from the prologue they were
  // SGPR-spilled into VGPR lanes and reloaded (v_writelane / v_readlane are VECTOR
  // instructions): ~45 per ray in the Newton kernels, which are short of SGPRs.
  struct {
    T* rays[8];
    T* prt;
    uint32_t* status;
    int64_t n;
    uint32_t flags;
  } late;
  if constexpr (NR != 0) {
    const auto* q = kernarg_again<TraceArgs<T>>(kTraceArgsKernargOffset);
#pragma unroll
    for (int f = 0; f < 8; ++f) late.rays[f] = q->rays[f];
    late.prt = q->prt;
    late.status = q->status;
    late.n = q->n;
    late.flags = q->flags;
  } else {
#pragma unroll
    for (int f = 0; f < 8; ++f) late.rays[f] = a.rays[f];
    late.prt = a.prt;
    late.status = a.status;
    late.n = a.n;
    late.flags = a.flags;
  }
  if ((late.flags & kTraceWriteRays) && (!SPOT || live)) {
    Ray<V> gv[NV];
    to_global_now(gv);
    Ray<T> g[RPT];
#pragma unroll
    for (int k = 0; k < RPT; ++k) g[k] = LP::ray(gv, k);
    T tmp[RPT];
#define OL_WB_FIELD(idx, fld)                                        \
  _Pragma("unroll") for (int k = 0; k < RPT; ++k) tmp[k] = g[k].fld; \
  store_plane<T, RPT>(late.rays[idx], base, cnt, tmp);
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
      for (int k = 0; k < RPT; ++k) tmp[k] = LP::pget(P, k, e);
      // (the plane's base as a SCALAR pointer, said outright: left to itself the compiler
      // folded e * n into the per-lane address -- a 64-bit multiply-add in vector registers
      // for every one of the 9 / 18 stores, ~3.5 vector instructions each)
      T* plane = late.prt + (int64_t)e * late.n;
#if OL_PRT_SCALAR_BASE
      plane = refresh(plane);
#endif
      store_plane<T, RPT>(plane, base, cnt, tmp);
    }
  }
  if constexpr (EPI) {
    static_assert(GEN == kGenUniform && POLK != 0,
                  "the update_intensity epilogue: generating polarised launches");
    // PolarizedRays.update_intensity (rays/polarized_rays.py:68-133) of the traced bundle, from
    // the matrix still in registers (ol_trace_extras.updated_intensity, ABI 7) -- instead of a
    // second launch that reads nine PRT planes, three direction planes and the intensity
    // plane back.  The launch direction and the initial intensity `_i0` -- 1, or the
    // apodization of the pupil point (polarized_rays.py:51, ray_generator.py:81-85) -- come
    // back from the LDS slots the prologue left them in (rounds 3-5 generated the ray again
    // rather than keep four registers live through the surface loop).
    const auto ka = kernargs<T, TraceArgs<T>>();
    T* upd = ka->a.i_updated;
    {
      PolFields<T> f;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        f.ar[k] = ka->a.pf.ar[k]; f.ai[k] = ka->a.pf.ai[k];
        f.br[k] = ka->a.pf.br[k]; f.bi[k] = ka->a.pf.bi[k];
      }
      f.nf = ka->a.pf.nf;
      T (*epi_launch)[kTraceBlock * RPT] = epi_launch_slots<T, RPT>();  // (the prologue's slots)
      T upd_k[RPT];
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        T Pm[9], Qm[9];
#pragma unroll
        for (int e = 0; e < 9; ++e) {
          Pm[e] = LP::pget(P, k, e);
          Qm[e] = POLK == 2 ? LP::pget(P, k, POLK == 2 ? 9 + e : e) : T(0);
        }
        const int slot = threadIdx.x * RPT + k;
        const T L0 = epi_launch[0][slot], M0 = epi_launch[1][slot], N0 = epi_launch[2][slot],
                i0 = epi_launch[3][slot];
        uint32_t st_k = 0;
        upd_k[k] = pol_intensity_one<T, POLK == 2>(f, L0, M0, N0, Pm, Qm, i0, st_k);
        if (k < cnt) status |= st_k;
      }
      if constexpr (RPT == 1) base.at(upd)[0] = upd_k[0];
      else store_plane<T, RPT>(upd, base, cnt, upd_k);
    }
  }
  if (status && late.status) atomicOr(late.status, status);
}

// --------------------------------------------------------------------------
// host-side launcher
// --------------------------------------------------------------------------
// Dynamic LDS nobody touches, asked for so that at most `cap` workgroups are resident on a CU
// (160 KB per CU): the record-all kernels of conic-only unpolarised ranges ("lean": ~76 vector
// instructions per ray and surface against 8 stores) write FASTER with fewer stores in flight
// (tools/microbench/pace_probe.hip; the real kernels: profiles/r05_ab_wgcap.txt):
//   fp64  three workgroups instead of the four its registers allow: -3 ... -4 % in the first
//         launches after an idle part, -1.6 ... -3 % sustained, placed block or not
//   fp32  two instead of eight: -2 ... -3.6 % (-7.5 % at 1e6 rays) into a block that is NOT in
//         a placed window (OL_TRACE_FEW_WAVES: the caller knows); in a placed window -3.8 %
//         sustained (7.37 TB/s) but +11 ... +13 % in the first 25 launches after an idle part
//         (throttled clocks want the latency hiding): not there by default.  The form that
//         READS its rays from eight planes (ol_trace) wants three (-1.3 %; two: +5 % in the
//         window); its fp64 sibling gains 7 % from three, placed or not
// Newton / polarised kernels lose 5 ... 65 % under any cap and are never capped.
// OL_TUNE_RECORD_WG_CAP: 0 = this policy, 1 = never, 2 ... 8 = every record launch (A/B).
// cap >= 2 keeps the request under the 64 KB a launch may ask for without a function attribute.
template <typename T>
static unsigned record_lds(const TraceArgs<T>& a, bool lean, bool generating) {
  if (a.record == nullptr) return 0;
  int cap = tuning().record_wg_cap;
  if (cap == 0)
    cap = !lean ? 1 : sizeof(T) == 8 ? 3 : !(a.flags & kTraceFewWaves) ? 1 : generating ? 2 : 3;
  if (cap < 2) return 0;
  // (63 KB at most: with the static LDS some variants carry -- the spot epilogue's partial sums,
  // the update_intensity slots -- the request stays under the 64 KB a launch may ask for without
  // a function attribute, and two blocks of 63 KB still keep a third off the CU)
  const unsigned want = 160u * 1024u / (unsigned)cap - 1024u;
  return want > 63u * 1024u ? 63u * 1024u : want;
}

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
  hipLaunchKernelGGL((trace_kernel<T, RPT, R, P, NR, S>), grid, block,                        \
                     record_lds(a, NR == 0 && P == 0, false), stream, a.surf, a.cold,         \
                     a.optics, a.coeffs, a)
#define OL_LAUNCH(R, P) OL_LAUNCH_S(R, P, false)
  if (a.spot != nullptr) {
    // the spot epilogue exists for unpolarised traces (the polarised intensity needs
    // the update_intensity epilogue first); not on reference-Newton ranges (capi.hip refuses)
    if (polk != 0 || NR == kNrReference) return hipErrorInvalidValue;
    if constexpr (NR != kNrReference) {
      if (rec) OL_LAUNCH_S(true, 0, true); else OL_LAUNCH_S(false, 0, true);
    }
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
  } else {
    // single-family instantiations exist for one ray per lane, without the spot epilogue
    if (a.spot == nullptr) {
      if (nr == kNrZernike) return launch_nr<T, 1, kNrZernike>(a, stream);
      if (nr == kNrEvenAsphere) return launch_nr<T, 1, kNrEvenAsphere>(a, stream);
    }
    // the reference's batch-global Newton stop rule (opt-in): its own instantiations, so that
    // every other kernel is what it was
    if (nr == kNrReference) return launch_nr<T, 1, kNrReference>(a, stream);
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
    hipLaunchKernelGGL((trace_kernel<T, 2, true, 0, 0, false>), grid, block,
                       record_lds(a, true, false), stream, a.surf, a.cold, a.optics, a.coeffs,
                       a);
  else
    hipLaunchKernelGGL((trace_kernel<T, 2, false, 0, 0, false>), grid, block, 0, stream, a.surf,
                       a.cold, a.optics, a.coeffs, a);
  return hipGetLastError();
}

// ol_trace_generate on the lean fp32 form (conic-only range, unpolarised, launch-uniform field,
// no apodization; with or without the spot epilogue): one packed PAIR of rays per lane, as
// launch_pair.
// 100 vector instructions per pair and surface against 2 x 76: in steady state the two forms
// tie on a placed block (HBM-bound either way); in the first ~25 launches after an idle part
// -- where the power management throttles a vector-ALU-heavy kernel, profiles/
// r04_clock_transient.txt -- the pair form is 5-7 % faster (profiles/r04_pair_window.txt).
#ifndef OL_GEN_PAIR
#define OL_GEN_PAIR 1
#endif
// (OL_POLZ_PAIR: trace_launch.h)
template <typename T>
static hipError_t launch_gen_pair(const TraceArgs<T>& a, hipStream_t stream) {
  const int64_t threads = (a.n + 1) / 2;
  const int64_t blocks = (threads + kTraceBlock - 1) / kTraceBlock;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const dim3 grid((unsigned)blocks), block(kTraceBlock);
  if (a.spot != nullptr)  // the per-step form of the sharded trace: moments as an epilogue
    hipLaunchKernelGGL((trace_kernel<T, 2, true, 0, 0, true, kGenUniform, false>), grid, block,
                       record_lds(a, true, true), stream, a.surf, a.cold, a.optics, a.coeffs, a);
  else
    hipLaunchKernelGGL((trace_kernel<T, 2, true, 0, 0, false, kGenUniform, false>), grid, block,
                       record_lds(a, true, true), stream, a.surf, a.cold, a.optics, a.coeffs, a);
  return hipGetLastError();
}

// ol_trace_generate: one ray per lane, recording
template <typename T, int NR>
static hipError_t launch_gen_nr(const TraceArgs<T>& a, bool pair_ok, hipStream_t stream) {
  if constexpr (sizeof(T) == 4 && NR == 0) {
    const int want = tuning().rays_per_thread;
    if (pair_ok && a.prt == nullptr && a.in.hx == nullptr && a.rgc.apod_kind == 0 &&
        (want == 3 || (want == 0 && OL_GEN_PAIR)))
      return launch_gen_pair<T>(a, stream);
  }
  if constexpr (sizeof(T) == 4 && NR == kNrZernike) {
    // OL_POLZ_PAIR: the polarised Zernike fp32 launch (configuration C5) on TWO rays per lane
    const int want = tuning().rays_per_thread;
    if (pair_ok && a.prt != nullptr && !(a.flags & kTracePrtComplex) && a.spot == nullptr &&
        a.n % 2 == 0 && reinterpret_cast<uintptr_t>(a.prt) % 8 == 0 &&
        reinterpret_cast<uintptr_t>(a.i_updated) % 8 == 0 &&
        (want == 3 || (want == 0 && OL_POLZ_PAIR))) {
      const int64_t threads = (a.n + 1) / 2;
      const int64_t blocks2 = (threads + kTraceBlock - 1) / kTraceBlock;
      if (blocks2 == 0) return hipSuccess;
      if (blocks2 > 0x7fffffffLL) return hipErrorInvalidValue;
      const dim3 grid2((unsigned)blocks2), block2(kTraceBlock);
      if (a.i_updated != nullptr)
        hipLaunchKernelGGL((trace_kernel<T, 2, true, 1, NR, false, kGenUniform, true>), grid2,
                           block2, 0, stream, a.surf, a.cold, a.optics, a.coeffs, a);
      else
        hipLaunchKernelGGL((trace_kernel<T, 2, true, 1, NR, false, kGenUniform, false>), grid2,
                           block2, 0, stream, a.surf, a.cold, a.optics, a.coeffs, a);
      return hipGetLastError();
    }
  }
  const int64_t blocks = (a.n + kTraceBlock - 1) / kTraceBlock;
  if (blocks == 0) return hipSuccess;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  dim3 grid((unsigned)blocks), block(kTraceBlock);
  const int polk = a.prt == nullptr ? 0 : ((a.flags & kTracePrtComplex) ? 2 : 1);
#define OL_LAUNCH_G(P, S, G, E)                                                              \
  hipLaunchKernelGGL((trace_kernel<T, 1, true, P, NR, S, G, E>), grid, block,                 \
                     record_lds(a, NR == 0 && P == 0, G == kGenUniform), stream, a.surf,     \
                     a.cold, a.optics, a.coeffs, a)
  const bool epi = polk != 0 && a.i_updated != nullptr;  // update_intensity epilogue (ABI 7)
  const bool fieldp = a.in.hx != nullptr, apod = a.rgc.apod_kind != 0;
  if (polk != 0) {
    // polarised launches: ONE form that takes field planes / an apodized pupil at run time
    if (a.spot != nullptr) return hipErrorInvalidValue;
    if (polk == 2) { if (epi) OL_LAUNCH_G(2, false, 1, true); else OL_LAUNCH_G(2, false, 1, false); }
    else { if (epi) OL_LAUNCH_G(1, false, 1, true); else OL_LAUNCH_G(1, false, 1, false); }
  } else if (a.spot != nullptr) {
    // spot epilogue of the generating launch (ABI 8): the per-step form of the sharded trace
    // (one field point per step; capi.hip refuses it with field planes / apodization)
    if (fieldp || apod) return hipErrorInvalidValue;
    OL_LAUNCH_G(0, true, 1, false);
  } else {
    if (fieldp && apod) OL_LAUNCH_G(0, false, 7, false);
    else if (fieldp) OL_LAUNCH_G(0, false, 3, false);
    else if (apod) OL_LAUNCH_G(0, false, 5, false);
    else OL_LAUNCH_G(0, false, 1, false);
  }
#undef OL_LAUNCH_G
  return hipGetLastError();
}

// pair_ok: px / py, the record block and the optional final-state planes allow 8-byte lane
// accesses (capi.hip checks addresses and the record stride)
template <typename T>
hipError_t launch_trace_generate(const TraceArgs<T>& a_in, int nr_family, bool pair_ok,
                                 hipStream_t stream) {
  TraceArgs<T> a = a_in;
  // (reference-Newton ranges generate and trace in two launches: the counting launches of
  // ol_newton_count read the generated rays)
  if (nr_family == kNrReference) return hipErrorInvalidValue;
  if (nr_family == kNrNone) return launch_gen_nr<T, 0>(a, pair_ok, stream);
  if (nr_family == kNrZernike) return launch_gen_nr<T, kNrZernike>(a, pair_ok, stream);
  if (nr_family == kNrEvenAsphere) return launch_gen_nr<T, kNrEvenAsphere>(a, false, stream);
  return launch_gen_nr<T, 1>(a, false, stream);
}

#if OL_TRACE_TU == 0 || OL_TRACE_TU == 1
Tuning& tuning() {
  static Tuning t = [] {
    Tuning v;
    if (const char* e = getenv("OL_TRACE_RPT")) v.rays_per_thread = atoi(e);
    if (const char* e = getenv("OL_RECORD_WG_CAP")) {
      const int c = atoi(e);
      if (c >= 0 && c <= 8) v.record_wg_cap = c;
    }
    return v;
  }();
  return t;
}
#endif

template <typename T>
hipError_t launch_trace(const TraceArgs<T>& a, bool vector_ok, int nr_family,
                        hipStream_t stream) {
  constexpr int kVec = 16 / sizeof(T);  // rays per 16-byte lane vector
  // nr_family: kNrNone, kNrGeneric or a single-family kind (capi.hip:newton_family)
  const int nr = nr_family == kNrNone
                     ? kNrNone
                     : ((a.flags & kTraceCompact) && tuning().compact ? kNrCompact : nr_family);
  if (nr_family == kNrReference) {
    if (a.nr_iters == nullptr) return hipErrorInvalidValue;
    return launch_rpt<T, 1>(a, kNrReference, stream);
  }
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
  const bool prefer_one = (nr != kNrNone && nr != kNrCompact) || a.record != nullptr;
  if (want == 1 || (want == 0 && prefer_one && nr != 2))
    return launch_rpt<T, 1>(a, nr == 2 ? 1 : nr, stream);
  return launch_rpt<T, kVec>(a, nr, stream);
}

// OL_TRACE_TU: 0 = everything in this translation unit; 1 / 2 = the fp32 / fp64
// instantiations only (trace_kernel_f32.hip / trace_kernel_f64.hip include this file so
// that the two halves compile in parallel); 3 = no launcher at all: the including file
// instantiates the kernels it wants to look at (tools/kernel_probe.py)
#ifndef OL_TRACE_TU
#define OL_TRACE_TU 0
#endif
#if OL_TRACE_TU == 0 || OL_TRACE_TU == 1
template hipError_t launch_trace<float>(const TraceArgs<float>&, bool, int, hipStream_t);
template hipError_t launch_trace_generate<float>(const TraceArgs<float>&, int, bool,
                                                 hipStream_t);
#endif
#if OL_TRACE_TU == 0 || OL_TRACE_TU == 2
template hipError_t launch_trace<double>(const TraceArgs<double>&, bool, int, hipStream_t);
template hipError_t launch_trace_generate<double>(const TraceArgs<double>&, int, bool,
                                                  hipStream_t);
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

// BATCH: blockIdx.y = cell of a (field, wavelength) grid (`batch`, see trace_launch.h); the
// single-cell form carries a one-byte stand-in so that its kernarg segment stays what it was.
struct NoBatch {
  char unused;
};
template <typename T, int RPT, int NR, bool FIELDP, bool APOD, bool BATCH = false>
__global__ __launch_bounds__(kTraceBlock) void spot_trace_kernel(
    const DevSurfHot<T>* __restrict__ surf_tab, const DevSurfCold<T>* __restrict__ cold_tab,
    const DevOptics<T>* __restrict__ optics_tab, const T* __restrict__ coeff_tab,
    SpotArgs<T> a, typename std::conditional<BATCH, SpotBatch<T>, NoBatch>::type batch) {
  static_assert(!(BATCH && FIELDP), "a batched cell has ONE field point");
  // Newton ranges (and every fp64 instance): nothing of the argument block is held across the
  // surface loop -- each phase of a tile reads what it needs from the kernarg segment
  constexpr bool kFetch = fetch_level<T, NR, true>() >= 1;      // argument block
  constexpr bool kFetchRows = fetch_level<T, NR, true>() >= 2;  // table rows
  // FIELDP: per-ray field planes.  A template parameter because the per-ray
  // double-precision tangent (tan_deg) would otherwise set the register budget of
  // the common launch-uniform-field case as well.
  constexpr bool field_planes = FIELDP;

  SpotAcc acc;
  uint32_t status = 0;
  bool prt_fresh = false;  // unpolarised kernel: unused
  constexpr int64_t kTileRays = (int64_t)kTraceBlock * RPT;
  const int tiles_per_block = a.tiles_per_block;

  for (int tt = 0; tt < tiles_per_block; ++tt) {
    const auto A0 = arg_view<kFetch, T>(a);
    const int64_t n_rays = A0->n;
    const int64_t ntiles = (n_rays + kTileRays - 1) / kTileRays;
    const int64_t tile = (int64_t)blockIdx.x * tiles_per_block + tt;
    if (tile >= ntiles) break;  // workgroup-uniform
    const int64_t base = (tile * kTraceBlock + threadIdx.x) * RPT;
    const int64_t left = n_rays - base;
    const int cnt = left >= RPT ? RPT : (left > 0 ? (int)left : 0);
    const auto& in_ = A0->in;
    const bool vig_planes = !BATCH && in_.vx != nullptr;
    // launch-uniform field: the two tangents come from the host (uniform_field_tangents)
    T tx0, ty0, vx0, vy0;
    if constexpr (BATCH) {
      const SpotCell<T>& cell = batch.c[blockIdx.y];
      tx0 = cell.tx; ty0 = cell.ty; vx0 = cell.vx; vy0 = cell.vy;
    } else {
      tx0 = in_.tx0; ty0 = in_.ty0; vx0 = in_.vx0; vy0 = in_.vy0;
    }
    const uint32_t rg_flags = in_.flags;

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
        in[4][k] = vig_planes ? vec_get<T, RPT>(v[4], k) : vx0;
        in[5][k] = vig_planes ? vec_get<T, RPT>(v[5], k) : vy0;
      }
    } else {
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        const bool ok = k < cnt;
        in[0][k] = ok ? in_.px[base + k] : T(0);
        in[1][k] = ok ? in_.py[base + k] : T(0);
        in[2][k] = (ok && field_planes) ? in_.hx[base + k] : T(0);
        in[3][k] = (ok && field_planes) ? in_.hy[base + k] : T(0);
        in[4][k] = (ok && vig_planes) ? in_.vx[base + k] : vx0;
        in[5][k] = (ok && vig_planes) ? in_.vy[base + k] : vy0;
      }
    }

    using LP = LanePack<T, RPT, 0, NR>;  // fp32 lean kernel: packed pairs of rays
    using V = typename LP::V;
    constexpr int NV = LP::NV;
    Ray<V> r[NV];
    {
      const RaygenConsts<T> c = consts_of(&A0->rgc);
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        T tx = tx0, ty = ty0, o[6];
        if constexpr (FIELDP) {
          if ((rg_flags & kRaygenCheckField) && (outside_unit(in[2][k]) || outside_unit(in[3][k])))
            status |= kStatusFieldRange;
          raygen_field<T>(c, in[2][k], in[3][k], tx, ty);
        }
        raygen_pupil<T>(rg_flags, in[4][k], in[5][k], in[0][k], in[1][k], status);
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
    }

    bool is_global = true;
    DevSurf<T> last_traced;
    last_traced.cold = as_const(cold_tab);
    Prt<T, 0> P[1];
    Ray<V> gv[NV];
    if constexpr (kFetchRows) {
      // table rows and loop bounds re-read where they are used (see trace_kernel); nothing
      // of the argument block is carried through the surface loop
      int last_idx = 0;
      const int first = kernargs<T, SpotArgs<T>>()->a.first;
      for (int s = first; s <= kernargs<T, SpotArgs<T>>()->a.last; ++s) {
        const SurfFetched<T> h = [&](int s_) {
          if constexpr (BATCH)
            return fetched_surface_of<T, SpotArgs<T>>(s_, batch.c[blockIdx.y].optics,
                                                      batch.c[blockIdx.y].n_wl,
                                                      batch.c[blockIdx.y].wl);
          else
            return fetched_surface<T, SpotArgs<T>>(s_);
        }(s);
        if (refresh(h.hot)->interaction != kRecordOnly) {
          surface_step<V, NV, 0, NR, (RPT > 1)>(h, as_const(kernargs<T, SpotArgs<T>>()->coeffs), is_global, r,
                                     P, status, prt_fresh);
          is_global = false;
          last_idx = s;
        }
      }
      const DevSurf<T> lt = fetched_surface<T, SpotArgs<T>>(last_idx).surf();  // (no optics row)
      // kSpotHitsLocal: the hits stay in the last surface's own frame (the flag word is read
      // again here: nothing of the argument block is carried through the surface loop)
      const bool keep =
          is_global || (kernargs<T, SpotArgs<T>>()->a.in.flags & kSpotHitsLocal) != 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) gv[j] = keep ? r[j] : to_global<V>(lt, r[j]);
    } else {
      constexpr bool kPrefetch =
          (OL_FUSED_NR_PREFETCH || NR == 0) && (OL_PREFETCH_F64 || sizeof(T) == 4);
      DevSurfHot<T> cur = surf_tab[a.first];
      for (int s = a.first; s <= a.last; ++s) {
        DevSurf<T> S;
        if constexpr (kPrefetch) {
          static_cast<DevSurfHot<T>&>(S) = cur;
          if (s < a.last) cur = surf_tab[s + 1];
        } else {
          static_cast<DevSurfHot<T>&>(S) = surf_tab[s];
        }
        S.cold = as_const(cold_tab) + s;
        if (S.interaction != kRecordOnly) {
          DevOptics<T> O;
          if constexpr (BATCH) {
            const SpotCell<T>& cell = batch.c[blockIdx.y];
            O = cell.optics[s * cell.n_wl + cell.wl];
          } else {
            O = optics_tab[s * a.n_wl + a.wl];
          }
          surface_step<V, NV, 0, NR, (RPT > 1)>(S, O, as_const(coeff_tab), is_global, r, P, status,
                                     prt_fresh);
          is_global = false;
          last_traced = S;
        }
      }
      const bool keep = is_global || (a.in.flags & kSpotHitsLocal) != 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) gv[j] = keep ? r[j] : to_global<V>(last_traced, r[j]);
    }

    T hx_[RPT], hy_[RPT], hi_[RPT];
    const auto A1 = arg_view<kFetch, T>(a);
    double cx, cy;
    if constexpr (BATCH) {
      cx = batch.c[blockIdx.y].cx; cy = batch.c[blockIdx.y].cy;
    } else {
      cx = A1->cx; cy = A1->cy;
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
      const Ray<T> g = LP::ray(gv, k);
      hx_[k] = g.x; hy_[k] = g.y; hi_[k] = g.i;
      if (k < cnt) acc.add(g.x, g.y, g.i, cx, cy);
    }
    if (A1->hits[0] != nullptr && cnt > 0) {
      const RayIndexT<false> at{tile * (kTraceBlock * RPT), (uint32_t)threadIdx.x * RPT};
      int64_t cell_off = 0;
      if constexpr (BATCH) cell_off = (int64_t)blockIdx.y * 3 * batch.hits_stride;
      store_plane<T, RPT>(A1->hits[0] + cell_off, at, cnt, hx_);
      store_plane<T, RPT>(A1->hits[1] + cell_off, at, cnt, hy_);
      store_plane<T, RPT>(A1->hits[2] + cell_off, at, cnt, hi_);
    }
  }

  const auto A2 = arg_view<kFetch, T>(a);
  if constexpr (BATCH) acc.flush(A2->out + 8 * blockIdx.y);
  else acc.flush(A2->out);
  uint32_t* status_out = A2->status;
  if (status && status_out) atomicOr(status_out, status);
}

template <typename T, int RPT, int NR, bool BATCH = false>
static hipError_t launch_spot_nr(const SpotArgs<T>& a_in, hipStream_t stream,
                                 const SpotBatch<T>* batch = nullptr) {
  SpotArgs<T> a = a_in;
  const int64_t tile_rays = (int64_t)kTraceBlock * RPT;
  const int64_t ntiles = (a.n + tile_rays - 1) / tile_rays;
  static const int forced = [] {
    const char* e = getenv("OL_SPOT_TILES");
    return e ? atoi(e) : 0;
  }();
  constexpr int64_t kTargetBlocks = 8192;  // ~8 rounds of 256 CUs x 4 resident workgroups
  const int64_t cells = BATCH ? batch->n_cells : 1;
  // (a batch fills the part with cells x blocks workgroups: fewer blocks per cell do)
  int64_t tpb = forced > 0 ? forced : (ntiles * cells + kTargetBlocks - 1) / kTargetBlocks;
  if (tpb < 1) tpb = 1;
  if (tpb > 1024) tpb = 1024;
  a.tiles_per_block = (int32_t)tpb;
  a.rgc = RaygenConsts<T>(a.rg);
  if (!BATCH && a.in.hx == nullptr) uniform_field_tangents<T>(a.rg, a.in);
  const int64_t blocks = (ntiles + tpb - 1) / tpb;
  if (blocks == 0 || cells == 0) return hipSuccess;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  const bool apod = a.rg.apod_kind != 0;
  if constexpr (BATCH) {
#define OL_SPOT_LAUNCH_B(A)                                                                   \
  hipLaunchKernelGGL((spot_trace_kernel<T, RPT, NR, false, A, true>),                         \
                     dim3((unsigned)blocks, (unsigned)cells), dim3(kTraceBlock), 0, stream,   \
                     a.surf, a.cold, a.optics, a.coeffs, a, *batch)
    if (apod) OL_SPOT_LAUNCH_B(true);
    else OL_SPOT_LAUNCH_B(false);
#undef OL_SPOT_LAUNCH_B
  } else {
#define OL_SPOT_LAUNCH(F, A)                                                               \
  hipLaunchKernelGGL((spot_trace_kernel<T, RPT, NR, F, A>), dim3((unsigned)blocks),        \
                     dim3(kTraceBlock), 0, stream, a.surf, a.cold, a.optics, a.coeffs, a,  \
                     NoBatch{})
    const bool fieldp = a.in.hx != nullptr;
    if (fieldp && apod) OL_SPOT_LAUNCH(true, true);
    else if (fieldp) OL_SPOT_LAUNCH(true, false);
    else if (apod) OL_SPOT_LAUNCH(false, true);
    else OL_SPOT_LAUNCH(false, false);
#undef OL_SPOT_LAUNCH
  }
  return hipGetLastError();
}

template <typename T, bool BATCH>
static hipError_t dispatch_spot(const SpotArgs<T>& a, const SpotBatch<T>* batch, bool vector_ok,
                                int nr_family, hipStream_t stream) {
  constexpr int kVec = 16 / sizeof(T);
  // same defaults as record-last traces (launch_trace): conic-only ranges are ALU
  // bound and want the 16-byte vector of rays per lane; Newton ranges one ray (and, like
  // launch_trace, the single-family instantiation when the range allows it)
  const int want = tuning().rays_per_thread;
  if (nr_family != kNrNone) {
    if (vector_ok && want == 2) return launch_spot_nr<T, kVec, 1, BATCH>(a, stream, batch);
    if (nr_family == kNrZernike) return launch_spot_nr<T, 1, kNrZernike, BATCH>(a, stream, batch);
    if (nr_family == kNrEvenAsphere)
      return launch_spot_nr<T, 1, kNrEvenAsphere, BATCH>(a, stream, batch);
    return launch_spot_nr<T, 1, 1, BATCH>(a, stream, batch);
  }
  if (!vector_ok || want == 1) return launch_spot_nr<T, 1, 0, BATCH>(a, stream, batch);
  return launch_spot_nr<T, kVec, 0, BATCH>(a, stream, batch);
}

template <typename T>
hipError_t launch_spot_trace(const SpotArgs<T>& a, bool vector_ok, int nr_family,
                             hipStream_t stream) {
  return dispatch_spot<T, false>(a, nullptr, vector_ok, nr_family, stream);
}

template <typename T>
hipError_t launch_spot_batch(const SpotArgs<T>& a, const SpotBatch<T>& batch, bool vector_ok,
                             int nr_family, hipStream_t stream) {
  if (batch.n_cells < 0 || batch.n_cells > kSpotBatchCells) return hipErrorInvalidValue;
  return dispatch_spot<T, true>(a, &batch, vector_ok, nr_family, stream);
}

#if OL_TRACE_TU == 0 || OL_TRACE_TU == 1
template hipError_t launch_spot_trace<float>(const SpotArgs<float>&, bool, int, hipStream_t);
template hipError_t launch_spot_batch<float>(const SpotArgs<float>&, const SpotBatch<float>&, bool,
                                             int, hipStream_t);
#endif
#if OL_TRACE_TU == 0 || OL_TRACE_TU == 2
template hipError_t launch_spot_trace<double>(const SpotArgs<double>&, bool, int, hipStream_t);
template hipError_t launch_spot_batch<double>(const SpotArgs<double>&, const SpotBatch<double>&,
                                              bool, int, hipStream_t);
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
// torch reductions) moves 8 (S + 2) + 13 planes.  Wavefront work is fp64 (an OPD good to
// lambda/1000 over a 200 mm path); one ray per lane where the range has a Newton surface (the
// register budget), two without (RPT).  out[] of the moments:
//   0 sum w   1 sum w X   2 sum w Y   3 sum w XX   4 sum w XY   5 sum w YY
//   6 sum w o 7 sum w o X 8 sum w o Y      (w = intensity, o = OPD, X/Y = pupil point)
//   9 #{i > 0}   10 sum o [i > 0]   11 sum o^2 [i > 0]
// --------------------------------------------------------------------------
// DEVREF: the reference sphere / plane is read from device memory (ol_trace_opd_dev) -- an
// instantiation of its own: as a launch-uniform branch in the one kernel it cost the 1e7-ray
// launches 0.9-1.8 % (profiles/r04_ab_opd_devref.txt)
// RPT: rays per lane -- 2 (one 16-byte vector of pupil coordinates, two independent chains of
// fp64 arithmetic per lane) for ranges without a Newton surface: one ray per lane is bound by
// the latency of its dependent chain there.  The arithmetic of a ray is the
// same either way (no shared reciprocals): the per-ray outputs are bit-identical.
template <typename T, int NR, bool APOD, bool DEVREF = false, int RPT = 1>
__global__ __launch_bounds__(kTraceBlock) void opd_trace_kernel(
    const DevSurfHot<T>* __restrict__ surf_tab, const DevSurfCold<T>* __restrict__ cold_tab,
    const DevOptics<T>* __restrict__ optics_tab, const T* __restrict__ coeff_tab,
    OpdArgs<T> a) {
  // every instance is fp64: nothing of the argument block is held across the surface loop,
  // each phase of a ray reads what it needs from the kernarg segment (see spot_trace_kernel)
  constexpr bool kFetch = fetch_level<T, NR, true>() >= 1;      // argument block
  constexpr bool kFetchRows = fetch_level<T, NR, true>() >= 2;  // table rows
  uint32_t status = 0;
  bool prt_fresh = false;
  double s[kOpdMoments];
#pragma unroll
  for (int k = 0; k < kOpdMoments; ++k) s[k] = 0.0;

  // lane l of tile t works on rays (t * kTraceBlock + l) * RPT ... + RPT - 1
  for (int64_t j = ((int64_t)blockIdx.x * kTraceBlock + threadIdx.x) * RPT;
       j < arg_view<kFetch, T>(a)->n; j += (int64_t)gridDim.x * kTraceBlock * RPT) {
    Ray<T> r[RPT];
    int cnt = RPT;
    {
      const auto A0 = arg_view<kFetch, T>(a);
      const auto& in_ = A0->in;
      T px[RPT], py[RPT];
      if constexpr (RPT == 1) {
        px[0] = in_.px[j]; py[0] = in_.py[j];
      } else {
        const int64_t left = A0->n - j;
        cnt = left >= RPT ? RPT : (int)left;
        if (cnt == RPT) {
          using PV = typename VecOf<T, RPT>::type;
          const PV vx_ = *reinterpret_cast<const PV*>(in_.px + j);
          const PV vy_ = *reinterpret_cast<const PV*>(in_.py + j);
#pragma unroll
          for (int k = 0; k < RPT; ++k) {
            px[k] = vec_get<T, RPT>(vx_, k); py[k] = vec_get<T, RPT>(vy_, k);
          }
        } else {
          // the last lane of an odd launch: the rays past the end trace the pupil's centre
          // and are kept out of every output
#pragma unroll
          for (int k = 0; k < RPT; ++k) {
            px[k] = k < cnt ? in_.px[j + k] : T(0); py[k] = k < cnt ? in_.py[j + k] : T(0);
          }
        }
      }
      const RaygenConsts<T> c = consts_of(&A0->rgc);
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        T vx = in_.vx0, vy = in_.vy0, o[6];
        uint32_t st = 0;
        raygen_pupil<T>(in_.flags, vx, vy, px[k], py[k], st);
        if (k < cnt) status |= st;
        raygen_one<T>(c, in_.tx0, in_.ty0, px[k], py[k], vx, vy, o);
        r[k].x = o[0]; r[k].y = o[1]; r[k].z = o[2];
        r[k].L = o[3]; r[k].M = o[4]; r[k].N = o[5];
        if constexpr (APOD) r[k].i = raygen_apodize<T>(c, px[k], py[k]); else r[k].i = T(1);
        r[k].opd = T(0);
      }
    }

    bool is_global = true;
    DevSurf<T> last_traced;
    last_traced.cold = as_const(cold_tab);
    Prt<T, 0> P[1];
    Ray<T> g[RPT];
    if constexpr (kFetchRows) {
      int last_idx = 0;
      const int first = kernargs<T, OpdArgs<T>>()->a.first;
      for (int sidx = first; sidx <= kernargs<T, OpdArgs<T>>()->a.last; ++sidx) {
        const SurfFetched<T> h = fetched_surface<T, OpdArgs<T>>(sidx);
        if (refresh(h.hot)->interaction != kRecordOnly) {
          surface_step<T, RPT, 0, NR>(h, as_const(kernargs<T, OpdArgs<T>>()->coeffs), is_global,
                                      r, P, status, prt_fresh);
          is_global = false;
          last_idx = sidx;
        }
      }
      const DevSurf<T> lt = fetched_surface<T, OpdArgs<T>>(last_idx).surf();
#pragma unroll
      for (int k = 0; k < RPT; ++k) g[k] = is_global ? r[k] : to_global<T>(lt, r[k]);
    } else {
      constexpr bool kPrefetch =
          (OL_FUSED_NR_PREFETCH || NR == 0) && (OL_PREFETCH_F64 || sizeof(T) == 4);
      DevSurfHot<T> cur = surf_tab[a.first];
      for (int sidx = a.first; sidx <= a.last; ++sidx) {
        DevSurf<T> S;
        if constexpr (kPrefetch) {
          static_cast<DevSurfHot<T>&>(S) = cur;
          if (sidx < a.last) cur = surf_tab[sidx + 1];
        } else {
          static_cast<DevSurfHot<T>&>(S) = surf_tab[sidx];
        }
        S.cold = as_const(cold_tab) + sidx;
        if (S.interaction != kRecordOnly) {
          const DevOptics<T> O = optics_tab[sidx * a.n_wl + a.wl];
          surface_step<T, RPT, 0, NR>(S, O, as_const(coeff_tab), is_global, r, P, status,
                                      prt_fresh);
          is_global = false;
          last_traced = S;
        }
      }
#pragma unroll
      for (int k = 0; k < RPT; ++k) g[k] = is_global ? r[k] : to_global<T>(last_traced, r[k]);
    }
    const auto A1 = arg_view<kFetch, T>(a);
    // the pupil coordinates of the tilt term are the ones the CALLER passed (the
    // reference corrects with the distribution's points, strategy.py:88-139)
    T ov[RPT], gi[RPT], pu[3][RPT];
    {
      // the reference sphere / plane: from the argument block, or (ol_trace_opd_dev) from the
      // device structure ol_wavefront_reference left -- a launch-uniform choice, scalar loads
      // from the constant address space either way
      WavefrontConsts<T> w;
      if constexpr (DEVREF) w = load_consts(as_const(A1->wf_dev));
      else w = consts_of(&A1->wfc);
#pragma unroll
      for (int k = 0; k < RPT; ++k) {
        final_propagate<T, false>(w, g[k]);
        const int64_t jk = k < cnt ? j + k : j;
        T q[3];
        ov[k] = wavefront_one<T>(w, g[k].x, g[k].y, g[k].z, g[k].L, g[k].M, g[k].N, g[k].opd,
                                 A1->in.px[jk], A1->in.py[jk], q);
        gi[k] = g[k].i;
        pu[0][k] = q[0]; pu[1][k] = q[1]; pu[2][k] = q[2];
      }
    }
    T* const pup0 = A1->pupil[0];
    if constexpr (RPT == 1) {
      A1->opd[j] = ov[0];
      A1->inten[j] = gi[0];
      if (pup0) {
        pup0[j] = pu[0][0];
        A1->pupil[1][j] = pu[1][0];
        A1->pupil[2][j] = pu[2][0];
      }
    } else {
      const RayIndexT<false> at{j, 0u};
      store_plane<T, RPT>(A1->opd, at, cnt, ov);
      store_plane<T, RPT>(A1->inten, at, cnt, gi);
      if (pup0) {
        store_plane<T, RPT>(pup0, at, cnt, pu[0]);
        store_plane<T, RPT>(A1->pupil[1], at, cnt, pu[1]);
        store_plane<T, RPT>(A1->pupil[2], at, cnt, pu[2]);
      }
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k)
      if (k < cnt)
        opd_accumulate(s, (double)gi[k], (double)ov[k], (double)pu[0][k], (double)pu[1][k],
                       gi[k] > T(0));
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
    if (v != 0.0) unsafeAtomicAdd(&arg_view<kFetch, T>(a)->mom[threadIdx.x], v);
  }
  uint32_t* status_out = arg_view<kFetch, T>(a)->status;
  if (status && status_out) atomicOr(status_out, status);
}

// vector_ok: px / py and every output plane allow 16-byte lane accesses (capi.hip)
template <typename T>
hipError_t launch_opd_trace(const OpdArgs<T>& a_in, bool vector_ok, int nr_family,
                            hipStream_t stream) {
  OpdArgs<T> a = a_in;
  uniform_field_tangents<T>(a.rg, a.in);
  a.rgc = RaygenConsts<T>(a.rg);
  a.wfc = WavefrontConsts<T>(a.wf);
  if (a.n == 0) return hipSuccess;
  // two rays per lane where there is no Newton surface and the launch fills the part either
  // way (>= 2048 workgroups; OL_TUNE_RAYS_PER_THREAD = 1: always one)
#ifndef OL_OPD_TWO_RAYS
#define OL_OPD_TWO_RAYS 1
#endif
  const bool two = OL_OPD_TWO_RAYS && vector_ok && nr_family == kNrNone &&
                   tuning().rays_per_thread != 1 && a.n >= (int64_t)4096 * kTraceBlock;
  const int64_t per_block = (int64_t)kTraceBlock * (two ? 2 : 1);
  int64_t blocks = (a.n + per_block - 1) / per_block;
  if (blocks > 8192) blocks = 8192;  // grid-stride beyond: keeps the atomics few
  const bool apod = a.rg.apod_kind != 0;
#define OL_OPD_LAUNCH_R(N, A, R)                                                             \
  do {                                                                                       \
    if (a.wf_dev != nullptr)                                                                 \
      hipLaunchKernelGGL((opd_trace_kernel<T, N, A, true, R>), dim3((unsigned)blocks),       \
                         dim3(kTraceBlock), 0, stream, a.surf, a.cold, a.optics, a.coeffs, a); \
    else                                                                                     \
      hipLaunchKernelGGL((opd_trace_kernel<T, N, A, false, R>), dim3((unsigned)blocks),      \
                         dim3(kTraceBlock), 0, stream, a.surf, a.cold, a.optics, a.coeffs, a); \
  } while (0)
#define OL_OPD_LAUNCH(N, A) OL_OPD_LAUNCH_R(N, A, 1)
  if (nr_family == kNrZernike) {
    if (apod) OL_OPD_LAUNCH(kNrZernike, true); else OL_OPD_LAUNCH(kNrZernike, false);
  } else if (nr_family == kNrEvenAsphere) {
    if (apod) OL_OPD_LAUNCH(kNrEvenAsphere, true); else OL_OPD_LAUNCH(kNrEvenAsphere, false);
  } else if (nr_family != kNrNone) {
    if (apod) OL_OPD_LAUNCH(1, true); else OL_OPD_LAUNCH(1, false);
  } else if (two) {
    if (apod) OL_OPD_LAUNCH_R(0, true, 2); else OL_OPD_LAUNCH_R(0, false, 2);
  } else {
    if (apod) OL_OPD_LAUNCH(0, true); else OL_OPD_LAUNCH(0, false);
  }
#undef OL_OPD_LAUNCH
#undef OL_OPD_LAUNCH_R
  return hipGetLastError();
}

#if OL_TRACE_TU == 0 || OL_TRACE_TU == 2
template hipError_t launch_opd_trace<double>(const OpdArgs<double>&, bool, int, hipStream_t);
#endif

// --------------------------------------------------------------------------
// chief ray -> reference sphere / plane, device-resident (ol_wavefront_reference)
// --------------------------------------------------------------------------
// One lane generates the chief ray of the field point (pupil point (0, 0)), walks the surface
// table like opd_trace_kernel and writes the WavefrontConsts the OPD launch reads:
// centre = the chief ray's image point, R = its distance to the exit pupil on the axis
// (strategy.py:228-243), or the plane through it normal to the chief ray (:260-284);
// opd_ref = the chief ray's own path length to that surface (:181-184: its pupil point is
// (0, 0), so the tilt term vanishes).  A cold kernel (one wave, once per wavefront).
template <typename T, int NR>
__global__ __launch_bounds__(64) void chief_ref_kernel(
    const DevSurfHot<T>* __restrict__ surf_tab, const DevSurfCold<T>* __restrict__ cold_tab,
    const DevOptics<T>* __restrict__ optics_tab, const T* __restrict__ coeff_tab,
    ChiefArgs<T> a) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  using m = Math<T>;
  uint32_t status = 0;
  bool prt_fresh = false;
  Ray<T> r[1];
  {
    T px = T(0), py = T(0), vx = a.in.vx0, vy = a.in.vy0, o[6];
    raygen_pupil<T>(a.in.flags, vx, vy, px, py, status);
    raygen_one<T>(a.rgc, a.in.tx0, a.in.ty0, px, py, vx, vy, o);
    r[0].x = o[0]; r[0].y = o[1]; r[0].z = o[2];
    r[0].L = o[3]; r[0].M = o[4]; r[0].N = o[5];
    r[0].i = T(1);
    r[0].opd = T(0);
  }
  bool is_global = true;
  int last_idx = a.first;
  Prt<T, 0> P[1];
  for (int s = a.first; s <= a.last; ++s) {
    const SurfFetched<T> h{as_const(surf_tab) + s, as_const(cold_tab) + s,
                           as_const(optics_tab) + (s * a.n_wl + a.wl)};
    if (refresh(h.hot)->interaction != kRecordOnly) {
      surface_step<T, 1, 0, NR>(h, as_const(coeff_tab), is_global, r, P, status, prt_fresh);
      is_global = false;
      last_idx = s;
    }
  }
  Ray<T> g = r[0];
  if (!is_global) {
    const SurfFetched<T> h{as_const(surf_tab) + last_idx, as_const(cold_tab) + last_idx,
                           as_const(optics_tab) + (last_idx * a.n_wl + a.wl)};
    g = to_global<T>(h.surf(), r[0]);
  }
  WavefrontConsts<T> w = a.wfc;
  final_propagate<T, true>(w, g);
  w.xc = g.x; w.yc = g.y; w.zc = g.z;
  T t_back;
  if (w.planar) {
    w.R = T(0);
    w.nx = g.L; w.ny = g.M; w.nz = g.N;
    t_back = T(0);  // the chief ray's image point lies ON the plane
  } else {
    const T dz = g.z - a.pupil_z;
    w.R = m::sqrt(g.x * g.x + g.y * g.y + dz * dz);
    w.nx = w.ny = w.nz = T(0);
    // path_length of the chief ray itself: the sphere is centred on its image point
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
  if (status && a.status) atomicOr(a.status, status);
}

template <typename T>
hipError_t launch_chief_reference(const ChiefArgs<T>& a_in, int nr_family, hipStream_t stream) {
  ChiefArgs<T> a = a_in;
  uniform_field_tangents<T>(a.rg, a.in);
  a.rgc = RaygenConsts<T>(a.rg);
#define OL_CHIEF_LAUNCH(N)                                                                   \
  hipLaunchKernelGGL((chief_ref_kernel<T, N>), dim3(1), dim3(64), 0, stream, a.surf, a.cold, \
                     a.optics, a.coeffs, a)
  if (nr_family == kNrNone) OL_CHIEF_LAUNCH(0);
  else OL_CHIEF_LAUNCH(1);   // (cold: the generic Newton instantiation serves every family)
#undef OL_CHIEF_LAUNCH
  return hipGetLastError();
}
#if OL_TRACE_TU == 0 || OL_TRACE_TU == 2
template hipError_t launch_chief_reference<double>(const ChiefArgs<double>&, int, hipStream_t);
#endif

}  // namespace ol
