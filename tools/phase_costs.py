#!/usr/bin/env python3
"""Instruction budget of ONE surface of the fused trace, phase by phase (CPU only).

Every phase of `surface_step` (surface_math.h) -- frame change, conic intersection, move to the
hit + normal, interaction (OPD, clip, Snell), the local -> global transform of a recorded row,
the spot accumulation -- and the ray generator / reference-sphere OPD that bracket the fused
kernels is compiled BY ITSELF into a tiny gfx950 kernel (inputs loaded from planes, outputs
stored), and its vector instructions are counted in the ISA, minus those of an empty kernel
with the same loads and stores.  The phases are branch-free, so the static count IS the
per-ray count; per type: float, the packed pair f32x2 (two rays per instruction stream: the
figure is per PAIR) and double.  Transcendental (quarter-rate) and fp64 instructions are
listed separately.

    python tools/phase_costs.py            # prints the table (profiles/r04_phase_costs.txt)
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "optiland_amd", "csrc")

SRC = r'''
#include <hip/hip_runtime.h>
#include "device_table.h"
#include "trace_launch.h"
#include "raygen_device.h"
#include "wavefront_device.h"
#include "epilogue_device.h"
#include "surface_math.h"
using namespace ol;

template <typename V> struct Sc { using T = typename Math<V>::scalar; };

#define LOAD8(r) do { (r).x = in[0*n+i]; (r).y = in[1*n+i]; (r).z = in[2*n+i]; (r).L = in[3*n+i]; \
  (r).M = in[4*n+i]; (r).N = in[5*n+i]; (r).i = in[6*n+i]; (r).opd = in[7*n+i]; } while (0)
#define STORE8(r) do { out[0*n+i] = (r).x; out[1*n+i] = (r).y; out[2*n+i] = (r).z; out[3*n+i] = (r).L; \
  out[4*n+i] = (r).M; out[5*n+i] = (r).N; out[6*n+i] = (r).i; out[7*n+i] = (r).opd; } while (0)

// the surface constants arrive as kernel arguments (SGPRs), as in the kernels
template <typename V>
__global__ void k_empty(const V* in, V* out, int n, DevSurfHot<typename Sc<V>::T> s,
                        DevOptics<typename Sc<V>::T> o) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<V> r; LOAD8(r); STORE8(r);
}
template <typename V>
__global__ void k_frame(const V* in, V* out, int n, DevSurfHot<typename Sc<V>::T> s,
                        DevOptics<typename Sc<V>::T> o) {
  using T = typename Sc<V>::T;
  int i = blockIdx.x * 256 + threadIdx.x; Ray<V> r[1]; LOAD8(r[0]);
  DevSurf<T> S; static_cast<DevSurfHot<T>&>(S) = s; S.cold = nullptr;
  S.flags &= ~(kSurfRotated | kSurfRelRotated);
  into_local_frame<V, 1>(S, false, r);      // unrotated: three additions
  STORE8(r[0]);
}
template <typename V>
__global__ void k_distance(const V* in, V* out, int n, DevSurfHot<typename Sc<V>::T> s,
                           DevOptics<typename Sc<V>::T> o) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<V> r; LOAD8(r);
  r.opd = curved_distance<V>(s.cv, s.kp1, r.x, r.y, r.z, r.L, r.M, r.N);
  STORE8(r);
}
template <typename V, bool SPHERE>
__global__ void k_hit_normal(const V* in, V* out, int n, DevSurfHot<typename Sc<V>::T> s,
                             DevOptics<typename Sc<V>::T> o) {
  using T = typename Sc<V>::T; using m = Math<V>;
  int i = blockIdx.x * 256 + threadIdx.x; Ray<V> r; LOAD8(r);
  const V t = r.opd;
  r.x = m::fma(t, r.L, r.x); r.y = m::fma(t, r.M, r.y); r.z = m::fma(t, r.N, r.z);
  V nx, ny, nz;
  conic_normal<V>(s.cv, SPHERE ? T(1) : s.kp1, r.x, r.y, r.z, nx, ny, nz);
  r.L = nx; r.M = ny; r.N = nz;
  STORE8(r);
}
template <typename V, int APER>
__global__ void k_interact(const V* in, const V* nrm, V* out, int n,
                           DevSurfHot<typename Sc<V>::T> s, DevOptics<typename Sc<V>::T> o,
                           const DevSurfCold<typename Sc<V>::T>* cold) {
  using T = typename Sc<V>::T;
  int i = blockIdx.x * 256 + threadIdx.x; Ray<V> r[1]; LOAD8(r[0]);
  DevSurf<T> S; static_cast<DevSurfHot<T>&>(S) = s; S.cold = as_const(cold);
  S.aperture_kind = APER ? kApRadial : kApNone; S.coating_kind = kCoatNone; S.interaction = 0;
  DevOptics<T> O = o; O.absorb = T(0);
  V t[1] = {nrm[3*n+i]}, nx[1] = {nrm[0*n+i]}, ny[1] = {nrm[1*n+i]}, nz[1] = {nrm[2*n+i]};
  Prt<T, 0> P[1]; bool fresh = false;
  interact<V, 1, 0, false>(S, O, nullptr, t, nx, ny, nz, r, P, fresh);
  STORE8(r[0]);
}
template <typename V>
__global__ void k_to_global(const V* in, V* out, int n, DevSurfHot<typename Sc<V>::T> s,
                            DevOptics<typename Sc<V>::T> o) {
  using T = typename Sc<V>::T;
  int i = blockIdx.x * 256 + threadIdx.x; Ray<V> r; LOAD8(r);
  DevSurf<T> S; static_cast<DevSurfHot<T>&>(S) = s; S.cold = nullptr; S.flags &= ~kSurfRotated;
  Ray<V> g = to_global<V>(S, r);
  STORE8(g);
}
template <typename T>
__global__ void k_raygen(const T* in, T* out, int n, RaygenConsts<T> c, T tx, T ty) {
  int i = blockIdx.x * 256 + threadIdx.x; T o[6];
  Ray<T> r; LOAD8(r);
  raygen_one<T>(c, tx, ty, r.x, r.y, T(1), T(1), o);
  r.x = o[0]; r.y = o[1]; r.z = o[2]; r.L = o[3]; r.M = o[4]; r.N = o[5];
  STORE8(r);
}
template <typename T>
__global__ void k_wavefront(const T* in, T* out, int n, WavefrontConsts<T> w) {
  int i = blockIdx.x * 256 + threadIdx.x; T pu[3];
  Ray<T> r; LOAD8(r);
  r.opd = wavefront_one<T>(w, r.x, r.y, r.z, r.L, r.M, r.N, r.opd, r.i, r.i, pu);
  r.x = pu[0]; r.y = pu[1]; r.z = pu[2];
  STORE8(r);
}
template <typename T>
__global__ void k_spot(const T* in, T* out, int n, double cx, double cy, double* acc) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; LOAD8(r);
  double s[6] = {0, 0, 0, 0, 0, 0}, rmax = 0;
  spot_accumulate<T>(s, rmax, r.x, r.y, r.i, cx, cy);
  STORE8(r);
  for (int k = 0; k < 6; ++k) acc[k * n + i] = s[k];
  acc[6 * n + i] = rmax;
}
template <typename T>
__global__ void k_spot_empty(const T* in, T* out, int n, double cx, double cy, double* acc) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; LOAD8(r);
  STORE8(r);
  for (int k = 0; k < 7; ++k) acc[k * n + i] = cx;
}

#define INST_V(V) \
  template __global__ void k_empty<V>(const V*, V*, int, DevSurfHot<Sc<V>::T>, DevOptics<Sc<V>::T>); \
  template __global__ void k_frame<V>(const V*, V*, int, DevSurfHot<Sc<V>::T>, DevOptics<Sc<V>::T>); \
  template __global__ void k_distance<V>(const V*, V*, int, DevSurfHot<Sc<V>::T>, DevOptics<Sc<V>::T>); \
  template __global__ void k_hit_normal<V, true>(const V*, V*, int, DevSurfHot<Sc<V>::T>, DevOptics<Sc<V>::T>); \
  template __global__ void k_hit_normal<V, false>(const V*, V*, int, DevSurfHot<Sc<V>::T>, DevOptics<Sc<V>::T>); \
  template __global__ void k_interact<V, 0>(const V*, const V*, V*, int, DevSurfHot<Sc<V>::T>, DevOptics<Sc<V>::T>, const DevSurfCold<Sc<V>::T>*); \
  template __global__ void k_interact<V, 1>(const V*, const V*, V*, int, DevSurfHot<Sc<V>::T>, DevOptics<Sc<V>::T>, const DevSurfCold<Sc<V>::T>*); \
  template __global__ void k_to_global<V>(const V*, V*, int, DevSurfHot<Sc<V>::T>, DevOptics<Sc<V>::T>);
INST_V(float)
INST_V(f32x2)
INST_V(double)
#define INST_T(T) \
  template __global__ void k_raygen<T>(const T*, T*, int, RaygenConsts<T>, T, T); \
  template __global__ void k_wavefront<T>(const T*, T*, int, WavefrontConsts<T>); \
  template __global__ void k_spot<T>(const T*, T*, int, double, double, double*); \
  template __global__ void k_spot_empty<T>(const T*, T*, int, double, double, double*);
INST_T(float)
INST_T(double)
'''


# Round 6: the phases of the polarised Zernike kernel (configuration C5,
# trace_kernel<T, 1, RECORD, POLK = 1, kNrZernike, ..., GEN, EPI>), one ray per lane.  The PRT
# matrix travels with the ray: these kernels load and store 9 more planes, and so does their
# empty baseline.
SRC_POL = r'''
#include <hip/hip_runtime.h>
#include "device_table.h"
#include "trace_launch.h"
#include "raygen_device.h"
#include "wavefront_device.h"
#include "epilogue_device.h"
#include "surface_math.h"
using namespace ol;

#define LOAD8(r) do { (r).x = in[0*n+i]; (r).y = in[1*n+i]; (r).z = in[2*n+i]; (r).L = in[3*n+i]; \
  (r).M = in[4*n+i]; (r).N = in[5*n+i]; (r).i = in[6*n+i]; (r).opd = in[7*n+i]; } while (0)
#define STORE8(r) do { out[0*n+i] = (r).x; out[1*n+i] = (r).y; out[2*n+i] = (r).z; out[3*n+i] = (r).L; \
  out[4*n+i] = (r).M; out[5*n+i] = (r).N; out[6*n+i] = (r).i; out[7*n+i] = (r).opd; } while (0)
#define LOADP(P) _Pragma("unroll") for (int e = 0; e < 9; ++e) (P).m[e] = in[(8+e)*n+i]
#define STOREP(P) _Pragma("unroll") for (int e = 0; e < 9; ++e) out[(8+e)*n+i] = (P).m[e]
#define ARGS(T) const T* in, T* out, int n, DevSurfHot<T> s, DevOptics<T> o, \
             const DevSurfCold<T>* cold, const T* coeffs
#define SURF(S) DevSurf<T> S; static_cast<DevSurfHot<T>&>(S) = s; S.cold = as_const(cold)

template <typename T> __global__ void p_empty(ARGS(T)) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; LOAD8(r); STORE8(r);
}
template <typename T> __global__ void p_empty_prt(ARGS(T)) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; Prt<T, 1> P; LOAD8(r); LOADP(P);
  STORE8(r); STOREP(P);
}
// the generating prologue of a polarised launch: the run-time general form (pupil check and
// pre-scale by flags, launch-uniform field, apodisation switch)
template <typename T> __global__ void p_raygen(const T* in, T* out, int n, RaygenConsts<T> c,
                                               T tx, T ty, T vx, T vy, uint32_t flags,
                                               uint32_t* st) {
  int i = blockIdx.x * 256 + threadIdx.x; T o[6]; Ray<T> r; LOAD8(r);
  // (C5: uniform pupil, object at infinity, not telecentric -- the launch-uniform switches
  // are scalar branches, only the taken side is counted)
  c.apod_kind = 0; c.height = 0; c.infinite = 1; c.telecentric = 0;
  T px = r.x, py = r.y; uint32_t status = 0;
  raygen_pupil<T>(flags, vx, vy, px, py, status);
  raygen_one<T>(c, tx, ty, px, py, vx, vy, o);
  r.x = o[0]; r.y = o[1]; r.z = o[2]; r.L = o[3]; r.M = o[4]; r.N = o[5];
  r.i = raygen_apodize<T>(c, px, py);
  STORE8(r);
  if (status) atomicOr(st, status);
}
// start of the Newton iteration: the base conic's intersection and the re-based point
template <typename T> __global__ void p_conic_start(ARGS(T)) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; LOAD8(r); SURF(S);
  using m = Math<T>;
  const T t = conic_distance<T, false>(S, r.x, r.y, r.z, r.L, r.M, r.N);
  r.x = m::fma(t, r.L, r.x); r.y = m::fma(t, r.M, r.y); r.z = m::fma(t, r.N, r.z); r.opd = t;
  STORE8(r);
}
// ONE Newton iteration on a degree-4 one-polynomial Zernike surface (the 12 fringe terms of
// C5): evaluation + stop rule + update, as newton_iterate<kNrZernike> compiles it when the
// geometry kind and the degree are known (they are wave-uniform at run time)
template <typename T> __global__ void p_newton_iter(ARGS(T), uint32_t* st, int it) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; LOAD8(r); SURF(S);
  S.geom = kGeomZernikeMono; S.n_coeff = 4;
  NewtonRay<T> q; q.xb = r.x; q.yb = r.y; q.zb = r.z; q.dt = r.i; q.fprev = r.opd;
  q.gx = q.gy = T(0); q.active = true;
  uint32_t status = 0;
  newton_iterate<kNrZernike>(S, as_const(coeffs), q, r.L, r.M, r.N, it, status);
  r.x = q.dt; r.y = q.fprev; r.z = q.gx; r.i = q.gy; r.opd = q.active ? T(1) : T(0);
  STORE8(r);
  if (status) atomicOr(st, status);
}
// ... of which: the conic base + normalised coordinates + range check
template <typename T> __global__ void p_zern_begin(ARGS(T), uint32_t* st) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; LOAD8(r); SURF(S);
  T sag, fx, fy, xn, yn, u; uint32_t status = 0;
  zernike_begin(S, r.x, r.y, sag, fx, fy, xn, yn, u, status);
  r.x = sag; r.y = fx; r.z = fy; r.L = xn; r.M = yn; r.N = u;
  STORE8(r);
  if (status) atomicOr(st, status);
}
// ... the degree-4 polynomial: sag series and the gradient of the normal's series
template <typename T> __global__ void p_zern_mono4(ARGS(T)) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; LOAD8(r);
  T zsum, gx, gy;
  zernike_mono_fixed<T, 4>(as_const(coeffs), r.x, r.y, zsum, gx, gy);
  r.z = zsum; r.L = gx; r.M = gy;
  STORE8(r);
}
// ... the chain rule back to (x, y), vertex regularisation branch included
template <typename T> __global__ void p_zern_finish(ARGS(T)) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; LOAD8(r); SURF(S);
  T sag = r.opd, fx = r.i, fy = r.N;
  zernike_finish(S, r.x, r.y, r.z, r.L, r.M, r.i, sag, fx, fy);
  r.x = sag; r.y = fx; r.z = fy;
  STORE8(r);
}
// end of the iteration: the hit point and the unit normal from the last gradient
template <typename T> __global__ void p_nr_normal(ARGS(T)) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; LOAD8(r); using m = Math<T>;
  const T dt = r.opd, gx = r.i;
  const T x = m::fma(dt, r.L, r.x), y = m::fma(dt, r.M, r.y), z = m::fma(dt, r.N, r.z);
  const T gy = r.x * r.y;
  const T im = m::rsqrt(m::fma(gx, gx, m::fma(gy, gy, T(1))));
  r.x = x; r.y = y; r.z = z; r.L = gx * im; r.M = gy * im; r.N = -im;
  STORE8(r);
}
// interact on a Fresnel-coated refracting surface of a polarised trace: OPD, Snell, the
// (s, p) triads, the Fresnel amplitudes, the PRT update.  FIRST: the matrix is still the
// identity of a fresh trace (prt_first_diag), else the rank-2 update (prt_apply_diag)
template <typename T, bool FIRST, int COAT> __global__ void p_interact(ARGS(T), const T* nrm) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r[1]; Prt<T, 1> P[1]; LOAD8(r[0]); LOADP(P[0]);
  SURF(S);
  S.aperture_kind = kApNone; S.coating_kind = COAT; S.interaction = kRefract;
  DevOptics<T> O = o; O.absorb = T(0);
  T t[1] = {nrm[3*n+i]}, nx[1] = {nrm[0*n+i]}, ny[1] = {nrm[1*n+i]}, nz[1] = {nrm[2*n+i]};
  bool fresh = FIRST;
  interact<T, 1, 1, true>(S, O, as_const(coeffs), t, nx, ny, nz, r, P, fresh);
  STORE8(r[0]); STOREP(P[0]);
}
// ... of which: the unpolarised part (OPD + Snell), same loads
template <typename T> __global__ void p_interact_plain(ARGS(T), const T* nrm) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r[1]; Prt<T, 0> P[1]; Prt<T, 1> Q; LOAD8(r[0]);
  LOADP(Q); SURF(S);
  S.aperture_kind = kApNone; S.coating_kind = kCoatNone; S.interaction = kRefract;
  DevOptics<T> O = o; O.absorb = T(0);
  T t[1] = {nrm[3*n+i]}, nx[1] = {nrm[0*n+i]}, ny[1] = {nrm[1*n+i]}, nz[1] = {nrm[2*n+i]};
  bool fresh = false;
  interact<T, 1, 0, true>(S, O, as_const(coeffs), t, nx, ny, nz, r, P, fresh);
  STORE8(r[0]); STOREP(Q);
}
// ... the (s, p0, p1) triads alone
template <typename T> __global__ void p_pol_basis(ARGS(T), const T* nrm) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; LOAD8(r);
  const PolBasis<T> b = pol_basis(r.x, r.y, r.z, r.L, r.M, r.N, nrm[0*n+i], nrm[1*n+i],
                                  nrm[2*n+i]);
  r.x = b.sx + b.p0x + b.p1x; r.y = b.sy + b.p0y + b.p1y; r.z = b.sz + b.p0z + b.p1z;
  STORE8(r);
}
// ... the two PRT updates alone (triads and amplitudes given)
template <typename T, bool FIRST> __global__ void p_prt(ARGS(T), const T* bas) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; Prt<T, 1> P; LOAD8(r); LOADP(P);
  PolBasis<T> b;
  b.sx = bas[0*n+i]; b.sy = bas[1*n+i]; b.sz = bas[2*n+i];
  b.p0x = bas[3*n+i]; b.p0y = bas[4*n+i]; b.p0z = bas[5*n+i];
  b.p1x = bas[6*n+i]; b.p1y = bas[7*n+i]; b.p1z = bas[8*n+i];
  const T k1x = bas[9*n+i], k1y = bas[10*n+i], k1z = bas[11*n+i];
  if (FIRST) prt_first_diag<T, 1>(P, b, r.L, r.M, r.N, k1x, k1y, k1z, r.x, r.y, r.z);
  else prt_apply_diag<T, 1>(P, b, r.L, r.M, r.N, k1x, k1y, k1z, r.x, r.y, r.z);
  STORE8(r); STOREP(P);
}
// update_intensity as the epilogue of the generating launch: the launch direction is generated
// again (p_raygen above), then |P E0|^2 for NF incident states (1: a polarised state, 2: the
// unpolarised mean)
template <typename T, int NF> __global__ void p_pol_intensity(ARGS(T), PolFields<T> f, uint32_t* st) {
  int i = blockIdx.x * 256 + threadIdx.x; Ray<T> r; Prt<T, 1> P; LOAD8(r); LOADP(P);
  T Pm[9], Qm[9];
#pragma unroll
  for (int e = 0; e < 9; ++e) { Pm[e] = P.m[e]; Qm[e] = T(0); }
  f.nf = NF; uint32_t status = 0;
  // (launch-uniform facts of the state, known here so that only the taken form is counted:
  // the unpolarised mean has real amplitudes, the polarised state of C5 a phase)
  if (NF == 2) { f.ai[0] = f.bi[0] = f.ai[1] = f.bi[1] = T(0); } else { f.bi[0] = T(0.5); }
  r.i = pol_intensity_one<T, false>(f, r.L, r.M, r.N, Pm, Qm, r.i, status);
  STORE8(r); STOREP(P);
  if (status) atomicOr(st, status);
}

#define INST(T) \
  template __global__ void p_empty<T>(ARGS(T)); \
  template __global__ void p_empty_prt<T>(ARGS(T)); \
  template __global__ void p_raygen<T>(const T*, T*, int, RaygenConsts<T>, T, T, T, T, uint32_t, uint32_t*); \
  template __global__ void p_conic_start<T>(ARGS(T)); \
  template __global__ void p_newton_iter<T>(ARGS(T), uint32_t*, int); \
  template __global__ void p_zern_begin<T>(ARGS(T), uint32_t*); \
  template __global__ void p_zern_mono4<T>(ARGS(T)); \
  template __global__ void p_zern_finish<T>(ARGS(T)); \
  template __global__ void p_nr_normal<T>(ARGS(T)); \
  template __global__ void p_interact<T, true, kCoatFresnel>(ARGS(T), const T*); \
  template __global__ void p_interact<T, false, kCoatFresnel>(ARGS(T), const T*); \
  template __global__ void p_interact<T, false, kCoatNone>(ARGS(T), const T*); \
  template __global__ void p_interact_plain<T>(ARGS(T), const T*); \
  template __global__ void p_pol_basis<T>(ARGS(T), const T*); \
  template __global__ void p_prt<T, true>(ARGS(T), const T*); \
  template __global__ void p_prt<T, false>(ARGS(T), const T*); \
  template __global__ void p_pol_intensity<T, 1>(ARGS(T), PolFields<T>, uint32_t*); \
  template __global__ void p_pol_intensity<T, 2>(ARGS(T), PolFields<T>, uint32_t*);
INST(float)
INST(double)
'''


# What a vector instruction costs on gfx950, in units of a plain fp32 instruction on vector
# registers (tools/microbench/valu_rate.hip, profiles/r06_valu_rate.txt: 8 waves per SIMD)
PRICE = {"trans": 3.0, "pk_fma": 1.82, "pk": 1.6, "fma_f64": 1.87, "f64": 1.57, "cmp": 1.6,
         "cndmask": 1.5, "lane": 1.56, "sgpr_operand": 1.56, "plain": 1.0}


def price_of(s):
    op = s.split()[0]
    if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op):
        return PRICE["trans"]
    if op.startswith("v_pk_fma"):
        return PRICE["pk_fma"]
    if op.startswith("v_pk_"):
        return PRICE["pk"]
    if op.endswith("_f64") or "_f64_" in op:
        return PRICE["fma_f64"] if "fma" in op else PRICE["f64"]
    if op.startswith("v_cmp"):
        return PRICE["cmp"]
    if op.startswith("v_cndmask"):
        return PRICE["cndmask"]
    if re.match(r"v_(readlane|writelane|readfirstlane)", op):
        return PRICE["lane"]
    operands = s.split(None, 1)[1] if " " in s else ""
    if re.search(r"(^|[ ,\-|])s(\d+|\[\d+:\d+\])", operands):
        return PRICE["sgpr_operand"]
    return PRICE["plain"]


def count(lines):
    v = tr = f64 = 0
    price = 0.0
    for ln in lines:
        s = ln.strip()
        if not s.startswith("v_"):
            continue
        op = s.split()[0]
        v += 1
        price += price_of(s.split(";")[0])
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op):
            tr += 1
        if op.endswith("_f64") or "_f64_" in op:
            f64 += 1
    return v, tr, f64, price


def compile_counts(src, defs, name):
    """demangled kernel name (arguments dropped) -> (VALU, transcendental, fp64) of its ISA."""
    d = tempfile.mkdtemp()
    path = os.path.join(d, name + ".hip")
    with open(path, "w") as f:
        f.write(src)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on",
           "-fno-math-errno", "--cuda-device-only", f"-I{CSRC}", *defs, "-S", path, "-o",
           os.path.join(d, name + ".s")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode:
        sys.stderr.write(p.stderr)
        raise SystemExit(1)
    fns, cur = {}, None
    for ln in open(os.path.join(d, name + ".s")):
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = fns.setdefault(m.group(1), [])
            continue
        if ln.startswith(".Lfunc_end"):
            cur = None
        elif cur is not None:
            cur.append(ln)
    names = list(fns)
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True,
                         text=True).stdout.splitlines()
    got = {}
    for n, dn in zip(names, dem):
        dn = re.sub(r"float\s*(__vector\(2\)|vector\[2\]|__attribute__\(\(ext_vector_type\(2\)\)\))",
                    "f32x2", dn)
        dn = re.sub(r"\(.*$", "", dn.replace("void ", ""))
        dn = dn.replace("(ol::", "").replace("ol::", "")
        got[dn] = count(fns[n])
    return got


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    got = compile_counts(SRC, defs, "phases")
    types = (("float", "float"), ("f32x2", "f32x2 (per PAIR of rays)"), ("double", "double"))
    rows = (("frame change (unrotated: + offset)", "k_frame<{}>", "k_empty<{}>"),
            ("conic intersection (curved_distance)", "k_distance<{}>", "k_empty<{}>"),
            ("move to the hit + normal, sphere", "k_hit_normal<{}, true>", "k_empty<{}>"),
            ("move to the hit + normal, conic (k != 0)", "k_hit_normal<{}, false>", "k_empty<{}>"),
            ("interact: OPD + Snell, no aperture", "k_interact<{}, 0>", "k_empty<{}>"),
            ("interact: OPD + radial clip + Snell", "k_interact<{}, 1>", "k_empty<{}>"),
            ("local -> global of a recorded row (unrotated)", "k_to_global<{}>", "k_empty<{}>"),
            ("ray generator (raygen_one, object at infinity)", "k_raygen<{}>", "k_empty<{}>"),
            ("reference-sphere OPD (wavefront_one)", "k_wavefront<{}>", "k_empty<{}>"),
            ("spot accumulation (7 fp64 moments)", "k_spot<{}>", "k_spot_empty<{}>"))
    print("# vector instructions per ray of each phase (static = dynamic: branch-free), gfx950, "
          + (" ".join(defs) or "product knobs"))
    print(f"# {'phase':<48} " + " ".join(f"{t[1]:>26}" for t in types))
    print(f"# {'':<48} " + " ".join(f"{'VALU (transc., fp64) price':>26}" for _ in types))
    per_surface = {t[0]: [0, 0, 0, 0.0] for t in types}
    priced = {}
    for label, kern, base in rows:
        cells = []
        for t, _ in types:
            k, b = got.get(kern.format(t)), got.get(base.format(t))
            if k is None or b is None:
                cells.append(f"{'-':>26}")
                continue
            v, tr, f64, pr = k[0] - b[0], k[1] - b[1], k[2] - b[2], k[3] - b[3]
            cells.append(f"{v:>8d} ({tr:d}, {f64:d}) {pr:6.0f}".rjust(26))
            priced.setdefault(t, {})[label] = (v, pr)
            if label.startswith(("frame", "conic inter", "move to the hit + normal, sphere",
                                 "interact: OPD + radial")):
                for q, x in enumerate((v, tr, f64, pr)):
                    per_surface[t][q] += x
        print(f"  {label:<48} " + " ".join(cells))
    print(f"  {'ONE spherical surface with a radial aperture':<48} " + " ".join(
        f"{per_surface[t][0]:>8d} ({per_surface[t][1]:d}, {per_surface[t][2]:d}) "
        f"{per_surface[t][3]:6.0f}".rjust(26) for t, _ in types))
    print("# last number of a cell = PRICE: issue cost in units of one plain fp32 instruction on "
          "vector registers (1.13 ns per wave-instruction and SIMD at 8 waves, "
          "profiles/r06_valu_rate.txt): " + ", ".join(f"{k} {v}" for k, v in PRICE.items()))
    polarised(defs, priced)


def polarised(defs, priced=None):
    """Round 6: where the vector instructions of the polarised Zernike kernel (configuration C5:
    raygen -> Zernike surface with a Fresnel coating -> spherical surface with a Fresnel coating
    -> image plane, every row recorded, update_intensity epilogue) go.  Static counts of branch-
    free phases; the Newton loop's trip count is the one dynamic quantity (2 evaluations for
    most rays of C5, 3 where a lane of the wave needs them: tools/gpu_nr_iters.py)."""
    got = compile_counts(SRC_POL, defs, "phases_pol")
    types = ("float", "double")
    E, EP = "p_empty<{}>", "p_empty_prt<{}>"
    rows = (
        ("generating prologue (pupil check, ray, apodisation switch)", "p_raygen<{}>", E, "R"),
        ("Newton start: base conic + re-based point", "p_conic_start<{}>", E, "Z"),
        ("ONE Newton iteration, degree-4 one-polynomial Zernike", "p_newton_iter<{}>", E, "I"),
        ("  of which conic base, normalised x y, range check", "p_zern_begin<{}>", E, ""),
        ("  of which the polynomial: sag + gradient (36 coeff.)", "p_zern_mono4<{}>", E, ""),
        ("  of which chain rule + vertex branch", "p_zern_finish<{}>", E, ""),
        ("Newton end: hit point + unit normal", "p_nr_normal<{}>", E, "Z"),
        ("interact, Fresnel, FIRST update of a fresh PRT", "p_interact<{}, true, 2>", EP, "Z"),
        ("interact, Fresnel, rank-2 PRT update", "p_interact<{}, false, 2>", EP, "S"),
        ("interact, uncoated (identity Jones), rank-2 update", "p_interact<{}, false, 0>", EP, ""),
        ("  of which OPD + Snell (the unpolarised interact)", "p_interact_plain<{}>", EP, ""),
        ("  of which the (s, p0, p1) triads", "p_pol_basis<{}>", E, ""),
        ("  of which prt_first_diag", "p_prt<{}, true>", EP, ""),
        ("  of which prt_apply_diag", "p_prt<{}, false>", EP, ""),
        ("update_intensity epilogue, polarised state", "p_pol_intensity<{}, 1>", EP, "P1"),
        ("update_intensity epilogue, unpolarised mean", "p_pol_intensity<{}, 2>", EP, "P2"),
    )
    print()
    print("# round 6: the polarised Zernike kernel (configuration C5), one ray per lane")
    print(f"# {'phase':<58} " + " ".join(f"{t:>26}" for t in types))
    print(f"# {'':<58} " + " ".join(f"{'VALU (transc., fp64) price':>26}" for _ in types))
    cost = {t: {} for t in types}
    pcost = {t: {} for t in types}
    for label, kern, base, tag in rows:
        cells = []
        for t in types:
            k, b = got.get(kern.format(t)), got.get(base.format(t))
            if k is None or b is None:
                cells.append(f"{'-':>26}")
                continue
            v, tr, f64, pr = k[0] - b[0], k[1] - b[1], k[2] - b[2], k[3] - b[3]
            cells.append(f"{v:>8d} ({tr:d}, {f64:d}) {pr:6.0f}".rjust(26))
            if tag:
                cost[t][tag] = cost[t].get(tag, 0) + v
                pcost[t][tag] = pcost[t].get(tag, 0.0) + pr
        print(f"  {label:<58} " + " ".join(cells))
    # the budget of one C5 ray from these rows and the conic rows above (float: 3 / 32 / 13 for
    # frame change / intersection / hit + normal; double: 3 / 64 / 15), for K Newton evaluations
    # per wave -- against SQ_INSTS_VALU per ray of the real kernel (profiles/r0N_kernel_table.txt)
    conic = {"float": (3, 32, 13, 24), "double": (3, 64, 15, 37)}
    for t in types:
        c = cost[t]
        fr, dist, hit, plain = conic[t]
        fixed = (c["R"] + fr + c["Z"] + fr + dist + hit + c["S"] + fr + 10 + plain + 4 * 3)
        for epi, name in (("P1", "polarised state"), ("P2", "unpolarised mean")):
            tot = {k: fixed + k * c["I"] + c["R"] + c[epi] for k in (2, 3, 4)}
            print(f"# {t}: one C5 ray, {name}: prologue {c['R']} + Zernike surface "
                  f"{fr + c['Z']} + K x {c['I']} + spherical surface {fr + dist + hit + c['S']} + "
                  f"image plane {fr + 10 + plain} + 4 rows to global 12 + epilogue "
                  f"{c['R'] + c[epi]} = " + ", ".join(f"{v} (K = {k})" for k, v in tot.items()))
    # the same budget PRICED (K = 2): what the launch costs in issue time, against the measured
    # engine cycles (profiles/r06_cycles.txt: 5.3e5 per launch and XCD at 1e7 rays = 3470 per wave
    # and SIMD slot; one price unit = 1.13 ns x the clock of the valu_rate run, ~2.4 cycles)
    if priced:
        names = {"fr": "frame change (unrotated: + offset)",
                 "dist": "conic intersection (curved_distance)",
                 "hit": "move to the hit + normal, sphere",
                 "plain": "interact: OPD + Snell, no aperture",
                 "glob": "local -> global of a recorded row (unrotated)"}
        for t in types:
            if t not in priced:
                continue
            q = {k: priced[t][v][1] for k, v in names.items()}
            c = pcost[t]
            fixed = (c["R"] + q["fr"] + c["Z"] + q["fr"] + q["dist"] + q["hit"] + c["S"] + q["fr"]
                     + 10 + q["plain"] + 4 * q["glob"])
            tot = fixed + 2 * c["I"] + c["R"] + c["P1"]
            print(f"# {t}: the same ray PRICED (K = 2, polarised state): {tot:.0f} units = "
                  f"{tot * 1.13:.0f} ns of issue per wave and SIMD slot at the valu_rate clock; "
                  f"152.6 waves per SIMD at 1e7 rays: {tot * 1.13 * 152.6 * 1e-6:.3f} ms of vector "
                  "issue per launch (stores, address arithmetic and spill reloads not in the "
                  "phases: +~20 %)")
    return cost


if __name__ == "__main__":
    main()
