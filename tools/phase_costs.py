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


def count(lines):
    v = tr = f64 = 0
    for ln in lines:
        s = ln.strip()
        if not s.startswith("v_"):
            continue
        op = s.split()[0]
        v += 1
        if re.match(r"v_(rcp|rsq|sqrt|exp|log|sin|cos)", op):
            tr += 1
        if op.endswith("_f64") or "_f64_" in op:
            f64 += 1
    return v, tr, f64


def main():
    defs = [a for a in sys.argv[1:] if a.startswith("-D")]
    d = tempfile.mkdtemp()
    path = os.path.join(d, "phases.hip")
    with open(path, "w") as f:
        f.write(SRC)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=on",
           "-fno-math-errno", "--cuda-device-only", f"-I{CSRC}", *defs, "-S", path, "-o",
           os.path.join(d, "phases.s")]
    p = subprocess.run(cmd, capture_output=True, text=True)
    if p.returncode:
        sys.stderr.write(p.stderr)
        raise SystemExit(1)
    fns, cur = {}, None
    for ln in open(os.path.join(d, "phases.s")):
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
        got[dn] = count(fns[n])
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
    print(f"# {'':<48} " + " ".join(f"{'VALU (transc., fp64)':>26}" for _ in types))
    per_surface = {t[0]: [0, 0, 0] for t in types}
    for label, kern, base in rows:
        cells = []
        for t, _ in types:
            k, b = got.get(kern.format(t)), got.get(base.format(t))
            if k is None or b is None:
                cells.append(f"{'-':>26}")
                continue
            v, tr, f64 = k[0] - b[0], k[1] - b[1], k[2] - b[2]
            cells.append(f"{v:>12d} ({tr:d}, {f64:d})".rjust(26))
            if label.startswith(("frame", "conic inter", "move to the hit + normal, sphere",
                                 "interact: OPD + radial")):
                for q, x in enumerate((v, tr, f64)):
                    per_surface[t][q] += x
        print(f"  {label:<48} " + " ".join(cells))
    print(f"  {'ONE spherical surface with a radial aperture':<48} " + " ".join(
        f"{per_surface[t][0]:>12d} ({per_surface[t][1]:d}, {per_surface[t][2]:d})".rjust(26)
        for t, _ in types))
    print("# issue cycles per wave: full-rate VALU 4, transcendental (v_rcp / v_rsq / v_sqrt) 16;"
          " fp64 FMA / MUL / ADD issue at half the fp32 rate on this part")


if __name__ == "__main__":
    main()
