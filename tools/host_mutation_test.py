#!/usr/bin/env python
"""Mutation test of the KERNEL SOURCE on the CPU: how many injected faults do the tests see?

Each mutant is one small textual change to a copy of optiland_amd/csrc (a flipped sign, an
off-by-one comparison, a dropped term, a swapped argument ...).  The host harness
(tests/hostmath) is built from the mutated copy and the host-math tests are run against it;
a mutant is KILLED when any of them fails.  A surviving mutant is a hole in the tests (or an
equivalent mutant: a change with no observable effect, listed as such).

    python tools/host_mutation_test.py [--only N ...]   ->  profiles/r02_mutation_test.txt

CPU only; ~25 s per mutant (rebuild + tests/test_hostmath.py + the fuzz file).
"""
import argparse
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = "surface_math.h"

# (file under csrc, exact old text, new text, what it breaks)
MUTANTS = [
    (H, "  return -m::div(z, Ns);\n}", "  return m::div(z, Ns);\n}", "flat_distance: sign of the plane distance"),
    (H, "V q = -(E + m::copysign(sq, E));", "V q = -(E - m::copysign(sq, E));", "curved_distance: the cancelling root"),
    (H, "V disc = m::fma(E, E, -A * C);", "V disc = m::fma(E, E, A * C);", "curved_distance: discriminant sign"),
    (H, "V t = m::select(m::le(z1, z2), t1, t2);", "V t = m::select(m::le(z1, z2), t2, t1);", "curved_distance: root selection rule"),
    (H, "t = m::select(m::eq(A, zero), ta, t);", "t = m::select(m::eq(A, zero), tb, t);", "curved_distance: a == 0 branch"),
    (H, "      return (r2 <= ap[1]) && (r2 >= ap[0]);\n    }\n    case kApOffsetRadial", "      return (r2 < ap[1]) && (r2 >= ap[0]);\n    }\n    case kApOffsetRadial", "radial aperture: rim is inside (<=)"),
    (H, "T dx = x - ap[2], dy = y - ap[3];\n      T r2 = m::fma(dx, dx, dy * dy);", "T dx = x + ap[2], dy = y - ap[3];\n      T r2 = m::fma(dx, dx, dy * dy);", "offset radial aperture: offset sign"),
    (H, "return (ap[0] <= x) && (x <= ap[1]) && (ap[2] <= y) && (y <= ap[3]);", "return (ap[0] <= x) && (x <= ap[1]) && (ap[2] <= y) && (y < ap[3]);", "rectangular aperture: upper edge inclusive"),
    (H, "return m::fma(dx * dx, ap[0], dy * dy * ap[1]) <= T(1);", "return m::fma(dx * dx, ap[1], dy * dy * ap[0]) <= T(1);", "elliptical aperture: axes swapped"),
    (H, "V root = m::sqrt(m::fma(m::splat(-u * u), m::fma(-adot[k], adot[k], one), one));", "V root = m::sqrt(m::fma(m::splat(-u), m::fma(-adot[k], adot[k], one), one));", "refract: u instead of u^2 under the root"),
    (H, "V k2 = m::splat(-2) * adot[k];", "V k2 = m::splat(2) * adot[k];", "reflect: sign of the reflected component"),
    (H, "const V sgn = m::select(m::ne(dot, zero), m::copysign(one, dot), zero);", "const V sgn = m::copysign(one, dot);", "align normal: sign(0) = 0 kept"),
    (H, "r[k].opd = r[k].opd + m::abs(t[k] * o.n1);", "r[k].opd = r[k].opd + m::abs(t[k] * o.n2);", "opd: index of the medium the ray came through"),
    (H, "r[k].opd = r[k].opd + m::abs(t[k] * o.n1);", "r[k].opd = r[k].opd + (t[k] * o.n1);", "opd: |t n| (negative distances)"),
    (H, "const V arg = -o.absorb * t[k];", "const V arg = o.absorb * t[k];", "absorption: sign of the exponent"),
    (H, "    const T f = s.interaction == kReflect ? s.cold->coat[1] : s.cold->coat[0];", "    const T f = s.interaction == kReflect ? s.cold->coat[0] : s.cold->coat[1];", "simple coating: R and T swapped"),
    (H, "    if (ck == kCoatSimple) return;", "    if (false) return;", "simple coating leaves the PRT untouched"),
    (H, "            j0 = m::div(T(2) * ci, ci + root);", "            j0 = m::div(T(2) * ci, ci - root);", "Fresnel ts"),
    (H, "            j1 = m::div(T(2) * nn * ci, m::fma(nn * nn, ci, root));", "            j1 = m::div(T(2) * ci, m::fma(nn * nn, ci, root));", "Fresnel tp: factor n"),
    (H, "            j2 = T(-1);", "            j2 = T(1);", "Fresnel reflection: k k^T sign"),
    (H, "  const T bx = m::fma(j2, k1x, -(j0 * k0x)), by = m::fma(j2, k1y, -(j0 * k0y)),\n          bz = m::fma(j2, k1z, -(j0 * k0z));\n#pragma unroll\n  for (int c = 0; c < NP; ++c) {", "  const T bx = m::fma(j2, k1x, (j0 * k0x)), by = m::fma(j2, k1y, -(j0 * k0y)),\n          bz = m::fma(j2, k1z, -(j0 * k0z));\n#pragma unroll\n  for (int c = 0; c < NP; ++c) {", "prt_apply_diag: one component of the rank-2 update"),
    (H, "      P.m[3 * i + e] = m::fma(a[i], p0[e], m::fma(bb[i], k0[e], i == e ? j0 : T(0)));", "      P.m[3 * i + e] = m::fma(a[i], p0[e], m::fma(bb[i], k0[e], i == e ? j1 : T(0)));", "prt_first_diag: diagonal term"),
    (H, "  T sx = k0y * nz - k0z * ny, sy = k0z * nx - k0x * nz, sz = k0x * ny - k0y * nx;", "  T sx = k0y * nz - k0z * ny, sy = k0z * nx + k0x * nz, sz = k0x * ny - k0y * nx;", "pol_basis: s = k0 x n"),
    (H, "  bool done = !(af >= s.cold->tol);                       // converged, or NaN", "  bool done = !(af >= s.cold->tol * T(1e6));                       // converged, or NaN", "Newton: convergence tolerance 1e6 x looser"),
    (H, "  T df = m::fma(fx, L, m::fma(fy, M, -N));", "  T df = m::fma(fx, L, m::fma(fy, M, N));", "Newton: f' = fx L + fy M - N"),
    (H, "      const T im = m::rsqrt(m::fma(q[k].gx, q[k].gx, m::fma(q[k].gy, q[k].gy, T(1))));\n      nx[k] = q[k].gx * im;", "      const T im = m::rsqrt(m::fma(q[k].gx, q[k].gx, m::fma(q[k].gy, q[k].gy, T(1))));\n      nx[k] = -q[k].gx * im;", "Newton surfaces: normal from the sag gradient"),
    (H, "    P = i == N ? q : m::fma(P, xn, q);", "    P = i == N ? q : m::fma(P, yn, q);", "Zernike one-polynomial form: outer Horner variable"),
    (H, "      q = q * yy + cj;", "      q = q * xx + cj;", "Zernike one-polynomial form: gradient inner variable"),
    (H, "    const T t1s = (t1.x + t1.y) * T(2);", "    const T t1s = (t1.x + t1.y);", "Zernike level form: dQ/dx = 2 x Q'"),
    (H, "    const V2 w = qn * T(mg);", "    const V2 w = qn * T(mg + 1);", "Zernike level form: harmonic derivative factor m"),
    (H, "  if (u < T(1e-8)) {  // the reference's eps-regularised chain rule near / at the vertex", "  if (false) {  // the reference's eps-regularised chain rule near / at the vertex", "Zernike: vertex regularisation"),
    (H, "  if (m::abs(xn) > T(1) || m::abs(yn) > T(1)) status |= 0x1u;  // OL_STATUS_ZERNIKE_RANGE", "  if (m::abs(xn) > T(2) || m::abs(yn) > T(2)) status |= 0x1u;  // OL_STATUS_ZERNIKE_RANGE", "Zernike: range check threshold"),
    (H, "    g.x = R[0] * r.x + R[3] * r.y + R[6] * r.z;\n    g.y = R[1] * r.x + R[4] * r.y + R[7] * r.z;", "    g.x = R[0] * r.x + R[1] * r.y + R[2] * r.z;\n    g.y = R[1] * r.x + R[4] * r.y + R[7] * r.z;", "to_global: inverse rotation = transpose"),
    (H, "        r[k].z -= oz;\n      }\n    }\n  } else if", "        r[k].z += oz;\n      }\n    }\n  } else if", "into_local_frame: translation sign (from global)"),
    ("raygen_device.h", "    x0 = px * c.EPD / T(2) * vx + (-tx * c.off_epl);", "    x0 = px * c.EPD / T(2) * vx + (tx * c.off_epl);", "ray generation: field offset sign at infinity"),
    ("raygen_device.h", "  const bool is_zero = mag < T(1e-9);  // paraxial.py:96-104", "  const bool is_zero = mag < T(1e9);  // paraxial.py:96-104", "ray generation: zero-length direction guard"),
    ("wavefront_device.h", "    t = t1 < T(0) ? t2 : t1;", "    t = t1 < T(0) ? t1 : t2;", "wavefront: root choice on the reference sphere"),
    ("epilogue_device.h", "  return acc * i0 / T(f.nf);", "  return acc * i0;", "update_intensity: mean over the two unpolarised states"),
    ("epilogue_device.h", "  if (v == e[nb]) return nb - 1;", "  if (v == e[nb]) return -1;", "histogram: right-most edge folded into the last bin"),
    ("epilogue_device.h", "    if (r <= steps[mid]) hi = mid; else lo = mid + 1;", "    if (r < steps[mid]) hi = mid; else lo = mid + 1;", "encircled energy: radii <= r"),
    ("capi.hip", "      out.push_back((i + 1) * Q[(i + 1) * W + j]);  // dQ/dx", "      out.push_back((i + 2) * Q[(i + 1) * W + j]);  // dQ/dx", "host: monomial derivative coefficients"),
    ("capi.hip", "          const double sign = ((h / 2) & 1) ? -1.0 : 1.0;  // i^h: +1, (+i), -1, (-i), ...", "          const double sign = 1.0;  // i^h: +1, (+i), -1, (-i), ...", "host: harmonic expansion signs"),
    # second batch: the other sag functors, polygon / boolean apertures, polarizer / retarder,
    # the ray generator's other branches, the host's table conversion
    (H, "    dp = m::fma(dp, r2, T(i + 1) * ci);\n    p = m::fma(p, r2, ci);\n  }\n  sag = m::fma(p, r2, sag);\n  f = m::fma(T(2), dp, f);", "    dp = m::fma(dp, r2, T(i + 1) * ci);\n    p = m::fma(p, r2, ci);\n  }\n  sag = m::fma(p, r2, sag);\n  f = m::fma(T(1), dp, f);", "even asphere: d(r^2)/dx = 2 x"),
    (H, "    if (i >= 1) q = m::fma(q, r, T(i + 1) * ci);", "    if (i >= 1) q = m::fma(q, r, T(i) * ci);", "odd asphere: derivative factor (i + 1)"),
    (H, "  T t0 = r > T(0) ? m::div(c0, r) : T(0);  // i = 0 term: x C_0 / r, 0 at r == 0", "  T t0 = T(0);  // i = 0 term: x C_0 / r, 0 at r == 0", "odd asphere: the linear term's gradient"),
    (H, "      dqi = m::fma(dqi, y, qi);\n      qi = m::fma(qi, y, cij);", "      dqi = m::fma(dqi, y, cij);\n      qi = m::fma(qi, y, cij);", "XY polynomial: d/dy by Horner"),
    (H, "      dq = m::fma(cij * T(j), Uj1, dq);", "      dq = m::fma(cij, Uj1, dq);", "Chebyshev: T_j' = j U_(j-1)"),
    (H, "    zy = m::div(cy * y * y, T(1) + m::sqrt(st0));", "    zy = m::div(cy * y * y, T(1) - m::sqrt(st0));", "biconic: y profile"),
    (H, "  const T d = R - zy;", "  const T d = R + zy;", "toroidal: rotation about the axis at R"),
    (H, "    if (yflag0 != yflag1 && (((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == yflag1))", "    if (yflag0 != yflag1 && (((y1 - ty) * (x0 - x1) > (x1 - tx) * (y0 - y1)) == yflag1))", "polygon: edge crossing with >="),
    (H, "const uint32_t v = op == kApOpUnion ? (a | b) : (op == kApOpIntersection ? (a & b) : (a & ~b & 1u));", "const uint32_t v = op == kApOpUnion ? (a | b) : (op == kApOpIntersection ? (a & b) : (b & ~a & 1u));", "boolean apertures: difference a \\ b operand order"),
    (H, "    J.a00 = uso * usi; J.a01 = uso * upi; J.a10 = upo * usi; J.a11 = upo * upi;", "    J.a00 = uso * usi; J.a01 = upo * usi; J.a10 = uso * upi; J.a11 = upo * upi;", "polarizer Jones: u_out u_in^T transposed"),
    (H, "    J.b01 = J.b10 = T(-2) * rs * q01;", "    J.b01 = J.b10 = T(2) * rs * q01;", "retarder Jones: off-diagonal sign"),
    (H, "    T px = T(0), py = k0z, pz = -k0y;\n    if (py == T(0) && pz == T(0)) {", "    T px = T(0), py = -k0z, pz = k0y;\n    if (py == T(0) && pz == T(0)) {", "pol_basis: normal-incidence fallback axis  [EQUIVALENT: s and p flip together; s s^T, p1 p0^T, and the axis projections of polarizer / retarder are even in that sign]"),
    (H, "      r[k].i = m::select(aperture_mask<V, FULL>(s, coeffs, r[k].x, r[k].y), r[k].i, zero);", "      r[k].i = m::select(aperture_mask<V, FULL>(s, coeffs, r[k].x, r[k].y), r[k].i, r[k].i);", "clip: outside rays lose their intensity"),
    ("raygen_device.h", "    x1 = px * vx + x0;\n    y1 = py * vy + y0;\n    z1 = c.tele_dz + z0;", "    x1 = px * vx - x0;\n    y1 = py * vy + y0;\n    z1 = c.tele_dz + z0;", "ray generation: telecentric target point"),
    ("raygen_device.h", "    case 1: return exp(-r2 / (T(2) * a * a));                                  // gaussian.py", "    case 1: return exp(-r2 / (a * a));                                  // gaussian.py", "apodization: Gaussian width"),
    ("raygen_device.h", "    x0 = -tx * c.epl_z;", "    x0 = tx * c.epl_z;", "ray generation: finite-object angle field origin"),
    ("capi.hip", "      out[0] = a[0] * a[0];\n      out[1] = a[1] * a[1];", "      out[0] = a[0];\n      out[1] = a[1] * a[1];", "host: radial aperture r_min squared"),
    ("capi.hip", "      out[0] = 1.0 / (a[0] * a[0]);", "      out[0] = 1.0 / (a[0]);", "host: elliptical aperture semi-axis squared"),
]

TESTS = ["tests/test_hostmath.py", "tests/test_hostmath_fuzz.py"]

# round 4: the reduction passes of ol_wavefront_fit (wavefront_fit_device.h) and the other
# arithmetic touched this round; killed or not by tests/test_wavefront_fit.py + the fused-generate
# tests (`--set r04` -> profiles/r04_mutation_test.txt)
F = "wavefront_fit_device.h"
MUTANTS_R04 = [
    (F, "fit_finite(r.N) && fit_finite(opd_t) && r.i != 0.0;", "fit_finite(r.N) && fit_finite(opd_t);", "validity: unlit rays (i == 0) are not samples"),
    (F, "OL_DEV bool fit_finite(double v) { return v - v == 0.0; }", "OL_DEV bool fit_finite(double v) { return v == v; }", "validity: infinities are not finite"),
    (F, "  const double s = opd_t / p.ni;\n  pts[0] = r.x - s * r.L;", "  const double s = opd_t / p.ni;\n  pts[0] = r.x + s * r.L;", "wavefront point: back along the ray"),
    (F, "    const double wi = r.i < 0.0 ? 0.0 : r.i;\n    s[0] += 1.0;", "    const double wi = r.i;\n    s[0] += 1.0;", "weights: negative intensities clamp to 0"),
    (F, "    const bool unit = s[1] == 0.0;  // strategy.py:411-414", "    const bool unit = false;  // strategy.py:411-414", "weights: all ones when they sum to 0"),
    (F, "    const double sd = ::sqrt(s[0] / (st.n_valid - (double)p.ddof));", "    const double sd = ::sqrt(s[0] / (st.n_valid));", "trimming: Bessel's correction of the torch flavour"),
    (F, "    st.thr = st.mean_d + p.trim_std * sd;", "    st.thr = st.mean_d + sd;", "trimming: k sigma"),
    (F, "    if (st.trim_on != 0.0 && s[0] >= 4.0) {  // strategy.py:426-429", "    if (st.trim_on != 0.0 && s[0] >= 400.0) {  // strategy.py:426-429", "trimming: used when at least 4 rays are kept"),
    (F, "    if (fit_norm3(r.x - st.c0[0], r.y - st.c0[1], r.z - st.c0[2]) <= st.thr) {", "    if (fit_norm3(r.x - st.c0[0], r.y - st.c0[1], r.z - st.c0[2]) >= st.thr) {", "trimming: keep the NEAR rays"),
    (F, "      s[0] += w * fit_norm3(pts[0] - st.cen[0], pts[1] - st.cen[1], pts[2] - st.cen[2]);", "      s[0] += fit_norm3(pts[0] - st.cen[0], pts[1] - st.cen[1], pts[2] - st.cen[2]);", "centroid sphere: WEIGHTED mean distance"),
    (F, "    if (len > 0.0) { n[0] /= len; n[1] /= len; n[2] /= len; }", "    if (false) { n[0] /= len; n[1] /= len; n[2] /= len; }", "centroid plane: unit normal"),
    (F, "    else if (s[0] < 4.0) *status |= kFitTooFew;", "    else if (s[0] < 3.0) *status |= kFitTooFew;", "best fit: at least 4 samples"),
    (F, "      st.sc[k] = var > 0.0 ? ::sqrt(var) : 1.0;  // a scale, not a statistic: any value > 0 works", "      st.sc[k] = var > 0.0 ? 7.0 * ::sqrt(var) : 3.0;  // a scale, not a statistic: any value > 0 works", "best fit: the scale of the normal equations  [EQUIVALENT by design: the fit is invariant under it]"),
    (F, "    const double b = u0 * u0 + u1 * u1 + u2 * u2;", "    const double b = u0 * u0 + u1 * u1 - u2 * u2;", "best-fit sphere: right-hand side |u|^2"),
    (F, "      const double R = ::sqrt(g[3] + a0 * a0 + a1 * a1 + a2 * a2);", "      const double R = ::sqrt(g[3] + a0 * a0 + a1 * a1);", "best-fit sphere: R^2 = e + |a|^2"),
    (F, "      const double a0 = g[0] * st.isc[0] * 0.5,", "      const double a0 = g[0] * st.isc[0],", "best-fit sphere: centre = g / (2 s)"),
    (F, "  if (C[1][1] < C[lo][lo]) lo = 1;", "  if (C[1][1] > C[lo][lo]) lo = 1;", "best-fit plane: the SMALLEST eigenvalue"),
    (F, "      if (!(p.skip_nan && o != o)) {  // backend/torch_backend.py:969-989: be.mean drops NaN", "      if (true) {  // backend/torch_backend.py:969-989: be.mean drops NaN", "piston: the torch flavour's mean ignores NaN"),
    (F, "    if (r.i > 0.0) {\n      double pu[3];", "    if (r.i != 0.0) {\n      double pu[3];", "piston: over the rays with i > 0"),
    (F, "    out->opd_ref = s[1] / s[2];", "    out->opd_ref = s[1] / (s[2] + 1.0);", "piston: the mean"),
    ("wavefront_device.h", "  const T opd = TILT_FIRST ? (opd_in + tilt) - opd_img : opd_in - opd_img + tilt;\n  const T tt = m::div(opd_img, w.ni);", "  const T opd = TILT_FIRST ? (opd_in - tilt) - opd_img : opd_in - opd_img + tilt;\n  const T tt = m::div(opd_img, w.ni);", "fitted OPD map: launch-plane tilt sign"),
    ("raygen_device.h", "m::rsqrt(m2)", "m::rsqrt(m2 + T(1e-3))", "ray generator: normalisation by the reciprocal square root"),
]
TESTS_R04 = ["tests/test_wavefront_fit.py", "tests/test_generate_fused.py", "tests/test_hostmath.py"]


def run(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, text=True, **kw)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", type=int, nargs="*")
    ap.add_argument("--out", default=None)
    ap.add_argument("--set", default="r02", choices=("r02", "r04"))
    args = ap.parse_args()
    global MUTANTS, TESTS
    if args.set == "r04":
        MUTANTS, TESTS = [(f, o.replace("\\n", "\n"), n.replace("\\n", "\n"), w)
                          for f, o, n, w in MUTANTS_R04], TESTS_R04
    if args.out is None:
        args.out = os.path.join(ROOT, "profiles", f"{args.set}_mutation_test.txt")
    sys.path.insert(0, ROOT)
    import importlib.util
    spec = importlib.util.spec_from_file_location("hb", os.path.join(ROOT, "tests", "hostmath", "build.py"))
    hb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(hb)
    flags = ["--offload-host-only", "-O1", "-std=c++17", "-fPIC", "-ffp-contract=on",
             "-fno-math-errno"] + (["-mfma"] if hb._cpu_has_fma() else [])
    rows = []
    todo = range(len(MUTANTS)) if not args.only else args.only
    for idx in todo:
        fname, old, new, what = MUTANTS[idx]
        t0 = time.time()
        with tempfile.TemporaryDirectory() as tmp:
            # the harness includes "../../optiland_amd/csrc/..." and "../../include/...": same tree
            shutil.copytree(os.path.join(ROOT, "optiland_amd", "csrc"), os.path.join(tmp, "optiland_amd", "csrc"))
            shutil.copytree(os.path.join(ROOT, "include"), os.path.join(tmp, "include"))
            os.makedirs(os.path.join(tmp, "tests", "hostmath"))
            shutil.copy(os.path.join(ROOT, "tests", "hostmath", "harness.hip"), os.path.join(tmp, "tests", "hostmath"))
            path = os.path.join(tmp, "optiland_amd", "csrc", fname)
            src = open(path).read()
            if src.count(old) < 1:
                rows.append((idx, what, "STALE (pattern not found)", 0.0))
                continue
            open(path, "w").write(src.replace(old, new, 1))
            objs = []
            ok = True
            for s in (os.path.join(tmp, "optiland_amd", "csrc", "capi.hip"),
                      os.path.join(tmp, "tests", "hostmath", "harness.hip")):
                o = s + ".o"
                r = run(["/opt/rocm/bin/hipcc", *flags, "-c", s, "-o", o])
                if r.returncode != 0:
                    ok = False
                    break
                objs.append(o)
            if not ok:
                rows.append((idx, what, "DOES NOT COMPILE", time.time() - t0))
                continue
            lib = os.path.join(tmp, "libmut.so")
            subprocess.check_call(["g++", "-shared", "-Wl,-Bsymbolic", *objs, "-o", lib])
            env = dict(os.environ, OL_HOSTMATH_LIBRARY=lib, PYTHONPATH=ROOT)
            r = run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider",
                     "-m", "not gpu", *TESTS],
                    cwd=ROOT, env=env, timeout=900)
            if r.returncode == 0:
                verdict = "SURVIVED"
            else:
                first = [ln for ln in r.stdout.splitlines() if ln.startswith("FAILED")]
                verdict = "killed by " + (first[0].split("::", 1)[1][:70] if first else "a crash")
            rows.append((idx, what, verdict, time.time() - t0))
            print(f"[{idx:2d}] {what:60s} {verdict}  ({time.time() - t0:.0f} s)", flush=True)
    killed = sum(1 for r in rows if r[2].startswith("killed"))
    equivalent = sum(1 for r in rows if "[EQUIVALENT" in r[1])
    with open(args.out, "w") as f:
        f.write(f"# {args.set}: mutation test of the kernel source on the CPU (tools/host_mutation_test.py --set {args.set}): one\n"
                "# small fault per mutant in a copy of optiland_amd/csrc, host harness rebuilt from it,\n"
                f"# {' + '.join(TESTS)} run against it (-x: the first failing test is named).\n"
                f"# {killed} of {len(rows)} mutants killed ({equivalent} marked equivalent: no observable effect).\n")
        for idx, what, verdict, dt in rows:
            f.write(f"{idx:3d}  {what[:110]:62s} {verdict}\n")
    print(open(args.out).read())


if __name__ == "__main__":
    main()
