// wavefront_fit_device.h -- the reference sphere / plane of a wavefront FITTED to the traced
// bundle (wavefront/strategy.py:287-620: CentroidStrategy, BestFitStrategy), as a chain of
// device reductions that leaves a WavefrontConsts<double> in device memory -- the structure
// ol_wavefront_reference writes for the chief-ray strategy -- so that the OPD map that follows
// needs no read-back.  ONE definition of every per-ray term and of every between-pass decision,
// shared by the kernels (aux_kernels.hip) and the host-math harness.
//
// The reference's steps (torch backend) and the pass that carries each:
//   _points_from_rays (:367-393)   valid = all finite & i != 0;  pts = p - (opd_t / n_image) d
//   _calculate_weights (:395-431)  w = max(i, 0), all ones if their sum is 0            [C1]
//                                  trimming: c0 = sum(w p) / sum(w), dist = |p - c0|,
//                                  mean(dist) [C2], std(dist) (n - 1) [C3],
//                                  keep = dist <= mean + k std, used if sum(keep) >= 4  [C4]
//   _create_reference_geometry     centroid = sum(w p) / sum(w)                        [C1/C4]
//   _create_spherical_ref (:457)   R = sum(w |pts - centroid|) / sum(w)                 [C5]
//   _create_planar_ref (:485)      normal = normalised sum(w d) / sum(w)               [C1/C4]
//   BestFit sphere (:556-582)      least squares [x y z 1] c = |pts|^2                 [B1, B2]
//   BestFit plane (:584-605)       smallest right singular vector of pts - mean        [B1, B2]
//   compute_wavefront_data :331    piston = mean OPD of the rays with i > 0             [M]
// The least-squares problems are solved from their normal equations in CENTRED, per-axis
// SCALED coordinates (the fit is invariant under both; the Gram matrix of such columns has a
// condition number of 1e0 - 1e2 where the reference's raw [x y z 1] has 1e4 - 1e6), by
// Cholesky / Jacobi rotations in the pass's finishing thread.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cmath>

#include "wavefront_device.h"

namespace ol {

// between-pass state, device memory (the head of the workspace)
struct FitState {
  double n_valid, unit_w, sw, c0[3], swd[3];        // C1
  double mean_d, thr, trim_on;                       // C2, C3
  double trimmed, sw_final, cen[3];                  // C4 (or C1 without trimming)
  double m[3], sc[3], isc[3], deficient;             // B1
  uint32_t ticket[kFitPasses];                       // blocks finished, per pass
};
static_assert(sizeof(FitState) <= kFitStateDoubles * sizeof(double), "FitState outgrew its slot");

struct FitRay {
  double x, y, z, L, M, N, opd, i, px, py;
};

OL_DEV bool fit_finite(double v) { return v - v == 0.0; }  // false for NaN and +-inf

// strategy.py:88-139 (_correct_tilt) + :367-393: validity and the wavefront point of one ray
OL_DEV bool fit_point(const FitParams& p, const FitRay& r, double (&pts)[3]) {
  const double opd_t = r.opd + (p.ux * (r.px * p.half_epd) + p.uy * (r.py * p.half_epd));
  const bool valid = fit_finite(r.x) && fit_finite(r.y) && fit_finite(r.z) && fit_finite(r.L) &&
                     fit_finite(r.M) && fit_finite(r.N) && fit_finite(opd_t) && r.i != 0.0;
  const double s = opd_t / p.ni;
  pts[0] = r.x - s * r.L;
  pts[1] = r.y - s * r.M;
  pts[2] = r.z - s * r.N;
  return valid;
}

OL_DEV double fit_norm3(double a, double b, double c) { return ::sqrt(a * a + b * b + c * c); }

// The terms ONE ray adds to the running sums of a pass.  `s` has kFitSums entries.
template <int PASS>
OL_DEV void fit_accumulate(const FitParams& p, const FitState& st,
                           const WavefrontConsts<double>& ref, const FitRay& r, double* s) {
  if constexpr (PASS == kPassMean) {
    // strategy.py:325-340: opd = rays.opd - opd_img over the rays with intensity > 0
    if (r.i > 0.0) {
      double pu[3];
      const double o = wavefront_opd_mm<double, true>(ref, r.x, r.y, r.z, r.L, r.M, r.N, r.opd,
                                                      r.px, r.py, pu);
      s[0] += 1.0;
      if (!(p.skip_nan && o != o)) {  // backend/torch_backend.py:969-989: be.mean drops NaN
        s[1] += o;
        s[2] += 1.0;
      }
    }
    return;
  }
  double pts[3];
  if (!fit_point(p, r, pts)) return;
  const double w = st.unit_w != 0.0 ? 1.0 : (r.i < 0.0 ? 0.0 : r.i);
  if constexpr (PASS == kPassC1) {
    const double wi = r.i < 0.0 ? 0.0 : r.i;
    s[0] += 1.0;
    s[1] += wi;
    s[2] += wi * r.x; s[3] += wi * r.y; s[4] += wi * r.z;
    s[5] += r.x;      s[6] += r.y;      s[7] += r.z;
    s[8] += wi * r.L; s[9] += wi * r.M; s[10] += wi * r.N;
    s[11] += r.L;     s[12] += r.M;     s[13] += r.N;
  } else if constexpr (PASS == kPassC2) {
    s[0] += fit_norm3(r.x - st.c0[0], r.y - st.c0[1], r.z - st.c0[2]);
  } else if constexpr (PASS == kPassC3) {
    const double d = fit_norm3(r.x - st.c0[0], r.y - st.c0[1], r.z - st.c0[2]) - st.mean_d;
    s[0] += d * d;
  } else if constexpr (PASS == kPassC4) {
    if (fit_norm3(r.x - st.c0[0], r.y - st.c0[1], r.z - st.c0[2]) <= st.thr) {
      s[0] += 1.0;
      s[1] += w;
      s[2] += w * r.x; s[3] += w * r.y; s[4] += w * r.z;
      s[5] += w * r.L; s[6] += w * r.M; s[7] += w * r.N;
    }
  } else if constexpr (PASS == kPassC5) {
    const bool keep = st.trimmed == 0.0 ||
                      fit_norm3(r.x - st.c0[0], r.y - st.c0[1], r.z - st.c0[2]) <= st.thr;
    if (keep)
      s[0] += w * fit_norm3(pts[0] - st.cen[0], pts[1] - st.cen[1], pts[2] - st.cen[2]);
  } else if constexpr (PASS == kPassB1) {
    s[0] += 1.0;
    s[1] += pts[0]; s[2] += pts[1]; s[3] += pts[2];
    s[4] += pts[0] * pts[0]; s[5] += pts[1] * pts[1]; s[6] += pts[2] * pts[2];
  } else if constexpr (PASS == kPassB2) {
    const double u0 = pts[0] - st.m[0], u1 = pts[1] - st.m[1], u2 = pts[2] - st.m[2];
    const double q0 = u0 * st.isc[0], q1 = u1 * st.isc[1], q2 = u2 * st.isc[2];
    const double b = u0 * u0 + u1 * u1 + u2 * u2;
    // Gram matrix of [q0 q1 q2 1], upper triangle row by row, then the right-hand side
    s[0] += q0 * q0; s[1] += q0 * q1; s[2] += q0 * q2; s[3] += q0;
    s[4] += q1 * q1; s[5] += q1 * q2; s[6] += q1;
    s[7] += q2 * q2; s[8] += q2;
    s[9] += 1.0;
    s[10] += q0 * b; s[11] += q1 * b; s[12] += q2 * b; s[13] += b;
  }
}

// the reference left for the OPD kernels: everything but opd_ref (the piston pass sets it)
OL_DEV void fit_write_reference(const FitParams& p, WavefrontConsts<double>* out, double xc,
                                double yc, double zc, double R, double nx, double ny, double nz) {
  out->xc = xc; out->yc = yc; out->zc = zc; out->R = R;
  out->ni = p.ni; out->inv_w = p.inv_w; out->ux = p.ux; out->uy = p.uy;
  out->half_epd = p.half_epd; out->opd_ref = 0.0;
  out->nx = nx; out->ny = ny; out->nz = nz; out->planar = p.planar ? 1 : 0;
  out->last_t = 0.0; out->last_absorb = 0.0;   // (the fitted strategies read TRACED bundles)
}

// strategy.py:433-455 once the weights are final: the centroid, and for a planar reference the
// weighted mean direction (:485-517)
OL_DEV void fit_centroid_done(const FitParams& p, FitState& st, const double* swp,
                              const double* swd, double sw, WavefrontConsts<double>* out) {
  st.sw_final = sw;
  for (int k = 0; k < 3; ++k) st.cen[k] = swp[k] / sw;
  if (p.planar) {
    double n[3] = {swd[0] / sw, swd[1] / sw, swd[2] / sw};
    const double len = fit_norm3(n[0], n[1], n[2]);
    if (len > 0.0) { n[0] /= len; n[1] /= len; n[2] /= len; }
    fit_write_reference(p, out, st.cen[0], st.cen[1], st.cen[2], 0.0, n[0], n[1], n[2]);
  }
}

// 4 x 4 symmetric positive definite solve (Cholesky); false if a pivot is not positive -- or,
// round 5, not clearly positive: below 1e-10 of the LARGEST diagonal entry.  In the scaled
// coordinates every diagonal entry is ~n when the per-axis scales are right; a direction the
// points do not span -- a tilted PLANE of wavefront points, or an axis that is flat to
// rounding and was "scaled" by the square root of the cancellation noise of E[z^2] - m^2 --
// leaves a pivot of ~1e-15 n, and what the elimination makes of it is noise.  The caller then
// reports kFitSingular and the host solves the raw system the way the reference does (SVD,
// minimum norm: wavefront.py).
OL_DEV bool fit_solve4(double (&A)[4][4], double (&b)[4]) {
  double dmax = 0.0;
  for (int j = 0; j < 4; ++j)
    if (A[j][j] > dmax) dmax = A[j][j];
  for (int j = 0; j < 4; ++j) {
    double d = A[j][j];
    for (int k = 0; k < j; ++k) d -= A[j][k] * A[j][k];
    if (!(d > 1e-10 * dmax)) return false;
    d = ::sqrt(d);
    A[j][j] = d;
    for (int i = j + 1; i < 4; ++i) {
      double v = A[i][j];
      for (int k = 0; k < j; ++k) v -= A[i][k] * A[j][k];
      A[i][j] = v / d;
    }
  }
  for (int i = 0; i < 4; ++i) {
    double v = b[i];
    for (int k = 0; k < i; ++k) v -= A[i][k] * b[k];
    b[i] = v / A[i][i];
  }
  for (int i = 3; i >= 0; --i) {
    double v = b[i];
    for (int k = i + 1; k < 4; ++k) v -= A[k][i] * b[k];
    b[i] = v / A[i][i];
  }
  return true;
}

// eigenvector of the smallest eigenvalue of a symmetric 3 x 3 matrix (cyclic Jacobi)
OL_DEV void fit_smallest_eigenvector(double (&C)[3][3], double (&v)[3]) {
  double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
  for (int sweep = 0; sweep < 30; ++sweep) {
    const double off = C[0][1] * C[0][1] + C[0][2] * C[0][2] + C[1][2] * C[1][2];
    const double dia = C[0][0] * C[0][0] + C[1][1] * C[1][1] + C[2][2] * C[2][2];
    if (!(off > 1e-60 * dia)) break;
    for (int a = 0; a < 2; ++a)
      for (int b = a + 1; b < 3; ++b) {
        if (C[a][b] == 0.0) continue;
        const double theta = (C[b][b] - C[a][a]) / (2.0 * C[a][b]);
        const double t =
            (theta >= 0.0 ? 1.0 : -1.0) / (::fabs(theta) + ::sqrt(theta * theta + 1.0));
        const double c = 1.0 / ::sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {  // columns a, b of C and V
          const double ca = C[k][a], cb = C[k][b];
          C[k][a] = c * ca - s * cb;
          C[k][b] = s * ca + c * cb;
          const double va = V[k][a], vb = V[k][b];
          V[k][a] = c * va - s * vb;
          V[k][b] = s * va + c * vb;
        }
        for (int k = 0; k < 3; ++k) {  // rows a, b of C
          const double ra = C[a][k], rb = C[b][k];
          C[a][k] = c * ra - s * rb;
          C[b][k] = s * ra + c * rb;
        }
      }
  }
  int lo = 0;
  if (C[1][1] < C[lo][lo]) lo = 1;
  if (C[2][2] < C[lo][lo]) lo = 2;
  for (int k = 0; k < 3; ++k) v[k] = V[k][lo];
}

// What ONE thread does with the complete sums of a pass: the decisions the reference takes on
// the host between its array operations.
template <int PASS>
OL_DEV void fit_finish(const FitParams& p, FitState& st, const double* s,
                       WavefrontConsts<double>* out, uint32_t* status) {
  if constexpr (PASS == kPassC1) {
    st.n_valid = s[0];
    if (s[0] == 0.0) *status |= kFitNoValid;
    const bool unit = s[1] == 0.0;  // strategy.py:411-414
    st.unit_w = unit ? 1.0 : 0.0;
    st.sw = unit ? s[0] : s[1];
    const double* swp = unit ? s + 5 : s + 2;
    const double* swd = unit ? s + 11 : s + 8;
    for (int k = 0; k < 3; ++k) {
      st.c0[k] = swp[k] / st.sw;
      st.swd[k] = swd[k];
    }
    st.trimmed = 0.0;
    st.trim_on = 0.0;
    st.thr = 0.0;
    // without trimming the weights are final here; with it C4 decides
    double swp3[3] = {swp[0], swp[1], swp[2]};
    fit_centroid_done(p, st, swp3, st.swd, st.sw, out);
  } else if constexpr (PASS == kPassC2) {
    st.mean_d = s[0] / st.n_valid;
  } else if constexpr (PASS == kPassC3) {
    const double sd = ::sqrt(s[0] / (st.n_valid - (double)p.ddof));
    st.trim_on = sd > 0.0 ? 1.0 : 0.0;                    // (NaN for one ray: no trimming)
    st.thr = st.mean_d + p.trim_std * sd;
  } else if constexpr (PASS == kPassC4) {
    if (st.trim_on != 0.0 && s[0] >= 4.0) {  // strategy.py:426-429
      st.trimmed = 1.0;
      fit_centroid_done(p, st, s + 2, s + 5, s[1], out);
    }
  } else if constexpr (PASS == kPassC5) {
    fit_write_reference(p, out, st.cen[0], st.cen[1], st.cen[2], s[0] / st.sw_final, 0, 0, 0);
  } else if constexpr (PASS == kPassB1) {
    st.n_valid = s[0];
    if (s[0] == 0.0) *status |= kFitNoValid;
    else if (s[0] < 4.0) *status |= kFitTooFew;
    double var[3], vmax = 0.0;
    st.deficient = 0.0;
    for (int k = 0; k < 3; ++k) {
      st.m[k] = s[1 + k] / s[0];
      var[k] = s[4 + k] / s[0] - st.m[k] * st.m[k];
      if (var[k] > vmax) vmax = var[k];
    }
    for (int k = 0; k < 3; ++k) {
      // Round 5: an axis along which the points do not spread (a collimated beam: the
      // wavefront points of an afocal system lie in ONE plane, z = const to 1e-14) has a
      // "variance" that is the cancellation noise of E[x^2] - m^2, ~eps m^2; scaled by its
      // square root it would look like a fourth dimension and the sphere would be fitted to
      // noise (radius 1e2 ... 1e13 where NumPy's rank-3 minimum-norm solution says 4.7).
      // Such a cloud is reported as singular; the host follows the reference from there.
      if (!p.planar && s[0] >= 4.0 && !(var[k] > 64.0 * 2.220446049250313e-16 *
                                                      (st.m[k] * st.m[k] + vmax)))
        st.deficient = 1.0;
      st.sc[k] = var[k] > 0.0 ? ::sqrt(var[k]) : 1.0;  // a scale: any value > 0 works
      st.isc[k] = 1.0 / st.sc[k];
    }
  } else if constexpr (PASS == kPassB2) {
    if (p.planar) {
      // covariance of the centred points = the scaled Gram block (m IS their mean)
      double C[3][3] = {{s[0], s[1], s[2]}, {s[1], s[4], s[5]}, {s[2], s[5], s[7]}};
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i][j] *= st.sc[i] * st.sc[j];
      double n[3];
      fit_smallest_eigenvector(C, n);
      fit_write_reference(p, out, st.m[0], st.m[1], st.m[2], 0.0, n[0], n[1], n[2]);
    } else {
      // |u|^2 = sum_k g_k q_k + e,  u = pts - m = sc q  ->  centre = m + g / (2 sc),
      // R^2 = e + |centre - m|^2  (strategy.py:576-582 in the centred frame)
      double A[4][4] = {{s[0], s[1], s[2], s[3]}, {s[1], s[4], s[5], s[6]},
                        {s[2], s[5], s[7], s[8]}, {s[3], s[6], s[8], s[9]}};
      double g[4] = {s[10], s[11], s[12], s[13]};
      if (st.deficient != 0.0 || !fit_solve4(A, g)) {
        *status |= kFitSingular;
        g[0] = g[1] = g[2] = g[3] = __builtin_nan("");
      }
      const double a0 = g[0] * st.isc[0] * 0.5, a1 = g[1] * st.isc[1] * 0.5,
                   a2 = g[2] * st.isc[2] * 0.5;
      const double R = ::sqrt(g[3] + a0 * a0 + a1 * a1 + a2 * a2);
      fit_write_reference(p, out, st.m[0] + a0, st.m[1] + a1, st.m[2] + a2, R, 0, 0, 0);
    }
  } else if constexpr (PASS == kPassMean) {
    if (s[0] == 0.0) *status |= kFitNoAlive;
    out->opd_ref = s[1] / s[2];  // (no ray left after dropping NaN: 0 / 0, as the backend)
  }
}

}  // namespace ol
