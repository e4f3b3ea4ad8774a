/*
 * trace_oracle.c -- CPU restatement (fp64, plain C99) of Optiland's batched
 * sequential real-ray trace.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this file's shared object; the product path (optiland_amd/) never
 * imports, links or executes anything under oracle/.
 *
 * Parity status: PINNED.  tests/test_oracle_*.py check this restatement against
 *  (1) the known-answer scalars hard-coded in the reference's own unit tests
 *      (tests/test_geometries.py, tests/test_rays.py, tests/test_coatings.py),
 *  (2) golden traces generated in the build container by importing the
 *      reference itself with its NumPy backend (tools/make_golden.py ->
 *      tests/golden/*.npz: 65 systems incl. every lens of optiland.samples),
 *  (3) the Zemax OpticStudio ray data and the ray-generator values the reference's
 *      tests hard-code (tests/test_external_known_answers.py),
 *  (4) the installed matplotlib for the point-in-polygon test the reference
 *      delegates to it (tests/test_oracle_polygon.py).
 *
 * The structure mirrors the reference: an outer loop over surfaces and, per
 * surface, whole-batch passes -- so batch-global decisions (the Newton-Raphson
 * stop rule, `any(k > 0)`) are reproduced exactly.  Every function cites the
 * reference lines it follows (paths relative to /root/reference/optiland).
 */
#include <complex.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/optiland_hip.h"

/* plane indices in rays[] / record rows */
enum { PX = 0, PY, PZ, PL, PM, PN, PI_, POPD };

typedef struct {
  int64_t n;
  double *x, *y, *z, *L, *M, *N, *i, *opd;
  double *L0, *M0, *N0;       /* pre-interaction cosines (real_rays.py:170-172) */
  double complex* p;          /* n x 3 x 3 PRT matrices or NULL                 */
} rays_t;

/* ---- coordinate_system.py:73-89 (localize) / 91-107 (globalize) -----------
 * The packed descriptor carries the composed rotation R and origin; see
 * optiland_amd/packer.py:cs_to_affine for the composition.                   */
static void localize(const ol_surface_desc* s, rays_t* r) {
  const double* R = s->rot;
  const int rotated = (s->flags & OL_SURF_ROTATED) != 0;
  for (int64_t j = 0; j < r->n; ++j) {
    double x = r->x[j] - s->origin[0]; /* rays/base.py:51-65 translate */
    double y = r->y[j] - s->origin[1];
    double z = r->z[j] - s->origin[2];
    if (rotated) {
      double L = r->L[j], M = r->M[j], N = r->N[j];
      r->x[j] = R[0] * x + R[1] * y + R[2] * z;
      r->y[j] = R[3] * x + R[4] * y + R[5] * z;
      r->z[j] = R[6] * x + R[7] * y + R[8] * z;
      r->L[j] = R[0] * L + R[1] * M + R[2] * N;
      r->M[j] = R[3] * L + R[4] * M + R[5] * N;
      r->N[j] = R[6] * L + R[7] * M + R[8] * N;
    } else {
      r->x[j] = x; r->y[j] = y; r->z[j] = z;
    }
  }
}

static void globalize(const ol_surface_desc* s, rays_t* r) {
  const double* R = s->rot;
  const int rotated = (s->flags & OL_SURF_ROTATED) != 0;
  for (int64_t j = 0; j < r->n; ++j) {
    double x = r->x[j], y = r->y[j], z = r->z[j];
    if (rotated) { /* inverse rotation = transpose */
      double L = r->L[j], M = r->M[j], N = r->N[j];
      double gx = R[0] * x + R[3] * y + R[6] * z;
      double gy = R[1] * x + R[4] * y + R[7] * z;
      double gz = R[2] * x + R[5] * y + R[8] * z;
      x = gx; y = gy; z = gz;
      r->L[j] = R[0] * L + R[3] * M + R[6] * N;
      r->M[j] = R[1] * L + R[4] * M + R[7] * N;
      r->N[j] = R[2] * L + R[5] * M + R[8] * N;
    }
    r->x[j] = x + s->origin[0];
    r->y[j] = y + s->origin[1];
    r->z[j] = z + s->origin[2];
  }
}

/* ---- geometries/standard.py:81-95 (sag of the base conic) ----------------- */
static double conic_sag(double R, double k, double x, double y) {
  double r2 = x * x + y * y;
  return r2 / (R * (1.0 + sqrt(1.0 - (1.0 + k) * r2 / (R * R))));
}

/* ---- zernike/base.py:216-239 (_radial_term) / 260-299 (_radial_derivative) */
static double factorial_d(int n) {
  double f = 1.0;
  for (int i = 2; i <= n; ++i) f *= (double)i;
  return f;
}

static double zern_radial(int n, int m, double r) {
  int ma = m < 0 ? -m : m;
  int s_max = (n - ma) / 2 + 1;
  double value = 0.0;
  for (int k = 0; k < s_max; ++k) {
    double num = factorial_d(n - k);
    double den = factorial_d(k) * factorial_d((n + ma) / 2 - k) *
                 factorial_d((n - ma) / 2 - k);
    double coeff = ((k & 1) ? -1.0 : 1.0) * num / den;
    value += coeff * pow(r, (double)(n - 2 * k));
  }
  return value;
}

static double zern_radial_deriv(int n, int m, double r) { /* m >= 0 here */
  int s_max = (n - m) / 2 + 1;
  double value = 0.0;
  for (int k = 0; k < s_max; ++k) {
    double num = factorial_d(n - k);
    double den = factorial_d(k) * factorial_d((n + m) / 2 - k) *
                 factorial_d((n - m) / 2 - k);
    int factor = n - 2 * k;
    if (factor < 0) continue;
    double power_term = (n - 2 * k - 1) >= 0 ? pow(r, (double)(n - 2 * k - 1)) : 0.0;
    value += ((k & 1) ? -1.0 : 1.0) * (num / den) * (double)factor * power_term;
  }
  return value;
}

/* ---- toroidal.py:86-160: Y-Z profile and its derivative -------------------- */
static double tor_zy(const ol_surface_desc* s, const double* c, double y) {
  double y2 = y * y, z_y = 0.0;
  double R_yz = s->radius, k = c[1];
  if (isfinite(R_yz) && R_yz != 0.0) {
    double cv = 1.0 / R_yz;
    double root_val = 1.0 - (1.0 + k) * cv * cv * y2;
    double root = root_val < 0 ? 0.0 : root_val;
    double denom = 1.0 + sqrt(root);
    double safe = fabs(denom) < 1e-14 ? 1e-14 : denom;
    z_y = (cv * y2) / safe;
  }
  if (s->n_coeff > 2) {
    double poly = 0.0, pw = y2;
    for (int i = 2; i < s->n_coeff; ++i) { poly = poly + c[i] * pw; pw = pw * y2; }
    z_y = z_y + poly;
  }
  return z_y;
}

static double tor_dzy(const ol_surface_desc* s, const double* c, double y) {
  double y2 = y * y, d = 0.0;
  double R_yz = s->radius, k = c[1];
  if (isfinite(R_yz) && R_yz != 0.0) {
    double cv = 1.0 / R_yz;
    double root_val = 1.0 - (1.0 + k) * cv * cv * y2;
    double root = root_val < 1e-14 ? 1e-14 : root_val;
    double sq = sqrt(root);
    double safe = fabs(sq) < 1e-14 ? 1e-14 : sq;
    d = (cv * y) / safe;
  }
  if (s->n_coeff > 2) {
    double poly = 0.0, pw = y;
    for (int i = 2; i < s->n_coeff; ++i) {
      poly = poly + c[i] * (2.0 * ((i - 2) + 1.0)) * pw;
      pw = pw * y2;
    }
    d = d + poly;
  }
  return d;
}

/* chebyshev.py:197-225 */
static double cheb_T(int n, double x) { return cos(n * acos(x)); }
static double cheb_dT(int n, double x) { return n * sin(n * acos(x)) / sqrt(1 - x * x); }

/* ---- sag(x, y) per geometry -------------------------------------------------
 * even_asphere.py:93-109, odd_asphere.py:86-104, polynomial.py:105-126,
 * zernike.py:153-180 (+ zernike/base.py:42-98 get_term/poly)                 */
static double geom_sag(const ol_surface_desc* s, const double* coeffs, double x,
                       double y, uint32_t* status) {
  const double* c = coeffs + s->coeff_offset;
  double r2 = x * x + y * y;
  double z = conic_sag(s->radius, s->conic, x, y);
  switch (s->geom_kind) {
    case OL_GEOM_EVEN_ASPHERE:
      for (int i = 0; i < s->n_coeff; ++i) z = z + c[i] * pow(r2, (double)(i + 1));
      return z;
    case OL_GEOM_ODD_ASPHERE: {
      double r = sqrt(r2);
      for (int i = 0; i < s->n_coeff; ++i) z = z + c[i] * pow(r, (double)(i + 1));
      return z;
    }
    case OL_GEOM_POLYNOMIAL: {
      int cols = s->poly_cols, rows = cols ? s->n_coeff / cols : 0;
      for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j)
          z = z + c[i * cols + j] * pow(x, (double)i) * pow(y, (double)j);
      return z;
    }
    case OL_GEOM_CHEBYSHEV: { /* chebyshev.py:126-152 */
      double xn = x / c[0], yn = y / c[1];
      if (fabs(xn) > 1.0 || fabs(yn) > 1.0) *status |= OL_STATUS_CHEBYSHEV_RANGE;
      int cols = s->poly_cols, rows = cols ? s->n_coeff / cols : 0;
      for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j) {
          double cij = c[2 + i * cols + j];
          if (cij != 0.0) z = z + cij * cheb_T(i, xn) * cheb_T(j, yn);
        }
      return z;
    }
    case OL_GEOM_BICONIC: { /* biconic.py:69-103 (radius/conic = Rx/kx, c = {Ry, ky}) */
      double Rx = s->radius, kx = s->conic, Ry = c[0], ky = c[1];
      double cx = (isinf(Rx) || Rx == 0) ? 0.0 : 1.0 / Rx;
      double cy = (isinf(Ry) || Ry == 0) ? 0.0 : 1.0 / Ry;
      double zx = 0.0, zy = 0.0;
      if (cx != 0.0) {
        double v = 1.0 - (1.0 + kx) * cx * cx * x * x;
        double st = v < 1e-14 ? 0.0 : v;
        double den = 1.0 + sqrt(st);
        double sd = fabs(den) < 1e-14 ? 1e-14 : den;
        zx = (cx * x * x) / sd;
      }
      if (cy != 0.0) {
        double v = 1.0 - (1.0 + ky) * cy * cy * y * y;
        double st = v < 1e-14 ? 0.0 : v;
        double den = 1.0 + sqrt(st);
        double sd = fabs(den) < 1e-14 ? 1e-14 : den;
        zy = (cy * y * y) / sd;
      }
      return zx + zy;
    }
    case OL_GEOM_TOROIDAL: { /* toroidal.py:162-190 */
      double R = c[0];
      double z_y = tor_zy(s, c, y);
      if (isinf(R)) return z_y;
      double term = (R - z_y) * (R - z_y) - x * x;
      if (term < 0) return NAN;
      double d = R - z_y;
      double sg = (d > 0) - (d < 0);
      return z_y + (d - sg * sqrt(term));
    }
    case OL_GEOM_ZERNIKE: {
      double xn = x / s->norm_radius, yn = y / s->norm_radius;
      if (fabs(xn) > 1.0 || fabs(yn) > 1.0) *status |= OL_STATUS_ZERNIKE_RANGE;
      double rho = sqrt(xn * xn + yn * yn);
      double phi = atan2(yn, xn);
      double sum = 0.0;
      for (int j = 0; j < s->n_coeff; ++j) {
        double cj = c[4 * j];
        int n = (int)c[4 * j + 1], m = (int)c[4 * j + 2];
        double Nj = c[4 * j + 3];
        double az = m >= 0 ? cos(m * phi) : sin(-m * phi); /* base.py:241-258 */
        sum += cj * Nj * zern_radial(n, m, rho) * az;
      }
      return z + sum;
    }
    default:
      return z;
  }
}

/* ---- _surface_normal(x, y) per Newton-Raphson geometry ----------------------
 * even_asphere.py:111-140, odd_asphere.py:106-143, polynomial.py:128-155,
 * zernike.py:182-252                                                          */
static __thread uint32_t g_normal_status; /* chebyshev validates inside _surface_normal too */

static void geom_normal_nr(const ol_surface_desc* s, const double* coeffs, double x,
                           double y, int all_rho_zero, double* nx, double* ny,
                           double* nz) {
  const double* c = coeffs + s->coeff_offset;
  const double R = s->radius, k = s->conic;
  if (s->geom_kind == OL_GEOM_BICONIC) { /* biconic.py:105-158 */
    double Rx = s->radius, kx = s->conic, Ry = c[0], ky = c[1];
    double cx = (isinf(Rx) || Rx == 0) ? 0.0 : 1.0 / Rx;
    double cy = (isinf(Ry) || Ry == 0) ? 0.0 : 1.0 / Ry;
    double dfdx = 0.0, dfdy = 0.0;
    if (cx != 0.0) {
      double v = 1.0 - (1.0 + kx) * cx * cx * x * x;
      double st = v < 1e-14 ? 1e-14 : v;
      double sq = sqrt(st);
      double sd = fabs(sq) < 1e-14 ? 1e-14 : sq;
      dfdx = (cx * x) / sd;
    }
    if (cy != 0.0) {
      double v = 1.0 - (1.0 + ky) * cy * cy * y * y;
      double st = v < 1e-14 ? 1e-14 : v;
      double sq = sqrt(st);
      double sd = fabs(sq) < 1e-14 ? 1e-14 : sq;
      dfdy = (cy * y) / sd;
    }
    double mag = sqrt(dfdx * dfdx + dfdy * dfdy + 1.0);
    double sm = mag < 1e-14 ? 1.0 : mag;
    *nx = dfdx / sm; *ny = dfdy / sm; *nz = -1.0 / sm;
    return;
  }
  if (s->geom_kind == OL_GEOM_TOROIDAL) { /* toroidal.py:192-242 */
    const double eps = 1e-14;
    double Rr = c[0];
    double z_y = tor_zy(s, c, y), dz_dy = tor_dzy(s, c, y);
    double fx, fy, term;
    if (isinf(Rr)) {
      fx = 0.0; fy = dz_dy; term = INFINITY;
    } else {
      term = (Rr - z_y) * (Rr - z_y) - x * x;
      int valid = term >= 0;
      double safe_term = valid ? term : eps;
      double sq = sqrt(safe_term);
      double ssq = fabs(sq) < eps ? eps : sq;
      double sg = (Rr > 0) - (Rr < 0);
      fx = valid ? sg * x / ssq : 0.0;
      fy = valid ? sg * (Rr - z_y) * dz_dy / ssq : 0.0;
    }
    double mag = sqrt(fx * fx + fy * fy + 1.0);
    double sm = mag < eps ? 1.0 : mag;
    if (term >= 0) { *nx = fx / sm; *ny = fy / sm; *nz = -1.0 / sm; }
    else { *nx = 0.0; *ny = 0.0; *nz = -1.0; }
    return;
  }
  double r2 = x * x + y * y;
  double denom = R * sqrt(1.0 - (1.0 + k) * r2 / (R * R));
  double dfdx = x / denom, dfdy = y / denom;
  switch (s->geom_kind) {
    case OL_GEOM_EVEN_ASPHERE:
      for (int i = 0; i < s->n_coeff; ++i) {
        dfdx = dfdx + 2 * (i + 1) * x * c[i] * pow(r2, (double)i);
        dfdy = dfdy + 2 * (i + 1) * y * c[i] * pow(r2, (double)i);
      }
      break;
    case OL_GEOM_ODD_ASPHERE: {
      /* odd_asphere.py:120-134: non-finite terms (r == 0, i == 0) are zeroed */
      double r = sqrt(r2);
      for (int i = 0; i < s->n_coeff; ++i) {
        double x_term = (i + 1) * x * c[i] * pow(r, (double)(i - 1));
        double y_term = (i + 1) * y * c[i] * pow(r, (double)(i - 1));
        if (!isfinite(x_term)) x_term = 0.0;
        if (!isfinite(y_term)) y_term = 0.0;
        dfdx = dfdx + x_term;
        dfdy = dfdy + y_term;
      }
      break;
    }
    case OL_GEOM_POLYNOMIAL: {
      int cols = s->poly_cols, rows = cols ? s->n_coeff / cols : 0;
      for (int i = 1; i < rows; ++i)
        for (int j = 0; j < cols; ++j)
          dfdx = dfdx + i * c[i * cols + j] * pow(x, (double)(i - 1)) * pow(y, (double)j);
      for (int i = 0; i < rows; ++i)
        for (int j = 1; j < cols; ++j)
          dfdy = dfdy + j * c[i * cols + j] * pow(x, (double)i) * pow(y, (double)(j - 1));
      break;
    }
    case OL_GEOM_CHEBYSHEV: { /* chebyshev.py:154-195: no 1/norm on the derivative */
      double xn = x / c[0], yn = y / c[1];
      if (fabs(xn) > 1.0 || fabs(yn) > 1.0) g_normal_status |= OL_STATUS_CHEBYSHEV_RANGE;
      int cols = s->poly_cols, rows = cols ? s->n_coeff / cols : 0;
      for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j) {
          double cij = c[2 + i * cols + j];
          if (cij == 0.0) continue;
          dfdx = dfdx + (cheb_dT(i, xn) * cij * cheb_T(j, yn));
          dfdy = dfdy + (cheb_dT(j, yn) * cij * cheb_T(i, xn));
        }
      break;
    }
    case OL_GEOM_ZERNIKE: {
      const double eps = 1e-14, nr = s->norm_radius;
      double xn = x / nr, yn = y / nr;
      double rho = sqrt(xn * xn + yn * yn);
      double phi = atan2(yn, xn);
      double drho_dx = all_rho_zero ? 0.0 : ((x / (nr * nr)) / (rho + eps));
      double drho_dy = all_rho_zero ? 0.0 : ((y / (nr * nr)) / (rho + eps));
      double dphi_dx = -(yn) / (rho * rho + eps) * (1.0 / nr);
      double dphi_dy = +(xn) / (rho * rho + eps) * (1.0 / nr);
      for (int j = 0; j < s->n_coeff; ++j) {
        double cj = c[4 * j];
        if (cj == 0.0) continue;
        int n = (int)c[4 * j + 1], m = (int)c[4 * j + 2];
        int ma = m < 0 ? -m : m;
        /* zernike/base.py:100-137 get_derivative (no norm constant!) */
        double Rt = zern_radial(n, ma, rho);
        double Rd = zern_radial_deriv(n, ma, rho);
        double dZdrho, dZdphi;
        if (m == 0) { dZdrho = Rd; dZdphi = 0.0; }
        else if (m > 0) { dZdrho = Rd * cos(m * phi); dZdphi = -m * Rt * sin(m * phi); }
        else { dZdrho = Rd * sin(ma * phi); dZdphi = ma * Rt * cos(ma * phi); }
        dfdx += cj * (dZdrho * drho_dx + dZdphi * dphi_dx);
        dfdy += cj * (dZdrho * drho_dy + dZdphi * dphi_dy);
      }
      double norm = sqrt(dfdx * dfdx + dfdy * dfdy + 1.0);
      if (norm < eps) norm = 1.0;
      *nx = dfdx / norm; *ny = dfdy / norm; *nz = -1.0 / norm;
      return;
    }
    default:
      break;
  }
  double mag = sqrt(dfdx * dfdx + dfdy * dfdy + 1.0);
  *nx = dfdx / mag; *ny = dfdy / mag; *nz = -1.0 / mag;
}

/* ---- standard.py:97-148 (conic distance) ----------------------------------- */
static double conic_distance(double R, double k, double x, double y, double z,
                             double L, double M, double N) {
  if (isinf(R)) {
    double N_safe = fabs(N) > 1e-14 ? N : 1e-14;
    return -z / N_safe;
  }
  /* (NumPy's association: `self.k * rays.N**2` is k (N N), not (k N) N -- where the formula
   * is ill conditioned, |1 + k| << 1, that last bit shows at 1e-9 mm) */
  double a = k * (N * N) + L * L + M * M + N * N;
  double b = 2 * k * N * z + 2 * L * x + 2 * M * y - 2 * N * R + 2 * N * z;
  double c = k * (z * z) - 2 * R * z + x * x + y * y + z * z;
  double d = b * b - 4 * a * c;
  double sq = sqrt(d); /* NaN for d < 0, silently (standard.py:132-137) */
  double t1 = (-b + sq) / (2 * a);
  double t2 = (-b - sq) / (2 * a);
  double z1 = z + t1 * N, z2 = z + t2 * N;
  double t = (fabs(z1) <= fabs(z2)) ? t1 : t2; /* NaN compare false -> t2 */
  if (a == 0) t = -c / b;
  return t;
}

/* distance for the whole batch; Newton loop with the GLOBAL stop rule of
 * newton_raphson.py:119-168.                                                  */
static void batch_distance(const ol_surface_desc* s, const double* coeffs,
                           const rays_t* r, double* t, uint32_t* status) {
  const int64_t n = r->n;
  if (s->geom_kind == OL_GEOM_PLANE) { /* plane.py:72-88 */
    for (int64_t j = 0; j < n; ++j) t[j] = -r->z[j] / r->N[j];
    return;
  }
  for (int64_t j = 0; j < n; ++j)
    t[j] = conic_distance(s->radius, s->conic, r->x[j], r->y[j], r->z[j], r->L[j],
                          r->M[j], r->N[j]);
  if (s->geom_kind == OL_GEOM_STANDARD) return;

  double* f = (double*)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
  for (int it = 0; it < s->max_iter; ++it) {
    double fmax = 0.0; /* numpy max propagates NaN */
    int has_nan = 0, all_rho_zero = 1;
    for (int64_t j = 0; j < n; ++j) {
      double xi = r->x[j] + t[j] * r->L[j];
      double yi = r->y[j] + t[j] * r->M[j];
      double zi = r->z[j] + t[j] * r->N[j];
      f[j] = geom_sag(s, coeffs, xi, yi, status) - zi;
      double af = fabs(f[j]);
      if (isnan(af)) has_nan = 1;
      else if (af > fmax) fmax = af;
      if (!(xi == 0.0 && yi == 0.0)) all_rho_zero = 0;
    }
    if (n == 0) break;
    if (!has_nan && fmax < s->tol) break;
    for (int64_t j = 0; j < n; ++j) {
      double xi = r->x[j] + t[j] * r->L[j];
      double yi = r->y[j] + t[j] * r->M[j];
      double nx, ny, nz;
      geom_normal_nr(s, coeffs, xi, yi, all_rho_zero, &nx, &ny, &nz);
      double nz_safe = fabs(nz) > 1e-14 ? nz : 1e-14;
      double fx = -nx / nz_safe, fy = -ny / nz_safe;
      double df = fx * r->L[j] + fy * r->M[j] - r->N[j];
      double df_safe = fabs(df) > 1e-14 ? df : 1e-14;
      t[j] = t[j] - f[j] / df_safe;
    }
  }
  free(f);
}

/* surface normal at the hit point, per geometry (interact step) */
static void hit_normal(const ol_surface_desc* s, const double* coeffs, double x,
                       double y, int all_rho_zero, double* nx, double* ny,
                       double* nz) {
  if (s->geom_kind == OL_GEOM_PLANE) { /* plane.py:90-109 */
    *nx = 0.0; *ny = 0.0; *nz = 1.0;
    return;
  }
  if (s->geom_kind == OL_GEOM_STANDARD) { /* standard.py:150-175 */
    double R = s->radius, k = s->conic;
    double r2 = x * x + y * y;
    double denom = R * sqrt(1.0 - (1.0 + k) * r2 / (R * R));
    double dfdx = x / denom, dfdy = y / denom, dfdz = -1.0;
    double mag = sqrt(dfdx * dfdx + dfdy * dfdy + dfdz * dfdz);
    *nx = dfdx / mag; *ny = dfdy / mag; *nz = dfdz / mag;
    return;
  }
  geom_normal_nr(s, coeffs, x, y, all_rho_zero, nx, ny, nz);
}

/* ---- physical_apertures/ contains() ------------------------------------ */
/* polygon.py:54-71 -> backend/numpy_backend.py:1125-1137 ->
 * matplotlib.path.Path(vertices).contains_points(points) (radius 0): matplotlib's
 * point_in_path_impl (src/_path.h), the crossings test of Graphics Gems IV over the
 * implicitly closed polygon; non-finite points are outside.  (matplotlib is a pinned
 * dependency of the reference, not vendored; tests/test_oracle_polygon.py checks this
 * restatement against the installed matplotlib where it is importable.)            */
static int polygon_contains(const double* v, int nv, double tx, double ty) {
  if (!(isfinite(tx) && isfinite(ty)) || nv < 1) return 0;
  int inside = 0;
  double x0 = v[0], y0 = v[1]; /* edge start; the first "edge" is degenerate */
  int yflag0 = y0 >= ty;
  for (int k = 1; k <= nv; ++k) {
    const double x1 = v[2 * (k % nv)], y1 = v[2 * (k % nv) + 1]; /* k == nv closes */
    const int yflag1 = y1 >= ty;
    if (yflag0 != yflag1 &&
        (((y1 - ty) * (x0 - x1) >= (x1 - tx) * (y0 - y1)) == yflag1))
      inside ^= 1;
    yflag0 = yflag1;
    x0 = x1;
    y0 = y1;
  }
  return inside;
}

static int leaf_contains(int kind, const double* a, const double* coeffs, double x, double y) {
  switch (kind) {
    case OL_AP_POLYGON: return polygon_contains(coeffs + (int)a[0], (int)a[1], x, y);
    case OL_AP_RADIAL: { /* radial.py:56-70 */
      double r2 = x * x + y * y;
      return (r2 <= a[1] * a[1]) && (r2 >= a[0] * a[0]);
    }
    case OL_AP_OFFSET_RADIAL: { /* offset_radial.py:48-61 */
      double r2 = (x - a[2]) * (x - a[2]) + (y - a[3]) * (y - a[3]);
      return (r2 <= a[1] * a[1]) && (r2 >= a[0] * a[0]);
    }
    case OL_AP_RECTANGULAR: /* rectangular.py:42-59 */
      return (a[0] <= x) && (x <= a[1]) && (a[2] <= y) && (y <= a[3]);
    case OL_AP_ELLIPTICAL: { /* elliptical.py:42-56 */
      double xx = x - a[2], yy = y - a[3];
      return (xx * xx / (a[0] * a[0]) + yy * yy / (a[1] * a[1])) <= 1.0;
    }
    default:
      return 1;
  }
}

/* base.py:259-340: Union / Intersection / Difference, evaluated from the
 * reverse-Polish token list the packer emits.                                */
static int aperture_contains(const ol_surface_desc* s, const double* coeffs, double x,
                             double y) {
  if (s->aperture_kind != OL_AP_COMPOSITE)
    return leaf_contains(s->aperture_kind, s->aperture, coeffs, x, y);
  const double* tok = coeffs + (int)s->aperture[0];
  int n_tok = (int)s->aperture[1];
  int stack[OL_AP_MAX_DEPTH], sp = 0;
  for (int i = 0; i < n_tok; ++i, tok += OL_AP_TOKEN_DOUBLES) {
    int op = (int)tok[0];
    if (op < OL_AP_OP_UNION) {
      stack[sp++] = leaf_contains(op, tok + 1, coeffs, x, y);
    } else {
      int b = stack[--sp], a = stack[--sp];
      stack[sp++] = op == OL_AP_OP_UNION ? (a || b)
                    : op == OL_AP_OP_INTERSECTION ? (a && b) : (a && !b);
    }
  }
  return sp > 0 ? stack[sp - 1] : 1;
}

/* ---- rays/polarized_rays.py:136-202 (get_local_basis + update) ------------- */
static void cross3(const double* a, const double* b, double* o) {
  o[0] = a[1] * b[2] - a[2] * b[1];
  o[1] = a[2] * b[0] - a[0] * b[2];
  o[2] = a[0] * b[1] - a[1] * b[0];
}
static double norm3(const double* a) {
  return sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
}

/* local basis of rays/polarized_rays.py:136-178 for ray j */
static void local_basis(const rays_t* r, int64_t j, double* s, double* p0, double* p1,
                        double* k0, double* k1) {
  k0[0] = r->L0[j]; k0[1] = r->M0[j]; k0[2] = r->N0[j];
  k1[0] = r->L[j]; k1[1] = r->M[j]; k1[2] = r->N[j];
  cross3(k0, k1, s);
  double mag = norm3(s);
  if (mag == 0.0) { /* k0 parallel k1 (NaN != 0, falls through like numpy) */
    double xh[3] = {1, 0, 0}, yh[3] = {0, 1, 0}, pf[3];
    cross3(k0, xh, pf);
    if (norm3(pf) == 0.0) cross3(k0, yh, pf);
    cross3(pf, k0, s);
    mag = norm3(s);
  }
  for (int a = 0; a < 3; ++a) s[a] /= mag;
  cross3(k0, s, p0);
  cross3(k1, s, p1);
}

/* PolarizedRays.update (polarized_rays.py:180-202) with a full 3x3 Jones matrix
 * J (row-major) or J = NULL for the identity.                                 */
static void prt_update(rays_t* r, int64_t j, const double complex* J) {
  double s[3], p0[3], p1[3], k0[3], k1[3];
  local_basis(r, j, s, p0, p1, k0, k1);
  /* o_in rows (s, p0, k0); o_out columns (s, p1, k1) */
  double oin[3][3], oout[3][3];
  for (int a = 0; a < 3; ++a) {
    oin[0][a] = s[a]; oin[1][a] = p0[a]; oin[2][a] = k0[a];
    oout[a][0] = s[a]; oout[a][1] = p1[a]; oout[a][2] = k1[a];
  }
  double complex JO[3][3], P[3][3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      double complex acc = 0;
      for (int c = 0; c < 3; ++c) acc += (J ? J[3 * a + c] : (a == c ? 1.0 : 0.0)) * oin[c][b];
      JO[a][b] = acc;
    }
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      double complex acc = 0;
      for (int c = 0; c < 3; ++c) acc += oout[a][c] * JO[c][b];
      P[a][b] = acc;
    }
  double complex* p = r->p + 9 * j;
  double complex q[9];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) {
      double complex acc = 0;
      for (int c = 0; c < 3; ++c) acc += P[a][c] * p[3 * c + b];
      q[3 * a + b] = acc;
    }
  memcpy(p, q, sizeof(q));
}

/* jones.py:120-181 (linear polarizer) and jones.py:331-393 (linear retarder) */
static void axis_jones(const rays_t* r, int64_t j, const double* axis, int retarder,
                       double d, double complex* J) {
  double s[3], p0[3], p1[3], k0[3], k1[3];
  local_basis(r, j, s, p0, p1, k0, k1);
  double ts_in = axis[0] * s[0] + axis[1] * s[1] + axis[2] * s[2];
  double tp_in = axis[0] * p0[0] + axis[1] * p0[1] + axis[2] * p0[2];
  double norm_in = sqrt(ts_in * ts_in + tp_in * tp_in);
  if (norm_in == 0) norm_in = 1.0;
  double us_in = ts_in / norm_in, up_in = tp_in / norm_in;
  for (int a = 0; a < 9; ++a) J[a] = 0;
  if (retarder) {
    double complex em = cexp(-I * d / 2), ep = cexp(I * d / 2);
    J[0] = em * us_in * us_in + ep * up_in * up_in;
    J[1] = -2.0 * I * sin(d / 2) * us_in * up_in;
    J[3] = J[1];
    J[4] = ep * us_in * us_in + em * up_in * up_in;
  } else {
    double ts_out = ts_in;
    double tp_out = axis[0] * p1[0] + axis[1] * p1[1] + axis[2] * p1[2];
    double norm_out = sqrt(ts_out * ts_out + tp_out * tp_out);
    if (norm_out == 0) norm_out = 1.0;
    double us_out = ts_out / norm_out, up_out = tp_out / norm_out;
    J[0] = us_out * us_in; J[1] = us_out * up_in;
    J[3] = up_out * us_in; J[4] = up_out * up_in;
  }
  J[8] = 1.0;
}

/* ---- one surface, whole batch: standard_surface.py:200-274 ----------------- */
static void trace_surface(const ol_surface_desc* s, const double* coeffs,
                          const ol_surface_optics* o, rays_t* r, double* t,
                          uint32_t* status) {
  const int64_t n = r->n;
  if (s->interaction == OL_INTERACT_RECORD_ONLY) return; /* object_surface.py:56-93 */
  localize(s, r);
  batch_distance(s, coeffs, r, t, status);
  /* propagation/homogeneous.py:30-57 */
  for (int64_t j = 0; j < n; ++j) {
    r->x[j] = r->x[j] + t[j] * r->L[j];
    r->y[j] = r->y[j] + t[j] * r->M[j];
    r->z[j] = r->z[j] + t[j] * r->N[j];
  }
  if (o->absorb > 0.0) /* == any(k > 0) for a scalar k */
    for (int64_t j = 0; j < n; ++j) r->i[j] = r->i[j] * exp(-o->absorb * t[j]);
  /* standard_surface.py:244 */
  for (int64_t j = 0; j < n; ++j) r->opd[j] = r->opd[j] + fabs(t[j] * o->n1);
  /* clip: physical_apertures/base.py:71-82, real_rays.py:154-161 */
  if (s->aperture_kind != OL_AP_NONE)
    for (int64_t j = 0; j < n; ++j)
      if (!aperture_contains(s, coeffs, r->x[j], r->y[j])) r->i[j] = 0.0;

  /* interactions/refractive_reflective_model.py:32-55 */
  int all_rho_zero = 1;
  for (int64_t j = 0; j < n; ++j)
    if (!(r->x[j] == 0.0 && r->y[j] == 0.0)) { all_rho_zero = 0; break; }
  const double u = o->n1 / o->n2;
  for (int64_t j = 0; j < n; ++j) {
    double nx, ny, nz;
    hit_normal(s, coeffs, r->x[j], r->y[j], all_rho_zero, &nx, &ny, &nz);
    double L0 = r->L[j], M0 = r->M[j], N0 = r->N[j];
    r->L0[j] = L0; r->M0[j] = M0; r->N0[j] = N0;
    /* real_rays.py:535-571 _align_surface_normal */
    double dot = L0 * nx + M0 * ny + N0 * nz;
    double sgn = (dot > 0) - (dot < 0);
    if (isnan(dot)) sgn = NAN;
    double ax = nx * sgn, ay = ny * sgn, az = nz * sgn;
    dot = fabs(dot);
    if (s->interaction == OL_INTERACT_REFLECT) { /* real_rays.py:189-205 */
      r->L[j] = L0 - 2 * dot * ax;
      r->M[j] = M0 - 2 * dot * ay;
      r->N[j] = N0 - 2 * dot * az;
    } else { /* real_rays.py:163-187 */
      double root = sqrt(1 - u * u * (1 - dot * dot));
      r->L[j] = u * L0 + ax * root - u * ax * dot;
      r->M[j] = u * M0 + ay * root - u * ay * dot;
      r->N[j] = u * N0 + az * root - u * az * dot;
    }
    /* interactions/base.py:111-128 _apply_coating_and_bsdf */
    if (s->coating_kind == OL_COAT_SIMPLE) { /* coatings.py:213,236 */
      r->i[j] = r->i[j] *
                (s->interaction == OL_INTERACT_REFLECT ? s->coat[1] : s->coat[0]);
    } else if (s->coating_kind == OL_COAT_FRESNEL && r->p) {
      /* coatings.py:72-92 _compute_aoi uses the un-aligned normal */
      double d = fabs(nx * L0 + ny * M0 + nz * N0);
      if (d > 1.0) d = 1.0;
      if (d < -1.0) d = -1.0;
      double aoi = acos(d);
      /* jones.py:71-117 */
      double cosi = cos(aoi);
      double nn = o->n2 / o->n1;
      double complex root = csqrt((double complex)(nn * nn - sin(aoi) * sin(aoi)));
      double complex J[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
      if (s->interaction == OL_INTERACT_REFLECT) {
        double complex sj = (cosi - root) / (cosi + root);
        double complex pj = (nn * nn * cosi - root) / (nn * nn * cosi + root);
        J[0] = sj; J[4] = -pj; J[8] = -1.0;
      } else {
        double complex sj = 2 * cosi / (cosi + root);
        double complex pj = 2 * nn * cosi / (nn * nn * cosi + root);
        J[0] = sj; J[4] = pj; J[8] = 1.0;
      }
      prt_update(r, j, J);
    } else if ((s->coating_kind == OL_COAT_POLARIZER || s->coating_kind == OL_COAT_RETARDER) &&
               r->p) {
      const double* ax = coeffs + (int)s->coat[0];
      double complex J[9];
      axis_jones(r, j, ax, s->coating_kind == OL_COAT_RETARDER,
                 s->coating_kind == OL_COAT_RETARDER ? ax[3] : 0.0, J);
      prt_update(r, j, J);
    } else if (r->p) {
      prt_update(r, j, NULL); /* rays.update() with no Jones matrix */
    }
  }
  globalize(s, r);
}

/* ===========================================================================
 * Exported entry points (loaded with ctypes by tests/ and bench.py only)
 * ========================================================================= */

/* SurfaceGroup.trace(rays, skip=first) -- surfaces/surface_group.py:245-257.
 * rays[8]: x,y,z,L,M,N,i,opd (updated in place).  record: rows x 8 x n or NULL.
 * prt: n x 9 complex128 (interleaved) or NULL.  Returns OL_STATUS_* bits.     */
uint32_t oracle_trace(const ol_surface_desc* surf, int32_t n_surf,
                      const double* coeffs, const ol_surface_optics* optics,
                      int32_t n_wl, int32_t wl_index, int64_t n, double* const rays[8],
                      double* record, double* prt, double* pre_dir /*3 x n or NULL*/,
                      int32_t first, int32_t last) {
  (void)n_surf;
  uint32_t status = 0;
  g_normal_status = 0;
  rays_t r;
  r.n = n;
  r.x = rays[PX]; r.y = rays[PY]; r.z = rays[PZ];
  r.L = rays[PL]; r.M = rays[PM]; r.N = rays[PN];
  r.i = rays[PI_]; r.opd = rays[POPD];
  size_t nn = (size_t)(n > 0 ? n : 1);
  double* scratch = (double*)malloc(sizeof(double) * nn * 4);
  r.L0 = scratch; r.M0 = scratch + nn; r.N0 = scratch + 2 * nn;
  double* t = scratch + 3 * nn;
  for (int64_t j = 0; j < n; ++j) { r.L0[j] = r.L[j]; r.M0[j] = r.M[j]; r.N0[j] = r.N[j]; }
  r.p = (double complex*)prt;
  for (int32_t s = first; s <= last; ++s) {
    trace_surface(&surf[s], coeffs, &optics[(size_t)s * n_wl + wl_index], &r, t, &status);
    if (record) { /* standard_surface.py:260-274 */
      double* row = record + (size_t)(s - first) * 8 * (size_t)n;
      for (int k = 0; k < 8; ++k) memcpy(row + (size_t)k * n, rays[k], sizeof(double) * (size_t)n);
    }
  }
  if (pre_dir) {
    memcpy(pre_dir, r.L0, sizeof(double) * (size_t)n);
    memcpy(pre_dir + n, r.M0, sizeof(double) * (size_t)n);
    memcpy(pre_dir + 2 * n, r.N0, sizeof(double) * (size_t)n);
  }
  free(scratch);
  return status | g_normal_status;
}

/* optiland/apodization/{uniform,gaussian,cosine_squared,hann,polynomial,
 * super_gaussian,tukey}.py get_intensity, applied at rays/ray_generator.py:81-85     */
static double apodization(const ol_raygen_params* p, double Px, double Py) {
  const double a = p->apod_a, b = p->apod_b;
  const double r2 = Px * Px + Py * Py;
  const double r = sqrt(r2);
  switch (p->apod_kind) {
    case OL_APOD_GAUSSIAN: return exp(-r2 / (2 * a * a));
    case OL_APOD_COSINE_SQUARED: {
      double c = cos(M_PI * r / (2 * a));
      return r < a ? c * c : 0.0;
    }
    case OL_APOD_HANN: return r < a / 2 ? 0.5 * (1 - cos(2 * M_PI * r / a)) : 0.0;
    case OL_APOD_POLYNOMIAL: {
      double q = (r / a) * (r / a);
      return r < a ? pow(1 - q, b) : 0.0;
    }
    case OL_APOD_SUPER_GAUSSIAN: return exp(-pow(r / a, b));
    case OL_APOD_TUKEY: {
      double flat = a * (1 - b / 2);
      double taper = 0.5 * (1 + cos(M_PI * (r - flat) / (a * b / 2)));
      double i = r <= flat ? 1.0 : 0.0;
      return (r > flat && r < a) ? taper : i;
    }
    default: return 1.0;
  }
}

/* rays/ray_generator.py:47-99 + rays/ray_aiming/paraxial.py:33-106 +
 * fields/field_types/angle.py:17-58 + fields/field_types/object_height.py:19-47
 * (planar object).  out[7] = x,y,z,L,M,N,i.                                      */
void oracle_generate_rays(const ol_raygen_params* p, int64_t n, const double* hx,
                          const double* hy, const double* px, const double* py,
                          const double* vx, const double* vy, double* const out[7]) {
  const double d2r = M_PI / 180.0;
  for (int64_t j = 0; j < n; ++j) {
    double vxx = vx ? vx[j] : 1.0, vyy = vy ? vy[j] : 1.0;
    double field_x = p->max_field * hx[j], field_y = p->max_field * hy[j];
    double x0, y0, z0;
    const int linear = p->field_kind != OL_FIELD_ANGLE;
    if (p->field_kind == OL_FIELD_OBJECT_HEIGHT ||
        (p->field_kind == OL_FIELD_PARAXIAL_IMAGE_HEIGHT && !p->object_infinite)) {
      /* object_height.py:36-47: x0 = field_x, y0 = field_y, z0 = sag(x0,y0) + obj z
       * (sag = 0: only planar object surfaces are packed); paraxial_image_height.py:
       * 50-60: the same with field = y_obj_unit * (max_field H / y_img_unit), the
       * scale folded into max_field by the packer                                 */
      x0 = field_x;
      y0 = field_y;
      z0 = p->z_first;
    } else if (p->object_infinite) {
      /* angle.py:40-47; paraxial_image_height.py:39-49 with the slope
       * u_obj = u_obj_unit * (max_field H / y_img_unit) in place of the tangent   */
      double sx = linear ? field_x : tan(field_x * d2r);
      double sy = linear ? field_y : tan(field_y * d2r);
      double x = -sx * (p->offset + p->EPL);
      double y = -sy * (p->offset + p->EPL);
      z0 = p->z_first - p->offset;
      x0 = px[j] * p->EPD / 2 * vxx + x;
      y0 = py[j] * p->EPD / 2 * vyy + y;
    } else {
      z0 = p->z_first;
      x0 = -tan(field_x * d2r) * (p->EPL - z0);
      y0 = -tan(field_y * d2r) * (p->EPL - z0);
    }
    double x1, y1, z1;
    if (p->tele_dz > 0.0) { /* paraxial.py:82-87: object-space telecentric */
      z1 = p->tele_dz + z0;
      x1 = px[j] * vxx + x0;
      y1 = py[j] * vyy + y0;
    } else { /* :88-94 */
      x1 = px[j] * p->EPD * vxx / 2;
      y1 = py[j] * p->EPD * vyy / 2;
      z1 = p->EPL;
    }
    double mag = sqrt((x1 - x0) * (x1 - x0) + (y1 - y0) * (y1 - y0) + (z1 - z0) * (z1 - z0));
    int is_zero = mag < 1e-9;
    if (is_zero) mag = 1.0;
    out[0][j] = x0; out[1][j] = y0; out[2][j] = z0;
    out[3][j] = is_zero ? 0.0 : (x1 - x0) / mag;
    out[4][j] = is_zero ? 0.0 : (y1 - y0) / mag;
    out[5][j] = is_zero ? 1.0 : (z1 - z0) / mag;
    out[6][j] = apodization(p, px[j], py[j]);
  }
}

/* rays/polarized_rays.py:68-133, 204-233: update_intensity.
 * prt: n x 9 complex128.  Returns status bits.                               */
uint32_t oracle_polarized_intensity(int64_t n, const double* prt, const double* L0,
                                    const double* M0, const double* N0, const double* i0,
                                    const ol_polarization_state* st, double* intensity) {
  uint32_t status = 0;
  const double complex* P = (const double complex*)prt;
  int nf = st->is_polarized ? 1 : 2;
  for (int64_t j = 0; j < n; ++j) {
    double k[3] = {L0[j], M0[j], N0[j]}, xh[3] = {1, 0, 0}, p[3], s[3];
    cross3(k, xh, p);
    double nrm = norm3(p);
    if (nrm == 0.0) status |= OL_STATUS_K_PARALLEL_X;
    for (int a = 0; a < 3; ++a) p[a] /= nrm;
    cross3(p, k, s);
    double acc = 0.0;
    for (int f = 0; f < nf; ++f) {
      double Ex, Ey, phx, phy;
      if (st->is_polarized) { Ex = st->Ex; Ey = st->Ey; phx = st->phase_x; phy = st->phase_y; }
      else { Ex = f == 0 ? 1.0 : 0.0; Ey = f == 0 ? 0.0 : 1.0; phx = phy = 0.0; }
      double complex E0[3], E1[3];
      for (int a = 0; a < 3; ++a)
        E0[a] = Ex * cexp(I * phx) * s[a] + Ey * cexp(I * phy) * p[a];
      for (int a = 0; a < 3; ++a) {
        E1[a] = 0;
        for (int b = 0; b < 3; ++b) E1[a] += P[9 * j + 3 * a + b] * E0[b];
        acc += cabs(E1[a]) * cabs(E1[a]);
      }
    }
    intensity[j] = acc * i0[j] / nf;
  }
  return status;
}

/* wavefront/strategy.py:163-215 (steps 4-5), reference_geometry.py:41-79,
 * strategy.py:83-139.  rays[7]: x,y,z,L,M,N,opd at the image surface.           */
void oracle_wavefront_opd(const ol_wavefront_params* p, int64_t n, double* const rays[7],
                          const double* px, const double* py, double* opd_waves,
                          double* const pupil[3]) {
  for (int64_t j = 0; j < n; ++j) {
    double xr = rays[0][j], yr = rays[1][j], zr = rays[2][j];
    double L = -rays[3][j], M = -rays[4][j], N = -rays[5][j];
    double xc = p->xc, yc = p->yc, zc = p->zc, R = p->R;
    double a = L * L + M * M + N * N;
    double b = 2 * (L * (xr - xc) + M * (yr - yc) + N * (zr - zc));
    double c = xr * xr + yr * yr + zr * zr - 2 * (xr * xc + yr * yc + zr * zc) + xc * xc +
               yc * yc + zc * zc - R * R;
    double d = b * b - 4 * a * c;
    if (d < 0) d = 0;
    double t1 = (-b - sqrt(d)) / (2 * a), t2 = (-b + sqrt(d)) / (2 * a);
    double t = t1 < 0 ? t2 : t1;
    if (p->nx != 0.0 || p->ny != 0.0 || p->nz != 0.0) {
      /* wavefront/reference_geometry.py:104-124 PlanarReference.path_length */
      double num = (xr - xc) * p->nx + (yr - yc) * p->ny + (zr - zc) * p->nz;
      double den = L * p->nx + M * p->ny + N * p->nz;
      if (fabs(den) < 1e-12) den = 1e-12;
      t = -num / den;
    }
    double opd_img = p->n_image * t;
    double opd = rays[6][j] - opd_img;
    double X_m = px[j] * p->half_epd, Y_m = py[j] * p->half_epd;
    opd = opd + (p->ux * X_m + p->uy * Y_m);
    opd_waves[j] = (p->opd_ref - opd) / (p->wavelength_um * 1e-3);
    if (pupil) {
      double tt = opd_img / p->n_image;
      pupil[0][j] = xr - tt * rays[3][j];
      pupil[1][j] = yr - tt * rays[4][j];
      pupil[2][j] = zr - tt * rays[5][j];
    }
  }
}

/* single-point helpers for the reference's known-answer unit tests */
int oracle_polygon_contains(const double* vertices, int nv, double x, double y) {
  return polygon_contains(vertices, nv, x, y);
}

double oracle_sag(const ol_surface_desc* s, const double* coeffs, double x, double y) {
  uint32_t st = 0;
  if (s->geom_kind == OL_GEOM_PLANE) return 0.0;
  if (s->geom_kind == OL_GEOM_STANDARD) return conic_sag(s->radius, s->conic, x, y);
  return geom_sag(s, coeffs, x, y, &st);
}

void oracle_normal(const ol_surface_desc* s, const double* coeffs, double x, double y,
                   double* out3) {
  hit_normal(s, coeffs, x, y, (x == 0.0 && y == 0.0), &out3[0], &out3[1], &out3[2]);
}

void oracle_distance(const ol_surface_desc* s, const double* coeffs, int64_t n,
                     const double* x, const double* y, const double* z, const double* L,
                     const double* M, const double* N, double* t) {
  rays_t r;
  memset(&r, 0, sizeof(r));
  r.n = n;
  r.x = (double*)x; r.y = (double*)y; r.z = (double*)z;
  r.L = (double*)L; r.M = (double*)M; r.N = (double*)N;
  uint32_t st = 0;
  batch_distance(s, coeffs, &r, t, &st);
}
