// device_table.h -- device-resident image of the surface table (gfx950).
//
// The public `ol_surface_desc` (include/optiland_hip.h) is what the boundary
// accepts; at ol_system_create time it is re-expressed per arithmetic type T as
// `DevSurf<T>`: curvature instead of radius, squared aperture bounds, the
// surface-to-surface relative transform (so fp32 never adds a 6 m vertex offset
// to a 10 um spot coordinate), Zernike terms regrouped into per-|m| radial
// polynomials.  Every field is wave-uniform at run time: the kernel indexes the
// table with the (uniform) surface counter, so the compiler emits scalar loads
// (s_load_dwordx8/x16 through the scalar cache) and the values live in SGPRs --
// they cost no VGPRs and no LDS traffic.  Variable-length coefficient blocks
// (aspheres, polynomials, Zernike groups) are staged per block into LDS.
#pragma once
#include <stdint.h>

// Qualifier of the per-ray arithmetic (surface_math.h, raygen_device.h).  Product build:
// device code only.  OL_HOST_MATH is defined by tests/hostmath/harness.hip alone, which
// executes the same functions on the host as a check that needs no GPU.
#ifdef OL_HOST_MATH
#define OL_DEV __host__ __device__ inline
#else
#define OL_DEV __device__ __forceinline__
#endif

// Table memory as the kernels address it.  In device code every pointer into the surface
// table (hot / cold / optics rows, coefficient blocks) is a CONSTANT-address-space pointer
// (amdgcn address space 4): loads through it are scalar loads (s_load_dword*, the K$ path)
// by construction -- not by an alias analysis that has to prove the record stores cannot
// clobber the table -- and, unlike loads through a `__restrict__` kernel argument, they stay
// scalar loads when the pointer value has been passed through `refresh()` below.  On the
// host (the argument marshalling of the launchers; tests/hostmath) it is a plain pointer.
#if defined(__HIP_DEVICE_COMPILE__)
#define OL_CONST_AS __attribute__((address_space(4)))
#else
#define OL_CONST_AS
#endif

namespace ol {
template <typename X>
using cptr = const OL_CONST_AS X*;

// global -> constant address space (the same addresses on amdgcn)
template <typename X>
OL_DEV cptr<X> as_const(const X* p) {
  return (cptr<X>)p;
}

// `refresh(p)`: the same pointer, but opaque to the optimiser from here on.  Loads through
// the returned value cannot be merged with, or hoisted above, anything before this point,
// so a table field is fetched WHERE it is used (a scalar load that hits the K$) instead of
// once in the prologue and then kept live across the whole surface body -- in the
// Newton-Raphson and fp64 kernels those long-lived scalars overflowed the ~100 SGPRs a
// wave has and were spilled to VGPR lanes (v_writelane / v_readlane: VECTOR instructions,
// in kernels that are bound by vector issue).  Costs nothing at run time: the asm is empty.
template <typename P>
OL_DEV P refresh(P p) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+s"(p));
#else
  asm volatile("" : "+r"(p));
#endif
  return p;
}

// refresh(p) that is additionally ORDERED AFTER the computation of `acc` (the asm takes it
// as an in/out operand): loads through the result cannot be issued before `acc` exists.
// Used to meter a long coefficient stream -- chunk k + 2 is requested when chunk k has been
// consumed -- so that only a bounded window of it occupies SGPRs (CoeffStream below).
template <typename P, typename A>
OL_DEV P refresh_after(P p, A& acc) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+s"(p), "+v"(acc));
#else
  asm volatile("" : "+r"(p), "+x"(acc));
#endif
  return p;
}
}  // namespace ol

namespace ol {

enum : int {
  kGeomPlane = 0,
  kGeomStandard = 1,
  kGeomEvenAsphere = 2,
  kGeomZernike = 3,
  kGeomOddAsphere = 4,
  kGeomPolynomial = 5,
  kGeomChebyshev = 6,  // block: {1/norm_x, 1/norm_y, c[i][j]...}
  kGeomBiconic = 7,    // block: {cy, 1 + ky}; cv/kp1 hold cx, 1 + kx
  kGeomToroidal = 8,   // block: {R_rot (or inf), 1/R_rot, 1 + k_yz, c_yz, a_1, a_2, ...}
  // internal (never in ol_surface_desc): a Zernike surface of low order re-expressed by
  // ol_system_create as one bivariate polynomial of degree n_coeff in (x, y) / norm_radius
  kGeomZernikeMono = 9
};
enum : int { kRecordOnly = 0, kRefract = 1, kReflect = 2 };
enum : int {
  kApNone = 0, kApRadial = 1, kApOffsetRadial = 2, kApRect = 3, kApElliptical = 4,
  kApComposite = 5, kApPolygon = 6, kApOpUnion = 10, kApOpIntersection = 11, kApOpDifference = 12
};
constexpr int kApTokenLen = 5;   // {op, p0..p3} per reverse-Polish token
enum : int {
  kCoatNone = 0, kCoatSimple = 1, kCoatFresnel = 2, kCoatPolarizer = 3, kCoatRetarder = 4
};

// Newton-Raphson kernel families (template parameter NR of surface_step and of the kernels;
// surface_math.h: nr_eval)
constexpr int kNrNone = 0;       // conic-only range: no Newton code at all (lean kernel)
constexpr int kNrGeneric = 1;    // every functor
constexpr int kNrCompact = 2;    // every functor + wavefront straggler compaction (RPT > 1)
constexpr int kNrZernike = 3;    // Zernike surfaces only (level form and one-polynomial form)
constexpr int kNrEvenAsphere = 4;  // even aspheres only
constexpr int kNrReference = 5;  // every functor + the reference's batch-global stop rule
                                 // (OL_SURF_REFERENCE_NEWTON, opt-in: newton_reference)

// launch-uniform facts of a polarised trace (surface_math.h: interact)
constexpr uint32_t kPolNonUnitK = 0x1u;     // OL_TRACE_NONUNIT_K: direction cosines not unit

constexpr uint32_t kSurfRotated = 0x1u;     // this surface's own frame is rotated
constexpr uint32_t kSurfRelRotated = 0x2u;  // transform from the previous frame rotates
constexpr uint32_t kSurfRadiusInf = 0x4u;   // |R| = inf (standard.py:108-111 branch)
constexpr uint32_t kSurfReferenceRoot = 0x8u;  // OL_SURF_REFERENCE_ROOT on a curved standard surface

// Hot part: everything a plain conic surface needs, exactly 16 elements so the
// kernel fetches it with ONE s_load_dwordx16 (fp32) / two (fp64) per surface and
// can prefetch the next surface's block while it works on the current one.
template <typename T>
struct alignas(16) DevSurfHot {
  int32_t geom;
  int32_t interaction;
  int32_t aperture_kind;
  int32_t coating_kind;
  uint32_t flags;
  int32_t coeff_off;   // offset into the T coefficient buffer
  int32_t n_coeff;     // asphere: #C_i; polynomial: rows*cols; zernike: #groups
  int32_t max_iter;
  T cv;                // curvature 1/R (0 for |R| = inf)
  T kp1;               // 1 + conic
  T rel_off[3];        // local_s = rel_rot * local_{s-1} + rel_off
  T origin[3];         // global position of the local origin
};

// Cold part: only touched when a flag / kind in the hot part says so.
template <typename T>
struct DevSurfCold {
  int32_t poly_cols;
  int32_t coeff_len;   // number of T elements of this surface's coefficient block
  int32_t ap_off;      // composite aperture: first token in the coefficient buffer
  int32_t ap_len;      // composite aperture: number of tokens
  T tol;
  T inv_norm;          // 1 / norm_radius (zernike)
  T rot[9];            // local = rot * (global - origin)
  T rel_rot[9];
  T ap[4];             // radial: rmin^2, rmax^2, ox, oy; rect: xmin,xmax,ymin,ymax;
                       // elliptical: 1/a^2, 1/b^2, ox, oy
  T coat[2];           // simple coating: T, R
  T axis[3];           // polarizer / retarder axis (normalised)
  T ret_cos, ret_sin;  // retarder: cos(d/2), sin(d/2)
  T radius, conic;     // as given (kSurfReferenceRoot: the reference's own R-scaled quadratic)
};

// What the device functions see: the hot block BY VALUE (SGPRs) + a pointer to
// the cold block.
template <typename T>
struct DevSurf : DevSurfHot<T> {
  cptr<DevSurfCold<T>> cold;
};

template <typename T>
struct DevOptics {
  T n1;      // material_pre.n(lambda)
  T n2;      // material_post.n(lambda)
  T u;       // n1 / n2 (formed in double)
  T nn;      // n2 / n1 (Fresnel: jones.py:95)
  T absorb;  // 4 pi k1 / lambda * 1e3, 0 => skip
  T pad[3];
};

// Field-wise copies out of the constant address space (a struct assignment would have to
// bind a generic reference to constant-address-space memory).
template <typename T>
OL_DEV DevSurfHot<T> load_hot(cptr<DevSurfHot<T>> p) {
  DevSurfHot<T> h;
  h.geom = p->geom;
  h.interaction = p->interaction;
  h.aperture_kind = p->aperture_kind;
  h.coating_kind = p->coating_kind;
  h.flags = p->flags;
  h.coeff_off = p->coeff_off;
  h.n_coeff = p->n_coeff;
  h.max_iter = p->max_iter;
  h.cv = p->cv;
  h.kp1 = p->kp1;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    h.rel_off[k] = p->rel_off[k];
    h.origin[k] = p->origin[k];
  }
  return h;
}

template <typename T>
OL_DEV DevOptics<T> load_optics(cptr<DevOptics<T>> p) {
  DevOptics<T> o;
  o.n1 = p->n1;
  o.n2 = p->n2;
  o.u = p->u;
  o.nn = p->nn;
  o.absorb = p->absorb;
  return o;
}

// How a kernel hands one surface's table rows to surface_step() (surface_math.h).
//  * SurfLoaded: the hot block and the optics row were loaded ONCE, by value, and stay in
//    SGPRs for the whole surface body -- the lean (conic-only) kernels, where that is a
//    single s_load_dwordx16 prefetched one surface ahead and nothing spills;
//  * SurfFetched: three table pointers; every phase of the surface body (frame change,
//    intersection, each Newton iteration, interaction, record) re-reads the few fields it
//    needs through refresh() -- the Newton-Raphson kernels, which are short of SGPRs.
template <typename T>
struct SurfLoaded {
  const DevSurf<T>& s;
  const DevOptics<T>& o;
  OL_DEV const DevSurf<T>& surf() const { return s; }
  OL_DEV const DevOptics<T>& optics() const { return o; }
};

template <typename T>
struct SurfFetched {
  cptr<DevSurfHot<T>> hot;
  cptr<DevSurfCold<T>> cold;
  cptr<DevOptics<T>> opt;
  OL_DEV DevSurf<T> surf() const {
    DevSurf<T> S;
#ifdef OL_EXP_NOREFRESH
    static_cast<DevSurfHot<T>&>(S) = load_hot<T>(hot);
    S.cold = cold;
    return S;
  }
  OL_DEV DevOptics<T> optics() const { return load_optics<T>(opt); }
#else
    static_cast<DevSurfHot<T>&>(S) = load_hot<T>(refresh(hot));
    S.cold = refresh(cold);
    return S;
  }
  OL_DEV DevOptics<T> optics() const { return load_optics<T>(refresh(opt)); }
#endif
};

// Zernike block inside the coefficient array: one LEVEL per azimuthal order m that has a
// non-zero term, ascending in m (surface.n_coeff = number of levels):
//   [0] m >= 0   [1] K = number of powers of u = rho^2       -- INTEGER bit patterns
//                                                              (int32 in a float slot,
//                                                               int64 in a double slot)
//   then, for k = 0 .. K-1, six values (cos part, sin part interleaved so that a pair is
//   one 2-vector / one s_load_dwordx2):
//     a_c[k] a_s[k]   sag coefficient of u^k   (sum_j c_j N_j R_n^m, rho^m taken out)
//     b_c[k] b_s[k]   the same without N_j     (the reference's normal, zernike.py:234)
//     d_c[k] d_s[k]   (k + 1) b[k + 1]         (coefficients of dQ_b/du; 0 for k = K-1)
//   (m = 0 has no sin part: zeros.)
//
// kGeomZernikeMono block (surface.n_coeff = degree n), coefficients in Horner order -- rows
// by descending power of x, within a row by descending power of y:
//   S:  (n + 1)(n + 2) / 2 values   s_ij of x^i y^j, i + j <= n          (the sag, with N_j)
//   G:  n (n + 1) / 2 PAIRS         (dQ/dx, dQ/dy)_ij, i + j <= n - 1    (Q: without N_j)
constexpr int kZernLevelHeader = 2;
constexpr int kZernLevelStride = 6;

}  // namespace ol
