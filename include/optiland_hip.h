/*
 * optiland_hip.h -- C ABI of the MI355X-native sequential ray-trace hot path.
 *
 * This is the drop-in boundary for Optiland's batched real-ray trace
 * (reference: optiland/surfaces/surface_group.py:245-257 `SurfaceGroup.trace`,
 * optiland/surfaces/standard_surface.py:200-274 `Surface.trace/_trace_real/
 * _record_real`, driven by optiland/raytrace/real_ray_tracer.py:58-154).
 * The reference has no native interface (it is pure Python); these entry points
 * are what a ctypes binding inside `optiland.backend` would call -- see
 * INTEGRATION.md for the reference-side stub.
 *
 * Conventions
 *   - plain C, POD structs, no exceptions cross the boundary;
 *   - every function returns 0 on success or a negative OL_E* code and leaves a
 *     thread-local message readable through ol_last_error();
 *   - the library NEVER owns ray memory: every ray / record / PRT buffer is a
 *     caller-allocated DEVICE pointer (e.g. torch tensor .data_ptr());
 *   - the library owns only the opaque `ol_system` (surface table in HBM);
 *   - all launches go to the caller's stream (`hipStream_t` passed as void*).
 */
#ifndef OPTILAND_HIP_H
#define OPTILAND_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OL_ABI_VERSION 11

/* ---- error codes ------------------------------------------------------- */
#define OL_OK 0
#define OL_EINVAL (-1)      /* bad argument                                  */
#define OL_EUNSUPPORTED (-2)/* surface/coating kind the fast path refuses     */
#define OL_EHIP (-3)        /* HIP runtime error (message has hipGetErrorString) */
#define OL_ENOMEM (-4)

/* ---- arithmetic type of the ray buffers -------------------------------- */
/* A plain 32-bit integer, not an enum type: a foreign caller (ctypes, cgo, JNI ...) can pass
 * any value, and in the C++ implementation merely LOADING an out-of-range value of an enum
 * type is undefined behaviour (found by the UBSAN pass, tests/test_capi_sanitized.py); as an
 * integer it is validated and answered with OL_EINVAL.  Same size and calling convention. */
typedef int32_t ol_dtype;
enum { OL_F32 = 0, OL_F64 = 1 };

/* ---- geometry kinds (reference class -> enum) ---------------------------
 * PLANE         optiland/geometries/plane.py:72-109
 * STANDARD      optiland/geometries/standard.py:97-175 (conic; |R|=inf allowed)
 * EVEN_ASPHERE  optiland/geometries/even_asphere.py:93-140 (+ Newton-Raphson,
 *               optiland/geometries/newton_raphson.py:119-168)
 * ZERNIKE       optiland/geometries/zernike.py:153-266
 * ODD_ASPHERE   optiland/geometries/odd_asphere.py:86-143
 * POLYNOMIAL    optiland/geometries/polynomial.py:105-155 (x^i y^j freeform)
 * CHEBYSHEV     optiland/geometries/chebyshev.py:126-241; coefficient block =
 *               {norm_x, norm_y, c[0][0], c[0][1], ...}: n_coeff = rows*cols,
 *               poly_cols = cols
 * BICONIC       optiland/geometries/biconic.py:69-158; radius/conic = Rx/kx,
 *               coefficient block = {Ry, ky} (n_coeff = 2)
 * TOROIDAL      optiland/geometries/toroidal.py:86-242; radius = YZ radius (base
 *               conic of the Newton start, conic = 0), coefficient block =
 *               {R_rot, k_yz, a_1, a_2, ...} (n_coeff = 2 + number of y^2i terms)
 */
typedef enum ol_geom_kind {
  OL_GEOM_PLANE = 0,
  OL_GEOM_STANDARD = 1,
  OL_GEOM_EVEN_ASPHERE = 2,
  OL_GEOM_ZERNIKE = 3,
  OL_GEOM_ODD_ASPHERE = 4,
  OL_GEOM_POLYNOMIAL = 5,
  OL_GEOM_CHEBYSHEV = 6,
  OL_GEOM_BICONIC = 7,
  OL_GEOM_TOROIDAL = 8
} ol_geom_kind;

/* ---- what happens at the surface ----------------------------------------
 * RECORD_ONLY  ObjectSurface.trace, optiland/surfaces/object_surface.py:56-93
 * REFRACT      RealRays.refract,    optiland/rays/real_rays.py:163-187
 * REFLECT      RealRays.reflect,    optiland/rays/real_rays.py:189-205
 */
typedef enum ol_interaction {
  OL_INTERACT_RECORD_ONLY = 0,
  OL_INTERACT_REFRACT = 1,
  OL_INTERACT_REFLECT = 2
} ol_interaction;

/* ---- physical apertures (optiland/physical_apertures/ *.py) --------------
 * aperture[] meaning per kind:
 *   RADIAL         r_min, r_max                        radial.py:56-70
 *   OFFSET_RADIAL  r_min, r_max, offset_x, offset_y    offset_radial.py:48-61
 *   RECTANGULAR    x_min, x_max, y_min, y_max          rectangular.py:42-59
 *   ELLIPTICAL     a, b, offset_x, offset_y            elliptical.py:42-56
 *   COMPOSITE      boolean tree of the above (Union / Intersection / Difference,
 *                  base.py:259-340) flattened to reverse-Polish tokens stored in
 *                  coeffs[]: aperture[0] = first token's index in coeffs[],
 *                  aperture[1] = token count; a token is 5 doubles
 *                  {op, p0, p1, p2, p3}: op = a leaf kind above (p = its
 *                  parameters) or OL_AP_OP_UNION / _INTERSECTION / _DIFFERENCE
 *                  (pops b then a, pushes a|b, a&b, a&~b).
 */
typedef enum ol_aperture_kind {
  OL_AP_NONE = 0,
  OL_AP_RADIAL = 1,
  OL_AP_OFFSET_RADIAL = 2,
  OL_AP_RECTANGULAR = 3,
  OL_AP_ELLIPTICAL = 4,
  OL_AP_COMPOSITE = 5,
  OL_AP_POLYGON = 6   /* PolygonAperture / FileAperture (physical_apertures/polygon.py):
                         parameters {first vertex index in the coefficient buffer,
                         vertex count}; vertices stored x0, y0, x1, y1, ...; inside =
                         matplotlib's crossings test (Path.contains_points, radius 0),
                         which is what the NumPy backend calls                      */
} ol_aperture_kind;

#define OL_AP_OP_UNION 10
#define OL_AP_OP_INTERSECTION 11
#define OL_AP_OP_DIFFERENCE 12
#define OL_AP_TOKEN_DOUBLES 5
#define OL_AP_MAX_DEPTH 16

/* ---- coatings (optiland/coatings.py) --------------------------------------
 *   SIMPLE   i *= T (refract) | R (reflect)            coatings.py:164-237
 *   FRESNEL  Jones diag(s,p,1) from Fresnel equations  coatings.py:362-386,
 *            jones.py:71-117 (needs a PRT buffer, i.e. polarized rays)
 *   POLARIZER linear polarizer, transmission axis a    coatings.py:418-447,
 *            jones.py:120-181; coat[0] = index in coeffs[] of {ax, ay, az}
 *            (normalised, as JonesLinearPolarizer stores it)
 *   RETARDER linear retarder, fast axis a, retardance d coatings.py:450-485,
 *            jones.py:331-393; coat[0] = index in coeffs[] of {ax, ay, az, d};
 *            its Jones matrix is complex: needs OL_TRACE_PRT_COMPLEX
 */
typedef enum ol_coating_kind {
  OL_COAT_NONE = 0,
  OL_COAT_SIMPLE = 1,
  OL_COAT_FRESNEL = 2,
  OL_COAT_POLARIZER = 3,
  OL_COAT_RETARDER = 4
} ol_coating_kind;

#define OL_SURF_ROTATED 0x1u /* rot[] is not the identity */
#define OL_SURF_REFERENCE_ROOT 0x2u /* ABI 10, OL_GEOM_STANDARD with a finite radius only: the
                                       intersection in the reference's own form, (-b +- sqrt d) /
                                       2a with R-scaled coefficients (geometries/standard.py:
                                       112-146), instead of the cancellation-free one -- for
                                       callers who need the reference's NUMBERS where its formula
                                       is ill conditioned (|1 + k| << 1: goldens generated with it
                                       encode its systematic error, e.g. tests/test_operand.py
                                       test_opd_diff_on_axis).  Default off: less accurate         */
#define OL_SURF_REFERENCE_NEWTON 0x4u /* ABI 11, Newton-Raphson geometries only: the reference's
                                       OWN stop rule.  NewtonRaphsonGeometry.distance
                                       (geometries/newton_raphson.py:119-168) iterates the whole
                                       batch in lockstep and stops when max_j |f_j| < tol, so every
                                       ray of a trace call takes the SAME number K of updates
                                       (max_iter when any ray of the batch is NaN), and the surface
                                       normal is evaluated at the end point
                                       (surfaces/standard_surface.py:200-258).  The default kernels
                                       stop per ray and converge further (<= 1e-7 with the factory
                                       tolerance; more with a user-set loose one).  K is a property
                                       of the batch: it is found by ol_newton_count and handed to
                                       ol_trace_ex in ol_trace_extras.newton_iterations; the fused
                                       entry points refuse such a system (OL_EUNSUPPORTED)         */

/* One traced surface.  All lengths in mm, angles already folded into rot[]. */
typedef struct ol_surface_desc {
  int32_t geom_kind;     /* ol_geom_kind                                      */
  int32_t interaction;   /* ol_interaction                                    */
  int32_t aperture_kind; /* ol_aperture_kind                                  */
  int32_t coating_kind;  /* ol_coating_kind                                   */
  int32_t coeff_offset;  /* first element of this surface's block in coeffs[] */
  int32_t n_coeff;       /* EVEN/ODD_ASPHERE: number of C_i;
                            ZERNIKE: number of terms (4 doubles per term:
                            c_j, n_j, m_j, N_j -- coefficient, radial order,
                            azimuthal order, normalisation constant);
                            POLYNOMIAL: rows*cols of the c[i][j] grid (cols in
                            poly_cols)                                         */
  int32_t max_iter;      /* Newton-Raphson iteration cap (geometry.max_iter)  */
  uint32_t flags;        /* OL_SURF_*                                         */
  int32_t poly_cols;     /* POLYNOMIAL only: number of columns (y powers)     */
  int32_t reserved_;
  double radius;         /* radius of curvature (may be +-inf)                */
  double conic;          /* conic constant k                                  */
  double tol;            /* Newton-Raphson tolerance (geometry.tol)           */
  double norm_radius;    /* ZERNIKE normalisation radius                      */
  double origin[3];      /* global position of the local origin               */
  double rot[9];         /* row-major R: local = R * (global - origin)
                            (= Rx(-rx) Ry(-ry) Rz(-rz), with parent frames
                            folded in; coordinate_system.py:73-107)           */
  double aperture[4];
  double coat[2];        /* SIMPLE: transmittance, reflectance                */
} ol_surface_desc;

/* Per (surface, wavelength) optical constants, laid out [surface][wavelength].
 *   n1      = material_pre.n(lambda)   standard_surface.py:244, real_rays.py:175
 *   n2      = material_post.n(lambda)
 *   absorb  = 4*pi*k_pre(lambda)/lambda * 1e3  [1/mm]; 0 => no Beer-Lambert
 *             step (propagation/homogeneous.py:44-53)
 */
typedef struct ol_surface_optics {
  double n1;
  double n2;
  double absorb;
} ol_surface_optics;

typedef struct ol_system ol_system; /* opaque */

/* ---- status bits written (OR-ed) into the device status word -------------
 * Data-dependent conditions the reference turns into Python exceptions.   */
#define OL_STATUS_ZERNIKE_RANGE 0x1u /* |x/norm|>1 or |y/norm|>1
                                        (geometries/zernike.py:254-266)       */
#define OL_STATUS_K_PARALLEL_X 0x2u  /* initial k parallel to x-hat
                                        (rays/polarized_rays.py:221-222)      */
#define OL_STATUS_FIELD_RANGE 0x8u   /* a normalised field coordinate outside [-1, 1]
                                        (real_ray_tracer.py:156-173)             */
#define OL_STATUS_PUPIL_RANGE 0x10u  /* a normalised pupil coordinate outside [-1, 1] */
#define OL_STATUS_CHEBYSHEV_RANGE 0x4u /* |x/norm_x|>1 or |y/norm_y|>1
                                        (geometries/chebyshev.py:227-240)     */
#define OL_STATUS_NAN_DIRECTION 0x20u /* ABI 10, INFORMATIONAL (no exception in the reference):
                                        ol_trace / ol_trace_generate left a ray with a finite
                                        position and a direction that is not a number -- total
                                        internal reflection at the LAST traced surface.  The
                                        reference's trace ends with `x += t L` by the last
                                        thickness (real_ray_tracer.py:104-110), which turns such
                                        a position into NaN even for t = 0; a caller that hands
                                        out the final state as the reference's returned rays
                                        applies that when (and only when) this bit is set.   */

/* ---- trace flags --------------------------------------------------------- */
#define OL_TRACE_WRITE_RAYS 0x1u   /* write the final ray state back into rays[] */
#define OL_TRACE_PRT_COMPLEX 0x4u  /* prt holds 18 planes: 9 real then 9 imaginary */
#define OL_TRACE_PRT_IDENTITY 0x8u /* prt is write-only: start from the identity instead
                                      of reading it (PolarizedRays.__init__,
                                      rays/polarized_rays.py:50) -- saves 9 plane reads
                                      and the caller's fill                          */
#define OL_TRACE_COMPACT 0x2u      /* allow wavefront straggler compaction in the
                                      Newton loop (only if OL_TUNE_COMPACT=1) */
#define OL_TRACE_FEW_WAVES 0x10u   /* ABI 10, record-all launches of ol_trace /
                                      ol_trace_generate: the caller's word that the record
                                      block is NOT in a placed window (an ordinary
                                      allocation).  The fp32 conic-only unpolarised kernels
                                      then run with at most two workgroups resident per CU
                                      (an untouched dynamic-LDS request): fewer stores in
                                      flight write ~3 % faster there.  Results are
                                      identical; ignored by every other kernel.         */
#define OL_TRACE_NONUNIT_K 0x20u   /* ABI 11, polarised ol_trace / ol_trace_ex launches: the
                                      caller's word that the direction cosines of rays[] are
                                      NOT unit vectors -- what the reference's iterative /
                                      robust ray aimers hand out (|k|^2 - 1 ~ 1e-3,
                                      rays/ray_aiming/iterative.py:339-366).  The reference's
                                      PRT algebra takes k as it comes (its triads
                                      (s, k0 x s, k0) are then not orthonormal,
                                      rays/polarized_rays.py:136-202); with this flag the
                                      kernels form the same matrices: the rank-2 update on
                                      the normalised directions with the p and k amplitudes
                                      scaled by |k0| |k1|.  Without it the update equals the
                                      reference's only for |k| = 1 to rounding.  An uncoated
                                      refracting surface between EQUAL indices is then not
                                      the identity: the reference's s there is the rounding
                                      noise of k0 x k1, the kernels take s = k0 x n.       */

/* Polarisation state for the update_intensity epilogue
 * (rays/polarized_rays.py:122-133, rays/polarization_state.py:29-56).      */
typedef struct ol_polarization_state {
  int32_t is_polarized; /* 0 => unpolarised: mean of the x and y states      */
  int32_t reserved_;
  double Ex, Ey, phase_x, phase_y;
} ol_polarization_state;

/* Build the device-resident surface table.
 *   surf[n_surf]                  traced surfaces in order (object first)
 *   coeffs[n_coeffs]              flat coefficient buffer (may be NULL if 0)
 *   optics[n_surf*n_wavelengths]  per-surface, per-wavelength constants
 * Replaces: the per-call walk over Surface objects in SurfaceGroup.trace
 * (surfaces/surface_group.py:245-257).                                      */
int ol_system_create(const ol_surface_desc* surf, int32_t n_surf,
                     const double* coeffs, int32_t n_coeffs,
                     const ol_surface_optics* optics, int32_t n_wavelengths,
                     ol_system** out);
/* ABI 6.  Rewrite the device tables of an EXISTING system from a new description -- same
 * arguments as ol_system_create -- in place: no allocation, four small host-to-device copies
 * per precision queued on `stream` (so ordered after every launch already queued there that
 * reads the old tables).  For the callers that edit a prescription between traces
 * (optimisers, tolerancing loops: `Optic.updater.set_radius(...)` then `Optic.trace(...)`),
 * where create + destroy per edit would dominate a small trace.
 * Returns OL_EUNSUPPORTED -- the system is untouched, create a new one -- when the new
 * description does not fit the existing allocations (other surface / wavelength counts, a
 * device coefficient block beyond the allocated capacity).                               */
int ol_system_update(ol_system* sys, const ol_surface_desc* surf, int32_t n_surf,
                     const double* coeffs, int32_t n_coeffs,
                     const ol_surface_optics* optics, int32_t n_wavelengths, void* stream);
void ol_system_destroy(ol_system* sys);
int32_t ol_system_num_surfaces(const ol_system* sys);

/* Trace n_rays through surfaces [first_surface, last_surface] (inclusive).
 *   rays[8]  device pointers x,y,z,L,M,N,i,opd (each n_rays elements of dt);
 *            read as the initial state; rewritten iff OL_TRACE_WRITE_RAYS.
 *   record   nullable; (last-first+1) rows x 8 planes x record_stride
 *            elements: row s, plane k at record + ((s*8+k)*record_stride).
 *            Planes are x,y,z,L,M,N,intensity,opd in the GLOBAL frame, as
 *            Surface._record_real stores them (standard_surface.py:260-274).
 *            Zero-copy object row: if rays[k] == record + k*record_stride for all
 *            k (the rays were generated straight into row 0) and the first
 *            surface is RECORD_ONLY, row 0 is left as is instead of rewritten.
 *   prt      nullable; 9 planes x n_rays (row-major 3x3 polarisation
 *            ray-tracing matrix, real part) read-modify-write:
 *            PolarizedRays.update (rays/polarized_rays.py:180-202).  With
 *            OL_TRACE_PRT_COMPLEX: 18 planes, real part then imaginary part
 *            (required when the range holds a RETARDER coating).
 *   status   nullable device uint32; OL_STATUS_* bits are OR-ed in.
 * Replaces: SurfaceGroup.trace(rays, skip) -- `first_surface` generalises
 * `skip`.                                                                    */
int ol_trace(const ol_system* sys, ol_dtype dt, int64_t n_rays,
             void* const rays[8], int32_t wavelength_index,
             void* record, int64_t record_stride, void* prt,
             int32_t first_surface, int32_t last_surface, uint32_t flags,
             uint32_t* status, void* stream);

/* ol_trace with optional extras.  `spot_slots` (nullable): the launch also reduces the
 * FINAL ray state (global frame, last traced surface) to the masked spot moments of
 * ol_trace_spot about (cx, cy), as an epilogue of the same kernel -- no second pass
 * over the image-plane planes.  Layout: OL_SPOT_SLOTS x 8 device doubles, ACCUMULATED
 * (zero them first); workgroups spread their atomics over the slots, the consumer
 * adds slots up: elements 0..5 = {count, sum dx, sum dy, sum dx^2, sum dy^2, sum i}
 * (sum over slots), element 6 = max r^2 (max over slots), element 7 unused.      */
#define OL_SPOT_SLOTS 64
typedef struct ol_trace_extras {
  double* spot_slots;
  double cx, cy;
  /* ABI 6.  First RECORDED surface: row 0 of `record` then holds surface
   * `record_first_surface` and only last_surface - record_first_surface + 1 rows are
   * written -- e.g. last_surface - 1 for a consumer that reads the image plane and the
   * pre-interaction direction cosines only (RealRays.L0/M0/N0, real_rays.py:170-172), which
   * is what the drop-in's lazy-record mode launches.  Values <= first_surface (0 in a
   * zero-initialised struct) record every traced surface, as before.                    */
  int32_t record_first_surface;
  int32_t reserved_;
  /* ABI 7, ol_trace_generate with a `prt` only.  PolarizedRays.update_intensity
   * (rays/polarized_rays.py:68-133, what RealRayTracer.trace applies to a polarised bundle,
   * raytrace/real_ray_tracer.py:112-113) as an EPILOGUE of the same launch: the intensity
   * i0 |P E0|^2 of every ray from the polarisation ray-tracing matrix the kernel still holds
   * in registers and the direction it generated the ray with -- instead of a second launch
   * (ol_polarized_intensity) that reads the nine PRT planes, three direction planes and the
   * intensity plane back.  `updated_intensity`: n values of the ray dtype (device), written;
   * NULL (or a NULL state): no epilogue.  The PRT planes are written either way.        */
  const ol_polarization_state* update_intensity_state;
  void* updated_intensity;
  /* ABI 11, ranges with OL_SURF_REFERENCE_NEWTON surfaces (required there, ignored elsewhere).
   * `newton_iterations`: DEVICE array of 2 * n_surfaces int32 (ol_system_num_surfaces), zeroed
   * by the caller before the first ol_newton_count of a trace call: element s = the number of
   * Newton updates every ray takes at surface s; element n_surfaces + s is raised (non-zero)
   * by any launch that finds a ray NOT below the surface's tolerance after those updates
   * although the count is below max_iter -- the caller then adds one to element s, clears the
   * word and asks again (ol_newton_count with `verify`).  `newton_count_surface`: leave 0
   * (ol_newton_count sets it).                                                             */
  int32_t* newton_iterations;
  int32_t newton_count_surface;
  int32_t reserved2_;
} ol_trace_extras;

int ol_trace_ex(const ol_system* sys, ol_dtype dt, int64_t n_rays,
                void* const rays[8], int32_t wavelength_index,
                void* record, int64_t record_stride, void* prt,
                int32_t first_surface, int32_t last_surface, uint32_t flags,
                uint32_t* status, const ol_trace_extras* extras, void* stream);

/* ABI 11.  The iteration count of ONE reference-rule Newton surface for this batch of rays
 * (see OL_SURF_REFERENCE_NEWTON): the rays are traced from `first_surface` up to `surface` --
 * nothing is recorded or written back; every Newton surface of the range before `surface`
 * takes the count already in `iterations` -- and
 *   verify == 0:  element `surface` of `iterations` becomes the maximum over the batch of each
 *                 ray's first k with |f_k| < tol (max_iter for a ray that is NaN or never gets
 *                 there): newton_raphson.py:148's break index, unless a ray that was below tol
 *                 is above it again at that k;
 *   verify != 0:  every ray takes exactly iterations[surface] updates and the violation word
 *                 n_surfaces + surface is raised if the stop rule does not hold then.
 * Call it for the reference-rule surfaces of a range in ascending order (each count depends on
 * the ones before it), with the SAME rays, wavelength and first surface as the ol_trace_ex that
 * follows; a sharded batch takes the maximum of element `surface` over its shards after every
 * call (one int32 all-reduce).  Asynchronous on `stream`.                                   */
int ol_newton_count(const ol_system* sys, ol_dtype dt, int64_t n_rays, void* const rays[8],
                    int32_t wavelength_index, int32_t first_surface, int32_t surface,
                    int32_t* iterations, int32_t verify, void* stream);

/* Generate rays on device from normalised field/pupil coordinates: angle fields
 * (object at infinity or finite, fields/field_types/angle.py:17-58) and object-height
 * fields on a planar object (fields/field_types/object_height.py:19-47), paraxial
 * aiming incl. the object-space-telecentric branch (rays/ray_generator.py:47-99,
 * rays/ray_aiming/paraxial.py:33-106).  See ol_raygen_params.                   */
#define OL_FIELD_ANGLE 0
#define OL_FIELD_OBJECT_HEIGHT 1
#define OL_FIELD_PARAXIAL_IMAGE_HEIGHT 2 /* fields/field_types/paraxial_image_height.py:
                                            the normalised field maps LINEARLY to the
                                            object-space chief-ray slope (object at
                                            infinity) or to the object height (finite
                                            object); `max_field` then carries that
                                            host-computed scale: u_obj_unit (or
                                            y_obj_unit) * max_field / y_img_unit        */
typedef struct ol_raygen_params {
  int32_t object_infinite; /* obj.is_infinite                                 */
  int32_t field_kind;      /* OL_FIELD_ANGLE | _OBJECT_HEIGHT | _PARAXIAL_IMAGE_HEIGHT */
  double EPL, EPD;         /* paraxial entrance pupil location / diameter     */
  double max_field;        /* degrees (angle), lens units (object height), or the
                              slope / height scale of a paraxial image height field */
  double offset;           /* AngleField._get_starting_z_offset               */
  double z_first;          /* surfaces.positions[1] (infinite) or [0] (finite)*/
  double tele_dz;          /* 0: aim at the paraxial entrance pupil
                              (ray_aiming/paraxial.py:88-94); > 0: object-space
                              telecentric, sqrt(1 - sin^2)/sin of the object NA
                              (:82-87): the target plane sits tele_dz behind the
                              object point and the pupil offsets are ABSOLUTE     */
  double apod_a, apod_b;   /* apodization parameters, see OL_APOD_*                */
  int32_t apod_kind;       /* initial intensity i = A(Px, Py), ray_generator.py:81-85 */
  int32_t reserved_;
} ol_raygen_params;

/* optiland/apodization/: r = sqrt(Px^2 + Py^2) of the pupil coordinates handed to the
 * ray generator (after trace_generic's pre-scaling)                                */
#define OL_APOD_NONE 0           /* 1 (uniform.py)                                 */
#define OL_APOD_GAUSSIAN 1       /* exp(-r^2 / (2 a^2)),            a = sigma      */
#define OL_APOD_COSINE_SQUARED 2 /* cos^2(pi r / (2 a)) for r < a else 0, a = R    */
#define OL_APOD_HANN 3           /* (1 - cos(2 pi r / a)) / 2 for r < a/2 else 0, a = D */
#define OL_APOD_POLYNOMIAL 4     /* (1 - (r/a)^2)^b for r < a else 0, a = R, b = p */
#define OL_APOD_SUPER_GAUSSIAN 5 /* exp(-(r / a)^b),                a = w, b = n   */
#define OL_APOD_TUKEY 6          /* 1 for r <= a(1 - b/2); (1 + cos(pi (r - a(1-b/2)) /
                                    (a b / 2))) / 2 up to r < a; else 0; a = R, b = alpha */

/* Normalised coordinates of one ray block.  Each of the pairs (hx,hy) and (vx,vy)
 * is either two device planes of n elements or both NULL, in which case the
 * launch-uniform scalars are used (one field point per launch is the common case:
 * Optic.trace(Hx, Hy, ...) with scalar Hx, Hy).                                  */
typedef struct ol_raygen_inputs {
  const void *hx, *hy;  /* normalised field, per ray, or NULL -> hx0, hy0          */
  const void *px, *py;  /* normalised pupil, per ray (required)                    */
  const void *vx, *vy;  /* 1 - vignetting factor, per ray, or NULL -> vx0, vy0     */
  double hx0, hy0;
  double vx0, vy0;      /* 1 when unvignetted                                       */
  uint32_t flags;       /* OL_RAYGEN_*                                              */
  uint32_t reserved_;
} ol_raygen_inputs;

#define OL_RAYGEN_CHECK_FIELD 0x1u /* OR OL_STATUS_FIELD_RANGE into *status when a field
                                      coordinate lies outside [-1, 1]
                                      (RealRayTracer._validate_normalized_coordinates,
                                      real_ray_tracer.py:156-173) -- checked where the
                                      kernel reads the value anyway                 */
#define OL_RAYGEN_CHECK_PUPIL 0x2u /* same for the pupil coordinates
                                      (OL_STATUS_PUPIL_RANGE); trace_generic only   */
#define OL_RAYGEN_PRESCALE_PUPIL 0x4u /* trace_generic semantics: the pupil is scaled
                                      by (1 - v) BEFORE ray generation
                                      (real_ray_tracer.py:134-137) and the aimer
                                      scales it again (ray_aiming/paraxial.py:60-62,
                                      90-91) -- SURVEY.md Appendix D                */

/* ABI 10, ol_trace_spot / ol_trace_spot_batch only:                                        */
#define OL_SPOT_POLARIZED_OK 0x8u  /* the optic HAS a polarization state.  What a spot diagram
                                      reads of a polarised trace is the RECORDED last row
                                      (analysis/spot_diagram/core.py:462-468): positions and the
                                      geometric intensity -- clipping, absorption, SimpleCoating;
                                      Fresnel / polarizer / retarder coatings only act on the PRT
                                      matrix (interactions/base.py:111-128, rays/polarized_rays.py:
                                      136-202) and update_intensity's result never reaches that row
                                      (SURVEY.md Appendix D).  With this flag systems with
                                      polarization-dependent coatings are traced instead of refused */
#define OL_SPOT_HITS_LOCAL 0x10u   /* hits and moments in the LAST surface's own frame instead of
                                      the global one: SpotDiagram(coordinates="local") on a tilted
                                      image surface (visualization/system/utils.py:17-47)           */

/* out[0..6]: x,y,z,L,M,N,i planes (i = 1); out[7]: optional opd plane, zero-filled
 * (NULL to skip).  status: nullable unless a CHECK flag is set.                 */
int ol_generate_rays(const ol_raygen_params* p, ol_dtype dt, int64_t n,
                     const ol_raygen_inputs* in, void* const out[8],
                     uint32_t* status, void* stream);

/* ABI 6.  ol_generate_rays + ol_trace in ONE launch ("fuse generation into
 * the trace kernel"): each lane builds its ray from the normalised pupil point and the
 * launch-uniform field exactly like ol_generate_rays (same device code), records it as the
 * object row and walks surfaces [0, num_surfaces).  Replaces the chain
 * RayGenerator.generate_rays -> SurfaceGroup.trace of Optic.trace / trace_generic
 * (raytrace/real_ray_tracer.py:58-154, rays/ray_generator.py:47-99) for one field point:
 * no generator launch, and the object row is written once instead of written by one kernel
 * and read back by the next.
 *   in        px, py planes; launch-uniform field (hx0, hy0) and vignetting (vx0, vy0).
 *             ABI 8, unpolarised launches (prt == NULL): per-ray field planes hx, hy (and,
 *             with them, per-ray vignetting planes vx, vy) -- trace_generic(Hx[], Hy[], Px[],
 *             Py[]) and the fields x pupil expansion of a multi-field trace
 *             (real_ray_tracer.py:88-98, 120-154) -- and apodized pupils (p->apod_kind: the
 *             initial intensity is the apodization, ray_generator.py:81-85) are ONE launch
 *             too -- since ABI 10 also for POLARISED launches (prt != NULL, with or without the
 *             update_intensity epilogue, whose initial intensity is then the apodization:
 *             rays/polarized_rays.py:51).  vx, vy planes without hx, hy are refused with
 *             OL_EUNSUPPORTED
 *   record    required; rows as in ol_trace (row 0 = the generated rays unless
 *             extras->record_first_surface says otherwise)
 *   rays_out  NULL, or 8 planes receiving the final state (what OL_TRACE_WRITE_RAYS writes)
 *   prt       NULL, or the 9- / 18-plane PRT buffer, WRITE-ONLY (starts from the identity)
 *   flags     OL_TRACE_PRT_COMPLEX only; extras: record_first_surface, (ABI 7) the
 *             update_intensity epilogue of a polarised launch, and (ABI 8) spot_slots /
 *             cx / cy -- the masked image-plane moments of ol_trace_ex as an epilogue of
 *             the GENERATING launch (unpolarised, one field point, no apodization): the
 *             per-step form of a sharded trace whose exchange is the moments
 *             (distributed.py) -- generate, trace, record and reduce in one launch        */
int ol_trace_generate(const ol_system* sys, ol_dtype dt, int64_t n_rays,
                      const ol_raygen_params* p, const ol_raygen_inputs* in,
                      int32_t wavelength_index, void* record, int64_t record_stride,
                      void* const rays_out[8], void* prt, uint32_t flags, uint32_t* status,
                      const ol_trace_extras* extras, void* stream);


/* update_intensity epilogue for polarised traces (trace() only):
 * i = sum_fields |P E0|^2 * i0 / n_fields.  k0[3]: initial direction planes.  */
int ol_polarized_intensity(ol_dtype dt, int64_t n_rays, const void* prt,
                           int32_t prt_complex /* 0: 9 planes, 1: 18 planes */,
                           const void* const k0[3], const void* i0,
                           const ol_polarization_state* state, void* intensity,
                           uint32_t* status, void* stream);

/* ABI 8.  The reference's deterministic pupil samplers (optiland/distribution.py:161-220) on
 * the device, one point per lane from its index: no host sampling pass and no upload in front
 * of the first trace of a `num_rays` / distribution pair (1e7 hexapolar points: ~0.1 s of
 * NumPy + an 80 MB copy before).  Same values in the same ORDER as the reference's samplers
 * (fp64 arithmetic as NumPy's, then rounded to the ray dtype; cos / sin are the device libm's:
 * <= 1 ulp from NumPy's in fp64).
 *   OL_PUPIL_HEXAPOLAR  param = number of rings R; n_points must be 1 + 3 R (R + 1)
 *   OL_PUPIL_UNIFORM    param = grid side n (>= 2); the disc mask comes from the caller as
 *                       two DEVICE tables: row_first[n] = first kept column of row j,
 *                       row_offset[n + 1] = index of row j's first point (row_offset[n] =
 *                       n_points); both NULL for the hexapolar sampler                     */
#define OL_PUPIL_HEXAPOLAR 0
#define OL_PUPIL_UNIFORM 1
int ol_pupil_points(int32_t kind, int32_t param, ol_dtype dt, int64_t n_points,
                    const int32_t* row_first, const int64_t* row_offset, void* x, void* y,
                    void* stream);

/* ABI 8, diagnostics.  The kernels' arithmetic primitives applied element-wise on the
 * device: op 0 = 1 / a, 1 = a / b, 2 = sqrt(a), 3 = 1 / sqrt(a), exactly as the trace kernels
 * form them (fp32: v_rcp_f32 / v_sqrt_f32 / v_rsq_f32, 1 ulp; fp64: the hardware seeds
 * refined by two Newton steps, ~1 ulp, IEEE special values except an infinite numerator /
 * a denormal divisor).  For tests that pin those error bounds; b may be NULL unless op == 1. */
int ol_math_probe(int32_t op, ol_dtype dt, int64_t n, const void* a, const void* b, void* out,
                  void* stream);

/* ABI 8, diagnostics.  Write-only streaming yardstick: the store pattern of a record-all
 * trace launch with the arithmetic taken out.  `dst` is viewed as `planes` planes of
 * bytes / planes each; every lane stores ONE element of `store_bytes` (4, 8 or 16) into each
 * plane with the non-temporal stores of the trace kernels.  bench.py times it on a buffer the
 * size and shape of what the trace launch writes: the ceiling a kernel that only writes
 * reaches on this part for that footprint, next to the 8 TB/s of the data sheet.        */
int ol_stream_fill(void* dst, int64_t bytes, int32_t store_bytes, int32_t planes,
                   uint32_t pattern, void* stream);

/* ABI 11.  Device memory of the library's OWN: the arenas in which the host side looks for
 * fast record windows (engine.py: RecordPool, alloc_record_placed) come from here -- hipMalloc
 * on the current device -- and go back with ol_arena_free (hipFree: waits for the device).
 * Until round 6 they were blocks of the caller's caching allocator, and handing an arena
 * back meant emptying THAT cache: the drop-in now never touches it.  OL_EHIP when the device
 * cannot spare `bytes` (nothing is allocated; the caller goes without a pool).            */
int ol_arena_alloc(int64_t bytes, void** out);
int ol_arena_free(void* arena);

/* Image-plane reductions for one ray block (analysis/spot_diagram/core.py:
 * 329-372): out[0..5] += {sum w, sum w x, sum w y, sum w x^2, sum w y^2,
 * count} with w = (i>0 ? 1 : 0) -- the masked centroid / RMS building blocks;
 * out[6] = max r^2 about (cx,cy) is filled by ol_spot_max_r2.  Device doubles. */
int ol_spot_moments(ol_dtype dt, int64_t n_rays, const void* x, const void* y,
                    const void* intensity, double* out6, void* stream);
int ol_spot_max_r2(ol_dtype dt, int64_t n_rays, const void* x, const void* y,
                   const void* intensity, double cx, double cy, double* out1,
                   void* stream);

/* Encircled-energy building block (analysis/encircled_energy.py:147-160): for the radii
 * r_step[0..n_steps) (ascending, device doubles) accumulate
 *   bins[j] += sum of intensity over rays whose radius about (cx, cy) satisfies
 *              r_step[j-1] < r <= r_step[j]        (r <= r_step[0] for j = 0),
 * so that cumsum(bins)[j] = nansum(energy[radii <= r_step[j]]) -- the reference's
 * curve.  Rays with NaN radius or NaN intensity are skipped (nansum / `<=`), rays
 * beyond r_step[n_steps-1] fall outside.  bins: n_steps device doubles, accumulated. */
int ol_radial_energy(ol_dtype dt, int64_t n_rays, const void* x, const void* y,
                     const void* intensity, double cx, double cy, const double* r_step,
                     int32_t n_steps, double* bins, void* stream);

/* Detector irradiance building block (analysis/irradiance.py:341-353): numpy.histogram2d
 * of the ray hits weighted by power, with explicit edges:
 *   hist[ix * ny + iy] += power  for rays with power > 0 and
 *   x_edges[ix] <= x < x_edges[ix+1], y_edges[iy] <= y < y_edges[iy+1]
 * (the LAST bin of each axis also takes its right edge, as numpy does); rays outside
 * the edges, with non-finite coordinates or with power <= 0 / NaN are dropped.
 * x_edges: nx + 1, y_edges: ny + 1 ascending device doubles; hist: nx * ny device
 * doubles, accumulated (so ray shards -- and ranks, through an all-reduce of the
 * bins -- add up: the reduction that replaces the all-gather of hits for imaging
 * consumers).                                                        */
int ol_irradiance(ol_dtype dt, int64_t n_rays, const void* x, const void* y,
                  const void* power, const double* x_edges, int32_t nx,
                  const double* y_edges, int32_t ny, double* hist, void* stream);

/* Fused spot pipeline for one ray block: generate -> trace the whole sequence ->
 * reduce, in ONE kernel (SURVEY.md 8 f1 + f2).  The rays never exist in HBM: each
 * lane builds its rays from the normalised coordinates exactly like
 * ol_generate_rays, walks surfaces [0, num_surfaces) like ol_trace and folds the
 * image-plane hit into masked sums about (cx, cy).  Replaces, for unpolarised
 * systems, the chain Optic.trace -> surface_group.x/y/intensity[-1] ->
 * SpotDiagram.centroid / rms_spot_radius / geometric_spot_radius
 * (raytrace/real_ray_tracer.py:58-118, analysis/spot_diagram/core.py:329-372,
 * 440-481).
 *   in       normalised coordinates as for ol_generate_rays (flags honoured)
 *   hits     NULL, or 3 planes receiving the global x, y and intensity at the last
 *            surface (what surface_group.x[-1], .y[-1], .intensity[-1] hold)
 *   out7     device doubles, ACCUMULATED (zero them first), rays with i > 0 only:
 *            {count, sum dx, sum dy, sum dx^2, sum dy^2, sum i, max(dx^2+dy^2)}
 *            with dx = x - cx, dy = y - cy; (cx, cy) = centre in global image-plane
 *            coordinates (e.g. the chief-ray hit)
 *   status   as ol_trace / ol_generate_rays.
 * Systems with polarization-dependent coatings are refused (OL_EINVAL) like
 * ol_trace without a prt.                                                       */
int ol_trace_spot(const ol_system* sys, ol_dtype dt, int64_t n_rays,
                  const ol_raygen_params* p, const ol_raygen_inputs* in,
                  double cx, double cy, int32_t wavelength_index,
                  void* const hits[3], double* out7, uint32_t* status, void* stream);

/* ABI 10.  ol_trace_spot for a whole GRID of (field, wavelength) cells in ONE launch: what
 * SpotDiagram._generate_data / EncircledEnergy._generate_data loop over
 * (analysis/spot_diagram/core.py:420-438, analysis/encircled_energy.py: fields x wavelengths,
 * one Optic.trace each) -- every cell traces the SAME pupil planes; the cell is the launch
 * grid's second dimension, so its field, vignetting factors, wavelength row, centre and
 * outputs are workgroup-uniform.  A 3 x 3 spot diagram at 6 rings is ~10 us of kernel: nine
 * launches, nine status read-backs and nine result objects of Python cost 1.3 ms, this 0.1.
 *   in        px, py planes and flags only (hx / hy / vx / vy planes must be NULL; hx0 ...
 *             vy0 are ignored: the cells carry them)
 *   cells     n_cells (<= OL_SPOT_BATCH_MAX_CELLS) of HOST memory
 *   hits      NULL, or ONE block of n_cells x 3 planes (x, y, intensity of cell c at
 *             hits + (3 c + k) hits_stride elements), hits_stride >= n_rays
 *   out8      n_cells x 8 DEVICE doubles, ACCUMULATED (zero them first): per cell the seven
 *             of ol_trace_spot, element 7 unused
 * Same refusals as ol_trace_spot.                                                         */
#define OL_SPOT_BATCH_MAX_CELLS 32
typedef struct ol_spot_cell {
  double hx, hy;            /* normalised field of the cell                               */
  double vx, vy;            /* 1 - vignetting factor at that field                        */
  double cx, cy;            /* centre the moments are taken about                         */
  int32_t wavelength_index;
  int32_t reserved_;
  /* NULL, or ANOTHER system of the same optic whose refractive-index rows this cell traces with
   * (wavelength_index then counts in THAT system): a host that keeps one packed table per
   * wavelength -- the drop-in does -- still gets the whole fields x wavelengths grid in one
   * launch.  Geometry, apertures and coatings are `sys`'s; the two must have the same number
   * of surfaces, live on the same device and be unpolarised alike.                        */
  const ol_system* optics_of;
} ol_spot_cell;
int ol_trace_spot_batch(const ol_system* sys, ol_dtype dt, int64_t n_rays,
                        const ol_raygen_params* p, const ol_raygen_inputs* in,
                        int32_t n_cells, const ol_spot_cell* cells, void* hits,
                        int64_t hits_stride, double* out8, uint32_t* status, void* stream);

/* Wavefront OPD against a spherical reference centred on the chief-ray image point
 * (wavefront/strategy.py:163-215 ChiefRayStrategy.
 * compute_wavefront_data, wavefront/reference_geometry.py:41-79
 * SphericalReference.path_length, strategy.py:83-139 _correct_tilt).  Per ray:
 *   t       = back-propagation distance from the image point to the sphere,
 *   opd_img = n_image * t,  opd = ray_opd - opd_img + (ux X + uy Y),
 *             X = px * half_epd, Y = py * half_epd,
 *   opd_waves = (opd_ref - opd) / (wavelength_um * 1e-3),
 *   pupil    = r - t * k  (optional).
 * rays[7]: x,y,z,L,M,N,opd at the image surface.  Meant for fp64 traces.       */
typedef struct ol_wavefront_params {
  double xc, yc, zc, R;  /* reference sphere                                      */
  double n_image;        /* index of image space at the primary wavelength       */
  double opd_ref;        /* chief-ray OPD to the sphere (tilt-corrected)          */
  double ux, uy;         /* launch-plane tilt direction cosines (0 when n/a)      */
  double half_epd;
  double wavelength_um;
  /* ABI 4: planar reference for afocal systems (wavefront/reference_geometry.py:87-128
   * PlanarReference.path_length).  All zero = the sphere above; otherwise the reference is
   * the plane through (xc, yc, zc) with this normal, R is ignored and
   *   t = -((r - c) . n) / (k_back . n),  |k_back . n| < 1e-12 replaced by 1e-12.       */
  double nx, ny, nz;
  /* ABI 10, the FUSED entry points only (ol_trace_opd, ol_wavefront_reference -- through whose
   * device structure ol_trace_opd_dev reads them): what ends Optic.trace / trace_generic when
   * the last surface has a thickness (raytrace/real_ray_tracer.py:104-110, 145-149,
   * propagation/homogeneous.py:30-57) -- the rays go on by `last_thickness` through that
   * surface's post-medium before the reference sphere is intersected; the optical path is not
   * extended.  `last_absorb` (4 pi k / lambda_um * 1e3 per mm, 0 when k = 0) attenuates the
   * chief ray ol_wavefront_reference hands back (`chief8[6]`, what trace_generic returns) and
   * nothing else: the intensity a wavefront reports is the RECORDED last row
   * (wavefront/strategy.py:188), which the reference's write-back never reaches.  0 / 0 (a
   * zero-initialised struct; every sample lens) = the trace ends at the last surface.
   * ol_wavefront_opd, which is handed finished rays, ignores them.                      */
  double last_thickness, last_absorb;
} ol_wavefront_params;

int ol_wavefront_opd(const ol_wavefront_params* p, ol_dtype dt, int64_t n_rays,
                     const void* const rays[7], const void* px, const void* py,
                     void* opd_waves, void* const pupil[3], void* stream);

/* ABI 5.  Fused generate -> trace -> OPD (the wavefront analogue of
 * ol_trace_spot): ONE launch takes the normalised pupil coordinates of one field point to
 * the OPD map.  Replaces, for unpolarised systems and the chief-ray strategy, the chain
 * RayGenerator.generate_rays -> SurfaceGroup.trace -> ChiefRayStrategy.compute_wavefront_data
 * (wavefront/strategy.py:163-215; reference_geometry.py:41-128) and the reductions the
 * consumers apply to the map (wavefront/wavefront.py:103-148 fit_and_remove_tilt,
 * wavefront/opd.py:145-159 rms; psf/fft.py:101-137 reads opd + intensity).
 *   in         px, py planes; launch-uniform field (hx0, hy0) and vignetting (vx0, vy0);
 *              per-ray field / vignetting planes are refused (one field per wavefront)
 *   w          reference sphere / plane incl. opd_ref (see ol_wavefront_params)
 *   opd_waves, intensity   n_rays outputs (what ol_wavefront_opd and
 *              surface_group.intensity[-1] hold)
 *   pupil      NULL, or 3 planes: the reference-surface intersection points
 *   moments12  device doubles, ACCUMULATED (zero them first), w = intensity, o = OPD,
 *              (X, Y) = pupil point:
 *              {sum w, sum wX, sum wY, sum wXX, sum wXY, sum wYY, sum wo, sum woX, sum woY,
 *               #{i > 0}, sum o [i > 0], sum o^2 [i > 0]}
 * fp64 only (OL_F32 -> OL_EUNSUPPORTED): an OPD in waves needs 1e-9 of the path length.
 * Systems with polarization-dependent coatings are refused like ol_trace_spot.          */
int ol_trace_opd(const ol_system* sys, ol_dtype dt, int64_t n_rays,
                 const ol_raygen_params* p, const ol_raygen_inputs* in,
                 const ol_wavefront_params* w, int32_t wavelength_index,
                 void* opd_waves, void* intensity, void* const pupil[3],
                 double* moments12, uint32_t* status, void* stream);

/* ABI 8.  The reference sphere / plane of a wavefront, DEVICE-RESIDENT.
 * ol_wavefront_reference traces the chief ray of one field point (pupil point (0, 0)) with one
 * lane and leaves in `reference_dev` (OL_WAVEFRONT_REFERENCE_DOUBLES doubles of device memory,
 * opaque but for [0..2] = centre, [3] = radius) what ChiefRayStrategy derives from it on the
 * host (wavefront/strategy.py:176-184, 228-284): the sphere centred on the chief ray's image
 * point with radius to (0, 0, pupil_z), or (planar != 0) the plane through that point normal
 * to the chief ray, and the chief ray's own optical path to it.  `w` supplies n_image,
 * wavelength_um, ux, uy, half_epd (the tilt term); its centre / radius / normal are ignored.
 * chief8 (nullable): 8 doubles receiving x, y, z, L, M, N, i, opd of the chief ray.
 * ol_trace_opd_dev is ol_trace_opd reading the reference from such a structure: an OPD map
 * is two launches and no read-back in between.  fp64 only.                                */
#define OL_WAVEFRONT_REFERENCE_DOUBLES 16
int ol_wavefront_reference(const ol_system* sys, ol_dtype dt, const ol_raygen_params* p,
                           const ol_raygen_inputs* in, const ol_wavefront_params* w,
                           double pupil_z, int32_t planar, int32_t wavelength_index,
                           void* reference_dev, void* chief8, uint32_t* status, void* stream);
int ol_trace_opd_dev(const ol_system* sys, ol_dtype dt, int64_t n_rays,
                     const ol_raygen_params* p, const ol_raygen_inputs* in,
                     const void* reference_dev, int32_t wavelength_index, void* opd_waves,
                     void* intensity, void* const pupil[3], double* moments12, uint32_t* status,
                     void* stream);

/* ABI 9.  The FITTED reference of a wavefront, device-resident: what CentroidStrategy and
 * BestFitStrategy (wavefront/strategy.py:287-620) derive from the traced bundle -- the sphere
 * centred on the intensity-weighted (optionally k-sigma trimmed) centroid of the image points
 * with the weighted mean wavefront distance as radius, or the least-squares sphere through the
 * wavefront points; planar != 0: the plane through that centroid normal to the weighted mean
 * direction, or the least-squares plane -- and the piston (:331-340: the mean optical path of
 * the rays with intensity > 0).  A chain of reduction passes on `stream`, no read-back; the
 * result is left in `reference_dev` (OL_WAVEFRONT_REFERENCE_DOUBLES doubles, the structure
 * ol_wavefront_reference writes: [0..2] centre / plane point, [3] radius).
 *   rays[8]    x, y, z, L, M, N, opd, intensity at the image surface (fp64 planes)
 *   px, py     normalised pupil coordinates (the launch-plane tilt of strategy.py:88-139)
 *   w          n_image, wavelength_um, ux, uy, half_epd; the rest is ignored
 *   trim_std   CentroidStrategy.robust_trim_std (<= 0: no trimming)
 *   flags      where the reference's two backends differ: OL_FIT_STD_DDOF1 = the trimming's
 *              standard deviation divides by n - 1 (torch.std; numpy.std: n);
 *              OL_FIT_PISTON_SKIPS_NAN = the piston is a mean over the non-NaN values (the
 *              torch backend's be.mean, backend/torch_backend.py:969-989; numpy: NaN spreads)
 *   workspace  OL_WAVEFRONT_FIT_WORKSPACE_DOUBLES doubles of device memory (contents undefined)
 *   fit_status device word, WRITTEN: OL_FIT_NO_VALID / _TOO_FEW / _NO_ALIVE -- the three
 *              ValueErrors of strategy.py:387, 536, 334 -- or OL_FIT_SINGULAR: the wavefront
 *              points do not span space (a collimated beam: ONE plane; a stigmatic image:
 *              one point).  A direction whose variance is within 64 eps of (mean^2 + the
 *              largest variance), or a Cholesky pivot below 1e-10 of the largest diagonal
 *              entry.  The sphere is not determined; the reference's backend returns what
 *              its lstsq makes of a rank-deficient system (NumPy: the minimum-norm solution
 *              in RAW coordinates), and the caller follows it from the points themselves
 *              (the centre / radius written here are NaN)
 * The sums are formed in a fixed order (bit-reproducible for a given n_rays).  The least-squares
 * fits solve their normal equations in centred, per-axis scaled coordinates.
 * ol_wavefront_opd_fitted is ol_wavefront_opd against such a reference, with the tilt added
 * before the image-to-reference path is subtracted, as those two strategies do (:318-325). */
#define OL_WAVEFRONT_FIT_WORKSPACE_DOUBLES 32832
#define OL_FIT_CENTROID 0
#define OL_FIT_BEST_FIT 1
#define OL_FIT_NO_VALID 1u
#define OL_FIT_TOO_FEW 2u
#define OL_FIT_NO_ALIVE 4u
#define OL_FIT_SINGULAR 8u
#define OL_FIT_STD_DDOF1 1u
#define OL_FIT_PISTON_SKIPS_NAN 2u
int ol_wavefront_fit(int32_t kind, const ol_wavefront_params* w, double trim_std,
                     uint32_t flags, int32_t planar, int64_t n_rays,
                     const double* const rays[8], const double* px, const double* py,
                     double* workspace, void* reference_dev, uint32_t* fit_status,
                     void* stream);
int ol_wavefront_opd_fitted(int64_t n_rays, const double* const rays[7], const double* px,
                            const double* py, const void* reference_dev, double* opd_waves,
                            double* const pupil[3], void* stream);

/* The pupil function of the scalar FFT PSF (psf/fft.py:101-137 _generate_pupil + the
 * zero padding of :139-160): sample j of the compacted pupil list -- cell `cell[j]`
 * (row-major) of the n_side x n_side sample grid -- becomes
 *   sqrt(intensity[j]) * exp(-i 2 pi (opd_waves[j] - (plane[0] + plane[1] X + plane[2] Y)))
 * at row + pad, column + pad of the grid_size x grid_size complex array `grid`
 * (interleaved re, im doubles; pad = (grid_size - n_side) / 2; ZEROED BY THE CALLER).
 * pupil_x / pupil_y / plane are NULL when no tilt is removed.                          */
int ol_pupil_fill(ol_dtype dt, int64_t n_rays, const void* opd_waves, const void* intensity,
                  const void* pupil_x, const void* pupil_y, const double plane[3],
                  const int32_t* cell, int32_t n_side, int32_t grid_size, double* grid,
                  void* stream);

/* Profiling knobs (process-wide, not part of the trace semantics).
 *   OL_TUNE_RAYS_PER_THREAD  0 = auto (16-byte vector of rays per lane for conic-only
 *                            ranges, one ray per lane when Newton surfaces are
 *                            present), 1 = one ray per lane, 2 = force the vector,
 *                            3 = fp32, one packed PAIR of rays per lane (8-byte loads /
 *                            stores): ol_trace on lean ranges (measured slower, kept for
 *                            A/B); ol_trace_generate of a polarised polynomial-Zernike range
 *                            (configuration C5; same bits, same time: round 6, A/B only)
 *   OL_TUNE_COMPACT          1 = wavefront straggler compaction in the Newton loop
 *                            (needs the vector layout; default 0, measured slower)
 *   OL_TUNE_FIT_GRID         most blocks an ol_wavefront_fit pass is launched with (0 = the
 *                            default, 768; at most 2048 -- the rows of its workspace)
 *   OL_TUNE_RECORD_WG_CAP    record-all launches, resident workgroups per CU: 0 = the
 *                            default policy (conic-only unpolarised ranges: fp64 three,
 *                            fp32 two under OL_TRACE_FEW_WAVES; nothing else is capped),
 *                            1 = never, 2 ... 8 = every record launch (A/B)
 * The environment variables OL_TRACE_RPT / OL_RECORD_WG_CAP seed OL_TUNE_RAYS_PER_THREAD /
 * OL_TUNE_RECORD_WG_CAP.
 *
 * Environment variables read by the library (A/B runs and parity tests; results are the
 * same to rounding either way):
 *   OPTILAND_HIP_ZERNIKE_MONO=0  ol_system_create keeps every Zernike surface on the
 *                                per-|m| level evaluator (default: radial order <= 8 is
 *                                re-expressed as one bivariate polynomial)
 *   OPTILAND_HIP_NR_FAMILY=0     every range with a Newton-Raphson surface runs the generic
 *                                Newton kernel (default: a range whose Newton surfaces are
 *                                all Zernike surfaces / all even aspheres runs a kernel
 *                                instantiation carrying only that family's code)        */
#define OL_TUNE_RAYS_PER_THREAD 0
#define OL_TUNE_COMPACT 1
#define OL_TUNE_FIT_GRID 2
#define OL_TUNE_RECORD_WG_CAP 3
int ol_set_tuning(int32_t knob, int32_t value);

const char* ol_last_error(void);
int32_t ol_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* OPTILAND_HIP_H */
