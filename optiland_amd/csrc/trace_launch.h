// trace_launch.h -- kernel argument block shared by the kernels and the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "device_table.h"

namespace ol {

#ifndef OL_TRACE_BLOCK
#define OL_TRACE_BLOCK 256
#endif
constexpr int kTraceBlock = OL_TRACE_BLOCK;  // threads per workgroup (waves of 64 lanes)
constexpr uint32_t kTraceWriteRays = 0x1u; // OL_TRACE_WRITE_RAYS
constexpr uint32_t kTraceCompact = 0x2u;   // OL_TRACE_COMPACT
constexpr uint32_t kTracePrtComplex = 0x4u;  // OL_TRACE_PRT_COMPLEX
constexpr uint32_t kTracePrtIdentity = 0x8u;  // OL_TRACE_PRT_IDENTITY
constexpr uint32_t kTraceFewWaves = 0x10u;    // OL_TRACE_FEW_WAVES
constexpr uint32_t kTraceNonUnitK = 0x20u;    // OL_TRACE_NONUNIT_K (polarised ol_trace launches)
constexpr uint32_t kTraceRow0IsInput = 0x100u;  // internal: rays[] ARE record row 0

// GEN template parameter of trace_kernel: 0 = the rays come from eight planes; otherwise the
// generating prologue, a bit set
constexpr int kGenUniform = 1;      // launch-uniform field point, unit initial intensity
constexpr int kGenFieldPlanes = 2;  // + per-ray field planes hx, hy (and, if given, vx, vy)
constexpr int kGenApod = 4;         // + pupil apodization as the initial intensity

// process-wide tuning knobs (ol_set_tuning)
// OL_POLZ_PAIR: the generating polarised Zernike fp32 launch (configuration C5) on TWO rays per
// lane, traced as one f32x2 through surface_step<f32x2, 1, 1, kNrZernike> (surface_math.h: the
// pair forms): every multiply-add of the two rays is one packed instruction, stores are 8 bytes
// per lane.  Taken when capi.hip finds the launch eligible (pair_ok); ol_set_tuning
// (OL_TUNE_RAYS_PER_THREAD: 1 = one ray per lane, 3 = pair) overrides the default for A/B runs.
// Round 6, measured (profiles/r06_polz_*.txt): the same bits; 645 instead of 797 vector
// instructions per ray and the SAME engine cycles at full clock (a packed instruction costs
// 1.6-1.8 plain ones on gfx950, tools/microbench/valu_rate.hip); 2-5 % faster in the driver's
// window and 1.5-3 % in steady state (three alternating bench runs, r06_c5_window.txt) -- ON.
#ifndef OL_POLZ_PAIR
#define OL_POLZ_PAIR 1
#endif

struct Tuning {
  int rays_per_thread = 0;  // 0 = default, 1 = one ray per lane, 2 = force vector
  int compact = 0;          // measured slower; opt-in (OL_TUNE_COMPACT)
  int fit_grid = 0;         // blocks of an ol_wavefront_fit pass: 0 = default cap
  int record_wg_cap = 0;    // record-all launches: resident workgroups per CU (0 = no cap)
};
Tuning& tuning();

// ray generation / epilogue / reductions (aux_kernels.hip)
struct RaygenDev {
  int32_t object_infinite;
  int32_t field_kind;  // 0 angle, 1 object height
  double EPL, EPD, max_field, offset, z_first, tele_dz;
  double apod_a, apod_b;
  int32_t apod_kind;
};

// launch-invariant scalars of the ray generator in the working precision.  Formed on the
// HOST by the launchers of the fused kernels (SpotArgs / OpdArgs carry them), so that a
// kernel can read a value from the kernarg segment where it uses it instead of converting
// the whole block in its prologue and holding it in SGPRs through the surface loop.
#if defined(__HIPCC__)
#define OL_HD __host__ __device__
#else
#define OL_HD
#endif
template <typename T>
struct RaygenConsts {
  T EPL, EPD, maxf, off_epl, z_inf, z_fin, epl_z, tele_dz, apod_a, apod_b;
  int32_t apod_kind;
  int32_t infinite, height, linear, telecentric;  // flags (0 / 1)
  RaygenConsts() = default;
  OL_HD explicit RaygenConsts(const RaygenDev& p)
      : EPL((T)p.EPL), EPD((T)p.EPD), maxf((T)p.max_field), off_epl((T)(p.offset + p.EPL)),
        z_inf((T)(p.z_first - p.offset)), z_fin((T)p.z_first), epl_z((T)(p.EPL - p.z_first)),
        tele_dz((T)p.tele_dz), apod_a((T)p.apod_a), apod_b((T)p.apod_b),
        apod_kind(p.apod_kind), infinite(p.object_infinite != 0),
        // the field quantity is a POSITION on the object for object-height fields and
        // for paraxial-image-height fields with a finite object; otherwise a slope
        height(p.field_kind == 1 || (p.field_kind == 2 && p.object_infinite == 0)),
        linear(p.field_kind != 0), telecentric(p.tele_dz > 0.0) {}
};

constexpr uint32_t kRaygenCheckField = 0x1u;    // OL_RAYGEN_CHECK_FIELD
constexpr uint32_t kRaygenCheckPupil = 0x2u;    // OL_RAYGEN_CHECK_PUPIL
constexpr uint32_t kRaygenPrescalePupil = 0x4u; // OL_RAYGEN_PRESCALE_PUPIL
constexpr uint32_t kSpotPolarizedOk = 0x8u;     // OL_SPOT_POLARIZED_OK (capi.hip only)
constexpr uint32_t kSpotHitsLocal = 0x10u;      // OL_SPOT_HITS_LOCAL
constexpr uint32_t kStatusFieldRange = 0x8u;    // OL_STATUS_FIELD_RANGE
constexpr uint32_t kStatusPupilRange = 0x10u;   // OL_STATUS_PUPIL_RANGE
constexpr uint32_t kStatusNanDirection = 0x20u;  // OL_STATUS_NAN_DIRECTION (informational)

// normalised coordinates of one ray block (ol_raygen_inputs in working precision)
template <typename T>
struct RaygenIn {
  const T *hx, *hy;  // per-ray field or nullptr -> hx0, hy0 (tangents tx0, ty0)
  const T *px, *py;  // per-ray pupil
  const T *vx, *vy;  // per-ray 1 - vignetting or nullptr -> vx0, vy0
  T hx0, hy0, vx0, vy0;
  T tx0, ty0;        // tan(field angle) / object height of the launch-uniform field (host)
  uint32_t flags;
};

// field quantity of a launch-uniform field (tan of the field angle, or the object
// height): same expression as raygen_field() / tan_deg() on the device (product in
// T, tangent in double) -- evaluated per lane the double-precision tan() cost ~15 %
// of the fused spot kernel.
template <typename T>
inline void uniform_field_tangents(const RaygenDev& rg, RaygenIn<T>& in) {
  const T maxf = (T)rg.max_field;
  if (rg.field_kind != 0) {
    in.tx0 = maxf * in.hx0;
    in.ty0 = maxf * in.hy0;
  } else {
    in.tx0 = (T)tan((double)(maxf * in.hx0) * 0.017453292519943295);
    in.ty0 = (T)tan((double)(maxf * in.hy0) * 0.017453292519943295);
  }
}

struct PolStateDev {
  int32_t is_polarized;
  double Ex, Ey, phase_x, phase_y;
};

// field amplitudes of the incident state: E0 = Ex e^{i phx} s_hat + Ey e^{i phy} p_hat
// (rays/polarization_state.py:29-56); an unpolarised state is the mean of the x and the y
// state (polarized_rays.py:122-133).  Plain data: formed once per launch (host or device).
template <typename T>
struct PolFields {
  T ar[2], ai[2], br[2], bi[2];
  int32_t nf;
  PolFields() = default;
  OL_HD explicit PolFields(const PolStateDev& st) {
    if (st.is_polarized) {
      nf = 1;
      ar[0] = (T)(st.Ex * cos(st.phase_x));
      ai[0] = (T)(st.Ex * sin(st.phase_x));
      br[0] = (T)(st.Ey * cos(st.phase_y));
      bi[0] = (T)(st.Ey * sin(st.phase_y));
      ar[1] = ai[1] = br[1] = bi[1] = T(0);
    } else {
      nf = 2;
      ar[0] = T(1); ai[0] = T(0); br[0] = T(0); bi[0] = T(0);
      ar[1] = T(0); ai[1] = T(0); br[1] = T(1); bi[1] = T(0);
    }
  }
};

template <typename T>
struct TraceArgs {
  const DevSurfHot<T>* surf;   // [n_surf] hot blocks
  const DevSurfCold<T>* cold;  // [n_surf] cold blocks
  const DevOptics<T>* optics;  // [n_surf][n_wl]
  const T* coeffs;             // coefficient blocks
  T* rays[8];                  // x,y,z,L,M,N,i,opd planes
  T* record;                   // rows x 8 x record_stride or nullptr
  T* prt;                      // 9 x n (18 x n with kTracePrtComplex) or nullptr
  uint32_t* status;            // device word or nullptr
  double* spot;                // [spot_slots][8] doubles (epilogue moments) or nullptr
  double cx, cy;               // centre of the epilogue moments
  int32_t spot_slots;
  int64_t n;
  int64_t record_stride;
  int32_t first, last;
  int32_t n_wl, wl;
  uint32_t flags;
  int32_t record_from;   // first RECORDED surface (row 0 of `record`); <= first: all of them
  // generating launches (ol_trace_generate; trace_kernel<..., GEN = true>): the normalised
  // coordinates and the generator constants instead of rays[]
  RaygenIn<T> in;
  RaygenConsts<T> rgc;
  // generating polarised launches: PolarizedRays.update_intensity as an epilogue of the same
  // kernel (ol_trace_extras.updated_intensity, ABI 7); nullptr = none
  T* i_updated;
  PolFields<T> pf;
  // reference-Newton launches (ABI 11, nr_family == kNrReference; surface_math.h NrRefCtl):
  // the per-surface iteration counts (device, [2 * n_surf]) and the surface whose count THIS
  // launch determines (-1: none)
  int32_t* nr_iters;
  int32_t nr_count_at;
  int32_t n_surf;
};

// nr_family: 0 = no Newton-Raphson geometry in the traced range (lean kernel), 1 = generic
// Newton kernel, 3 / 4 = every Newton surface of the range is a Zernike surface / an even
// asphere (single-family instantiations, surface_math.h kNr*)
template <typename T>
hipError_t launch_trace(const TraceArgs<T>& a, bool vector_ok, int nr_family,
                        hipStream_t stream);
// the generating variant: a.in / a.rgc instead of a.rays (one ray per lane, record-all form;
// pair_ok: every plane the launch touches may be accessed two rays = 8 bytes at a time, which
// lets the lean fp32 form run on packed pairs)
template <typename T>
hipError_t launch_trace_generate(const TraceArgs<T>& a, int nr_family, bool pair_ok,
                                 hipStream_t stream);


template <typename T>
hipError_t launch_raygen(const RaygenDev& p, const RaygenIn<T>& in, int64_t n, T* const out[8],
                         uint32_t* status, hipStream_t stream);

// fused generate -> trace -> reduce spot kernel (trace_kernel.hip, SURVEY.md 8 f1+f2)
template <typename T>
struct SpotArgs {
  const DevSurfHot<T>* surf;
  const DevSurfCold<T>* cold;
  const DevOptics<T>* optics;
  const T* coeffs;
  RaygenIn<T> in;
  RaygenDev rg;
  RaygenConsts<T> rgc;  // = RaygenConsts<T>(rg), set by the launcher
  double cx, cy;     // centre the moments are taken about (global image coordinates)
  T* hits[3];        // optional image-plane x, y, intensity planes (all or none)
  double* out;       // 7 doubles, accumulated
  uint32_t* status;
  int64_t n;
  int32_t first, last;
  int32_t n_wl, wl;
  int32_t tiles_per_block;  // set by the launcher
};

template <typename T>
hipError_t launch_spot_trace(const SpotArgs<T>& a, bool vector_ok, int nr_family,
                             hipStream_t stream);

// ONE launch for a grid of (field, wavelength) cells over the same pupil planes
// (ol_trace_spot_batch: SpotDiagram._generate_data, analysis/spot_diagram/core.py:420-438):
// blockIdx.y is the cell, so everything a cell changes -- the field tangents, the vignetting
// factors, the wavelength row of the optics table, the centre of the moments, where its hits
// and its eight sums go -- is workgroup-uniform and read from this block with scalar loads.
// It travels as a kernel parameter of its own BEHIND the SpotArgs block, whose kernarg layout
// (kernargs<T, SpotArgs<T>>) stays what the single-cell kernels read.
constexpr int kSpotBatchCells = 32;
template <typename T>
struct SpotCell {
  T tx, ty, vx, vy;   // field tangents (uniform_field_tangents), vignetting factors
  double cx, cy;
  const DevOptics<T>* optics;   // the optics table this cell reads (a.optics unless the cell
  int32_t n_wl, wl;             // brings another system's) and its wavelength slot in it
};
template <typename T>
struct SpotBatch {
  int32_t n_cells, pad_;
  int64_t hits_stride;   // elements between the planes of the hits block (cell-major, x / y / i)
  SpotCell<T> c[kSpotBatchCells];
};
// a.hits[k] = plane k of cell 0 (cell c: + c * 3 * hits_stride), a.out = the 8 doubles of cell 0
// (cell c: + 8 c); a.in.hx / hy / vx / vy must be null (one field per cell)
template <typename T>
hipError_t launch_spot_batch(const SpotArgs<T>& a, const SpotBatch<T>& batch, bool vector_ok,
                             int nr_family, hipStream_t stream);

template <typename T>
hipError_t launch_pol_intensity(int64_t n, const T* prt, bool prt_complex, const T* const k0[3],
                                const T* i0,
                                const PolStateDev& st, T* intensity, uint32_t* status,
                                hipStream_t stream);

struct WavefrontDev {
  double xc, yc, zc, R, n_image, opd_ref, ux, uy, half_epd, wavelength_um;
  double nx, ny, nz;  // all zero: spherical reference; else planar reference normal
  // ABI 10: the propagation that ends Optic.trace / trace_generic -- on by the LAST surface's
  // thickness through its post-medium (real_ray_tracer.py:104-110, 145-149) -- for the fused
  // kernels, which end at the last surface: 0 (every sample lens) = none
  double last_t = 0.0, last_absorb = 0.0;
};

// reference sphere / plane of the wavefront kernels in the working precision (host-formed,
// like RaygenConsts)
template <typename T>
struct WavefrontConsts {
  T xc, yc, zc, R, ni, inv_w, ux, uy, half_epd, opd_ref, nx, ny, nz;
  T last_t = T(0), last_absorb = T(0);   // (see WavefrontDev)
  int32_t planar;
  WavefrontConsts() = default;
  OL_HD explicit WavefrontConsts(const WavefrontDev& p)
      : xc((T)p.xc), yc((T)p.yc), zc((T)p.zc), R((T)p.R), ni((T)p.n_image),
        inv_w((T)(1.0 / (p.wavelength_um * 1e-3))), ux((T)p.ux), uy((T)p.uy),
        half_epd((T)p.half_epd), opd_ref((T)p.opd_ref), nx((T)p.nx), ny((T)p.ny), nz((T)p.nz),
        last_t((T)p.last_t), last_absorb((T)p.last_absorb),
        planar(p.nx != 0.0 || p.ny != 0.0 || p.nz != 0.0) {}
};

template <typename T>
hipError_t launch_wavefront(const WavefrontDev& p, int64_t n, const T* const rays[7], const T* px,
                            const T* py, T* opd_waves, T* const pupil[3], hipStream_t stream);

// fused generate -> trace -> OPD kernel (trace_kernel.hip)
constexpr int kOpdMoments = 12;
template <typename T>
struct OpdArgs {
  const DevSurfHot<T>* surf;
  const DevSurfCold<T>* cold;
  const DevOptics<T>* optics;
  const T* coeffs;
  RaygenIn<T> in;    // pupil planes; launch-uniform field and vignetting
  RaygenDev rg;
  WavefrontDev wf;
  RaygenConsts<T> rgc;     // = RaygenConsts<T>(rg), set by the launcher
  WavefrontConsts<T> wfc;  // = WavefrontConsts<T>(wf), set by the launcher
  T* opd;            // OPD in waves per ray
  T* inten;          // image-plane intensity per ray
  T* pupil[3];       // optional: reference-surface intersection point (all or none)
  double* mom;       // kOpdMoments doubles, accumulated (see ol_trace_opd)
  uint32_t* status;
  int64_t n;
  int32_t first, last;
  int32_t n_wl, wl;
  // ABI 8 (ol_trace_opd_dev): the reference sphere / plane from DEVICE memory -- written there
  // by ol_wavefront_reference (chief_ref_kernel) -- instead of the kernel argument `wfc`
  const WavefrontConsts<T>* wf_dev;
};
template <typename T>
hipError_t launch_opd_trace(const OpdArgs<T>& a, bool vector_ok, int nr_family,
                            hipStream_t stream);

// ol_wavefront_reference: the chief ray of one field point traced by ONE lane and turned into
// the reference sphere / plane of the wavefront kernels, left in device memory
// (wavefront/strategy.py:176-184, 228-284 -- what the host did with a one-ray trace, a
// read-back and a handful of scalar operations)
template <typename T>
struct ChiefArgs {
  const DevSurfHot<T>* surf;
  const DevSurfCold<T>* cold;
  const DevOptics<T>* optics;
  const T* coeffs;
  RaygenIn<T> in;          // launch-uniform field and vignetting; the pupil point is (0, 0)
  RaygenDev rg;
  RaygenConsts<T> rgc;     // set by the launcher
  WavefrontConsts<T> wfc;  // ni, inv_w, ux, uy, half_epd, planar (the rest is filled in)
  T pupil_z;               // exit-pupil position (spherical reference)
  WavefrontConsts<T>* out; // device
  T* chief;                // optional: 8 values x, y, z, L, M, N, i, opd of the chief ray
  uint32_t* status;
  int32_t first, last;
  int32_t n_wl, wl;
};
template <typename T>
hipError_t launch_chief_reference(const ChiefArgs<T>& a, int nr_family, hipStream_t stream);

// ol_wavefront_fit (aux_kernels.hip, wavefront_fit_device.h): the reference sphere / plane of
// CentroidStrategy / BestFitStrategy (wavefront/strategy.py:287-620) from the traced bundle,
// as a chain of reduction passes that leaves a WavefrontConsts<double> in device memory
constexpr int kFitSums = 16;        // running sums per pass (at most)
constexpr int kFitMaxBlocks = 2048;  // most blocks of a pass; one row of partial sums per block
constexpr int kFitDefaultBlocks = 768;  // 3 per CU: measured best, profiles/r04_fit_timing.txt
constexpr int kFitStateDoubles = 64;
constexpr int kFitWorkspaceDoubles = kFitStateDoubles + kFitMaxBlocks * kFitSums;

enum : int32_t { kFitCentroid = 0, kFitBestFit = 1 };
enum : int32_t { kPassC1 = 0, kPassC2, kPassC3, kPassC4, kPassC5, kPassB1, kPassB2, kPassMean,
                 kFitPasses };
// fit_status bits (ol_wavefront_fit): the reference's three ValueErrors and a singular fit
enum : uint32_t { kFitNoValid = 1u, kFitTooFew = 2u, kFitNoAlive = 4u, kFitSingular = 8u };

struct FitParams {
  double ni, inv_w, ux, uy, half_epd, trim_std;
  int32_t kind, planar;
  int32_t ddof;      // of the trimming's standard deviation: torch.std 1, numpy.std 0
  int32_t skip_nan;  // the piston is a mean that ignores NaN (the torch backend's be.mean)
};

struct FitArgs {
  FitParams p;
  const double* ray[8];  // x, y, z, L, M, N, opd, intensity at the image surface
  const double* px;
  const double* py;
  int64_t n;
  double* workspace;     // kFitWorkspaceDoubles: between-pass state + one row of sums per block
  WavefrontConsts<double>* out;
  uint32_t* status;      // kFit* bits, OR-ed in
};
hipError_t launch_wavefront_fit(const FitArgs& a, hipStream_t stream);
// the OPD map against such a device-resident reference (tilt added before the subtraction,
// strategy.py:318-340)
hipError_t launch_wavefront_fitted(const WavefrontConsts<double>* ref, int64_t n,
                                   const double* const rays[7], const double* px,
                                   const double* py, double* opd_waves, double* const pupil[3],
                                   hipStream_t stream);

template <typename T>
hipError_t launch_pupil_fill(int64_t n, const T* opd, const T* inten, const T* pupil_x,
                             const T* pupil_y, const double coef[3], const int32_t* cell,
                             int32_t n_side, int32_t grid, int32_t pad, double* out,
                             hipStream_t stream);

template <typename T>
hipError_t launch_spot_moments(int64_t n, const T* x, const T* y, const T* inten, double* out6,
                               hipStream_t stream);
template <typename T>
hipError_t launch_radial_energy(int64_t n, const T* x, const T* y, const T* inten, double cx,
                                double cy, const double* r_step, int n_steps, double* bins,
                                hipStream_t stream);
template <typename T>
hipError_t launch_irradiance(int64_t n, const T* x, const T* y, const T* power,
                             const double* x_edges, int nx, const double* y_edges, int ny,
                             double* hist, hipStream_t stream);
// deterministic pupil samplers on the device (aux_kernels.hip; ol_pupil_points)
template <typename T>
hipError_t launch_pupil_points(int kind, int32_t param, int64_t n, const int32_t* first,
                               const int64_t* offset, T* x, T* y, hipStream_t stream);

// Math<T> primitives element-wise (aux_kernels.hip; ol_math_probe)
template <typename T>
hipError_t launch_math_probe(int op, int64_t n, const T* a, const T* b, T* out,
                             hipStream_t stream);

// write-only streaming yardstick (aux_kernels.hip; ol_stream_fill)
hipError_t launch_stream_fill(void* dst, int64_t bytes, int width, int planes, uint32_t pattern,
                              hipStream_t stream);

template <typename T>
hipError_t launch_spot_max_r2(int64_t n, const T* x, const T* y, const T* inten, double cx,
                              double cy, double* out1, hipStream_t stream);

}  // namespace ol
