// aux_kernels.hip -- the steps either side of the trace kernel:
// on-device ray generation, the polarised update_intensity epilogue and the
// image-plane spot reductions.  All are streaming elementwise / reduction
// kernels (HBM-bound, grid-stride, coalesced SoA).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "raygen_device.h"
#include "wavefront_device.h"
#include "wavefront_fit_device.h"
#include "epilogue_device.h"
#include "trace_launch.h"
#include "surface_math.h"

namespace ol {

namespace {
constexpr int kBlock = 256;

inline unsigned grid_for(int64_t n) {
  int64_t b = (n + kBlock - 1) / kBlock;
  const int64_t cap = 256 * 8;  // 256 CUs x 8 blocks, grid-stride beyond that
  return (unsigned)(b < cap ? (b > 0 ? b : 1) : cap);
}

}  // namespace

// rays/ray_generator.py:47-99, rays/ray_aiming/paraxial.py:33-106,
// fields/field_types/angle.py:17-58; range validation real_ray_tracer.py:156-173
template <typename T>
__global__ __launch_bounds__(kBlock) void raygen_kernel(RaygenDev p, RaygenIn<T> in, int64_t n,
                                                        T* ox, T* oy, T* oz, T* oL, T* oM, T* oN,
                                                        T* oi, T* oopd, uint32_t* status) {
  const RaygenConsts<T> c(p);
  const bool field_planes = in.hx != nullptr, vig_planes = in.vx != nullptr;
  uint32_t st = 0;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * kBlock) {
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
    ox[j] = o[0];
    oy[j] = o[1];
    oz[j] = o[2];
    oL[j] = o[3];
    oM[j] = o[4];
    oN[j] = o[5];
    oi[j] = raygen_apodize<T>(c, px, py);
    if (oopd) oopd[j] = T(0);
  }
  if (st && status) atomicOr(status, st);
}

template <typename T>
hipError_t launch_raygen(const RaygenDev& p, const RaygenIn<T>& in_, int64_t n, T* const out[8],
                         uint32_t* status, hipStream_t stream) {
  RaygenIn<T> in = in_;
  if (in.hx == nullptr) uniform_field_tangents<T>(p, in);
  hipLaunchKernelGGL((raygen_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, stream, p, in, n,
                     out[0], out[1], out[2], out[3], out[4], out[5], out[6], out[7], status);
  return hipGetLastError();
}

// rays/polarized_rays.py:68-133, 204-233: per-ray arithmetic in epilogue_device.h
template <typename T, bool CPLX>
__global__ __launch_bounds__(kBlock) void pol_intensity_kernel(int64_t n, const T* __restrict__ prt,
                                                               const T* __restrict__ k0x,
                                                               const T* __restrict__ k0y,
                                                               const T* __restrict__ k0z,
                                                               const T* __restrict__ i0,
                                                               PolFields<T> fld, T* intensity,
                                                               uint32_t* status) {
  // (the incident state's amplitudes are formed by the launcher: fp64 sin / cos on the device
  // brought a private array -- LDS or scratch -- into this streaming kernel)
  uint32_t flag = 0;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * kBlock) {
    T P[9], Q[9];
#pragma unroll
    for (int e = 0; e < 9; ++e) {
      P[e] = prt[(int64_t)e * n + j];
      Q[e] = CPLX ? prt[(int64_t)(9 + e) * n + j] : T(0);
    }
    intensity[j] = pol_intensity_one<T, CPLX>(fld, k0x[j], k0y[j], k0z[j], P, Q, i0[j], flag);
  }
  if (flag && status) atomicOr(status, flag);
}

template <typename T>
hipError_t launch_pol_intensity(int64_t n, const T* prt, bool prt_complex, const T* const k0[3],
                                const T* i0, const PolStateDev& st, T* intensity,
                                uint32_t* status, hipStream_t stream) {
  const PolFields<T> fld(st);
  if (prt_complex)
    hipLaunchKernelGGL((pol_intensity_kernel<T, true>), dim3(grid_for(n)), dim3(kBlock), 0,
                       stream, n, prt, k0[0], k0[1], k0[2], i0, fld, intensity, status);
  else
    hipLaunchKernelGGL((pol_intensity_kernel<T, false>), dim3(grid_for(n)), dim3(kBlock), 0,
                       stream, n, prt, k0[0], k0[1], k0[2], i0, fld, intensity, status);
  return hipGetLastError();
}

// wavefront/strategy.py:163-215 + reference_geometry.py:41-79: OPD in waves against
// the chief-ray reference sphere.  Streaming elementwise (9 planes in, 1-4 out).
template <typename T>
__global__ __launch_bounds__(kBlock) void wavefront_kernel(
    WavefrontDev p, int64_t n, const T* __restrict__ x, const T* __restrict__ y,
    const T* __restrict__ z, const T* __restrict__ Ld, const T* __restrict__ Md,
    const T* __restrict__ Nd, const T* __restrict__ opd_in, const T* __restrict__ px,
    const T* __restrict__ py, T* opd_waves, T* pux, T* puy, T* puz) {
  const WavefrontConsts<T> w(p);
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * kBlock) {
    T pu[3];
    opd_waves[j] = wavefront_one<T>(w, x[j], y[j], z[j], Ld[j], Md[j], Nd[j], opd_in[j], px[j],
                                    py[j], pu);
    if (pux) {
      pux[j] = pu[0];
      puy[j] = pu[1];
      puz[j] = pu[2];
    }
  }
}

// psf/fft.py:101-137: the pupil function A exp(-i 2 pi OPD) scattered into the zero-padded
// FFT grid.  Sample j of the compacted 'uniform' pupil list lives in cell `cell[j]` of the
// n x n sample grid (row-major, the host sampler's own mask); the grid is grid x grid
// complex (re, im interleaved), zeroed by the caller, with the n x n block at offset
// `pad`.  An intensity-weighted plane a + b X + c Y (wavefront.py:103-148, tilt removal)
// is subtracted from the OPD when `pupil_x` is given.
template <typename T>
__global__ __launch_bounds__(kBlock) void pupil_fill_kernel(
    int64_t n, const T* __restrict__ opd, const T* __restrict__ inten,
    const T* __restrict__ pupil_x, const T* __restrict__ pupil_y, double c0, double c1,
    double c2, const int32_t* __restrict__ cell, int32_t n_side, int32_t grid, int32_t pad,
    double* __restrict__ out) {
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * kBlock) {
    double o = (double)opd[j];
    if (pupil_x) o -= c0 + c1 * (double)pupil_x[j] + c2 * (double)pupil_y[j];
    double re, im;
    pupil_sample(o, (double)inten[j], re, im);
    const int64_t at = pupil_cell_offset(cell[j], n_side, grid, pad);
    out[at] = re;
    out[at + 1] = im;
  }
}

template <typename T>
hipError_t launch_pupil_fill(int64_t n, const T* opd, const T* inten, const T* pupil_x,
                             const T* pupil_y, const double coef[3], const int32_t* cell,
                             int32_t n_side, int32_t grid, int32_t pad, double* out,
                             hipStream_t stream) {
  hipLaunchKernelGGL((pupil_fill_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, stream, n, opd,
                     inten, pupil_x, pupil_y, coef[0], coef[1], coef[2], cell, n_side, grid, pad,
                     out);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_wavefront(const WavefrontDev& p, int64_t n, const T* const rays[7], const T* px,
                            const T* py, T* opd_waves, T* const pupil[3], hipStream_t stream) {
  hipLaunchKernelGGL((wavefront_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, stream, p, n,
                     rays[0], rays[1], rays[2], rays[3], rays[4], rays[5], rays[6], px, py,
                     opd_waves, pupil ? pupil[0] : nullptr, pupil ? pupil[1] : nullptr,
                     pupil ? pupil[2] : nullptr);
  return hipGetLastError();
}

// wave-level max via shuffles (64 lanes)
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o = __shfl_down(v, off, 64);
    v = o > v ? o : v;
  }
  return v;
}

// analysis/spot_diagram/core.py:329-372, 440-481 building blocks: rays with
// intensity > 0 only (mask at :470-476).
template <typename T>
__global__ __launch_bounds__(kBlock) void spot_moments_kernel(int64_t n, const T* __restrict__ x,
                                                              const T* __restrict__ y,
                                                              const T* __restrict__ inten,
                                                              double* out6) {
  double s[6] = {0, 0, 0, 0, 0, 0};
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * kBlock) {
    const double xv = (double)x[j], yv = (double)y[j];
    if (inten[j] > T(0)) {
      s[0] += 1.0;
      s[1] += xv;
      s[2] += yv;
      s[3] += xv * xv;
      s[4] += yv * yv;
      s[5] += 1.0;
    }
  }
  __shared__ double part[kBlock / 64][6];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // step-major wave reduction: the six values move together (12 independent
  // ds_bpermute per step) instead of six serial chains -- see SpotAcc::flush
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) o[k] = __shfl_down(s[k], off, 64);
#pragma unroll
    for (int k = 0; k < 6; ++k) s[k] += o[k];
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) part[wave][k] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    double v = 0;
    for (int w = 0; w < kBlock / 64; ++w) v += part[w][threadIdx.x];
    atomicAdd(&out6[threadIdx.x], v);
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void spot_max_r2_kernel(int64_t n, const T* __restrict__ x,
                                                             const T* __restrict__ y,
                                                             const T* __restrict__ inten,
                                                             double cx, double cy, double* out1) {
  double best = 0.0;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * kBlock) {
    if (inten[j] > T(0)) {
      const double dx = (double)x[j] - cx, dy = (double)y[j] - cy;
      const double r2 = dx * dx + dy * dy;
      best = r2 > best ? r2 : best;
    }
  }
  best = wave_max(best);
  __shared__ double part[kBlock / 64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) part[wave] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    double v = part[0];
    for (int w = 1; w < kBlock / 64; ++w) v = part[w] > v ? part[w] : v;
    // non-negative doubles order like their bit patterns
    atomicMax(reinterpret_cast<unsigned long long*>(out1),
              (unsigned long long)__double_as_longlong(v));
  }
}

// analysis/encircled_energy.py:147-160: energy per radius step.  Each workgroup bins
// its rays into an LDS histogram (binary search over the caller's r_step array, so the
// `radii <= r` comparisons are the reference's own), then adds its non-empty bins to
// the global ones.
constexpr int kMaxEeSteps = 1024;

template <typename T>
__global__ __launch_bounds__(kBlock) void radial_energy_kernel(
    int64_t n, const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ inten,
    double cx, double cy, const double* __restrict__ r_step, int n_steps, double* bins) {
  __shared__ double hist[kMaxEeSteps];
  __shared__ double steps[kMaxEeSteps];
  for (int k = threadIdx.x; k < n_steps; k += kBlock) {
    hist[k] = 0.0;
    steps[k] = r_step[k];
  }
  __syncthreads();
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * kBlock) {
    const double e = (double)inten[j];
    const double dx = (double)x[j] - cx, dy = (double)y[j] - cy;
    const double r = sqrt(dx * dx + dy * dy);
    const int lo = radial_step_index(steps, n_steps, r, e);  // epilogue_device.h
    if (lo >= 0 && e != 0.0) unsafeAtomicAdd(&hist[lo], e);
  }
  __syncthreads();
  for (int k = threadIdx.x; k < n_steps; k += kBlock)
    if (hist[k] != 0.0) unsafeAtomicAdd(&bins[k], hist[k]);
}

template <typename T>
hipError_t launch_radial_energy(int64_t n, const T* x, const T* y, const T* inten, double cx,
                                double cy, const double* r_step, int n_steps, double* bins,
                                hipStream_t stream) {
  if (n_steps < 1 || n_steps > kMaxEeSteps) return hipErrorInvalidValue;
  hipLaunchKernelGGL((radial_energy_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, stream, n, x,
                     y, inten, cx, cy, r_step, n_steps, bins);
  return hipGetLastError();
}

// analysis/irradiance.py:341-353: numpy.histogram2d(x, y, bins=[x_edges, y_edges],
// weights=power) for rays with power > 0.  Bin search = numpy's
// searchsorted(edges, v, "right") - 1 with the right-most edge folded into the last bin.
// (edge_bin: epilogue_device.h)

constexpr int kMaxLdsBins = 4096;  // 32 KB of LDS doubles: small detectors are privatised

template <typename T, bool LDS>
__global__ __launch_bounds__(kBlock) void irradiance_kernel(
    int64_t n, const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ power,
    const double* __restrict__ xe, int nx, const double* __restrict__ ye, int ny, double* hist) {
  __shared__ double lh[LDS ? kMaxLdsBins : 1];
  const int nb = nx * ny;
  if constexpr (LDS) {
    for (int k = threadIdx.x; k < nb; k += kBlock) lh[k] = 0.0;
    __syncthreads();
  }
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * kBlock) {
    const double p = (double)power[j];
    if (!(p > 0.0)) continue;  // irradiance.py:346 (NaN power fails the test too)
    const int ix = edge_bin(xe, nx, (double)x[j]);
    if (ix < 0) continue;
    const int iy = edge_bin(ye, ny, (double)y[j]);
    if (iy < 0) continue;
    if constexpr (LDS) unsafeAtomicAdd(&lh[ix * ny + iy], p);
    else unsafeAtomicAdd(&hist[(int64_t)ix * ny + iy], p);
  }
  if constexpr (LDS) {
    __syncthreads();
    for (int k = threadIdx.x; k < nb; k += kBlock)
      if (lh[k] != 0.0) unsafeAtomicAdd(&hist[k], lh[k]);
  }
}

template <typename T>
hipError_t launch_irradiance(int64_t n, const T* x, const T* y, const T* power,
                             const double* x_edges, int nx, const double* y_edges, int ny,
                             double* hist, hipStream_t stream) {
  if (nx < 1 || ny < 1) return hipErrorInvalidValue;
  if ((int64_t)nx * ny <= kMaxLdsBins)
    hipLaunchKernelGGL((irradiance_kernel<T, true>), dim3(grid_for(n)), dim3(kBlock), 0, stream,
                       n, x, y, power, x_edges, nx, y_edges, ny, hist);
  else
    hipLaunchKernelGGL((irradiance_kernel<T, false>), dim3(grid_for(n)), dim3(kBlock), 0, stream,
                       n, x, y, power, x_edges, nx, y_edges, ny, hist);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_spot_moments(int64_t n, const T* x, const T* y, const T* inten, double* out6,
                               hipStream_t stream) {
  hipLaunchKernelGGL((spot_moments_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, stream, n, x,
                     y, inten, out6);
  return hipGetLastError();
}

template <typename T>
hipError_t launch_spot_max_r2(int64_t n, const T* x, const T* y, const T* inten, double cx,
                              double cy, double* out1, hipStream_t stream) {
  hipLaunchKernelGGL((spot_max_r2_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, stream, n, x, y,
                     inten, cx, cy, out1);
  return hipGetLastError();
}

// ol_pupil_points: the deterministic pupil samplers on the device (distribution.py:161-220
// of the reference: "hexapolar", "uniform"), one point per lane from its INDEX -- no host
// sampling pass, no upload.  Values are formed in fp64 exactly as the reference's NumPy code
// forms them (the same products and sums, no contraction) and then rounded to T; the one
// thing that is not NumPy's is the libm behind cos / sin (<= 1 ulp apart in fp64; after
// rounding to fp32 the planes agree bit for bit but for ~1 value in 1e8).
//   hexapolar(R): point 0 = the centre; ring i = 1..R holds 6 i points at radius
//     linspace(0, 1, R + 1)[i] = i * (1 / R) (the last one exactly 1) and azimuth
//     j * (2 pi / (6 i)).  Ring of point p (q = p - 1): the largest i with 3 i (i - 1) <= q,
//     from an fp64 square root corrected by integer arithmetic.
//   uniform(n): the n x n grid linspace(-1, 1, n)^2 masked to the unit disc, row-major.  The
//     mask is evaluated by the caller (the reference's own x^2 + y^2 <= 1 on the same
//     squares): row j keeps the columns [first[j], first[j] + offset[j + 1] - offset[j]);
//     the lane finds its row by bisection of `offset` (n + 1 entries, L2-resident).
namespace {
__device__ __forceinline__ double linspace_at(int64_t i, int64_t last, double start, double stop,
                                              double step) {
  // numpy.linspace: y = arange(num) * step + start (two roundings), y[-1] = stop
#pragma clang fp contract(off)
  const double v = (double)i * step;
  return i == last ? stop : v + start;
}
}  // namespace

template <typename T>
__global__ __launch_bounds__(kBlock) void pupil_hexapolar_kernel(int32_t rings, int64_t n,
                                                                 T* __restrict__ x,
                                                                 T* __restrict__ y) {
#pragma clang fp contract(off)
  const double step = 1.0 / (double)rings;
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n;
       p += (int64_t)gridDim.x * kBlock) {
    if (p == 0) {
      x[0] = T(0);
      y[0] = T(0);
      continue;
    }
    const int64_t q = p - 1;
    int64_t i = (int64_t)((3.0 + sqrt(9.0 + 12.0 * (double)q)) / 6.0);
    while (3 * i * (i - 1) > q) --i;
    while (3 * (i + 1) * i <= q) ++i;
    const int64_t j = q - 3 * i * (i - 1);
    const double theta = (double)j * (6.283185307179586 / (double)(6 * i));
    const double r = linspace_at(i, rings, 0.0, 1.0, step);
    x[p] = (T)(r * cos(theta));
    y[p] = (T)(r * sin(theta));
  }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void pupil_uniform_kernel(int32_t side, int64_t n,
                                                               const int32_t* __restrict__ first,
                                                               const int64_t* __restrict__ offset,
                                                               T* __restrict__ x,
                                                               T* __restrict__ y) {
#pragma clang fp contract(off)
  const double step = 2.0 / (double)(side - 1);
  for (int64_t p = (int64_t)blockIdx.x * kBlock + threadIdx.x; p < n;
       p += (int64_t)gridDim.x * kBlock) {
    int lo = 0, hi = side;  // invariant: offset[lo] <= p < offset[hi]
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (offset[mid] <= p) lo = mid; else hi = mid;
    }
    const int64_t col = (int64_t)first[lo] + (p - offset[lo]);
    x[p] = (T)linspace_at(col, side - 1, -1.0, 1.0, step);
    y[p] = (T)linspace_at(lo, side - 1, -1.0, 1.0, step);
  }
}

template <typename T>
hipError_t launch_pupil_points(int kind, int32_t param, int64_t n, const int32_t* first,
                               const int64_t* offset, T* x, T* y, hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  if (kind == 0)
    hipLaunchKernelGGL((pupil_hexapolar_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, stream,
                       param, n, x, y);
  else
    hipLaunchKernelGGL((pupil_uniform_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, stream,
                       param, n, first, offset, x, y);
  return hipGetLastError();
}
template hipError_t launch_pupil_points<float>(int, int32_t, int64_t, const int32_t*,
                                               const int64_t*, float*, float*, hipStream_t);
template hipError_t launch_pupil_points<double>(int, int32_t, int64_t, const int32_t*,
                                                const int64_t*, double*, double*, hipStream_t);

// ol_math_probe: the kernels' own arithmetic primitives (Math<T>, surface_math.h) applied
// element-wise -- lets the GPU tests hold the hardware-seed fp64 quotient / square root
// (OL_FAST_F64) and the fp32 1-ulp instructions to their stated error bounds and to IEEE's
// special values, on the device, outside any trace.
template <typename T>
__global__ __launch_bounds__(kBlock) void math_probe_kernel(int op, int64_t n,
                                                            const T* __restrict__ a,
                                                            const T* __restrict__ b,
                                                            T* __restrict__ out) {
  using m = Math<T>;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * kBlock) {
    const T x = a[j], y = b ? b[j] : T(0);
    T r;
    switch (op) {
      case 0: r = m::rcp(x); break;
      case 1: r = m::div(x, y); break;
      case 2: r = m::sqrt(x); break;
      default: r = m::rsqrt(x); break;
    }
    out[j] = r;
  }
}

template <typename T>
hipError_t launch_math_probe(int op, int64_t n, const T* a, const T* b, T* out,
                             hipStream_t stream) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL((math_probe_kernel<T>), dim3(grid_for(n)), dim3(kBlock), 0, stream, op, n,
                     a, b, out);
  return hipGetLastError();
}
template hipError_t launch_math_probe<float>(int, int64_t, const float*, const float*, float*,
                                             hipStream_t);
template hipError_t launch_math_probe<double>(int, int64_t, const double*, const double*, double*,
                                              hipStream_t);

// Write-only streaming yardstick (ol_stream_fill): the STORE PATTERN of a record-all trace
// launch with the arithmetic taken out -- the buffer is `planes` planes of n elements, every
// lane stores ONE element of WIDTH bytes into each plane (the same non-temporal stores,
// trace_kernel.hip: store_plane; one workgroup per 256 consecutive elements, like the trace
// kernels).  What a kernel that ONLY writes sustains on this part for that footprint and
// that pattern: the ceiling the record-all kernels are held against.  (A linear one-store-
// per-lane fill of the same bytes is bound by wave launch rate instead: 4.6-4.8 TB/s.)
template <typename V>
__global__ __launch_bounds__(kBlock) void stream_fill_kernel(V* __restrict__ dst, int64_t n,
                                                             int64_t stride, int planes,
                                                             V value) {
  const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  if (j >= n) return;
  V* p = dst + j;
  for (int k = 0; k < planes; ++k, p += stride) __builtin_nontemporal_store(value, p);
}

hipError_t launch_stream_fill(void* dst, int64_t bytes, int width, int planes, uint32_t pattern,
                              hipStream_t stream) {
  typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  if (bytes <= 0) return hipSuccess;
  if (planes < 1) planes = 1;
  const int64_t n = bytes / width / planes;  // elements per plane
  if (n <= 0) return hipSuccess;
  const int64_t blocks = (n + kBlock - 1) / kBlock;
  if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
  if (width == 4)
    hipLaunchKernelGGL((stream_fill_kernel<uint32_t>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       stream, static_cast<uint32_t*>(dst), n, n, planes, pattern);
  else if (width == 8)
    hipLaunchKernelGGL((stream_fill_kernel<u32x2>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       stream, static_cast<u32x2*>(dst), n, n, planes, u32x2{pattern, pattern});
  else
    hipLaunchKernelGGL((stream_fill_kernel<u32x4>), dim3((unsigned)blocks), dim3(kBlock), 0,
                       stream, static_cast<u32x4*>(dst), n, n, planes,
                       u32x4{pattern, pattern, pattern, pattern});
  return hipGetLastError();
}

// ---- ol_wavefront_fit: the fitted reference of CentroidStrategy / BestFitStrategy -----------
// One launch per pass (wavefront_fit_device.h).  Every block leaves its sums as one row of the
// workspace; the block that finishes LAST (a ticket taken after a release fence) adds the rows
// in a fixed order -- the result does not depend on which block that was -- and its thread 0
// takes the decision the reference takes on the host at that point.  The next pass is the next
// launch on the same stream: no read-back anywhere in the chain.
template <int PASS>
__global__ __launch_bounds__(kBlock) void fit_pass_kernel(FitArgs a) {
  FitState* st = reinterpret_cast<FitState*>(a.workspace);
  double* rows = a.workspace + kFitStateDoubles;
  const FitState seen = *st;  // as the previous pass left it
  WavefrontConsts<double> ref{};
  if constexpr (PASS == kPassMean) ref = load_consts(as_const(a.out));
  double s[kFitSums];
#pragma unroll
  for (int k = 0; k < kFitSums; ++k) s[k] = 0.0;
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < a.n;
       j += (int64_t)gridDim.x * kBlock) {
    const FitRay r{a.ray[0][j], a.ray[1][j], a.ray[2][j], a.ray[3][j], a.ray[4][j],
                   a.ray[5][j], a.ray[6][j], a.ray[7][j], a.px[j],     a.py[j]};
    fit_accumulate<PASS>(a.p, seen, ref, r, s);
  }
  __shared__ double part[kBlock / 64][kFitSums];
  __shared__ double fin[kBlock / kFitSums][kFitSums];
  __shared__ bool last;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    double o[kFitSums];
#pragma unroll
    for (int k = 0; k < kFitSums; ++k) o[k] = __shfl_down(s[k], off, 64);
#pragma unroll
    for (int k = 0; k < kFitSums; ++k) s[k] += o[k];
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < kFitSums; ++k) part[wave][k] = s[k];
  }
  __syncthreads();
  if (threadIdx.x < kFitSums) {
    double v = 0;
    for (int w = 0; w < kBlock / 64; ++w) v += part[w][threadIdx.x];
    rows[(int64_t)blockIdx.x * kFitSums + threadIdx.x] = v;
    __threadfence();  // the row before the ticket
  }
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(&st->ticket[PASS], 1u) == gridDim.x - 1;
  __syncthreads();
  if (!last) return;
  __threadfence();
  {  // rows of all blocks, in block order: 16 interleaved chains per sum, then those in order
    // (agent-scope loads: past this XCD's L2; eight in flight per thread, added in row order)
    const int k = threadIdx.x % kFitSums, g = threadIdx.x / kFitSums;
    constexpr unsigned kStep = kBlock / kFitSums, kFly = 8;
    double v = 0;
    for (unsigned b = g; b < gridDim.x; b += kStep * kFly) {
      double t[kFly];
#pragma unroll
      for (unsigned u = 0; u < kFly; ++u) {
        const unsigned row = b + u * kStep;
        t[u] = row < gridDim.x ? __hip_atomic_load(&rows[(int64_t)row * kFitSums + k],
                                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                               : 0.0;
      }
#pragma unroll
      for (unsigned u = 0; u < kFly; ++u) v += t[u];
    }
    fin[g][k] = v;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tot[kFitSums];
    for (int k = 0; k < kFitSums; ++k) {
      double v = 0;
      for (int g = 0; g < kBlock / kFitSums; ++g) v += fin[g][k];
      tot[k] = v;
    }
    uint32_t bits = 0;
    fit_finish<PASS>(a.p, *st, tot, a.out, &bits);
    // one thread per pass, the passes in stream order: the first one WRITES the word
    if constexpr (PASS == kPassC1 || PASS == kPassB1) *a.status = bits;
    else if (bits) atomicOr(a.status, bits);
  }
}

hipError_t launch_wavefront_fit(const FitArgs& a, hipStream_t stream) {
  static_assert(kBlock % kFitSums == 0 && kBlock / 64 <= kBlock / kFitSums, "fit reduction shape");
  hipError_t e = hipMemsetAsync(a.workspace, 0, kFitStateDoubles * sizeof(double), stream);
  if (e != hipSuccess) return e;
  // four rays per lane before another block is worth its row in the finishing sum
  int64_t b = (a.n + 4 * kBlock - 1) / (4 * kBlock);
  const int cap = tuning().fit_grid > 0 ? tuning().fit_grid : kFitDefaultBlocks;
  const dim3 grid((unsigned)(b < 1 ? 1 : (b > cap ? cap : b)));
#define OL_FIT_PASS(P) \
  hipLaunchKernelGGL((fit_pass_kernel<P>), grid, dim3(kBlock), 0, stream, a)
  if (a.p.kind == kFitBestFit) {
    OL_FIT_PASS(kPassB1);
    OL_FIT_PASS(kPassB2);
  } else {
    OL_FIT_PASS(kPassC1);
    if (a.p.trim_std > 0.0) {  // strategy.py:417 (`robust_trim_std and robust_trim_std > 0`)
      OL_FIT_PASS(kPassC2);
      OL_FIT_PASS(kPassC3);
      OL_FIT_PASS(kPassC4);
    }
    if (!a.p.planar) OL_FIT_PASS(kPassC5);
  }
  OL_FIT_PASS(kPassMean);
#undef OL_FIT_PASS
  return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void wavefront_fitted_kernel(
    const WavefrontConsts<double>* ref, int64_t n, const double* __restrict__ x,
    const double* __restrict__ y, const double* __restrict__ z, const double* __restrict__ Ld,
    const double* __restrict__ Md, const double* __restrict__ Nd,
    const double* __restrict__ opd_in, const double* __restrict__ px,
    const double* __restrict__ py, double* opd_waves, double* pux, double* puy, double* puz) {
  const WavefrontConsts<double> w = load_consts(as_const(ref));
  for (int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x; j < n;
       j += (int64_t)gridDim.x * kBlock) {
    double pu[3];
    opd_waves[j] = wavefront_one<double, true>(w, x[j], y[j], z[j], Ld[j], Md[j], Nd[j],
                                               opd_in[j], px[j], py[j], pu);
    if (pux) {
      pux[j] = pu[0];
      puy[j] = pu[1];
      puz[j] = pu[2];
    }
  }
}

hipError_t launch_wavefront_fitted(const WavefrontConsts<double>* ref, int64_t n,
                                   const double* const rays[7], const double* px,
                                   const double* py, double* opd_waves, double* const pupil[3],
                                   hipStream_t stream) {
  hipLaunchKernelGGL(wavefront_fitted_kernel, dim3(grid_for(n)), dim3(kBlock), 0, stream, ref, n,
                     rays[0], rays[1], rays[2], rays[3], rays[4], rays[5], rays[6], px, py,
                     opd_waves, pupil ? pupil[0] : nullptr, pupil ? pupil[1] : nullptr,
                     pupil ? pupil[2] : nullptr);
  return hipGetLastError();
}

#define OL_INST(T)                                                                             \
  template hipError_t launch_raygen<T>(const RaygenDev&, const RaygenIn<T>&, int64_t,            \
                                       T* const[8], uint32_t*, hipStream_t);                    \
  template hipError_t launch_pol_intensity<T>(int64_t, const T*, bool, const T* const[3],      \
                                              const T*, const PolStateDev&, T*, uint32_t*,     \
                                              hipStream_t);                                    \
  template hipError_t launch_wavefront<T>(const WavefrontDev&, int64_t, const T* const[7],     \
                                          const T*, const T*, T*, T* const[3], hipStream_t);   \
  template hipError_t launch_pupil_fill<T>(int64_t, const T*, const T*, const T*, const T*,    \
                                           const double[3], const int32_t*, int32_t, int32_t,  \
                                           int32_t, double*, hipStream_t);                     \
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
