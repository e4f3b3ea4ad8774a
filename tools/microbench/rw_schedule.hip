// Micro-benchmark (round 3): can the record-all store pattern -- 8 planes
// read, 12 rows x 8 planes written, one ray per lane, 4-byte non-temporal stores, 2 MiB-aligned
// plane stride -- be SCHEDULED or LAID OUT so that it gets closer to what a plain fill of the
// same bytes sustains?  Same bytes (4.16 GB), same light arithmetic in every variant:
//   soa          the product pattern: one workgroup per 256-ray tile
//   persistent   2048 / 4096 resident workgroups walk the tiles grid-stride (no launch tail,
//                tiles of one workgroup far apart)
//   chunked      workgroup w owns a CONTIGUOUS run of tiles (its stores sweep each plane
//                window sequentially: fewer open DRAM pages per workgroup over time)
//   rowsync      persistent workgroups, resident set advances through the rows together: a
//                grid-wide arrive counter per (round, row) in front of each row's stores
//   lds_burst    the row tile is staged in LDS and written plane by plane, each WAVE storing
//                ONE plane's 1 KB (4 consecutive 256-B wave stores to the same plane) instead
//                of 256 B to each of 8 planes
//   tiled        layout (rows, tiles, 8, 256): the 8 planes of a tile adjacent -- one 8 KB
//                burst per workgroup and row, 12 streams instead of 96 (NOT drop-in: Surface.x
//                would be a strided view)
//   fill / copy  references: the same 3.84 GB written / 4.16 GB copied by a streaming kernel
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ROWS 12
__device__ __forceinline__ void churn(float (&s)[8]) {
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = s[k] * s[(k + 1) & 7] + 0.5f;
}
__device__ __forceinline__ void load8(const float* in, long stride, long i, float (&s)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = in[(long)k * stride + i];
}
__device__ __forceinline__ void store_row(float* out, long stride, int r, long i, const float (&s)[8]) {
#pragma unroll
  for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(s[k], out + ((long)(r * 8 + k)) * stride + i);
}

__global__ __launch_bounds__(256) void k_soa(const float* in, float* out, long n, long stride) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s[8];
  load8(in, stride, i, s);
  for (int r = 0; r < ROWS; ++r) { churn(s); store_row(out, stride, r, i, s); }
}

// tiles visited grid-stride (CHUNK = false) or as one contiguous run per workgroup (true)
template <bool CHUNK>
__global__ __launch_bounds__(256) void k_persistent(const float* in, float* out, long n, long stride) {
  const long ntiles = (n + 255) / 256;
  const long per = (ntiles + gridDim.x - 1) / gridDim.x;
  for (long q = 0; q < per; ++q) {
    const long tile = CHUNK ? (long)blockIdx.x * per + q : q * gridDim.x + blockIdx.x;
    if (tile >= ntiles) break;
    const long i = tile * 256 + threadIdx.x;
    if (i >= n) continue;
    float s[8];
    load8(in, stride, i, s);
    for (int r = 0; r < ROWS; ++r) { churn(s); store_row(out, stride, r, i, s); }
  }
}

__global__ __launch_bounds__(256) void k_rowsync(const float* in, float* out, long n, long stride,
                                                 unsigned* ctr) {
  const long ntiles = (n + 255) / 256;
  const long per = (ntiles + gridDim.x - 1) / gridDim.x;
  for (long q = 0; q < per; ++q) {
    const long tile = q * gridDim.x + blockIdx.x;
    const long i = tile * 256 + threadIdx.x;
    const bool live = tile < ntiles && i < n;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live) load8(in, stride, i, s);
    for (int r = 0; r < ROWS; ++r) {
      churn(s);
      // every resident workgroup reaches (round q, row r) before anyone stores it
      if (threadIdx.x == 0) {
        const unsigned slot = (unsigned)(q * ROWS + r);
        __hip_atomic_fetch_add(&ctr[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (bounded: a grid that is not fully resident must not hang the box)
        for (int spin = 0; spin < (1 << 20) &&
             __hip_atomic_load(&ctr[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x;
             ++spin)
          __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      if (live) store_row(out, stride, r, i, s);
    }
  }
}

__global__ __launch_bounds__(256) void k_lds_burst(const float* in, float* out, long n, long stride) {
  __shared__ float tile[8][256];
  const long base = (long)blockIdx.x * 256;
  const long i = base + threadIdx.x;
  float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (i < n) load8(in, stride, i, s);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  for (int r = 0; r < ROWS; ++r) {
    churn(s);
#pragma unroll
    for (int k = 0; k < 8; ++k) tile[k][threadIdx.x] = s[k];
    __syncthreads();
    // wave w writes planes 2w and 2w + 1, 1 KB each, as four consecutive 256-B stores
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int k = 2 * wave + p;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const long j = base + c * 64 + lane;
        if (j < n) __builtin_nontemporal_store(tile[k][c * 64 + lane], out + ((long)(r * 8 + k)) * stride + j);
      }
    }
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void k_tiled(const float* in, float* out, long n, long stride) {
  const long ntiles = (n + 255) / 256;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s[8];
  load8(in, stride, i, s);
  for (int r = 0; r < ROWS; ++r) {
    churn(s);
    float* o = out + (((long)r * ntiles + blockIdx.x) * 8) * 256 + threadIdx.x;
#pragma unroll
    for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(s[k], o + k * 256);
  }
}

__global__ __launch_bounds__(256) void k_fill(float* out, long total) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  f4 v = {1.f, 2.f, 3.f, 4.f};
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 1024)
    *reinterpret_cast<f4*>(out + i) = v;
}
__global__ __launch_bounds__(256) void k_copy(const float* in, float* out, long total) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += (long)gridDim.x * 1024)
    *reinterpret_cast<f4*>(out + i) = *reinterpret_cast<const f4*>(in + i);
}

template <typename F>
float time_ms(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < reps; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const long n = 10000000, stride = 10485760;
  float *in, *out; unsigned* ctr;
  hipMalloc(&in, 4 * stride * 8); hipMalloc(&out, 4 * stride * 8 * (ROWS + 1));
  hipMalloc(&ctr, 4 * 65536);
  hipMemset(in, 0, 4 * stride * 8);
  const double gb = 4.0 * n * (8 + 8 * ROWS) / 1e9;
  const unsigned b256 = (unsigned)((n + 255) / 256);
  for (int pass = 0; pass < 3; ++pass) {
    auto rep = [&](const char* label, double bytes_gb, auto f) {
      float best = 1e9;
      for (int r = 0; r < 3; ++r) { float ms = time_ms(f, 10); best = ms < best ? ms : best; }
      printf("%-58s best %.3f ms  %.0f GB/s\n", label, best, bytes_gb / best * 1e3);
    };
    rep("soa         (product pattern, 39063 workgroups)", gb, [&] { hipLaunchKernelGGL(k_soa, dim3(b256), dim3(256), 0, 0, in, out, n, stride); });
    for (unsigned g : {2048u, 4096u}) {
      char l[96];
      snprintf(l, sizeof l, "persistent  %u workgroups, grid-stride tiles", g);
      rep(l, gb, [&] { hipLaunchKernelGGL((k_persistent<false>), dim3(g), dim3(256), 0, 0, in, out, n, stride); });
      snprintf(l, sizeof l, "chunked     %u workgroups, contiguous run of tiles each", g);
      rep(l, gb, [&] { hipLaunchKernelGGL((k_persistent<true>), dim3(g), dim3(256), 0, 0, in, out, n, stride); });
    }
    // (the resident set must hold the whole grid or the arrive counters deadlock: 4 / CU)
    rep("rowsync     1024 workgroups, grid-wide arrive per row", gb, [&] {
      hipMemsetAsync(ctr, 0, 4 * 65536, 0);
      hipLaunchKernelGGL(k_rowsync, dim3(1024), dim3(256), 0, 0, in, out, n, stride, ctr); });
    rep("lds_burst   one plane's 1 KB per wave via LDS", gb, [&] { hipLaunchKernelGGL(k_lds_burst, dim3(b256), dim3(256), 0, 0, in, out, n, stride); });
    rep("tiled       (rows, tiles, 8, 256): 8 KB burst per row", gb, [&] { hipLaunchKernelGGL(k_tiled, dim3(b256), dim3(256), 0, 0, in, out, n, stride); });
    rep("fill        3.84 GB written, 16-byte stores", 4.0 * n * 8 * ROWS / 1e9, [&] { hipLaunchKernelGGL(k_fill, dim3(8192), dim3(256), 0, 0, out, (long)n * 8 * ROWS); });
    const long half = (long)(gb * 1e9 / 8);  // floats: 2.08 GB read from the upper half of out
    rep("copy        2.08 GB read + 2.08 GB written", gb, [&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, out + half, out, half); });
  }
  return 0;
}
