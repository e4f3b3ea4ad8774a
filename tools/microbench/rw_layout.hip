// Micro-benchmark: layout of the record block.  Same bytes (8 planes read, 12 rows x 8
// values written per ray, one ray per lane, light arithmetic), three layouts:
//   soa   (rows, 8, stride): the product layout -- 8 four-byte stores per row, 96 streams
//   aos   (rows, n, 8)     : 32 contiguous bytes per ray and row -- 2 sixteen-byte stores,
//                            12 streams; surface.x would be a stride-8 view
//   soa4  (rows, 8, stride) with 4 rays per lane: 16-byte stores, 96 streams
#include <hip/hip_runtime.h>
#include <cstdio>

template <int BS>
__global__ __launch_bounds__(BS) void soa(const float* in, float* out, long n, long stride, int rows) {
  long base = (long)blockIdx.x * BS + threadIdx.x;
  if (base >= n) return;
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = in[(long)k * stride + base];
  for (int r = 0; r < rows; ++r) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] = s[k] * s[(k + 1) & 7] + 0.5f;
#pragma unroll
    for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(s[k], out + ((long)(r * 8 + k)) * stride + base);
  }
}

template <int BS, bool NT>
__global__ __launch_bounds__(BS) void aos(const float* in, float* out, long n, long stride, int rows) {
  long base = (long)blockIdx.x * BS + threadIdx.x;
  if (base >= n) return;
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = in[(long)k * stride + base];
  for (int r = 0; r < rows; ++r) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] = s[k] * s[(k + 1) & 7] + 0.5f;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4* p = reinterpret_cast<f4*>(out + ((long)r * stride + base) * 8);
    f4 v0 = {s[0], s[1], s[2], s[3]}, v1 = {s[4], s[5], s[6], s[7]};
    if (NT) {
      __builtin_nontemporal_store(v0, p);
      __builtin_nontemporal_store(v1, p + 1);
    } else {
      p[0] = v0;
      p[1] = v1;
    }
  }
}

template <int BS>
__global__ __launch_bounds__(BS) void soa4(const float* in, float* out, long n, long stride, int rows) {
  long base = ((long)blockIdx.x * BS + threadIdx.x) * 4;
  if (base >= n) return;
  float4 s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = *reinterpret_cast<const float4*>(in + (long)k * stride + base);
  for (int r = 0; r < rows; ++r) {
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        s[k].x = s[k].x * s[(k + 1) & 7].x + 0.5f; s[k].y = s[k].y * s[(k + 1) & 7].y + 0.5f;
        s[k].z = s[k].z * s[(k + 1) & 7].z + 0.5f; s[k].w = s[k].w * s[(k + 1) & 7].w + 0.5f;
      }
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<float4*>(out + ((long)(r * 8 + k)) * stride + base) = s[k];
  }
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
  const int rows = 12;
  float *in, *out;
  hipMalloc(&in, 4 * stride * 8); hipMalloc(&out, 4 * stride * 8 * rows);
  hipMemset(in, 0, 4 * stride * 8);
  const double gb = 4.0 * n * (8 + 8 * rows) / 1e9;
  for (int pass = 0; pass < 3; ++pass) {
    auto rep = [&](const char* label, auto f) {
      float best = 1e9;
      for (int r = 0; r < 3; ++r) { float ms = time_ms(f, 10); best = ms < best ? ms : best; }
      printf("%-44s best %.3f ms  %.0f GB/s\n", label, best, gb / best * 1e3);
    };
    unsigned b256 = (unsigned)((n + 255) / 256), b256_4 = (unsigned)((n / 4 + 255) / 256);
    rep("soa  (product layout), nt 4-byte stores", [&] { hipLaunchKernelGGL((soa<256>), dim3(b256), dim3(256), 0, 0, in, out, n, stride, rows); });
    rep("aos  32 B per ray-row, plain 16-byte stores", [&] { hipLaunchKernelGGL((aos<256, false>), dim3(b256), dim3(256), 0, 0, in, out, n, stride, rows); });
    rep("aos  32 B per ray-row, nt 16-byte stores", [&] { hipLaunchKernelGGL((aos<256, true>), dim3(b256), dim3(256), 0, 0, in, out, n, stride, rows); });
    rep("soa4 4 rays per lane, plain 16-byte stores", [&] { hipLaunchKernelGGL((soa4<256>), dim3(b256_4), dim3(256), 0, 0, in, out, n, stride, rows); });
  }
  return 0;
}
