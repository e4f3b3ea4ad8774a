// Micro-benchmark: what HBM write bandwidth can the record-all store pattern reach?
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 stream_write.hip -o sw && ./sw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

// each thread writes `planes` 16-byte vectors, one per plane (plane stride = n floats)
template <bool NT>
__global__ __launch_bounds__(256) void multi_plane_write(float* out, long n, int planes) {
  long base = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (base >= n) return;
  f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
  for (int p = 0; p < planes; ++p) {
    f4* dst = reinterpret_cast<f4*>(out + (long)p * n + base);
    if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
    v.x += 1.f;
  }
}

// same bytes, but each block owns a contiguous chunk of every plane: chunk = rpb rays
template <bool NT>
__global__ __launch_bounds__(256) void multi_plane_write_persistent(float* out, long n, int planes, int iters) {
  for (int it = 0; it < iters; ++it) {
    long base = (((long)blockIdx.x * iters + it) * 256 + threadIdx.x) * 4;
    if (base >= n) return;
    f4 v = {1.f, 2.f, 3.f, (float)threadIdx.x};
    for (int p = 0; p < planes; ++p) {
      f4* dst = reinterpret_cast<f4*>(out + (long)p * n + base);
      if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
      v.x += 1.f;
    }
  }
}

template <bool NT>
__global__ __launch_bounds__(256) void linear_fill(float* out, long total) {
  long stride = (long)gridDim.x * 256 * 4;
  f4 v = {1.f, 2.f, 3.f, 4.f};
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < total; i += stride) {
    f4* dst = reinterpret_cast<f4*>(out + i);
    if (NT) __builtin_nontemporal_store(v, dst); else *dst = v;
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
  const long n = 10000000; const int planes = 104;
  float* buf; hipMalloc(&buf, sizeof(float) * n * planes);
  const double gb = sizeof(float) * (double)n * planes / 1e9;
  unsigned blocks = (unsigned)((n / 4 + 255) / 256);
  float ms;
  ms = time_ms([&] { hipLaunchKernelGGL(multi_plane_write<false>, dim3(blocks), dim3(256), 0, 0, buf, n, planes); }, 10);
  printf("multi_plane_write      plain : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  ms = time_ms([&] { hipLaunchKernelGGL(multi_plane_write<true>, dim3(blocks), dim3(256), 0, 0, buf, n, planes); }, 10);
  printf("multi_plane_write      nt    : %.3f ms  %.0f GB/s\n", ms, gb / ms * 1e3);
  for (int iters : {2, 4, 8}) {
    unsigned b2 = (blocks + iters - 1) / iters;
    ms = time_ms([&] { hipLaunchKernelGGL(multi_plane_write_persistent<true>, dim3(b2), dim3(256), 0, 0, buf, n, planes, iters); }, 10);
    printf("multi_plane chunk x%d   nt    : %.3f ms  %.0f GB/s\n", iters, ms, gb / ms * 1e3);
  }
  for (unsigned g : {2048u, 4096u, 16384u}) {
    ms = time_ms([&] { hipLaunchKernelGGL(linear_fill<false>, dim3(g), dim3(256), 0, 0, buf, n * planes); }, 10);
    printf("linear_fill grid %5u plain : %.3f ms  %.0f GB/s\n", g, ms, gb / ms * 1e3);
    ms = time_ms([&] { hipLaunchKernelGGL(linear_fill<true>, dim3(g), dim3(256), 0, 0, buf, n * planes); }, 10);
    printf("linear_fill grid %5u nt    : %.3f ms  %.0f GB/s\n", g, ms, gb / ms * 1e3);
  }
  hipFree(buf);
  return 0;
}
