// Micro-benchmark: 8 plane reads + 104 plane writes per thread (the record-all traffic
// pattern) with a tunable amount of dependent ALU work between row stores.
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));

template <int RPT, int ALU>
__global__ __launch_bounds__(256) void rw_mix(const float* in, float* out, long n, int rows) {
  typedef float fv __attribute__((ext_vector_type(RPT)));
  long base = ((long)blockIdx.x * 256 + threadIdx.x) * RPT;
  if (base >= n) return;
  fv s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = *reinterpret_cast<const fv*>(in + (long)k * n + base);
  for (int r = 0; r < rows; ++r) {
#pragma unroll
    for (int a = 0; a < ALU; ++a) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] = s[k] * s[(k + 1) & 7] + 0.5f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<fv*>(out + ((long)(r * 8 + k)) * n + base) = s[k];
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

template <int RPT, int ALU>
void run(const float* in, float* out, long n) {
  unsigned blocks = (unsigned)((n / RPT + 255) / 256);
  float ms = time_ms([&] { hipLaunchKernelGGL((rw_mix<RPT, ALU>), dim3(blocks), dim3(256), 0, 0, in, out, n, 13); }, 10);
  double gb = 4.0 * n * (8 + 104) / 1e9;
  printf("RPT=%d ALU=%2d (%3d fma/ray/row): %.3f ms  %.0f GB/s\n", RPT, ALU, ALU * 8, ms, gb / ms * 1e3);
}

int main() {
  const long n = 10000000;
  float *in, *out;
  hipMalloc(&in, 4 * n * 8); hipMalloc(&out, 4 * n * 104);
  hipMemset(in, 0, 4 * n * 8);
  run<4, 0>(in, out, n); run<4, 4>(in, out, n); run<4, 8>(in, out, n); run<4, 16>(in, out, n); run<4, 32>(in, out, n);
  run<1, 0>(in, out, n); run<1, 8>(in, out, n); run<1, 16>(in, out, n); run<1, 32>(in, out, n);
  run<2, 0>(in, out, n); run<2, 16>(in, out, n);
  return 0;
}
