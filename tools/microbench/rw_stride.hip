// Micro-benchmark: does the PLANE STRIDE of the record block change the HBM rate?
// 8 plane reads + 96 plane writes (12 rows), one ray per lane, 4-byte stores.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ __launch_bounds__(256) void rw(const float* in, float* out, long n, long stride, int rows) {
  long base = (long)blockIdx.x * 256 + threadIdx.x;
  if (base >= n) return;
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = in[(long)k * stride + base];
  for (int r = 0; r < rows; ++r) {
#pragma unroll
    for (int a = 0; a < 8; ++a) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] = s[k] * s[(k + 1) & 7] + 0.5f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) out[((long)(r * 8 + k)) * stride + base] = s[k];
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
  const long n = 10000000;
  const long maxstride = n + (1 << 23);
  float *in, *out;
  hipMalloc(&in, 4 * maxstride * 8); hipMalloc(&out, 4 * maxstride * 96);
  hipMemset(in, 0, 4 * maxstride * 8);
  auto up = [&](long align_bytes) { long a = align_bytes / 4; return (n + a - 1) / a * a; };
  long strides[] = {n, up(4096), up(16384), up(65536), up(262144), up(1 << 20), up(2 << 20),
                    up(4 << 20), up(8 << 20), up(1 << 20) + (1 << 18), up(2 << 20) + (1 << 19)};
  unsigned blocks = (unsigned)((n + 255) / 256);
  for (int pass = 0; pass < 2; ++pass)
    for (long st : strides) {
      float best = 1e9;
      for (int rep = 0; rep < 3; ++rep) {
        float ms = time_ms([&] { hipLaunchKernelGGL(rw, dim3(blocks), dim3(256), 0, 0, in, out, n, st, 12); }, 10);
        best = ms < best ? ms : best;
      }
      double gb = 4.0 * n * (8 + 96) / 1e9;
      printf("stride n%+9ld (%10ld elems, %% 2^20 B = %7ld): best %.3f ms  %.0f GB/s\n", st - n, st,
             (st * 4) % (1 << 20), best, gb / best * 1e3);
    }
  return 0;
}
