// Micro-benchmark: does the block->ray mapping change the HBM rate of the
// record-all traffic pattern (8 plane reads + 104 plane writes, 16 B/lane)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int MAP, int BS>
__global__ __launch_bounds__(BS) void rw(const float* in, float* out, long n, int rows, unsigned nb) {
  unsigned b = blockIdx.x;
  if (MAP == 1) {  // XCD-contiguous: XCD k (= b % 8) owns the k-th eighth of the rays
    unsigned per = (nb + 7) / 8;
    b = (blockIdx.x % 8) * per + blockIdx.x / 8;
    if (b >= nb) return;
  }
  long base = ((long)b * BS + threadIdx.x) * 4;
  if (base >= n) return;
  f4 s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) s[k] = *reinterpret_cast<const f4*>(in + (long)k * n + base);
  for (int r = 0; r < rows; ++r) {
#pragma unroll
    for (int a = 0; a < 8; ++a) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] = s[k] * s[(k + 1) & 7] + 0.5f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) *reinterpret_cast<f4*>(out + ((long)(r * 8 + k)) * n + base) = s[k];
  }
}

// row-major order of stores within a block reversed: plane-major across two rows at a time
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

template <int MAP, int BS>
void run(const float* in, float* out, long n, const char* label) {
  unsigned nb = (unsigned)((n / 4 + BS - 1) / BS);
  unsigned grid = MAP == 1 ? ((nb + 7) / 8) * 8 : nb;
  float best = 1e9, sum = 0;
  for (int rep = 0; rep < 3; ++rep) {
    float ms = time_ms([&] { hipLaunchKernelGGL((rw<MAP, BS>), dim3(grid), dim3(BS), 0, 0, in, out, n, 13, nb); }, 10);
    best = ms < best ? ms : best; sum += ms;
  }
  double gb = 4.0 * n * (8 + 104) / 1e9;
  printf("%-28s best %.3f ms (%.0f GB/s)  mean %.3f\n", label, best, gb / best * 1e3, sum / 3);
}

int main() {
  const long n = 10000000;
  float *in, *out;
  hipMalloc(&in, 4 * n * 8); hipMalloc(&out, 4 * n * 104);
  hipMemset(in, 0, 4 * n * 8);
  for (int pass = 0; pass < 2; ++pass) {
    run<0, 256>(in, out, n, "linear map, block 256");
    run<1, 256>(in, out, n, "XCD-contiguous, block 256");
    run<0, 512>(in, out, n, "linear map, block 512");
    run<1, 512>(in, out, n, "XCD-contiguous, block 512");
    run<0, 1024>(in, out, n, "linear map, block 1024");
    run<0, 128>(in, out, n, "linear map, block 128");
    run<0, 64>(in, out, n, "linear map, block 64");
  }
  return 0;
}
