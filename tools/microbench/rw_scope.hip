// Micro-benchmark: store flavours for the record-all pattern (one ray per lane,
// 2 MiB-aligned plane stride): plain, nontemporal, agent-scope (sc1, write-through).
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__device__ __forceinline__ void st(float* p, float v) {
  if (MODE == 0) *p = v;
  else if (MODE == 1) __builtin_nontemporal_store(v, p);
  else if (MODE == 2) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int MODE, int LDMODE, int BS = 256>
__global__ __launch_bounds__(BS) void rw(const float* in, float* out, long n, long stride, int rows) {
  long base = (long)blockIdx.x * BS + threadIdx.x;
  if (base >= n) return;
  float s[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    s[k] = LDMODE ? __builtin_nontemporal_load(in + (long)k * stride + base) : in[(long)k * stride + base];
  for (int r = 0; r < rows; ++r) {
#pragma unroll
    for (int a = 0; a < 8; ++a) {
#pragma unroll
      for (int k = 0; k < 8; ++k) s[k] = s[k] * s[(k + 1) & 7] + 0.5f;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) st<MODE>(out + ((long)(r * 8 + k)) * stride + base, s[k]);
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

template <int MODE, int LDMODE, int BS = 256>
void run(const float* in, float* out, long n, long stride, const char* label) {
  unsigned blocks = (unsigned)((n + BS - 1) / BS);
  float best = 1e9;
  for (int rep = 0; rep < 3; ++rep) {
    float ms = time_ms([&] { hipLaunchKernelGGL((rw<MODE, LDMODE, BS>), dim3(blocks), dim3(BS), 0, 0, in, out, n, stride, 12); }, 10);
    best = ms < best ? ms : best;
  }
  double gb = 4.0 * n * (8 + 96) / 1e9;
  printf("%-32s best %.3f ms  %.0f GB/s\n", label, best, gb / best * 1e3);
}

int main() {
  const long n = 10000000, stride = 10485760;
  float *in, *out;
  hipMalloc(&in, 4 * stride * 8); hipMalloc(&out, 4 * stride * 96);
  hipMemset(in, 0, 4 * stride * 8);
  for (int pass = 0; pass < 2; ++pass) {
    run<0, 0>(in, out, n, stride, "plain store, plain load");
    run<1, 0>(in, out, n, stride, "nontemporal store");
    run<2, 0>(in, out, n, stride, "agent-scope (sc1) store");
    run<3, 0>(in, out, n, stride, "system-scope store");
    run<0, 1>(in, out, n, stride, "plain store, nt load");
    run<1, 0, 64>(in, out, n, stride, "nt store, block 64");
    run<1, 0, 128>(in, out, n, stride, "nt store, block 128");
    run<1, 0, 512>(in, out, n, stride, "nt store, block 512");
    run<1, 0, 1024>(in, out, n, stride, "nt store, block 1024");
    run<1, 1>(in, out, n, stride, "nt store, nt load");
  }
  return 0;
}
