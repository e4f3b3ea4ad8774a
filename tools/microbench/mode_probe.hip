// mode_probe.hip -- round 5: the record-all store pattern over TIME.  vmm_pairs.hip showed every
// pair of half-block pieces go from 5.7-6.5 to 7.1-7.2 TB/s at one moment ~20 s into the process
// (one probe at 3.7 TB/s, everything fast afterwards): a state of the PART, not of the place.
// This program writes one plain block and the magic window of a 40 GiB arena (+30 GiB) in turns,
// prints the rate of every launch that differs from the last printed one by > 2 % (and every
// 2 s anyway) with wall-clock stamps -- tools/gpu_r05.sh samples the clocks / power / temperature
// of the part beside it -- then looks at how long the fast state survives idle gaps.
// build: hipcc --offload-arch=gfx950 -O3 -o mode_probe mode_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d: %s\n", hipGetErrorString(e_), __FILE__, __LINE__, #x); exit(2); } } while (0)

__global__ __launch_bounds__(256) void fill(uint32_t* __restrict__ dst, int64_t n, int64_t stride,
                                            int planes, uint32_t v) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t* p = dst + j;
  for (int k = 0; k < planes; ++k, p += stride) __builtin_nontemporal_store(v, p);
}
// the same stores behind ~`work` dependent FMAs per plane: a stand-in for a trace kernel's arithmetic
__global__ __launch_bounds__(256) void fill_alu(uint32_t* __restrict__ dst, int64_t n, int64_t stride,
                                                int planes, int work) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t* p = dst + j;
  float a = (float)j * 1e-9f, b = 1.0001f;
  for (int k = 0; k < planes; ++k, p += stride) {
    for (int w = 0; w < work; ++w) a = __builtin_fmaf(a, b, 0.5f);
    __builtin_nontemporal_store(__float_as_uint(a), p);
  }
}

static double wall() {
  return std::chrono::duration<double>(std::chrono::system_clock::now().time_since_epoch()).count();
}
static hipEvent_t e0, e1;
static const int64_t n = 10485760;
static const int planes = 104;
static double one(void* va, int work = 0) {
  const unsigned blocks = (unsigned)((n + 255) / 256);
  CK(hipEventRecord(e0));
  if (work) hipLaunchKernelGGL(fill_alu, dim3(blocks), dim3(256), 0, 0, (uint32_t*)va, n, n, planes, work);
  else hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, (uint32_t*)va, n, n, planes, 1u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)n * planes * 4 / (ms * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const size_t GiB = 1ull << 30;
  const double run_s = argc > 1 ? atof(argv[1]) : 40.0;
  const int work = argc > 2 ? atoi(argv[2]) : 0;
  CK(hipSetDevice(0));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t bytes = (size_t)n * 4 * planes;
  char* arena; CK(hipMalloc((void**)&arena, 40 * GiB));
  char* plain; CK(hipMalloc((void**)&plain, bytes));
  char* magic = arena + 30 * GiB;
  const double t0 = wall();
  printf("# t0 = %.3f (unix time); work = %d FMAs per store; columns: seconds, plain TB/s, +30 GiB window TB/s\n", t0, work);
  double lp = 0, lm = 0, last_print = -10;
  long launches = 0;
  while (wall() - t0 < run_s) {
    const double p = one(plain, work), m = one(magic, work);
    launches += 2;
    const double t = wall() - t0;
    if (fabs(p - lp) > 0.02 * lp || fabs(m - lm) > 0.02 * lm || t - last_print > 2.0) {
      printf("%8.3f  %.2f  %.2f\n", t, p, m);
      lp = p; lm = m; last_print = t;
    }
  }
  printf("# %ld launches in %.1f s\n", launches, wall() - t0);
  // idle gaps: does the state survive?
  for (int gap_ms : {1, 10, 50, 200, 1000, 3000}) {
    std::this_thread::sleep_for(std::chrono::milliseconds(gap_ms));
    printf("gap %4d ms:", gap_ms);
    for (int k = 0; k < 6; ++k) printf("  %.2f/%.2f", one(plain, work), one(magic, work));
    printf("   (t = %.2f)\n", wall() - t0);
    // load again for a second before the next gap
    const double t1 = wall();
    while (wall() - t1 < 1.0) { one(plain, work); one(magic, work); }
    printf("          after 1 s of load again: %.2f/%.2f\n", one(plain, work), one(magic, work));
  }
  return 0;
}
