// pace_probe.hip -- round 5: do the record-all stores like to be PACED?
// mode_probe.hip: with 8 dependent FMAs in front of every store the arithmetic-free fill goes from
// 7.09 to 7.43 TB/s in a placed window (and from 5.86 to 5.66 in a plain block).  The trace kernels
// issue their stores in BURSTS: ~76 vector instructions of one surface, then the 8 planes of its
// row back to back.  Same arithmetic per store, two schedules:
//   even   `work` FMAs, one store, `work` FMAs, one store ...
//   burst  8 x `work` FMAs, then 8 stores back to back (a row), 13 times
// on the best window of a 40 GiB arena and on a plain block.  If even >> burst, the row's stores
// are worth spreading through the next surface's arithmetic.
// build: hipcc --offload-arch=gfx950 -O3 -o pace_probe pace_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d: %s\n", hipGetErrorString(e_), __FILE__, __LINE__, #x); exit(2); } } while (0)

template <int WORK, bool BURST>
__global__ __launch_bounds__(256) void fill(uint32_t* __restrict__ dst, int64_t n, int64_t stride) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t* p = dst + j;
  float a = (float)j * 1e-9f;
  const float b = 1.0001f;
  if constexpr (BURST) {
    for (int row = 0; row < 13; ++row) {
      float v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int w = 0; w < WORK; ++w) a = __builtin_fmaf(a, b, 0.5f);
        v[k] = a;
      }
#pragma unroll
      for (int k = 0; k < 8; ++k, p += stride) __builtin_nontemporal_store(__float_as_uint(v[k]), p);
    }
  } else {
    for (int k = 0; k < 104; ++k, p += stride) {
#pragma unroll
      for (int w = 0; w < WORK; ++w) a = __builtin_fmaf(a, b, 0.5f);
      __builtin_nontemporal_store(__float_as_uint(a), p);
    }
  }
}

// the plain fill with (a) an s_sleep after every store, (b) a cap on resident workgroups per CU
// (dynamic LDS: 160 KB per CU / cap)
template <int SLEEP>
__global__ __launch_bounds__(256) void fill_sleep(uint32_t* __restrict__ dst, int64_t n, int64_t stride) {
  extern __shared__ char lds_[];
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t* p = dst + j;
  for (int k = 0; k < 104; ++k, p += stride) {
    __builtin_nontemporal_store(1u, p);
    if constexpr (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
  }
}

static hipEvent_t e0, e1;
static const int64_t n = 10485760;
template <int WORK, bool BURST>
static double rate(void* va, int reps = 8, int cap = 0) {
  const unsigned blocks = (unsigned)((n + 255) / 256);
  const size_t lds = cap > 0 ? (size_t)(160 * 1024) / cap - 1024 : 0;
  if (lds > 64 * 1024)
    CK(hipFuncSetAttribute((const void*)fill<WORK, BURST>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((fill<WORK, BURST>), dim3(blocks), dim3(256), lds, 0, (uint32_t*)va, n, n);
  CK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((fill<WORK, BURST>), dim3(blocks), dim3(256), lds, 0, (uint32_t*)va, n, n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)n * 104 * 4 / (ms / reps * 1e-3) / 1e12;
}

template <int SLEEP>
static double rate_sleep(void* va, int cap, int reps = 8) {
  const unsigned blocks = (unsigned)((n + 255) / 256);
  const size_t lds = cap > 0 ? (size_t)(160 * 1024) / cap - 1024 : 0;
  if (lds > 64 * 1024)
    CK(hipFuncSetAttribute((const void*)fill_sleep<SLEEP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int k = 0; k < 3; ++k) hipLaunchKernelGGL((fill_sleep<SLEEP>), dim3(blocks), dim3(256), lds, 0, (uint32_t*)va, n, n);
  CK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL((fill_sleep<SLEEP>), dim3(blocks), dim3(256), lds, 0, (uint32_t*)va, n, n);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)n * 104 * 4 / (ms / reps * 1e-3) / 1e12;
}

template <int WORK>
static void line(const char* tag, void* va) {
  printf("%s work %2d  even %.2f  burst %.2f TB/s\n", tag, WORK, rate<WORK, false>(va), rate<WORK, true>(va));
}

int main() {
  const size_t GiB = 1ull << 30;
  CK(hipSetDevice(0));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t bytes = (size_t)n * 4 * 104;
  char* plain; CK(hipMalloc((void**)&plain, bytes));
  char* best = nullptr; double best_r = 0; int best_g = 0;
  char* arenas[3] = {nullptr, nullptr, nullptr};
  for (int a = 0; a < 3 && best_r < 6.6; ++a) {      // up to three arenas, like the product
    CK(hipMalloc((void**)&arenas[a], 40 * GiB));
    for (int k = 0; k < 20; ++k) rate<0, false>(arenas[a], 2);
    for (int g = 0; g <= 35; ++g) {
      const double r = rate<0, false>(arenas[a] + g * GiB, 3);
      if (r > best_r) { best_r = r; best = arenas[a] + g * GiB; best_g = g + 100 * a; }
    }
  }
  printf("best window: arena %d +%d GiB, %.2f TB/s; plain block %.2f TB/s\n", best_g / 100, best_g % 100, best_r,
         rate<0, false>(plain));
  for (int rep = 0; rep < 2; ++rep) {
    printf("s_sleep after every store  window: 0 %.2f  1 %.2f  2 %.2f  4 %.2f  8 %.2f  16 %.2f | plain: 0 %.2f  2 %.2f  8 %.2f\n",
           rate_sleep<0>(best, 0), rate_sleep<1>(best, 0), rate_sleep<2>(best, 0), rate_sleep<4>(best, 0),
           rate_sleep<8>(best, 0), rate_sleep<16>(best, 0), rate_sleep<0>(plain, 0), rate_sleep<2>(plain, 0),
           rate_sleep<8>(plain, 0));
    printf("work 4 per store (the fp32 record kernel's ratio), workgroups per CU capped  plain:");
    for (int cap : {0, 1, 2, 3, 4, 6}) printf("  %d %.2f", cap, rate<4, false>(plain, 8, cap));
    printf(" | window:");
    for (int cap : {0, 1, 2, 3, 4}) printf("  %d %.2f", cap, rate<4, false>(best, 8, cap));
    printf("\n");
    printf("plain fill, caps on the plain block:");
    for (int cap : {1, 2, 3}) printf("  %d %.2f", cap, rate_sleep<0>(plain, cap));
    printf("\n");
    printf("workgroups per CU capped   window:");
    for (int cap : {1, 2, 3, 4, 5, 6, 8}) printf("  %d %.2f", cap, rate_sleep<0>(best, cap));
    printf(" | plain:");
    for (int cap : {2, 4, 6, 8}) printf("  %d %.2f", cap, rate_sleep<0>(plain, cap));
    printf("\n");
    line<0>("window", best);  line<0>("plain ", plain);
    line<2>("window", best);  line<2>("plain ", plain);
    line<4>("window", best);  line<4>("plain ", plain);
    line<8>("window", best);  line<8>("plain ", plain);
    line<16>("window", best); line<16>("plain ", plain);
    line<32>("window", best); line<32>("plain ", plain);
  }
  return 0;
}
