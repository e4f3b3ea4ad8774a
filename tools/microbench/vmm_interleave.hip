// vmm_interleave.hip -- where does the record-all store pattern write fast, and can a block be
// BUILT to write fast?  One physical allocation (hipMemCreate), mapped into virtual address
// ranges in different ways with the HIP virtual-memory API; the arithmetic-free store pattern of
// the trace kernels (planes x n elements, one non-temporal dword per lane and plane) is timed on
// each mapping:
//   linear  : the block is H[off, off + bytes)                       (what hipMalloc gives)
//   inter P : pages of P bytes taken alternately from H[lo...] and H[hi...]
// build: hipcc --offload-arch=gfx950 -O3 -o vmm_interleave vmm_interleave.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d: %s\n", hipGetErrorString(e_), __FILE__, __LINE__, #x); exit(2); } } while (0)

__global__ __launch_bounds__(256) void fill(uint32_t* __restrict__ dst, int64_t n, int64_t stride,
                                            int planes, uint32_t v) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t* p = dst + j;
  for (int k = 0; k < planes; ++k, p += stride) __builtin_nontemporal_store(v, p);
}

static double time_fill(void* va, int64_t n, int planes, int reps = 6) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned blocks = (unsigned)((n + 255) / 256);
  for (int k = 0; k < 2; ++k) hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, (uint32_t*)va, n, n, planes, 1u);
  CK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, (uint32_t*)va, n, n, planes, 1u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
  return (double)n * planes * 4 / (ms / reps * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const size_t GiB = 1ull << 30, MiB = 1ull << 20;
  const size_t total = (argc > 1 ? atoll(argv[1]) : 150) * GiB;
  CK(hipSetDevice(0));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const int64_t n = 10485760;
  const int planes = 104;
  const size_t P = (size_t)n * 4;
  const size_t bytes = P * planes;
  void* blk; CK(hipMalloc(&blk, bytes));
  for (int k = 0; k < 60; ++k) time_fill(blk, n, planes, 2);
  printf("hipMalloc block, alone                         %.3f TB/s\n", time_fill(blk, n, planes));
  const size_t M = total / P;
  std::vector<hipMemGenericAllocationHandle_t> h(M);
  size_t made = 0;
  auto map_from = [&](size_t k0) {
    void* va = nullptr;
    CK(hipMemAddressReserve(&va, bytes, 2 * MiB, nullptr, 0));
    for (int p = 0; p < planes; ++p) CK(hipMemMap((char*)va + p * P, P, 0, h[k0 + p], 0));
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    return va;
  };
  // stage 1: only the block's own handles exist
  for (; made < (size_t)planes; ++made) CK(hipMemCreate(&h[made], P, &prop, 0));
  void* va0 = map_from(0);
  printf("VMM block (handles 0..103), nothing else        %.3f TB/s\n", time_fill(va0, n, planes));
  // stage 2: create the rest, in steps
  for (size_t target : {M / 8, M / 4, M / 2, 3 * M / 4, M}) {
    for (; made < target; ++made) if (hipMemCreate(&h[made], P, &prop, 0) != hipSuccess) break;
    printf("  same block, %5.1f GiB of handles exist         %.3f TB/s\n", (double)made * P / GiB,
           time_fill(va0, n, planes));
  }
  printf("hipMalloc block again                          %.3f TB/s\n", time_fill(blk, n, planes));
  for (size_t k0 : {(size_t)500, (size_t)1500, made - 200}) {
    if (k0 + planes > made) continue;
    void* va = map_from(k0);
    printf("VMM block from handles %zu..                    %.3f TB/s\n", k0, time_fill(va, n, planes));
    CK(hipMemUnmap(va, bytes)); CK(hipMemAddressFree(va, bytes));
  }
  // stage 3: release everything but the block's own handles
  for (size_t k = planes; k < made; ++k) CK(hipMemRelease(h[k]));
  printf("VMM block (handles 0..103), the rest released   %.3f TB/s\n", time_fill(va0, n, planes));
  printf("hipMalloc block, the rest released              %.3f TB/s\n", time_fill(blk, n, planes));
  return 0;
}
