// vmm_pairs.hip -- round 5, second map: WHICH two pieces of device memory make a fast record block?
// vmm_regions.hip (profiles/r05_vmm_regions.txt) found no fast pairing of 40 MiB pieces on a box
// whose 40 GiB hipMalloc arenas DO hold a 7 TB/s window at +30 GiB -- the place where the arena's
// 32 GiB buddy block meets its 8 GiB one.  So here the junction is rebuilt from pieces that can
// be released one by one:
//   1  a 40 GiB hipMalloc arena, 4.06 GiB windows in 1 GiB steps          (is there a fast window
//      in THIS process, and where)
//   2  the same arena from two VMM handles of 32 GiB and 8 GiB mapped back to back
//   3  NH handles of half a block (2.03 GiB) each; the block [H_i | H_j] for every ordered pair
//      i < j (and a sample of j < i): the whole matrix
//   4  the best pair kept, everything else released: does it stay fast?
// build: hipcc --offload-arch=gfx950 -O3 -o vmm_pairs vmm_pairs.hip
#include <hip/hip_runtime.h>
#include <chrono>
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

static double now() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

static hipEvent_t e0, e1;
static double time_fill(void* va, int64_t n, int planes, int reps = 3, int warm = 1) {
  const unsigned blocks = (unsigned)((n + 255) / 256);
  for (int k = 0; k < warm; ++k) hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, (uint32_t*)va, n, n, planes, 1u);
  CK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, (uint32_t*)va, n, n, planes, 1u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)n * planes * 4 / (ms / reps * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const size_t GiB = 1ull << 30, MiB = 1ull << 20;
  CK(hipSetDevice(0));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  size_t free_b, total_b; CK(hipMemGetInfo(&free_b, &total_b));
  printf("device memory: %.1f GiB free of %.1f GiB\n", (double)free_b / GiB, (double)total_b / GiB);
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
  const size_t HALF = bytes / 2;           // 2080 MiB

  // 1: hipMalloc arena
  {
    char* arena; CK(hipMalloc((void**)&arena, 40 * GiB));
    for (int k = 0; k < 40; ++k) time_fill(arena, n, planes, 2, 0);   // clocks up
    printf("-- 1: hipMalloc 40 GiB arena, window start in GiB: TB/s\n");
    for (int g = 0; g * GiB + bytes <= 40 * GiB; ++g)
      printf("1 +%2d %.2f\n", g, time_fill(arena + g * GiB, n, planes, 4));
    CK(hipFree(arena));
  }
  // 2: the same from a 32 GiB and an 8 GiB handle
  {
    hipMemGenericAllocationHandle_t h32, h8;
    double t0 = now();
    CK(hipMemCreate(&h32, 32 * GiB, &prop, 0));
    CK(hipMemCreate(&h8, 8 * GiB, &prop, 0));
    double t1 = now();
    char* va; CK(hipMemAddressReserve((void**)&va, 40 * GiB, 2 * MiB, nullptr, 0));
    CK(hipMemMap(va, 32 * GiB, 0, h32, 0));
    CK(hipMemMap(va + 32 * GiB, 8 * GiB, 0, h8, 0));
    CK(hipMemSetAccess(va, 40 * GiB, &acc, 1));
    double t2 = now();
    printf("-- 2: VMM handles of 32 + 8 GiB back to back (create %.1f ms, map %.1f ms)\n", (t1 - t0) * 1e3, (t2 - t1) * 1e3);
    for (int g = 0; g * GiB + bytes <= 40 * GiB; ++g)
      printf("2 +%2d %.2f\n", g, time_fill(va + g * GiB, n, planes, 4));
    CK(hipMemUnmap(va, 40 * GiB)); CK(hipMemAddressFree(va, 40 * GiB));
    CK(hipMemRelease(h32)); CK(hipMemRelease(h8));
  }
  // 3: half-block handles, the pair matrix
  CK(hipMemGetInfo(&free_b, &total_b));
  size_t want = (argc > 1 ? (size_t)atoll(argv[1]) : 230) * GiB;
  if (want + 12 * GiB > free_b) want = free_b - 12 * GiB;
  const int NH = (int)(want / HALF);
  std::vector<hipMemGenericAllocationHandle_t> h(NH);
  double t0 = now();
  int made = 0;
  for (; made < NH; ++made) if (hipMemCreate(&h[made], HALF, &prop, 0) != hipSuccess) break;
  double t1 = now();
  printf("-- 3: %d handles of %zu MiB (%.1f GiB) created in %.3f s\n", made, HALF / MiB, (double)made * HALF / GiB, t1 - t0);
  char* va; CK(hipMemAddressReserve((void**)&va, bytes, 2 * MiB, nullptr, 0));
  double t_map = 0; int n_map = 0;
  auto probe = [&](int i, int j, int reps) {
    double a = now();
    CK(hipMemMap(va, HALF, 0, h[i], 0));
    CK(hipMemMap(va + HALF, HALF, 0, h[j], 0));
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    t_map += now() - a; ++n_map;
    double r = time_fill(va, n, planes, reps);
    CK(hipMemUnmap(va, bytes));
    return r;
  };
  printf("   rows i, columns j = i+1 .. %d: [H_i | H_j] in 0.1 TB/s\n", made - 1);
  double best = 0; int bi = 0, bj = 1;
  std::vector<int> fast_count(made, 0);
  for (int i = 0; i < made; ++i) {
    printf("3 %3d:", i);
    for (int j = i + 1; j < made; ++j) {
      const double r = probe(i, j, 3);
      printf(" %2d", (int)(r * 10 + 0.5));
      if (r > best) { best = r; bi = i; bj = j; }
      if (r > 6.5) { ++fast_count[i]; ++fast_count[j]; }
    }
    printf("\n");
  }
  printf("   pairs above 6.5 TB/s per handle:");
  for (int i = 0; i < made; ++i) printf(" %d", fast_count[i]);
  printf("\n   a sample of reversed pairs [H_j | H_i], j > i:\n");
  for (int i = 0; i < made; i += 9) {
    printf("3r %3d:", i);
    for (int j = i + 1; j < made; j += 5) printf(" %2d", (int)(probe(j, i, 3) * 10 + 0.5));
    printf("\n");
  }
  printf("   map + access of two halves: %.2f ms (mean of %d)\n", t_map / n_map * 1e3, n_map);
  // 4: keep the best pair only
  printf("-- 4: best pair (%d, %d) %.3f TB/s\n", bi, bj, best);
  CK(hipMemMap(va, HALF, 0, h[bi], 0));
  CK(hipMemMap(va + HALF, HALF, 0, h[bj], 0));
  CK(hipMemSetAccess(va, bytes, &acc, 1));
  printf("4 mapped again                                     %.3f TB/s\n", time_fill(va, n, planes, 6, 2));
  t0 = now();
  for (int k = 0; k < made; ++k) if (k != bi && k != bj) CK(hipMemRelease(h[k]));
  t1 = now();
  CK(hipMemGetInfo(&free_b, &total_b));
  printf("4 released %d handles in %.3f s; %.1f GiB free\n", made - 2, t1 - t0, (double)free_b / GiB);
  for (int k = 0; k < 3; ++k) printf("4 the kept pair, everything else released          %.3f TB/s\n", time_fill(va, n, planes, 6, 2));
  void* big = nullptr;
  hipError_t e = hipMalloc(&big, 200 * GiB);
  printf("4 a 200 GiB hipMalloc afterwards: %s\n", hipGetErrorString(e));
  if (e == hipSuccess) {
    for (int g : {0, 30, 62, 94, 126, 158, 190})
      printf("4   window +%3d GiB of it                          %.3f TB/s\n", g, time_fill((char*)big + g * GiB, n, planes, 4));
  }
  printf("4 the kept pair, next to it                         %.3f TB/s\n", time_fill(va, n, planes, 6, 2));
  return 0;
}
