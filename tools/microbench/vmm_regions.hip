// vmm_regions.hip -- round 5: the MAP behind "where a record block writes fast".
// Round 4 found (vmm_interleave.hip) that a 104-plane block whose planes come alternately from two
// places of device memory >= 36 GiB apart (in handle-creation order) takes the record-all store
// pattern at 6.9 TB/s, consecutive handles 5.8-6.3.  This program draws the whole map so that an
// allocator can BUILD such a block instead of searching a 40 GiB arena for one:
//   A  planes alternately from chunk 0 and chunk k, all k        (is it distance, or a region id?)
//   B  the same with the middle chunk as the reference
//   C  planes round-robin over R places spaced D GiB apart        (do more regions help further?)
//   D  keep only the chosen handles, release the rest, time again (does a built block stay fast?)
//   E  what creating / mapping / releasing costs
// build: hipcc --offload-arch=gfx950 -O3 -o vmm_regions vmm_regions.hip
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

static double time_fill(void* va, int64_t n, int planes, int reps = 5) {
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
  CK(hipSetDevice(0));
  size_t free_b, total_b; CK(hipMemGetInfo(&free_b, &total_b));
  size_t want = (argc > 1 ? (size_t)atoll(argv[1]) : 240) * GiB;
  if (want + 12 * GiB > free_b) want = free_b - 12 * GiB;
  printf("device memory: %.1f GiB free of %.1f GiB; creating %.1f GiB of handles\n",
         (double)free_b / GiB, (double)total_b / GiB, (double)want / GiB);
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  const int64_t n = 10485760;
  const int planes = 104;
  const size_t P = (size_t)n * 4;          // one plane = one handle = 40 MiB
  const size_t bytes = P * planes;
  const int CH = planes / 2;               // a chunk: 52 handles = 2.03 GiB
  printf("granularity %zu; plane %zu MiB; block %.2f GiB; chunk %.2f GiB\n", gran, P / MiB,
         (double)bytes / GiB, (double)CH * P / GiB);

  // warm the clocks on an ordinary block
  void* blk; CK(hipMalloc(&blk, bytes));
  for (int k = 0; k < 40; ++k) time_fill(blk, n, planes, 2);
  printf("hipMalloc block                                  %.3f TB/s\n", time_fill(blk, n, planes));

  const size_t M = want / P;
  std::vector<hipMemGenericAllocationHandle_t> h(M);
  size_t made = 0;
  double t0 = now();
  for (; made < M; ++made) if (hipMemCreate(&h[made], P, &prop, 0) != hipSuccess) break;
  double t1 = now();
  printf("E created %zu handles (%.1f GiB) in %.3f s = %.1f us per handle, %.2f ms per GiB\n", made,
         (double)made * P / GiB, t1 - t0, (t1 - t0) / made * 1e6, (t1 - t0) / ((double)made * P / GiB) * 1e3);
  const int NC = (int)(made / CH);

  void* va = nullptr;
  CK(hipMemAddressReserve(&va, bytes, 2 * MiB, nullptr, 0));
  double t_map = 0, t_unmap = 0; int n_map = 0;
  auto probe = [&](const std::vector<size_t>& idx) {
    double a = now();
    for (int p = 0; p < planes; ++p) CK(hipMemMap((char*)va + (size_t)p * P, P, 0, h[idx[p]], 0));
    CK(hipMemSetAccess(va, bytes, &acc, 1));
    double b = now();
    double r = time_fill(va, n, planes);
    double c = now();
    CK(hipMemUnmap(va, bytes));
    double d = now();
    t_map += b - a; t_unmap += d - c; ++n_map;
    return r;
  };
  auto pair_idx = [&](int ca, int cb) {
    std::vector<size_t> idx(planes);
    for (int p = 0; p < planes; ++p) idx[p] = (size_t)((p & 1) ? cb : ca) * CH + p / 2;
    return idx;
  };
  auto linear_idx = [&](int c) {
    std::vector<size_t> idx(planes);
    for (int p = 0; p < planes; ++p) idx[p] = (size_t)c * CH + p;
    return idx;
  };

  printf("-- L: consecutive handles (chunks c, c+1)\n");
  for (int c = 0; c + 1 < NC; c += 8)
    printf("L c=%3d (+%6.1f GiB)  %.3f TB/s\n", c, (double)c * CH * P / GiB, probe(linear_idx(c)));

  for (int ref : {0, NC / 2}) {
    printf("-- %c: planes alternately from chunk %d (+%.1f GiB) and chunk k\n", ref ? 'B' : 'A', ref,
           (double)ref * CH * P / GiB);
    for (int k = 0; k < NC; ++k) {
      if (k == ref) continue;
      printf("%c k=%3d (+%6.1f GiB, distance %+7.1f GiB)  %.3f TB/s\n", ref ? 'B' : 'A', k,
             (double)k * CH * P / GiB, (double)(k - ref) * CH * P / GiB, probe(pair_idx(ref, k)));
    }
  }

  printf("-- C: planes round-robin over R places, D GiB apart (from +0)\n");
  for (int R : {2, 3, 4, 6, 8}) for (double D : {12.0, 18.0, 24.0, 32.0, 36.0, 40.0}) {
    const int per = (planes + R - 1) / R;           // handles taken from each place
    const int dc = (int)(D * GiB / (CH * P) + 0.5);   // spacing in chunks
    if ((size_t)(R - 1) * dc * CH + per > made) continue;
    std::vector<size_t> idx(planes);
    for (int p = 0; p < planes; ++p) idx[p] = (size_t)(p % R) * dc * CH + p / R;
    printf("C R=%d D=%4.0f GiB  %.3f TB/s\n", R, D, probe(idx));
  }
  printf("E map+access of %d planes: %.2f ms, unmap: %.2f ms (mean of %d)\n", planes, t_map / n_map * 1e3,
         t_unmap / n_map * 1e3, n_map);

  // D: build the block from the best pair of A (first k whose rate is within 1 % of the best), keep
  // only its handles
  int best_k = 1; double best = 0;
  {
    std::vector<double> r(NC, 0.0);
    for (int k = 1; k < NC; k += 2) { r[k] = probe(pair_idx(0, k)); if (r[k] > best) best = r[k]; }
    for (int k = 1; k < NC; k += 2) if (r[k] > 0.99 * best) { best_k = k; break; }
  }
  std::vector<size_t> keep = pair_idx(0, best_k);
  for (int p = 0; p < planes; ++p) CK(hipMemMap((char*)va + (size_t)p * P, P, 0, h[keep[p]], 0));
  CK(hipMemSetAccess(va, bytes, &acc, 1));
  printf("D built from chunk 0 and chunk %d (+%.1f GiB)          %.3f TB/s\n", best_k,
         (double)best_k * CH * P / GiB, time_fill(va, n, planes));
  std::vector<char> kept(made, 0);
  for (size_t i : keep) kept[i] = 1;
  t0 = now();
  for (size_t k = 0; k < made; ++k) if (!kept[k]) CK(hipMemRelease(h[k]));
  t1 = now();
  CK(hipMemGetInfo(&free_b, &total_b));
  printf("E released %zu handles in %.3f s; %.1f GiB free again\n", made - planes, t1 - t0, (double)free_b / GiB);
  for (int k = 0; k < 3; ++k) printf("D the built block, everything else released        %.3f TB/s\n", time_fill(va, n, planes));
  void* big = nullptr;
  hipError_t e = hipMalloc(&big, 100 * GiB);
  printf("D a 100 GiB hipMalloc afterwards: %s\n", hipGetErrorString(e));
  printf("D the built block, next to it                      %.3f TB/s\n", time_fill(va, n, planes));
  printf("hipMalloc block                                  %.3f TB/s\n", time_fill(blk, n, planes));
  return 0;
}
