// vmm_lottery.hip -- round 5: HOW OFTEN does a junction between two pieces make a fast block,
// by the sizes of the pieces?  (vmm_junctions.hip: one sample per configuration.)  Arenas are
// built from VMM handles of the listed sizes mapped back to back; for every junction the 4160 MiB
// block is timed centred on it (and 512 MiB to either side); > 6.6 TB/s counts as a hit.  The
// configurations go round robin, REPS times, in one process, each arena released before the
// next is made -- the driver reuses freed memory in an order that depends on history, which
// is the point.
// build: hipcc --offload-arch=gfx950 -O3 -o vmm_lottery vmm_lottery.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d: %s\n", hipGetErrorString(e_), __FILE__, __LINE__, #x); exit(2); } } while (0)

__global__ __launch_bounds__(256) void fill(uint32_t* __restrict__ dst, int64_t n, int64_t stride, int planes) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t* p = dst + j;
  for (int k = 0; k < planes; ++k, p += stride) __builtin_nontemporal_store(1u, p);
}

static hipEvent_t e0, e1;
static const int64_t n = 10485760;
static const int planes = 104;
static double rate(void* va, int reps = 3) {
  const unsigned blocks = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, (uint32_t*)va, n, n, planes);
  CK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, (uint32_t*)va, n, n, planes);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)n * planes * 4 / (ms / reps * 1e-3) / 1e12;
}

int main(int argc, char** argv) {
  const size_t MiB = 1ull << 20;
  const int REPS = argc > 1 ? atoi(argv[1]) : 6;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  CK(hipSetDevice(0));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t bytes = (size_t)n * 4 * planes;   // 4160 MiB
  { void* w; CK(hipMalloc(&w, bytes)); for (int k = 0; k < 40; ++k) rate(w, 2); printf("plain block %.2f TB/s\n", rate(w)); CK(hipFree(w)); }
  const std::vector<std::vector<size_t>> configs = {   // sizes in MiB
      {32768, 8192}, {8192, 32768}, {16384, 16384}, {16384, 8192}, {8192, 16384}, {8192, 8192},
      {4096, 4096}, {32768, 4096}, {4096, 32768}, {32768, 2560}, {2560, 32768}, {16384, 16384, 8192},
      {8192, 8192, 8192, 8192, 8192}, {16384, 4096}, {4096, 16384}, {65536, 8192}, {8192, 65536}};
    std::vector<int> hits(configs.size(), 0), tries(configs.size(), 0);
  std::vector<std::vector<double>> seen(configs.size());
  for (int rep = 0; rep < REPS; ++rep) {
    for (size_t c = 0; c < configs.size(); ++c) {
      const auto& sz = configs[c];
      std::vector<hipMemGenericAllocationHandle_t> h(sz.size());
      size_t total = 0;
      for (size_t i = 0; i < sz.size(); ++i) { CK(hipMemCreate(&h[i], sz[i] * MiB, &prop, 0)); total += sz[i] * MiB; }
      // (a reservation of exactly the arena's size, never given back: re-reserving after
      // hipMemAddressFree, and mapping into part of a larger reservation, both ended in GPU
      // memory access faults in this program; address space is not scarce)
      char* va; CK(hipMemAddressReserve((void**)&va, total, 2 * MiB, nullptr, 0));
      size_t off = 0;
      for (size_t i = 0; i < sz.size(); ++i) { CK(hipMemMap(va + off, sz[i] * MiB, 0, h[i], 0)); off += sz[i] * MiB; }
      CK(hipMemSetAccess(va, total, &acc, 1));
      size_t J = 0;
      for (size_t i = 0; i + 1 < sz.size(); ++i) {
        J += sz[i] * MiB;
        double best = 0;
        for (long d : {-512L, 0L, 512L}) {
          long start = (long)J - (long)(bytes / 2) + d * (long)MiB;
          start = start / (long)(2 * MiB) * (long)(2 * MiB);
          if (start < 0 || (size_t)start + bytes > total) continue;
          const double r = rate(va + start);
          if (r > best) best = r;
        }
        ++tries[c]; if (best > 6.6) ++hits[c];
        seen[c].push_back(best);
      }
      CK(hipDeviceSynchronize());
      CK(hipMemUnmap(va, total));
      for (auto x : h) CK(hipMemRelease(x));
      printf("rep %d config %zu done (%zu junction(s), last best %.2f)\n", rep, c, sz.size() - 1, seen[c].back());
    }
  }
  for (size_t c = 0; c < configs.size(); ++c) {
    printf("[");
    for (size_t i = 0; i < configs[c].size(); ++i) printf("%s%zu", i ? " | " : "", configs[c][i] / 1024);
    printf(" GiB]  hits %d / %d  :", hits[c], tries[c]);
    for (double v : seen[c]) printf(" %.1f", v);
    printf("\n");
  }
  return 0;
}
