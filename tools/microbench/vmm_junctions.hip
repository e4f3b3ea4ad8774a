// vmm_junctions.hip -- round 5, third map: WHAT about the place where a 40 GiB arena's 32 GiB
// piece meets its 8 GiB piece makes a record block written across it fast (7.1 against 5.85 TB/s,
// profiles/r05_vmm_pairs.txt stages 1 and 2)?  Arenas are built from VMM handles of chosen
// (power-of-two) sizes mapped back to back, and the 104-plane store pattern is timed on windows
// around every junction -- looking for the SMALLEST arena that still has the effect, so that a
// record block can be built on purpose and nothing but the block stays allocated.
// build: hipcc --offload-arch=gfx950 -O3 -o vmm_junctions vmm_junctions.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
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

static hipEvent_t e0, e1;
static double time_fill(void* va, int64_t n, int planes, int reps = 4, int warm = 1) {
  const unsigned blocks = (unsigned)((n + 255) / 256);
  for (int k = 0; k < warm; ++k) hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, (uint32_t*)va, n, n, planes, 1u);
  CK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k) hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, (uint32_t*)va, n, n, planes, 1u);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)n * planes * 4 / (ms / reps * 1e-3) / 1e12;
}

static const size_t GiB = 1ull << 30, MiB = 1ull << 20;
static hipMemAllocationProp prop = {};
static hipMemAccessDesc acc = {};

struct Arena {
  std::vector<hipMemGenericAllocationHandle_t> h;
  std::vector<size_t> sz;
  char* va = nullptr;
  size_t total = 0;
  void build(const std::vector<size_t>& sizes_mib, bool reverse_map = false) {
    for (size_t s : sizes_mib) {
      hipMemGenericAllocationHandle_t x;
      CK(hipMemCreate(&x, s * MiB, &prop, 0));
      h.push_back(x); sz.push_back(s * MiB); total += s * MiB;
    }
    CK(hipMemAddressReserve((void**)&va, total, 2 * MiB, nullptr, 0));
    size_t off = 0;
    const int nh = (int)h.size();
    for (int k = 0; k < nh; ++k) {
      const int i = reverse_map ? nh - 1 - k : k;
      CK(hipMemMap(va + off, sz[i], 0, h[i], 0));
      off += sz[i];
    }
    CK(hipMemSetAccess(va, total, &acc, 1));
  }
  void destroy() {
    CK(hipMemUnmap(va, total)); CK(hipMemAddressFree(va, total));
    for (auto x : h) CK(hipMemRelease(x));
    h.clear(); sz.clear(); total = 0; va = nullptr;
  }
};

// windows of `planes` x n dwords starting at the given MiB offsets
static void scan(const char* tag, Arena& a, int64_t n, int planes, const std::vector<size_t>& starts_mib) {
  const size_t bytes = (size_t)n * 4 * planes;
  printf("%s:", tag);
  for (size_t s : starts_mib) {
    if (s * MiB + bytes > a.total) continue;
    printf("  +%zu:%.2f", s, time_fill(a.va + s * MiB, n, planes));
  }
  printf("\n");
  fflush(stdout);
}

static std::vector<size_t> range(size_t lo, size_t hi, size_t step) {
  std::vector<size_t> r;
  for (size_t x = lo; x <= hi; x += step) r.push_back(x);
  return r;
}

int main() {
  CK(hipSetDevice(0));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const int64_t n = 10485760;   // 40 MiB planes: the 1e7-ray fp32 block, 4160 MiB
  const int planes = 104;
  {
    void* w; CK(hipMalloc(&w, (size_t)n * 4 * planes));
    for (int k = 0; k < 40; ++k) time_fill(w, n, planes, 2, 0);
    printf("plain hipMalloc block %.2f TB/s\n", time_fill(w, n, planes));
    CK(hipFree(w));
  }
  Arena a;
  printf("# window starts in MiB : TB/s.  The 4160 MiB block straddles a junction at J when start in (J - 4160, J)\n");
  a.build({32768, 8192});
  scan("a 32G+8G (control; junction 32768)", a, n, planes, range(27 * 1024, 33 * 1024, 512));
  // a smaller block (1e6 rays fp32: 4 MiB planes, 416 MiB) around the same junction
  scan("a 32G+8G, 416 MiB block (n = 1048576)", a, 1048576, planes,
       {1024, 20000, 32768 - 416, 32768 - 312, 32768 - 208, 32768 - 104, 32768 - 52, 32768, 33000});
  // and a bigger one (fp64 1e7: 80 MiB planes as 20971520 dwords, 8320 MiB)
  scan("a 32G+8G, 8320 MiB block (n = 20971520)", a, 20971520, planes, range(20 * 1024, 32 * 1024, 1024));
  a.destroy();
  a.build({8192, 8192});
  scan("b 8G+8G (junction 8192)", a, n, planes, range(0, 12 * 1024, 512));
  a.destroy();
  a.build({4096, 4096, 4096});
  scan("c 4G+4G+4G (junctions 4096, 8192)", a, n, planes, range(0, 8 * 1024, 512));
  a.destroy();
  a.build(std::vector<size_t>(12, 2048));
  scan("d 12 x 2G (junctions every 2048)", a, n, planes, range(0, 20 * 1024, 512));
  a.destroy();
  a.build(std::vector<size_t>(12, 2048), true);
  scan("e 12 x 2G mapped in reverse creation order", a, n, planes, range(0, 20 * 1024, 512));
  a.destroy();
  a.build({16384, 16384});
  scan("f 16G+16G (junction 16384)", a, n, planes, range(10 * 1024, 17 * 1024, 512));
  a.destroy();
  a.build({8192, 32768});
  scan("g 8G+32G (junction 8192)", a, n, planes, range(3 * 1024, 10 * 1024, 512));
  a.destroy();
  a.build({32768, 2048, 2048});
  scan("h 32G+2G+2G (junctions 32768, 34816)", a, n, planes, range(27 * 1024, 32 * 1024 + 512, 512));
  a.destroy();
  // i: does the junction have to be made of ONE allocation call each side?  32G as 2 x 16G
  a.build({16384, 16384, 8192});
  scan("i 16G+16G+8G (junctions 16384, 32768)", a, n, planes, range(11 * 1024, 33 * 1024, 1024));
  a.destroy();
  return 0;
}
