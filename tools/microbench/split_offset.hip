// split_offset.hip -- round 5: is the "junction" effect just an OFFSET between two groups of planes?
// A record block written across the place where two big pieces of an arena meet takes the
// 104-plane store pattern at 7.1 TB/s, a contiguous block at 5.85 (profiles/r05_vmm_junctions.txt:
// some junctions do it, some do not).  Across a junction the planes above it are displaced, in
// PHYSICAL address, by some large amount relative to the planes below.  Here that displacement is
// made on purpose inside ONE contiguous allocation: planes [0, split) at base + p * stride, planes
// [split, 104) at base + delta + p * stride, delta scanned from 256 B to 24 GiB.  If some delta
// is fast wherever the block lies, a record block can be LAID OUT fast (per-row base offsets in
// the kernel's argument block) instead of being searched for.
// build: hipcc --offload-arch=gfx950 -O3 -o split_offset split_offset.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d: %s\n", hipGetErrorString(e_), __FILE__, __LINE__, #x); exit(2); } } while (0)

// group g of a plane = (p / run) % groups; its planes are displaced by g * delta.  The plane
// offsets come ready-made in the argument block (scalar loads): the kernel is as arithmetic-free
// as the contiguous fill.
struct Offs { int64_t o[104]; };
__global__ __launch_bounds__(256) void fill(uint32_t* __restrict__ dst, int64_t n, Offs offs) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= n) return;
  uint32_t* p = dst + j;
#pragma unroll 8
  for (int k = 0; k < 104; ++k) __builtin_nontemporal_store(1u, p + offs.o[k]);
}

static hipEvent_t e0, e1;
static const int64_t n = 10485760;
static const int planes = 104;
static double rate(char* base, int run, int groups, int64_t delta, int reps = 4) {
  const unsigned blocks = (unsigned)((n + 255) / 256);
  Offs offs;
  for (int k = 0; k < planes; ++k) offs.o[k] = (int64_t)k * n + (int64_t)((k / run) % groups) * (delta / 4);
  hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, (uint32_t*)base, n, offs);
  CK(hipEventRecord(e0));
  for (int k = 0; k < reps; ++k)
    hipLaunchKernelGGL(fill, dim3(blocks), dim3(256), 0, 0, (uint32_t*)base, n, offs);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return (double)n * planes * 4 / (ms / reps * 1e-3) / 1e12;
}

int main() {
  const int64_t KiB = 1024, MiB = 1 << 20, GiB = 1ll << 30;
  CK(hipSetDevice(0));
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  char* arena; CK(hipMalloc((void**)&arena, 64 * GiB));
  for (int k = 0; k < 60; ++k) rate(arena, 104, 1, 0, 2);
  printf("contiguous block at +0: %.2f, at +8 GiB: %.2f, at +20 GiB: %.2f TB/s\n", rate(arena, 104, 1, 0),
         rate(arena + 8 * GiB, 104, 1, 0), rate(arena + 20 * GiB, 104, 1, 0));
  printf("plain windows of the 64 GiB arena in 2 GiB steps:");
  for (int g = 0; g <= 58; g += 2) printf(" +%d:%.2f", g, rate(arena + g * GiB, 104, 1, 0));
  printf("\n");
  std::vector<int64_t> deltas;
  for (int64_t d = 256; d <= 16 * GiB; d *= 2) deltas.push_back(d);
  for (int64_t d : {3 * KiB, 12 * KiB, 3 * MiB, 5 * MiB, 10 * MiB, 20 * MiB, 40 * MiB, 60 * MiB, 100 * MiB,
                    1 * GiB + 20 * MiB, 2 * GiB + 2 * MiB, 3 * GiB, 5 * GiB, 6 * GiB, 10 * GiB, 12 * GiB, 20 * GiB, 24 * GiB,
                    4 * GiB + 4 * KiB, 4 * GiB + 256, 8 * GiB + 1 * MiB, 30 * GiB, 34 * GiB, 40 * GiB, 48 * GiB})
    deltas.push_back(d);
  for (int64_t base_gib : {0, 6}) {
    char* base = arena + base_gib * GiB;
    printf("-- block at +%lld GiB; two halves (planes 0-51 | 52-103), the upper half displaced by delta\n", (long long)base_gib);
    for (int64_t d : deltas) {
      if (base_gib * GiB + d + (int64_t)planes * n * 4 > 64 * GiB) continue;
      printf("half  delta %14lld B (%9.3f MiB)  %.2f TB/s\n", (long long)d, (double)d / MiB, rate(base, 52, 2, d));
    }
  }
  printf("-- block at +0; odd planes displaced by delta\n");
  for (int64_t d : deltas) {
    if (d + (int64_t)planes * n * 4 > 64 * GiB) continue;
    printf("odd   delta %14lld B (%9.3f MiB)  %.2f TB/s\n", (long long)d, (double)d / MiB, rate(arena, 1, 2, d));
  }
  printf("-- block at +0; four quarters (26 planes each), quarter g displaced by g * delta\n");
  for (int64_t d : deltas) {
    if (3 * d + (int64_t)planes * n * 4 > 64 * GiB) continue;
    printf("quart delta %14lld B (%9.3f MiB)  %.2f TB/s\n", (long long)d, (double)d / MiB, rate(arena, 26, 4, d));
  }
  printf("-- block at +0; rows of 8 planes, row g displaced by g * delta (13 groups)\n");
  for (int64_t d : deltas) {
    if (12 * d + (int64_t)planes * n * 4 > 64 * GiB) continue;
    printf("rows  delta %14lld B (%9.3f MiB)  %.2f TB/s\n", (long long)d, (double)d / MiB, rate(arena, 8, 13, d));
  }
  return 0;
}
