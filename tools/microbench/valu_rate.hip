// valu_rate.hip -- round 6: what a vector instruction COSTS on gfx950, measured.
// The C5 kernel (polarised Zernike, fp32) on packed pairs of rays issues 19 % fewer vector
// instructions per ray than on one ray per lane (SQ_INSTS_VALU 797 -> 645) and takes the same
// engine cycles.  So: issue slots per instruction, per opcode.  One workgroup per CU-slot, W waves
// per SIMD, every wave runs ITER x 16 independent instructions of one kind (inline asm, so nothing
// is re-packed or folded) and the kernel is timed with s_memtime; cycles per wave-instruction and
// SIMD = elapsed / (ITER * 16 * waves per SIMD).
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  printf("HIP error %s at %s:%d: %s\n", hipGetErrorString(e_), __FILE__, __LINE__, #x); exit(2); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int ITER = 8192;

enum Kind { FMA = 0, PK_FMA, PK_MUL, PK_ADD, MUL, RCP, SQRT, RSQ, CNDMASK, CMP, FMA64, PK_FMA_BCAST,
            MOV, PK_MOV, CNDMASK_S, CMP_CND, READLANE, WRITELANE, ADD64, MUL64, FMA_SGPR, NKIND };
static const char* kNames[NKIND] = {"v_fma_f32", "v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32",
                                    "v_mul_f32", "v_rcp_f32", "v_sqrt_f32", "v_rsq_f32",
                                    "v_cndmask_b32", "v_cmp_gt_f32", "v_fma_f64",
                                    "v_pk_fma_f32 (op_sel broadcast)", "v_mov_b32", "v_pk_mov_b32",
                                    "v_cndmask_b32 (sgpr pair mask)", "v_cmp_gt_f32 + v_cndmask_b32 (pair)",
                                    "v_readlane_b32", "v_writelane_b32", "v_add_f64", "v_mul_f64",
                                    "v_fma_f32 (sgpr operand)"};

template <int K>
__global__ __launch_bounds__(256) void rate(uint64_t* out, float seed) {
  float a[16];
  f2 p[16];
  double d[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = seed + i + threadIdx.x * 1e-3f;
    p[i] = f2{a[i], a[i] + 0.5f};
    d[i] = a[i];
  }
  const float b = 1.0000001f, c = 1e-7f;
  const f2 pb = {b, b}, pc = {c, c};
  const double db = b, dc = c;
  const uint64_t smask = __builtin_amdgcn_read_exec() ^ 0x5555555555555555ull;
  const int sseed = __builtin_amdgcn_readfirstlane((int)seed);
  const float sb = __builtin_amdgcn_readfirstlane(__float_as_int(b)) == 0 ? 2.0f : __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(b)));
  int sl[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  __syncthreads();
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < ITER; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if constexpr (K == FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if constexpr (K == MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (K == PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(pb), "v"(pc));
      if constexpr (K == PK_FMA_BCAST)
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2 op_sel_hi:[1,0,0]" : "+v"(p[i]) : "v"(pb), "v"(pc));
      if constexpr (K == PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pb));
      if constexpr (K == PK_ADD) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(pc));
      if constexpr (K == RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
      if constexpr (K == SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
      if constexpr (K == RSQ) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]));
      if constexpr (K == CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
      if constexpr (K == CMP) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(b) : "vcc");
      if constexpr (K == FMA64) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i]) : "v"(db), "v"(dc));
      if constexpr (K == MOV) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(b));
      if constexpr (K == PK_MOV) asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(p[i]) : "v"(pb));
      if constexpr (K == CNDMASK_S)
        asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "s"(smask));
      if constexpr (K == CMP_CND)
        asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc"
                     : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
      if constexpr (K == READLANE) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sl[i]) : "v"(a[i]));
      if constexpr (K == ADD64) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(dc));
      if constexpr (K == MUL64) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(db));
      if constexpr (K == FMA_SGPR) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sb), "v"(c));
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i] + (float)sl[i];
  if (s == 123.456f) out[1] = 1;  // (keeps the chains alive)
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}

template <int K>
static double run(uint64_t* dev, int waves_per_simd, double ref_ms) {
  // workgroups of 4 waves (one per SIMD), `waves_per_simd` of them per CU
  const int blocks = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  rate<K><<<blocks, 256>>>(dev, 1.0f);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipEventRecord(e0));
    rate<K><<<blocks, 256>>>(dev, 1.0f);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  uint64_t host[2];
  CK(hipMemcpy(host, dev, sizeof(host), hipMemcpyDeviceToHost));
  const double instr = (double)ITER * 16 * waves_per_simd;  // wave-instructions per SIMD
  printf("%-34s waves/SIMD %d  launch %.4f ms  ns per wave-instruction and SIMD %6.3f  "
         "(x %.2f of v_fma_f32)   one wave: %.2f s_memtime ticks per instruction\n",
         kNames[K], waves_per_simd, best, best * 1e6 / instr,
         ref_ms > 0 ? best / ref_ms : 1.0, (double)host[0] / ((double)ITER * 16));
  return best;
}

int main() {
  uint64_t* dev;
  CK(hipMalloc(&dev, 16));
  CK(hipMemset(dev, 0, 16));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("%s  CUs %d  clock %d kHz  (s_memtime runs at a constant 100 MHz on this part if the "
         "tick column looks 20x small)\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  for (int w : {1, 2, 8}) {
    const double ref = run<FMA>(dev, w, 0.0);
    run<MUL>(dev, w, ref); run<PK_FMA>(dev, w, ref); run<PK_FMA_BCAST>(dev, w, ref);
    run<PK_MUL>(dev, w, ref); run<PK_ADD>(dev, w, ref); run<MOV>(dev, w, ref);
    run<PK_MOV>(dev, w, ref); run<CNDMASK>(dev, w, ref); run<CMP>(dev, w, ref);
    run<RCP>(dev, w, ref); run<SQRT>(dev, w, ref); run<RSQ>(dev, w, ref); run<FMA64>(dev, w, ref);
    run<ADD64>(dev, w, ref); run<MUL64>(dev, w, ref); run<CNDMASK_S>(dev, w, ref);
    run<CMP_CND>(dev, w, ref); run<READLANE>(dev, w, ref);
    run<FMA_SGPR>(dev, w, ref);
  }
  return 0;
}
