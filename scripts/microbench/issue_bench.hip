// issue_bench.hip — issue rate of the VALU instruction forms the sphere test is made of, on gfx950.
// Each kernel runs a block of 32 independent instructions of one form (8 accumulators, 4 rounds) in a loop.
#include <hip/hip_runtime.h>
#include <cstdio>

#define REP4(x) x x x x
#define BLOCK8(OP)                                                                                         \
  asm volatile(OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)                                       \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
               : "v"(x), "v"(y), "v"(z));

#define OP_FMAC(r) "v_fmac_f32_e32 " #r ", %8, %9\n"
#define OP_FMA3(r) "v_fma_f32 " #r ", %8, %9, " #r "\n"
#define OP_FMA3D(r) "v_fma_f32 " #r ", %8, %9, %10\n"
#define OP_FMANEG(r) "v_fma_f32 " #r ", %8, %9, -" #r "\n"
#define OP_SUB(r) "v_sub_f32_e32 " #r ", %8, " #r "\n"
#define OP_MUL(r) "v_mul_f32_e32 " #r ", %8, " #r "\n"
#define OP_MULNEG(r) "v_mul_f32_e64 " #r ", %8, -" #r "\n"
#define OP_MAX3(r) "v_max3_f32 " #r ", %8, %9, " #r "\n"
#define OP_MAX(r) "v_max_f32_e32 " #r ", %8, " #r "\n"
#define OP_PKFMA(r) "v_pk_fma_f32 " #r ", %8, %9, " #r "\n"

#define KERNEL(NAME, OP)                                                                                    \
  __global__ void __launch_bounds__(256) NAME(float *out, int iters) {                                       \
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;                         \
    float x = 1.0000001f, y = 0.9999999f, z = 0.5f;                                                          \
    for (int i = 0; i < iters; ++i) { REP4(BLOCK8(OP)) }                                                     \
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                            \
  }

KERNEL(k_fmac, OP_FMAC)
KERNEL(k_fma3, OP_FMA3)
KERNEL(k_fma3d, OP_FMA3D)
KERNEL(k_fmaneg, OP_FMANEG)
KERNEL(k_sub, OP_SUB)
KERNEL(k_mul, OP_MUL)
KERNEL(k_mulneg, OP_MULNEG)
KERNEL(k_max3, OP_MAX3)
KERNEL(k_max, OP_MAX)

__global__ void __launch_bounds__(256) k_pkfma(float *out, int iters) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f a0 = {1, 2}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, x = {1.0000001f, 1.f}, y = {0.9999999f, 1.f}, z = x;
  for (int i = 0; i < iters; ++i) { REP4(BLOCK8(OP_PKFMA)) }
  out[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.y + a2.x + a3.y + a4.x + a5.y + a6.x + a7.y;
}

template <typename K> void run(const char *name, K kern, float *out, int waves_per_simd) {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int grid = prop.multiProcessorCount * waves_per_simd, iters = 20000; // 256 threads = 4 waves = 1 per SIMD per block
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  kern<<<grid, 256>>>(out, 10);
  hipEventRecord(e0);
  kern<<<grid, 256>>>(out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double instr_per_simd = (double)waves_per_simd * iters * 32.0;
  const double cycles = ms * 1e-3 * prop.clockRate * 1e3;
  printf("%-34s %d waves/SIMD: %.2f cycles per wave64 instruction per SIMD\n", name, waves_per_simd, cycles / instr_per_simd);
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  for (int w : {1, 2, 4, 8}) {
    run("v_fmac_f32_e32 (VOP2, 4 B)", k_fmac, out, w);
    run("v_fma_f32 d=s2 (VOP3, 8 B)", k_fma3, out, w);
    run("v_fma_f32 3 srcs (VOP3)", k_fma3d, out, w);
    run("v_fma_f32 -s2 (VOP3 + neg)", k_fmaneg, out, w);
    run("v_sub_f32_e32", k_sub, out, w);
    run("v_mul_f32_e32", k_mul, out, w);
    run("v_mul_f32_e64 neg", k_mulneg, out, w);
    run("v_max_f32_e32", k_max, out, w);
    run("v_max3_f32", k_max3, out, w);
    run("v_pk_fma_f32 (2 lanes of work)", k_pkfma, out, w);
    printf("\n");
  }
  return 0;
}
