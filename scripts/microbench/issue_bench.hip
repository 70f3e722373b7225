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
#define OP_MULLO(r) "v_mul_lo_u32 " #r ", %8, " #r "\n"
#define OP_MULHI(r) "v_mul_hi_u32 " #r ", %8, " #r "\n"
#define OP_RCP(r) "v_rcp_f32_e32 " #r ", " #r "\n"
#define OP_SQRT(r) "v_sqrt_f32_e32 " #r ", " #r "\n"
#define OP_SIN(r) "v_sin_f32_e32 " #r ", " #r "\n"
#define OP_CVTFU(r) "v_cvt_f32_u32_e32 " #r ", " #r "\n"
#define OP_CVTUF(r) "v_cvt_u32_f32_e32 " #r ", " #r "\n"
#define OP_CNDMASK(r) "v_cndmask_b32_e32 " #r ", %8, " #r ", vcc\n"
#define OP_DPP(r) "v_add_u32_dpp " #r ", %8, " #r " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_ADDCO(r) "v_add_co_u32_e32 " #r ", vcc, %8, " #r "\n"
#define OP_ADDU(r) "v_add_u32_e32 " #r ", %8, " #r "\n"
#define OP_LSHL(r) "v_lshlrev_b32_e32 " #r ", 3, " #r "\n"
#define OP_XOR(r) "v_xor_b32_e32 " #r ", %8, " #r "\n"
#define OP_ALIGNBIT(r) "v_alignbit_b32 " #r ", %8, " #r ", 7\n"
#define OP_CMP(r) "v_cmp_lt_f32_e32 vcc, %8, " #r "\n"
#define OP_MAD24(r) "v_mad_u32_u24 " #r ", %8, %9, " #r "\n"
#define OP_XMOV(r) "v_mov_b32_e32 " #r ", %8\n"
#define OP_XAND(r) "v_and_b32_e32 " #r ", %8, " #r "\n"
#define OP_XOR2(r) "v_or_b32_e32 " #r ", %8, " #r "\n"
#define OP_XCNDE64(r) "v_cndmask_b32_e64 " #r ", %8, " #r ", s[10:11]\n"
#define OP_XCNDVCC(r) "v_cndmask_b32_e32 " #r ", %8, " #r ", vcc\n"
#define OP_XADDF(r) "v_add_f32_e32 " #r ", %8, " #r "\n"
#define OP_XSUBU(r) "v_sub_u32_e32 " #r ", %8, " #r "\n"
#define OP_XMINU(r) "v_min_u32_e32 " #r ", %8, " #r "\n"
#define OP_XMINF(r) "v_min_f32_e32 " #r ", %8, " #r "\n"
#define OP_XBFE(r) "v_bfe_u32 " #r ", " #r ", 3, 5\n"
#define OP_XLSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 2, %8\n"
#define OP_XADD3(r) "v_add3_u32 " #r ", " #r ", %8, %9\n"
#define OP_XANDOR(r) "v_and_or_b32 " #r ", " #r ", %8, %9\n"
#define OP_XLSHR(r) "v_lshrrev_b32_e32 " #r ", 3, " #r "\n"
#define OP_XASHR(r) "v_ashrrev_i32_e32 " #r ", 3, " #r "\n"
#define OP_XCMPE64(r) "v_cmp_lt_f32_e64 s[10:11], %8, " #r "\n"
#define OP_XCMPU(r) "v_cmp_lt_u32_e32 vcc, %8, " #r "\n"
#define OP_XFMAK(r) "v_fmaak_f32 " #r ", %8, " #r ", 0x3f000000\n"
#define OP_XMULSGPR(r) "v_mul_f32_e32 " #r ", s12, " #r "\n"
#define OP_XFMASGPR(r) "v_fma_f32 " #r ", s12, %8, " #r "\n"
#define OP_XMOVDPP(r) "v_mov_b32_dpp " #r ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP_XBCNT(r) "v_bcnt_u32_b32 " #r ", %8, " #r "\n"
#define OP_XMBCNT(r) "v_mbcnt_lo_u32_b32 " #r ", %8, " #r "\n"
#define OP_XFRACT(r) "v_fract_f32_e32 " #r ", " #r "\n"
#define OP_XCVTFI(r) "v_cvt_f32_i32_e32 " #r ", " #r "\n"
#define OP_XLDEXP(r) "v_ldexp_f32 " #r ", " #r ", 3\n"
#define OP_XMAX3I(r) "v_max3_i32 " #r ", " #r ", %8, %9\n"
#define OP_XXAD(r) "v_xad_u32 " #r ", " #r ", %8, %9\n"
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
KERNEL(k_mullo, OP_MULLO)
KERNEL(k_mulhi, OP_MULHI)
KERNEL(k_rcp, OP_RCP)
KERNEL(k_sqrt, OP_SQRT)
KERNEL(k_sin, OP_SIN)
KERNEL(k_cvtfu, OP_CVTFU)
KERNEL(k_cvtuf, OP_CVTUF)
KERNEL(k_cndmask, OP_CNDMASK)
KERNEL(k_dpp, OP_DPP)
KERNEL(k_addco, OP_ADDCO)
KERNEL(k_addu, OP_ADDU)
KERNEL(k_lshl, OP_LSHL)
KERNEL(k_xorb, OP_XOR)
KERNEL(k_alignbit, OP_ALIGNBIT)
KERNEL(k_cmp, OP_CMP)
KERNEL(k_mad24, OP_MAD24)
KERNEL(k_ymov, OP_XMOV)
KERNEL(k_yand, OP_XAND)
KERNEL(k_yor, OP_XOR2)
KERNEL(k_ycnde64, OP_XCNDE64)
KERNEL(k_ycndvcc, OP_XCNDVCC)
KERNEL(k_yaddf, OP_XADDF)
KERNEL(k_ysubu, OP_XSUBU)
KERNEL(k_yminu, OP_XMINU)
KERNEL(k_yminf, OP_XMINF)
KERNEL(k_ybfe, OP_XBFE)
KERNEL(k_ylshladd, OP_XLSHLADD)
KERNEL(k_yadd3, OP_XADD3)
KERNEL(k_yandor, OP_XANDOR)
KERNEL(k_ylshr, OP_XLSHR)
KERNEL(k_yashr, OP_XASHR)
KERNEL(k_ycmpe64, OP_XCMPE64)
KERNEL(k_ycmpu, OP_XCMPU)
KERNEL(k_yfmak, OP_XFMAK)
KERNEL(k_ymulsgpr, OP_XMULSGPR)
KERNEL(k_yfmasgpr, OP_XFMASGPR)
KERNEL(k_ymovdpp, OP_XMOVDPP)
KERNEL(k_ybcnt, OP_XBCNT)
KERNEL(k_ymbcnt, OP_XMBCNT)
KERNEL(k_yfract, OP_XFRACT)
KERNEL(k_ycvtfi, OP_XCVTFI)
KERNEL(k_yldexp, OP_XLDEXP)
KERNEL(k_ymax3i, OP_XMAX3I)
KERNEL(k_yxad, OP_XXAD)


__global__ void __launch_bounds__(256) k_pkfma(float *out, int iters) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f a0 = {1, 2}, a1 = a0, a2 = a0, a3 = a0, a4 = a0, a5 = a0, a6 = a0, a7 = a0, x = {1.0000001f, 1.f}, y = {0.9999999f, 1.f}, z = x;
  for (int i = 0; i < iters; ++i) { REP4(BLOCK8(OP_PKFMA)) }
  out[blockIdx.x * 256 + threadIdx.x] = a0.x + a1.y + a2.x + a3.y + a4.x + a5.y + a6.x + a7.y;
}

#define BLOCK8D(OP)                                                                                        \
  asm volatile(OP(%0) OP(%1) OP(%2) OP(%3) OP(%4) OP(%5) OP(%6) OP(%7)                                       \
               : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)             \
               : "v"(x), "v"(y), "v"(xs), "v"(ys));
#define OP_FMA64(r) "v_fma_f64 " #r ", %8, %9, " #r "\n"
#define OP_MAD64(r) "v_mad_u64_u32 " #r ", vcc, %10, %11, " #r "\n"
#define OP_PKMUL(r) "v_pk_mul_f32 " #r ", %8, " #r "\n"
#define OP_PKADD(r) "v_pk_add_f32 " #r ", %8, " #r "\n"
#define OP_LSHL64(r) "v_lshlrev_b64 " #r ", 3, " #r "\n"
#define OP_CVTF64(r) "v_cvt_f64_u32_e32 " #r ", %10\n"
#define KERNEL64(NAME, OP, T)                                                                               \
  __global__ void __launch_bounds__(256) NAME(float *out, int iters) {                                       \
    T a0 = (T)threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;                           \
    T x = (T)3, y = (T)5; unsigned xs = threadIdx.x | 1, ys = 12345;                                         \
    for (int i = 0; i < iters; ++i) { REP4(BLOCK8D(OP)) }                                                    \
    out[blockIdx.x * 256 + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);                   \
  }
KERNEL64(k_fma64, OP_FMA64, double)
KERNEL64(k_mad64, OP_MAD64, unsigned long long)
KERNEL64(k_pkmul, OP_PKMUL, double)
KERNEL64(k_pkadd, OP_PKADD, double)
KERNEL64(k_lshl64, OP_LSHL64, unsigned long long)
KERNEL64(k_cvtf64, OP_CVTF64, double)


__global__ void __launch_bounds__(256) k_cnd_vccset(float *out, int iters) {
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; float x = 1.0000001f, y = 0.9999999f, z = 0.5f;
  asm volatile("s_mov_b64 vcc, 0x5555" ::: "vcc");
  for (int i = 0; i < iters; ++i) { REP4(BLOCK8(OP_CNDMASK)) }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
#define OP_CMPCND(r) "v_cmp_lt_f32_e32 vcc, %8, " #r "\n v_cndmask_b32_e32 " #r ", %9, " #r ", vcc\n"
__global__ void __launch_bounds__(256) k_cmp_cnd(float *out, int iters) {
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; float x = 1.0000001f, y = 0.9999999f, z = 0.5f;
  for (int i = 0; i < iters; ++i) { REP4(BLOCK8(OP_CMPCND)) }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
#define OP_CMPCND64(r) "v_cmp_lt_f32_e64 s[10:11], %8, " #r "\n v_cndmask_b32_e64 " #r ", %9, " #r ", s[10:11]\n"
__global__ void __launch_bounds__(256) k_cmp_cnd64(float *out, int iters) {
  float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7; float x = 1.0000001f, y = 0.9999999f, z = 0.5f;
  for (int i = 0; i < iters; ++i) { REP4(BLOCK8(OP_CMPCND64)) }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
#define OP_MOVS(r) "v_mov_b32_e32 " #r ", s12\n"
KERNEL(k_movs, OP_MOVS)
#define OP_ADDS(r) "v_add_f32_e32 " #r ", s12, " #r "\n"
KERNEL(k_adds, OP_ADDS)
#define OP_MULC(r) "v_mul_f32_e32 " #r ", 2.0, " #r "\n"
KERNEL(k_mulc, OP_MULC)
#define OP_LSHLV(r) "v_lshlrev_b32_e32 " #r ", %8, " #r "\n"
KERNEL(k_lshlv, OP_LSHLV)
#define OP_ADDCV(r) "v_addc_co_u32_e32 " #r ", vcc, %8, " #r ", vcc\n"
KERNEL(k_addcv, OP_ADDCV)
#define OP_RDL(r) "v_readlane_b32 s12, " #r ", 3\n"
KERNEL(k_rdl, OP_RDL)

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
  for (int w : {8}) {
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
    run("v_pk_mul_f32", k_pkmul, out, w);
    run("v_pk_add_f32", k_pkadd, out, w);
    run("v_mul_lo_u32", k_mullo, out, w);
    run("v_mul_hi_u32", k_mulhi, out, w);
    run("v_mad_u64_u32", k_mad64, out, w);
    run("v_mad_u32_u24", k_mad24, out, w);
    run("v_rcp_f32", k_rcp, out, w);
    run("v_sqrt_f32", k_sqrt, out, w);
    run("v_sin_f32", k_sin, out, w);
    run("v_cvt_f32_u32", k_cvtfu, out, w);
    run("v_cvt_u32_f32", k_cvtuf, out, w);
    run("v_cndmask_b32", k_cndmask, out, w);
    run("v_add_u32_dpp row_shr", k_dpp, out, w);
    run("v_add_co_u32", k_addco, out, w);
    run("v_add_u32", k_addu, out, w);
    run("v_lshlrev_b32", k_lshl, out, w);
    run("v_xor_b32", k_xorb, out, w);
    run("v_alignbit_b32", k_alignbit, out, w);
    run("v_cmp_lt_f32", k_cmp, out, w);
    run("mov", k_ymov, out, w);
    run("and", k_yand, out, w);
    run("or", k_yor, out, w);
    run("cnde64", k_ycnde64, out, w);
    run("cndvcc", k_ycndvcc, out, w);
    run("addf", k_yaddf, out, w);
    run("subu", k_ysubu, out, w);
    run("minu", k_yminu, out, w);
    run("minf", k_yminf, out, w);
    run("bfe", k_ybfe, out, w);
    run("lshladd", k_ylshladd, out, w);
    run("add3", k_yadd3, out, w);
    run("andor", k_yandor, out, w);
    run("lshr", k_ylshr, out, w);
    run("ashr", k_yashr, out, w);
    run("cmpe64", k_ycmpe64, out, w);
    run("cmpu", k_ycmpu, out, w);
    run("fmak", k_yfmak, out, w);
    run("mulsgpr", k_ymulsgpr, out, w);
    run("fmasgpr", k_yfmasgpr, out, w);
    run("movdpp", k_ymovdpp, out, w);
    run("bcnt", k_ybcnt, out, w);
    run("mbcnt", k_ymbcnt, out, w);
    run("fract", k_yfract, out, w);
    run("cvtfi", k_ycvtfi, out, w);
    run("ldexp", k_yldexp, out, w);
    run("max3i", k_ymax3i, out, w);
    run("xad", k_yxad, out, w);
    run("cndmask vcc (vcc set by s_mov)", k_cnd_vccset, out, w);
    run("v_cmp vcc + cndmask vcc (2 instr)", k_cmp_cnd, out, w);
    run("v_cmp s[] + cndmask s[] (2 instr)", k_cmp_cnd64, out, w);
    run("v_mov from sgpr", k_movs, out, w);
    run("v_add_f32 sgpr", k_adds, out, w);
    run("v_mul_f32 inline const", k_mulc, out, w);
    run("v_lshlrev_b32 vgpr shift", k_lshlv, out, w);
    run("v_addc_co_u32 vcc", k_addcv, out, w);
    run("v_readlane", k_rdl, out, w);
    run("v_fma_f64", k_fma64, out, w);
    run("v_lshlrev_b64", k_lshl64, out, w);
    run("v_cvt_f64_u32", k_cvtf64, out, w);
    printf("\n");
  }
  return 0;
}
