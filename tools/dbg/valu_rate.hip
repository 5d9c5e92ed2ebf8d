// micro-benchmark: issue cost of the VALU instructions the row kernel is made of (gfx950).
//   cycles per wave-instruction with 1 wave per SIMD (256 threads) and with 2 (512 threads), one workgroup.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
#define BENCH(NAME, ASM)                                                                                    \
  __global__ void k_##NAME(unsigned long long *out, int iters) {                                            \
    unsigned int a0 = threadIdx.x, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 + 11, a5 = a0 + 13,        \
                 a6 = a0 + 17, a7 = a0 + 19;                                                                \
    unsigned long long d0 = a0, d1 = a1, d2 = a2, d3 = a3;                                                  \
    unsigned int sh = (a0 & 31) + 1;                                                                        \
    float f0 = a0 * 0.5f, f1 = a1 * 0.25f, f2 = 1.5f, f3 = 2.5f;                                            \
    __syncthreads();                                                                                        \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                             \
    for (int i = 0; i < iters; ++i) {                                                                       \
      asm volatile(REP8(ASM)                                                                                \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),         \
                     "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)          \
                   : "v"(sh)                                                                                \
                   : "vcc", "s10", "s11");                                                                                \
    }                                                                                                       \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                             \
    if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;                                           \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + d0 + d1 + d2 + d3 + f0 + f1 + f2 + f3 == 12345.f) out[63] = 1; \
  }
// each ASM body = 4 independent instructions (x8 repeats = 32 per loop trip)
BENCH(fma, "v_fma_f32 %12, %12, %14, %15\n v_fma_f32 %13, %13, %14, %15\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %1, %1, %14, %15\n")
BENCH(add_u32, "v_add_u32 %0, %0, %16\n v_add_u32 %1, %1, %16\n v_add_u32 %2, %2, %16\n v_add_u32 %3, %3, %16\n")
BENCH(and_b32, "v_and_b32 %0, %0, %16\n v_and_b32 %1, %1, %16\n v_and_b32 %2, %2, %16\n v_and_b32 %3, %3, %16\n")
BENCH(bfe_i32, "v_bfe_i32 %0, %4, 3, 1\n v_bfe_i32 %1, %5, 4, 1\n v_bfe_i32 %2, %6, 5, 1\n v_bfe_i32 %3, %7, 6, 1\n")
BENCH(lshr_b64, "v_lshrrev_b64 %8, %16, %8\n v_lshrrev_b64 %9, %16, %9\n v_lshrrev_b64 %10, %16, %10\n v_lshrrev_b64 %11, %16, %11\n")
BENCH(lshr_b32, "v_lshrrev_b32 %0, %16, %0\n v_lshrrev_b32 %1, %16, %1\n v_lshrrev_b32 %2, %16, %2\n v_lshrrev_b32 %3, %16, %3\n")
BENCH(add64, "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_add_co_u32 %2, vcc, %2, %6\n v_addc_co_u32 %3, vcc, %3, %7, vcc\n")
BENCH(cvt_u32, "v_cvt_u32_f32 %0, %12\n v_cvt_u32_f32 %1, %13\n v_cvt_u32_f32 %2, %14\n v_cvt_u32_f32 %3, %15\n")
BENCH(rndne, "v_rndne_f32 %12, %12\n v_rndne_f32 %13, %13\n v_rndne_f32 %14, %14\n v_rndne_f32 %15, %15\n")
BENCH(min_f32, "v_min_f32 %12, %12, %14\n v_min_f32 %13, %13, %14\n v_min_f32 %0, %0, %14\n v_min_f32 %1, %1, %14\n")
BENCH(mov, "v_mov_b32 %0, 0\n v_mov_b32 %1, 0\n v_mov_b32 %2, 0\n v_mov_b32 %3, 0\n")
BENCH(mad_u64, "v_mad_u64_u32 %8, vcc, %0, %16, %8\n v_mad_u64_u32 %9, vcc, %1, %16, %9\n v_mad_u64_u32 %10, vcc, %2, %16, %10\n v_mad_u64_u32 %11, vcc, %3, %16, %11\n")
BENCH(alignbit, "v_alignbit_b32 %0, %4, %5, %16\n v_alignbit_b32 %1, %5, %6, %16\n v_alignbit_b32 %2, %6, %7, %16\n v_alignbit_b32 %3, %7, %4, %16\n")
BENCH(max3, "v_max3_f32 %12, %12, %14, %15\n v_max3_f32 %13, %13, %14, %15\n v_max3_f32 %0, %0, %14, %15\n v_max3_f32 %1, %1, %14, %15\n")
BENCH(dpp_mov, "v_mov_b32_dpp %0, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
BENCH(fma_dep, "v_fma_f32 %12, %12, %14, %15\n v_fma_f32 %12, %12, %14, %15\n v_fma_f32 %12, %12, %14, %15\n v_fma_f32 %12, %12, %14, %15\n")
BENCH(add64_dep, "v_add_co_u32 %0, vcc, %0, %4\n v_addc_co_u32 %1, vcc, %1, %5, vcc\n v_add_co_u32 %0, vcc, %0, %6\n v_addc_co_u32 %1, vcc, %1, %7, vcc\n")
BENCH(lshl_add, "v_lshl_add_u32 %0, %4, 3, %0\n v_lshl_add_u32 %1, %5, 3, %1\n v_lshl_add_u32 %2, %6, 3, %2\n v_lshl_add_u32 %3, %7, 3, %3\n")
BENCH(cndmask, "v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %5, %6, vcc\n v_cndmask_b32 %2, %6, %7, vcc\n v_cndmask_b32 %3, %7, %4, vcc\n")
BENCH(ldexp, "v_ldexp_f32 %12, %12, %16\n v_ldexp_f32 %13, %13, %16\n v_ldexp_f32 %14, %14, %16\n v_ldexp_f32 %15, %15, %16\n")
BENCH(exp, "v_exp_f32 %12, %12\n v_exp_f32 %13, %13\n v_exp_f32 %14, %14\n v_exp_f32 %15, %15\n")
BENCH(add_f64, "v_add_f64 %8, %8, %9\n v_add_f64 %9, %9, %10\n v_add_f64 %10, %10, %11\n v_add_f64 %11, %11, %8\n")
BENCH(sub_co_lshl64, "v_lshlrev_b64 %8, %16, %8\n v_lshlrev_b64 %9, %16, %9\n v_lshlrev_b64 %10, %16, %10\n v_lshlrev_b64 %11, %16, %11\n")

BENCH(min_u32, "v_min_u32 %0, %0, %16\n v_min_u32 %1, %1, %16\n v_min_u32 %2, %2, %16\n v_min_u32 %3, %3, %16\n")
BENCH(sub_u32, "v_sub_u32 %0, %4, %0\n v_sub_u32 %1, %5, %1\n v_sub_u32 %2, %6, %2\n v_sub_u32 %3, %7, %3\n")
BENCH(max_f32, "v_max_f32 %12, %12, %14\n v_max_f32 %13, %13, %14\n v_max_f32 %0, %0, %14\n v_max_f32 %1, %1, %14\n")
BENCH(sub_f32, "v_sub_f32 %12, %12, %14\n v_sub_f32 %13, %13, %14\n v_sub_f32 %0, %0, %14\n v_sub_f32 %1, %1, %14\n")
BENCH(mul_f32, "v_mul_f32 %12, %12, %14\n v_mul_f32 %13, %13, %14\n v_mul_f32 %0, %0, %14\n v_mul_f32 %1, %1, %14\n")
BENCH(fmac_f32, "v_fmac_f32 %12, %14, %15\n v_fmac_f32 %13, %14, %15\n v_fmac_f32 %0, %14, %15\n v_fmac_f32 %1, %14, %15\n")
BENCH(fmaak, "v_fmaak_f32 %12, %12, %14, 0x3f317218\n v_fmaak_f32 %13, %13, %14, 0x3f317218\n v_fmaak_f32 %0, %0, %14, 0x3f317218\n v_fmaak_f32 %1, %1, %14, 0x3f317218\n")
BENCH(cvt_i32, "v_cvt_i32_f32 %0, %12\n v_cvt_i32_f32 %1, %13\n v_cvt_i32_f32 %2, %14\n v_cvt_i32_f32 %3, %15\n")
BENCH(ashr_i32, "v_ashrrev_i32 %0, %16, %0\n v_ashrrev_i32 %1, %16, %1\n v_ashrrev_i32 %2, %16, %2\n v_ashrrev_i32 %3, %16, %3\n")
BENCH(or_b32, "v_or_b32 %0, %0, %16\n v_or_b32 %1, %1, %16\n v_or_b32 %2, %2, %16\n v_or_b32 %3, %3, %16\n")
BENCH(bfi, "v_bfi_b32 %0, %4, %5, %0\n v_bfi_b32 %1, %5, %6, %1\n v_bfi_b32 %2, %6, %7, %2\n v_bfi_b32 %3, %7, %4, %3\n")
BENCH(and_or, "v_and_or_b32 %0, %4, %5, %0\n v_and_or_b32 %1, %5, %6, %1\n v_and_or_b32 %2, %6, %7, %2\n v_and_or_b32 %3, %7, %4, %3\n")
BENCH(add3, "v_add3_u32 %0, %4, %5, %0\n v_add3_u32 %1, %5, %6, %1\n v_add3_u32 %2, %6, %7, %2\n v_add3_u32 %3, %7, %4, %3\n")
BENCH(med3_u32, "v_med3_u32 %0, %4, %5, %0\n v_med3_u32 %1, %5, %6, %1\n v_med3_u32 %2, %6, %7, %2\n v_med3_u32 %3, %7, %4, %3\n")
BENCH(cmp_lt, "v_cmp_lt_u32 vcc, %0, %4\n v_cmp_lt_u32 vcc, %1, %5\n v_cmp_lt_u32 vcc, %2, %6\n v_cmp_lt_u32 vcc, %3, %7\n")
BENCH(add_f32, "v_add_f32 %12, %12, %14\n v_add_f32 %13, %13, %14\n v_add_f32 %0, %0, %14\n v_add_f32 %1, %1, %14\n")
BENCH(add_co_only, "v_add_co_u32 %0, vcc, %0, %4\n v_add_co_u32 %1, vcc, %1, %5\n v_add_co_u32 %2, vcc, %2, %6\n v_add_co_u32 %3, vcc, %3, %7\n")
BENCH(mul_u24, "v_mul_u32_u24 %0, %0, %16\n v_mul_u32_u24 %1, %1, %16\n v_mul_u32_u24 %2, %2, %16\n v_mul_u32_u24 %3, %3, %16\n")
BENCH(pk_fma, "v_pk_fma_f32 %8, %8, %9, %10\n v_pk_fma_f32 %9, %9, %10, %11\n v_pk_fma_f32 %10, %10, %11, %8\n v_pk_fma_f32 %11, %11, %8, %9\n")

BENCH(cvt_f64_f32, "v_cvt_f64_f32 %8, %12\n v_cvt_f64_f32 %9, %13\n v_cvt_f64_f32 %10, %14\n v_cvt_f64_f32 %11, %15\n")
BENCH(ldexp_f64, "v_ldexp_f64 %8, %8, %16\n v_ldexp_f64 %9, %9, %16\n v_ldexp_f64 %10, %10, %16\n v_ldexp_f64 %11, %11, %16\n")
BENCH(floor_f64, "v_floor_f64 %8, %8\n v_floor_f64 %9, %9\n v_floor_f64 %10, %10\n v_floor_f64 %11, %11\n")
BENCH(trunc_f64, "v_trunc_f64 %8, %8\n v_trunc_f64 %9, %9\n v_trunc_f64 %10, %10\n v_trunc_f64 %11, %11\n")
BENCH(fma_f64, "v_fma_f64 %8, %8, %9, %10\n v_fma_f64 %9, %9, %10, %11\n v_fma_f64 %10, %10, %11, %8\n v_fma_f64 %11, %11, %8, %9\n")
BENCH(cvt_f64_u32, "v_cvt_f64_u32 %8, %0\n v_cvt_f64_u32 %9, %1\n v_cvt_f64_u32 %10, %2\n v_cvt_f64_u32 %11, %3\n")
BENCH(mul_f64, "v_mul_f64 %8, %8, %9\n v_mul_f64 %9, %9, %10\n v_mul_f64 %10, %10, %11\n v_mul_f64 %11, %11, %8\n")

BENCH(fma_clamp, "v_fma_f32 %12, %12, %14, %15 clamp\n v_fma_f32 %13, %13, %14, %15 clamp\n v_fma_f32 %0, %0, %14, %15 clamp\n v_fma_f32 %1, %1, %14, %15 clamp\n")
BENCH(max_i32, "v_max_i32 %0, %0, %16\n v_max_i32 %1, %1, %16\n v_max_i32 %2, %2, %16\n v_max_i32 %3, %3, %16\n")
BENCH(exec_add, "s_mov_b64 exec, 0x5555\n v_add_f32 %12, %12, %14\n v_add_f32 %13, %13, %14\n s_mov_b64 exec, -1\n v_add_f32 %0, %0, %14\n v_add_f32 %1, %1, %14\n")
BENCH(setreg_fma, "s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n v_fma_f32 %12, %12, %14, %15\n v_fma_f32 %13, %13, %14, %15\n s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0\n v_fma_f32 %0, %0, %14, %15\n v_fma_f32 %1, %1, %14, %15\n")
BENCH(cndmask_s, "v_cndmask_b32 %0, %4, %5, s[10:11]\n v_cndmask_b32 %1, %5, %6, s[10:11]\n v_cndmask_b32 %2, %6, %7, s[10:11]\n v_cndmask_b32 %3, %7, %4, s[10:11]\n")
BENCH(lshl_b32, "v_lshlrev_b32 %0, 16, %0\n v_lshlrev_b32 %1, 16, %1\n v_lshlrev_b32 %2, 16, %2\n v_lshlrev_b32 %3, 16, %3\n")
BENCH(add_dpp, "v_add_u32_dpp %0, %4, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %5, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %6, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %7, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n")
BENCH(fract, "v_fract_f32 %12, %12\n v_fract_f32 %13, %13\n v_fract_f32 %14, %14\n v_fract_f32 %15, %15\n")
BENCH(and_sdwa, "v_and_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n v_and_b32_sdwa %1, %1, %5 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_and_b32_sdwa %2, %2, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_and_b32_sdwa %3, %3, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n")
BENCH(pk_add, "v_pk_add_f32 %8, %8, %9\n v_pk_add_f32 %9, %9, %10\n v_pk_add_f32 %10, %10, %11\n v_pk_add_f32 %11, %11, %8\n")
BENCH(mix8, "v_fma_f32 %12, %12, %14, %15\n v_add_u32 %0, %0, %16\n v_fma_f32 %13, %13, %14, %15\n v_and_b32 %1, %1, %16\n")

BENCH(cvt_bf16, "v_cvt_f32_bf16 %12, %0\n v_cvt_f32_bf16 %13, %1\n v_cvt_f32_bf16 %14, %2\n v_cvt_f32_bf16 %15, %3\n")
BENCH(cvt_bf16_hi, "v_cvt_f32_bf16_sdwa %12, %0 dst_sel:DWORD dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_cvt_f32_bf16_sdwa %13, %1 dst_sel:DWORD dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_cvt_f32_bf16_sdwa %14, %2 dst_sel:DWORD dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n v_cvt_f32_bf16_sdwa %15, %3 dst_sel:DWORD dst_unused:UNUSED_PRESERVE src0_sel:WORD_1\n")
BENCH(pk_max_f16, "v_pk_max_f16 %0, %0, %4\n v_pk_max_f16 %1, %1, %5\n v_pk_max_f16 %2, %2, %6\n v_pk_max_f16 %3, %3, %7\n")
BENCH(perm, "v_perm_b32 %0, %4, %5, %16\n v_perm_b32 %1, %5, %6, %16\n v_perm_b32 %2, %6, %7, %16\n v_perm_b32 %3, %7, %4, %16\n")
BENCH(pk_max_i16, "v_pk_max_i16 %0, %0, %4\n v_pk_max_i16 %1, %1, %5\n v_pk_max_i16 %2, %2, %6\n v_pk_max_i16 %3, %3, %7\n")

#define RUN(NAME)                                                                                           \
  for (int th : {256, 512, 1024}) {                                                                         \
    hipLaunchKernelGGL(k_##NAME, dim3(1), dim3(th), 0, 0, d, iters);                                        \
    hipLaunchKernelGGL(k_##NAME, dim3(1), dim3(th), 0, 0, d, iters);                                        \
    hipDeviceSynchronize();                                                                                 \
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);                                                      \
    unsigned long long mx = 0;                                                                              \
    for (int w = 0; w < th / 64; ++w) mx = h[w] > mx ? h[w] : mx;                                           \
    printf("%-14s %d waves/SIMD: %.2f cycles per wave-instruction (wave), %.2f per SIMD slot\n", #NAME,     \
           th / 256, (double)mx / (iters * 32.0), (double)mx / (iters * 32.0) / (th / 256));                \
  }
int main() {
  unsigned long long *d, h[64];
  hipMalloc(&d, sizeof(h));
  const int iters = 2000;
  RUN(fma) RUN(fma_dep) RUN(add_u32) RUN(and_b32) RUN(bfe_i32) RUN(lshr_b32) RUN(lshr_b64) RUN(sub_co_lshl64) RUN(add64) RUN(add64_dep)
  RUN(cvt_u32) RUN(rndne) RUN(min_f32) RUN(max3) RUN(mov) RUN(mad_u64) RUN(alignbit) RUN(dpp_mov) RUN(lshl_add) RUN(cndmask)
  RUN(ldexp) RUN(exp) RUN(add_f64)
  RUN(min_u32) RUN(sub_u32) RUN(max_f32) RUN(sub_f32) RUN(mul_f32) RUN(add_f32) RUN(fmac_f32) RUN(fmaak) RUN(cvt_i32) RUN(ashr_i32) RUN(or_b32)
  RUN(bfi) RUN(and_or) RUN(add3) RUN(med3_u32) RUN(cmp_lt) RUN(add_co_only) RUN(mul_u24) RUN(pk_fma)
  RUN(fma_clamp) RUN(max_i32) RUN(exec_add) RUN(setreg_fma) RUN(cndmask_s) RUN(lshl_b32) RUN(add_dpp) RUN(fract) RUN(and_sdwa) RUN(pk_add) RUN(mix8)
  RUN(cvt_bf16) RUN(cvt_bf16_hi) RUN(pk_max_f16) RUN(pk_max_i16) RUN(perm)
  RUN(cvt_f64_f32) RUN(ldexp_f64) RUN(floor_f64) RUN(trunc_f64) RUN(fma_f64) RUN(cvt_f64_u32) RUN(mul_f64)
  return 0;
}
