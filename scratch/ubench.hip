// VALU instruction throughput microbenchmark for gfx950: cycles per wave-instruction per SIMD with the
// SIMD saturated (8 waves/SIMD) -- independent instructions, 64 per loop body.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define REP4(x) x x x x
#define REP16(x) REP4(x) REP4(x) REP4(x) REP4(x)
#define REP64(x) REP16(x) REP16(x) REP16(x) REP16(x)
#define KERNEL(name, ASMSTR)                                                                   \
  __global__ __launch_bounds__(256) void name(uint32_t* out, int iters) {                      \
    uint32_t a = threadIdx.x * 3 + 0x3c003c00, b = threadIdx.x + 0x38003800, c = 0x3c003c00, d = 0x40004000; \
    double da = a, db = b;                                                                     \
    float fa = 1.0f + threadIdx.x, fb = 0.5f; uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0;         \
    uint64_t q0 = 0, q1 = 0;                                                                   \
    for (int i = 0; i < iters; ++i) {                                                          \
      asm volatile(REP16(ASMSTR) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(q0), "+v"(q1)  \
                   : "v"(a), "v"(b), "v"(c), "v"(d), "v"(fa), "v"(fb), "v"(da), "v"(db));      \
    }                                                                                          \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + (uint32_t)q0 + (uint32_t)q1; \
  }
// each ASMSTR has 4 independent instructions -> 64 per loop iteration
KERNEL(k_fma_f32,   "v_fma_f32 %0, %10, %11, %0\n v_fma_f32 %1, %10, %11, %1\n v_fma_f32 %2, %10, %11, %2\n v_fma_f32 %3, %10, %11, %3\n")
KERNEL(k_mul_f32,   "v_mul_f32 %0, %10, %11\n v_mul_f32 %1, %10, %11\n v_mul_f32 %2, %10, %11\n v_mul_f32 %3, %10, %11\n")
KERNEL(k_pk_fma_f32,"v_pk_fma_f32 %4, %12, %13, %4\n v_pk_fma_f32 %5, %12, %13, %5\n v_pk_fma_f32 %4, %12, %13, %4\n v_pk_fma_f32 %5, %12, %13, %5\n")
KERNEL(k_pk_mul_f32,"v_pk_mul_f32 %4, %12, %13\n v_pk_mul_f32 %5, %12, %13\n v_pk_mul_f32 %4, %12, %13\n v_pk_mul_f32 %5, %12, %13\n")
KERNEL(k_cvt_f32_f16,"v_cvt_f32_f16 %0, %6\n v_cvt_f32_f16 %1, %7\n v_cvt_f32_f16 %2, %8\n v_cvt_f32_f16 %3, %9\n")
KERNEL(k_cvt_f32_f16_sdwa,"v_cvt_f32_f16_sdwa %0, %6 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %1, %7 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %2, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n v_cvt_f32_f16_sdwa %3, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n")
KERNEL(k_cvt_f16_f32,"v_cvt_f16_f32 %0, %10\n v_cvt_f16_f32 %1, %11\n v_cvt_f16_f32 %2, %10\n v_cvt_f16_f32 %3, %11\n")
KERNEL(k_cvt_pk_f16_f32,"v_cvt_pk_f16_f32 %0, %10, %11\n v_cvt_pk_f16_f32 %1, %10, %11\n v_cvt_pk_f16_f32 %2, %10, %11\n v_cvt_pk_f16_f32 %3, %10, %11\n")
KERNEL(k_pk_add_f16,"v_pk_add_f16 %0, %6, %7\n v_pk_add_f16 %1, %6, %8\n v_pk_add_f16 %2, %6, %9\n v_pk_add_f16 %3, %7, %8\n")
KERNEL(k_pk_mul_f16,"v_pk_mul_f16 %0, %6, %7\n v_pk_mul_f16 %1, %6, %8\n v_pk_mul_f16 %2, %6, %9\n v_pk_mul_f16 %3, %7, %8\n")
KERNEL(k_pk_fma_f16,"v_pk_fma_f16 %0, %6, %7, %0\n v_pk_fma_f16 %1, %6, %8, %1\n v_pk_fma_f16 %2, %6, %9, %2\n v_pk_fma_f16 %3, %7, %8, %3\n")
KERNEL(k_pk_max_u16,"v_pk_max_u16 %0, %6, %7\n v_pk_max_u16 %1, %6, %8\n v_pk_max_u16 %2, %6, %9\n v_pk_max_u16 %3, %7, %8\n")
KERNEL(k_pk_min_f16,"v_pk_min_f16 %0, %6, %7\n v_pk_min_f16 %1, %6, %8\n v_pk_min_f16 %2, %6, %9\n v_pk_min_f16 %3, %7, %8\n")
KERNEL(k_and_b32,   "v_and_b32 %0, %6, %7\n v_and_b32 %1, %6, %8\n v_and_b32 %2, %6, %9\n v_and_b32 %3, %7, %8\n")
KERNEL(k_bfi_b32,   "v_bfi_b32 %0, %6, %7, %8\n v_bfi_b32 %1, %6, %8, %9\n v_bfi_b32 %2, %6, %9, %7\n v_bfi_b32 %3, %7, %8, %9\n")
KERNEL(k_perm_b32,  "v_perm_b32 %0, %6, %7, %8\n v_perm_b32 %1, %6, %8, %9\n v_perm_b32 %2, %6, %9, %7\n v_perm_b32 %3, %7, %8, %9\n")
KERNEL(k_rndne_f16, "v_rndne_f16 %0, %6\n v_rndne_f16 %1, %7\n v_rndne_f16 %2, %8\n v_rndne_f16 %3, %9\n")
KERNEL(k_fma_mix,   "v_fma_mix_f32 %0, %6, %7, %10 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %1, %6, %8, %10 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %2, %6, %9, %10 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %3, %7, %8, %10 op_sel_hi:[1,1,0]\n")
KERNEL(k_mov_dpp,   "v_mov_b32_dpp %0, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %8 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %9 row_half_mirror row_mask:0xf bank_mask:0xf\n")
KERNEL(k_add_f64,   "v_add_f64 %4, %12, %13\n v_add_f64 %5, %12, %13\n v_add_f64 %4, %12, %13\n v_add_f64 %5, %12, %13\n")
KERNEL(k_fma_f64,   "v_fma_f64 %4, %12, %13, %4\n v_fma_f64 %5, %12, %13, %5\n v_fma_f64 %4, %12, %13, %4\n v_fma_f64 %5, %12, %13, %5\n")
KERNEL(k_cvt_f64_f32,"v_cvt_f64_f32 %4, %10\n v_cvt_f64_f32 %5, %11\n v_cvt_f64_f32 %4, %10\n v_cvt_f64_f32 %5, %11\n")
KERNEL(k_rcp_f32,   "v_rcp_f32 %0, %10\n v_rcp_f32 %1, %11\n v_rcp_f32 %2, %10\n v_rcp_f32 %3, %11\n")
KERNEL(k_cvt_i32_f32,"v_cvt_i32_f32 %0, %10\n v_cvt_i32_f32 %1, %11\n v_cvt_i32_f32 %2, %10\n v_cvt_i32_f32 %3, %11\n")
KERNEL(k_cmp_cndmask,"v_cmp_lt_f32 vcc, %10, %11\n v_cndmask_b32 %0, %6, %7, vcc\n v_cmp_lt_f32 vcc, %11, %10\n v_cndmask_b32 %1, %6, %7, vcc\n")
KERNEL(k_med3_f16,  "v_med3_f16 %0, %6, %7, %8\n v_med3_f16 %1, %6, %8, %9\n v_med3_f16 %2, %6, %9, %7\n v_med3_f16 %3, %7, %8, %9\n")
KERNEL(k_lshl_add_u64,"v_lshl_add_u64 %4, %12, 0, %13\n v_lshl_add_u64 %5, %12, 0, %13\n v_lshl_add_u64 %4, %12, 0, %13\n v_lshl_add_u64 %5, %12, 0, %13\n")
KERNEL(k_pk_add_u16,"v_pk_add_u16 %0, %6, %7\n v_pk_add_u16 %1, %6, %8\n v_pk_add_u16 %2, %6, %9\n v_pk_add_u16 %3, %7, %8\n")
KERNEL(k_add_f16,   "v_add_f16 %0, %6, %7\n v_add_f16 %1, %6, %8\n v_add_f16 %2, %6, %9\n v_add_f16 %3, %7, %8\n")
KERNEL(k_cvt_f32_bf16,"v_lshlrev_b32 %0, 16, %6\n v_and_b32 %1, 0xffff0000, %6\n v_lshlrev_b32 %2, 16, %7\n v_and_b32 %3, 0xffff0000, %7\n")
KERNEL(k_dot2_f32_f16, "v_dot2_f32_f16 %0, %6, %7, %0\n v_dot2_f32_f16 %1, %6, %8, %1\n v_dot2_f32_f16 %2, %6, %9, %2\n v_dot2_f32_f16 %3, %7, %8, %3\n")
KERNEL(k_dot2c_f32_f16, "v_dot2c_f32_f16 %0, %6, %7\n v_dot2c_f32_f16 %1, %6, %8\n v_dot2c_f32_f16 %2, %6, %9\n v_dot2c_f32_f16 %3, %7, %8\n")
KERNEL(k_max3_f16_abs, "v_max3_f16 %0, |%6|, |%7|, %0 op_sel:[0,1,0,0]\n v_max3_f16 %1, |%6|, |%8|, %1 op_sel:[0,1,0,0]\n v_max3_f16 %2, |%6|, |%9|, %2 op_sel:[0,1,0,0]\n v_max3_f16 %3, |%7|, |%8|, %3 op_sel:[0,1,0,0]\n")
KERNEL(k_fma_mix_abs, "v_fma_mix_f32 %0, |%6|, %7, %0 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %1, |%6|, %8, %1 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %2, |%6|, %9, %2 op_sel_hi:[1,1,0]\n v_fma_mix_f32 %3, |%7|, %8, %3 op_sel_hi:[1,1,0]\n")
KERNEL(k_add_f32_abs, "v_add_f32 %0, |%10|, %0\n v_add_f32 %1, |%11|, %1\n v_add_f32 %2, |%10|, %2\n v_add_f32 %3, |%11|, %3\n")
KERNEL(k_pk_add_f32, "v_pk_add_f32 %4, %12, %13\n v_pk_add_f32 %5, %12, %13\n v_pk_add_f32 %4, %12, %13\n v_pk_add_f32 %5, %12, %13\n")
KERNEL(k_max_f16, "v_max_f16 %0, %6, %7\n v_max_f16 %1, %6, %8\n v_max_f16 %2, %6, %9\n v_max_f16 %3, %7, %8\n")
KERNEL(k_pk_max_f16, "v_pk_max_f16 %0, %6, %7\n v_pk_max_f16 %1, %6, %8\n v_pk_max_f16 %2, %6, %9\n v_pk_max_f16 %3, %7, %8\n")
KERNEL(k_max3_u32, "v_max3_u32 %0, %6, %7, %0\n v_max3_u32 %1, %6, %8, %1\n v_max3_u32 %2, %6, %9, %2\n v_max3_u32 %3, %7, %8, %3\n")
KERNEL(k_and_or_b32, "v_and_or_b32 %0, %6, %7, %8\n v_and_or_b32 %1, %6, %8, %9\n v_and_or_b32 %2, %6, %9, %7\n v_and_or_b32 %3, %7, %8, %9\n")
KERNEL(k_cvt_pk_bf16_f32,"v_cvt_pk_bf16_f32 %0, %10, %11\n v_cvt_pk_bf16_f32 %1, %10, %11\n v_cvt_pk_bf16_f32 %2, %10, %11\n v_cvt_pk_bf16_f32 %3, %10, %11\n")
KERNEL(k_lshlrev_b32, "v_lshlrev_b32 %0, 16, %6\n v_lshlrev_b32 %1, 16, %7\n v_lshlrev_b32 %2, 16, %8\n v_lshlrev_b32 %3, 16, %9\n")
KERNEL(k_and_literal, "v_and_b32 %0, 0xffff0000, %6\n v_and_b32 %1, 0xffff0000, %7\n v_and_b32 %2, 0xffff0000, %8\n v_and_b32 %3, 0xffff0000, %9\n")
KERNEL(k_pk_maximum3_f16, "v_pk_maximum3_f16 %0, %6, %7, %0\n v_pk_maximum3_f16 %1, %6, %8, %1\n v_pk_maximum3_f16 %2, %6, %9, %2\n v_pk_maximum3_f16 %3, %7, %8, %3\n")
KERNEL(k_mov_b32, "v_mov_b32 %0, %6\n v_mov_b32 %1, %7\n v_mov_b32 %2, %8\n v_mov_b32 %3, %9\n")
KERNEL(k_fma_mixlo_f16, "v_fma_mixlo_f16 %0, %10, %11, %10\n v_fma_mixlo_f16 %1, %10, %11, %11\n v_fma_mixlo_f16 %2, %11, %11, %10\n v_fma_mixlo_f16 %3, %10, %10, %11\n")
KERNEL(k_add_u32, "v_add_u32 %0, %6, %7\n v_add_u32 %1, %6, %8\n v_add_u32 %2, %6, %9\n v_add_u32 %3, %7, %8\n")
KERNEL(k_cvt_f32_bf16_insn, "v_cvt_f32_bf16 %0, %6\n v_cvt_f32_bf16 %1, %7\n v_cvt_f32_bf16 %2, %8\n v_cvt_f32_bf16 %3, %9\n")
KERNEL(k_cmp_only, "v_cmp_lt_f32 vcc, %10, %11\n v_cmp_lt_f32 vcc, %11, %10\n v_cmp_lt_u32 vcc, %6, %7\n v_cmp_lt_u32 vcc, %8, %9\n")
KERNEL(k_cndmask_only, "v_cndmask_b32 %0, %6, %7, vcc\n v_cndmask_b32 %1, %6, %8, vcc\n v_cndmask_b32 %2, %6, %9, vcc\n v_cndmask_b32 %3, %7, %8, vcc\n")
KERNEL(k_min_f16, "v_min_f16 %0, %6, %7\n v_min_f16 %1, %6, %8\n v_min_f16 %2, %6, %9\n v_min_f16 %3, %7, %8\n")
KERNEL(k_max_u32_dpp, "v_max_u32_dpp %0, %6, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %1, %7, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %2, %8, %2 row_mirror row_mask:0xf bank_mask:0xf\n v_max_u32_dpp %3, %9, %3 row_half_mirror row_mask:0xf bank_mask:0xf\n")
KERNEL(k_fmac_f32, "v_fmac_f32 %0, %10, %11\n v_fmac_f32 %1, %10, %11\n v_fmac_f32 %2, %10, %11\n v_fmac_f32 %3, %10, %11\n")
KERNEL(k_mul_f16, "v_mul_f16 %0, %6, %7\n v_mul_f16 %1, %6, %8\n v_mul_f16 %2, %6, %9\n v_mul_f16 %3, %7, %8\n")
KERNEL(k_salu,      "s_add_u32 s20, s20, 1\n s_add_u32 s21, s21, 1\n s_add_u32 s22, s22, 1\n s_add_u32 s23, s23, 1\n")

typedef void (*kfn)(uint32_t*, int);
struct Ent { const char* name; kfn fn; };
#define E(n) {#n, n}
int main() {
  Ent ents[] = {E(k_fma_f32), E(k_mul_f32), E(k_pk_fma_f32), E(k_pk_mul_f32), E(k_cvt_f32_f16), E(k_cvt_f32_f16_sdwa), E(k_cvt_f16_f32),
                E(k_cvt_pk_f16_f32), E(k_pk_add_f16), E(k_pk_mul_f16), E(k_pk_fma_f16), E(k_pk_max_u16), E(k_pk_min_f16), E(k_and_b32),
                E(k_bfi_b32), E(k_perm_b32), E(k_rndne_f16), E(k_fma_mix), E(k_mov_dpp), E(k_add_f64), E(k_fma_f64), E(k_cvt_f64_f32),
                E(k_rcp_f32), E(k_cvt_i32_f32), E(k_cmp_cndmask), E(k_med3_f16), E(k_lshl_add_u64), E(k_pk_add_u16), E(k_add_f16),
                E(k_cvt_f32_bf16), E(k_dot2_f32_f16), E(k_dot2c_f32_f16), E(k_max3_f16_abs), E(k_fma_mix_abs), E(k_add_f32_abs), E(k_pk_add_f32), E(k_max_f16), E(k_pk_max_f16), E(k_max3_u32), E(k_and_or_b32), E(k_cvt_pk_bf16_f32), E(k_lshlrev_b32), E(k_and_literal), E(k_pk_maximum3_f16), E(k_mov_b32), E(k_fma_mixlo_f16), E(k_add_u32), E(k_cvt_f32_bf16_insn), E(k_cmp_only), E(k_cndmask_only), E(k_min_f16), E(k_max_u32_dpp), E(k_fmac_f32), E(k_mul_f16), E(k_salu)};
  uint32_t* out; hipMalloc(&out, 256 * 8 * 256 * 4 * 2);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 2000;
  for (int occ : {4}) {
    printf("--- %d waves/SIMD (blocks/CU = %d)\n", occ, occ);
    for (auto& en : ents) {
      int grid = 256 * occ;  // 256 CUs x occ blocks of 4 waves -> occ waves per SIMD
      hipLaunchKernelGGL(en.fn, dim3(grid), dim3(256), 0, 0, out, 10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(en.fn, dim3(grid), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double insts_per_simd = (double)iters * 64 * occ;  // wave-instructions per SIMD
      printf("%-22s %8.3f ms  -> %6.2f ns/inst/SIMD = %5.2f cyc @2.4GHz\n", en.name, ms, ms * 1e6 / insts_per_simd, ms * 1e6 / insts_per_simd * 2.4);
    }
  }
  return 0;
}
