// Issue cost (cycles per wave-instruction on one SIMD) of the VALU instructions the chain leans on, measured with s_memtime
// around unrolled runs of 8 independent chains per instruction, one wavefront per SIMD.  Decides what is worth replacing.
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
#define CHAINS 8
#define TIME_ASM(NAME, DECL, BODY)                                                                  \
  __global__ void NAME(long long* out, double seed) {                                               \
    DECL;                                                                                           \
    long long t0 = __builtin_readcyclecounter();                                                    \
    for (int it = 0; it < 64; it++) {                                                               \
      _Pragma("unroll") for (int r = 0; r < REP; r++) { BODY; }                                      \
    }                                                                                               \
    long long t1 = __builtin_readcyclecounter();                                                    \
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;                                                \
    SINK;                                                                                           \
  }
#define D8 double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; double c = seed * 0.5
#define F8 float f0 = seed, f1 = seed + 1, f2 = seed + 2, f3 = seed + 3, f4 = seed + 4, f5 = seed + 5, f6 = seed + 6, f7 = seed + 7; float cf = seed * 0.5f
#define SINK if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[1000] = 1
#define EACH_D(OP) OP(a0) OP(a1) OP(a2) OP(a3) OP(a4) OP(a5) OP(a6) OP(a7)
#define ASM_D(INSTR) asm volatile(INSTR : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c))
TIME_ASM(k_fma64, D8, ASM_D("v_fma_f64 %0, %0, %8, %8\n v_fma_f64 %1, %1, %8, %8\n v_fma_f64 %2, %2, %8, %8\n v_fma_f64 %3, %3, %8, %8\n v_fma_f64 %4, %4, %8, %8\n v_fma_f64 %5, %5, %8, %8\n v_fma_f64 %6, %6, %8, %8\n v_fma_f64 %7, %7, %8, %8"))
TIME_ASM(k_mul64, D8, ASM_D("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8"))
TIME_ASM(k_add64, D8, ASM_D("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8"))
TIME_ASM(k_max64, D8, ASM_D("v_max_f64 %0, %0, %8\n v_max_f64 %1, %1, %8\n v_max_f64 %2, %2, %8\n v_max_f64 %3, %3, %8\n v_max_f64 %4, %4, %8\n v_max_f64 %5, %5, %8\n v_max_f64 %6, %6, %8\n v_max_f64 %7, %7, %8"))
TIME_ASM(k_rcp64, D8, ASM_D("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7"))
TIME_ASM(k_rsq64, D8, ASM_D("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7"))
TIME_ASM(k_rndne64, D8, ASM_D("v_rndne_f64 %0, %0\n v_rndne_f64 %1, %1\n v_rndne_f64 %2, %2\n v_rndne_f64 %3, %3\n v_rndne_f64 %4, %4\n v_rndne_f64 %5, %5\n v_rndne_f64 %6, %6\n v_rndne_f64 %7, %7"))
TIME_ASM(k_ldexp64, D8, ASM_D("v_ldexp_f64 %0, %0, 1\n v_ldexp_f64 %1, %1, 1\n v_ldexp_f64 %2, %2, 1\n v_ldexp_f64 %3, %3, 1\n v_ldexp_f64 %4, %4, 1\n v_ldexp_f64 %5, %5, 1\n v_ldexp_f64 %6, %6, 1\n v_ldexp_f64 %7, %7, 1"))
TIME_ASM(k_cmp64, D8, ASM_D("v_cmp_gt_f64 vcc, %0, %8\n v_cmp_gt_f64 vcc, %1, %8\n v_cmp_gt_f64 vcc, %2, %8\n v_cmp_gt_f64 vcc, %3, %8\n v_cmp_gt_f64 vcc, %4, %8\n v_cmp_gt_f64 vcc, %5, %8\n v_cmp_gt_f64 vcc, %6, %8\n v_cmp_gt_f64 vcc, %7, %8"))
// conversions: f64 -> f32 -> f64 round trip on each chain (2 instructions per chain)
#define CV(i, j) "v_cvt_f32_f64 %" #j ", %" #i "\n v_cvt_f64_f32 %" #i ", %" #j "\n"
#define CV1(i, j) "v_cvt_f32_f64 %" #j ", %" #i "\n"
#define G8 float g0, g1, g2, g3, g4, g5, g6, g7
TIME_ASM(k_cvt, D8; G8, asm volatile(CV(0, 8) CV(1, 9) CV(2, 10) CV(3, 11) CV(4, 12) CV(5, 13) CV(6, 14) CV(7, 15)
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3), "=v"(g4), "=v"(g5), "=v"(g6), "=v"(g7)))
TIME_ASM(k_cvt_to32, D8; G8, asm volatile(CV1(0, 8) CV1(1, 9) CV1(2, 10) CV1(3, 11) CV1(4, 12) CV1(5, 13) CV1(6, 14) CV1(7, 15)
                                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=v"(g0), "=v"(g1), "=v"(g2), "=v"(g3), "=v"(g4), "=v"(g5), "=v"(g6), "=v"(g7)))
#undef SINK
#define SINK if (f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 == 12345.678f) out[1000] = 1
#define ASM_F(INSTR) asm volatile(INSTR : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf))
TIME_ASM(k_movdpp, F8, asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n"
                                    "v_mov_b32_dpp %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_mov_b32_dpp %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)))
TIME_ASM(k_mov32, F8, asm volatile("v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8"
                                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf)))
TIME_ASM(k_cndmask, F8, asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc"
                                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf) : "vcc"))
TIME_ASM(k_cndmask_sgpr, F8, asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[10:11]\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]"
                                 : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf) : "s10", "s11"))
#define H4 float h0 = 1; float h1 = 2; float h2 = 3; float h3 = 4
TIME_ASM(k_cndmask_indep, F8; H4, asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc"
                                 : "=v"(h0), "=v"(h1), "=v"(h2), "=v"(h3) : "v"(f0), "v"(cf) : "vcc"); f1 += h0 + h1 + h2 + h3)
TIME_ASM(k_max32, F8, ASM_F("v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8"))
TIME_ASM(k_cmpsel_vcc, F8, asm volatile("v_cmp_gt_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\nv_cmp_gt_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\nv_cmp_gt_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\nv_cmp_gt_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\nv_cmp_gt_f32 vcc, %4, %8\n v_cndmask_b32 %4, %4, %8, vcc\nv_cmp_gt_f32 vcc, %5, %8\n v_cndmask_b32 %5, %5, %8, vcc\nv_cmp_gt_f32 vcc, %6, %8\n v_cndmask_b32 %6, %6, %8, vcc\nv_cmp_gt_f32 vcc, %7, %8\n v_cndmask_b32 %7, %7, %8, vcc\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf) : "vcc"))
TIME_ASM(k_cmpsel_sgpr, F8, asm volatile("v_cmp_gt_f32_e64 s[10:11], %0, %8\n v_cndmask_b32_e64 %0, %0, %8, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %1, %8\n v_cndmask_b32_e64 %1, %1, %8, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %2, %8\n v_cndmask_b32_e64 %2, %2, %8, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %3, %8\n v_cndmask_b32_e64 %3, %3, %8, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %4, %8\n v_cndmask_b32_e64 %4, %4, %8, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %5, %8\n v_cndmask_b32_e64 %5, %5, %8, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %6, %8\n v_cndmask_b32_e64 %6, %6, %8, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %7, %8\n v_cndmask_b32_e64 %7, %7, %8, s[10:11]\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf) : "s10", "s11"))
TIME_ASM(k_cmp32, F8, asm volatile("v_cmp_gt_f32 vcc, %0, %8\nv_cmp_gt_f32 vcc, %1, %8\nv_cmp_gt_f32 vcc, %2, %8\nv_cmp_gt_f32 vcc, %3, %8\nv_cmp_gt_f32 vcc, %4, %8\nv_cmp_gt_f32 vcc, %5, %8\nv_cmp_gt_f32 vcc, %6, %8\nv_cmp_gt_f32 vcc, %7, %8\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(cf) : "vcc"))
TIME_ASM(k_fma32, F8, ASM_F("v_fma_f32 %0, %0, %8, %8\n v_fma_f32 %1, %1, %8, %8\n v_fma_f32 %2, %2, %8, %8\n v_fma_f32 %3, %3, %8, %8\n v_fma_f32 %4, %4, %8, %8\n v_fma_f32 %5, %5, %8, %8\n v_fma_f32 %6, %6, %8, %8\n v_fma_f32 %7, %7, %8, %8"))
TIME_ASM(k_rcp32, F8, ASM_F("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7"))

#define G8F float g0 = 1, g1 = 2, g2 = 3, g3 = 4, g4 = 5, g5 = 6, g6 = 7, g7 = 8
TIME_ASM(k_sel2_vcc, F8; G8F, asm volatile("v_cmp_gt_f32 vcc, %0, %16\n v_cndmask_b32 %0, %0, %16, vcc\n v_cndmask_b32 %8, %8, %16, vcc\nv_cmp_gt_f32 vcc, %1, %16\n v_cndmask_b32 %1, %1, %16, vcc\n v_cndmask_b32 %9, %9, %16, vcc\nv_cmp_gt_f32 vcc, %2, %16\n v_cndmask_b32 %2, %2, %16, vcc\n v_cndmask_b32 %10, %10, %16, vcc\nv_cmp_gt_f32 vcc, %3, %16\n v_cndmask_b32 %3, %3, %16, vcc\n v_cndmask_b32 %11, %11, %16, vcc\nv_cmp_gt_f32 vcc, %4, %16\n v_cndmask_b32 %4, %4, %16, vcc\n v_cndmask_b32 %12, %12, %16, vcc\nv_cmp_gt_f32 vcc, %5, %16\n v_cndmask_b32 %5, %5, %16, vcc\n v_cndmask_b32 %13, %13, %16, vcc\nv_cmp_gt_f32 vcc, %6, %16\n v_cndmask_b32 %6, %6, %16, vcc\n v_cndmask_b32 %14, %14, %16, vcc\nv_cmp_gt_f32 vcc, %7, %16\n v_cndmask_b32 %7, %7, %16, vcc\n v_cndmask_b32 %15, %15, %16, vcc\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7) : "v"(cf) : "vcc"); f0 += g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7)
TIME_ASM(k_sel2_sgpr, F8; G8F, asm volatile("v_cmp_gt_f32_e64 s[10:11], %0, %16\n v_cndmask_b32_e64 %0, %0, %16, s[10:11]\n v_cndmask_b32_e64 %8, %8, %16, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %1, %16\n v_cndmask_b32_e64 %1, %1, %16, s[10:11]\n v_cndmask_b32_e64 %9, %9, %16, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %2, %16\n v_cndmask_b32_e64 %2, %2, %16, s[10:11]\n v_cndmask_b32_e64 %10, %10, %16, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %3, %16\n v_cndmask_b32_e64 %3, %3, %16, s[10:11]\n v_cndmask_b32_e64 %11, %11, %16, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %4, %16\n v_cndmask_b32_e64 %4, %4, %16, s[10:11]\n v_cndmask_b32_e64 %12, %12, %16, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %5, %16\n v_cndmask_b32_e64 %5, %5, %16, s[10:11]\n v_cndmask_b32_e64 %13, %13, %16, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %6, %16\n v_cndmask_b32_e64 %6, %6, %16, s[10:11]\n v_cndmask_b32_e64 %14, %14, %16, s[10:11]\nv_cmp_gt_f32_e64 s[10:11], %7, %16\n v_cndmask_b32_e64 %7, %7, %16, s[10:11]\n v_cndmask_b32_e64 %15, %15, %16, s[10:11]\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7) : "v"(cf) : "s10", "s11"); f0 += g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7)
TIME_ASM(k_sel_gap_vcc, F8; G8F, asm volatile("v_cmp_gt_f32 vcc, %0, %16\n v_add_f32 %8, %8, %16\n v_cndmask_b32 %0, %0, %16, vcc\nv_cmp_gt_f32 vcc, %1, %16\n v_add_f32 %9, %9, %16\n v_cndmask_b32 %1, %1, %16, vcc\nv_cmp_gt_f32 vcc, %2, %16\n v_add_f32 %10, %10, %16\n v_cndmask_b32 %2, %2, %16, vcc\nv_cmp_gt_f32 vcc, %3, %16\n v_add_f32 %11, %11, %16\n v_cndmask_b32 %3, %3, %16, vcc\nv_cmp_gt_f32 vcc, %4, %16\n v_add_f32 %12, %12, %16\n v_cndmask_b32 %4, %4, %16, vcc\nv_cmp_gt_f32 vcc, %5, %16\n v_add_f32 %13, %13, %16\n v_cndmask_b32 %5, %5, %16, vcc\nv_cmp_gt_f32 vcc, %6, %16\n v_add_f32 %14, %14, %16\n v_cndmask_b32 %6, %6, %16, vcc\nv_cmp_gt_f32 vcc, %7, %16\n v_add_f32 %15, %15, %16\n v_cndmask_b32 %7, %7, %16, vcc\n" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7), "+v"(g0), "+v"(g1), "+v"(g2), "+v"(g3), "+v"(g4), "+v"(g5), "+v"(g6), "+v"(g7) : "v"(cf) : "vcc"); f0 += g0 + g1 + g2 + g3 + g4 + g5 + g6 + g7)
int main() {
  long long* out; hipMallocManaged(&out, 8 * 2048);
  struct { const char* name; void (*k)(long long*, double); int per; } ks[] = {
    {"v_fma_f64", k_fma64, 8}, {"v_mul_f64", k_mul64, 8}, {"v_add_f64", k_add64, 8}, {"v_max_f64", k_max64, 8}, {"v_cmp_gt_f64", k_cmp64, 8},
    {"v_rcp_f64", k_rcp64, 8}, {"v_rsq_f64", k_rsq64, 8}, {"v_rndne_f64", k_rndne64, 8}, {"v_ldexp_f64", k_ldexp64, 8},
    {"v_cvt f64->f32->f64 (pair)", k_cvt, 16}, {"v_cvt_f32_f64", k_cvt_to32, 8}, {"v_mov_b32_dpp", k_movdpp, 8}, {"v_mov_b32", k_mov32, 8},
    {"v_cndmask_b32", k_cndmask, 8}, {"v_cndmask_b32_e64 sgpr", k_cndmask_sgpr, 8}, {"v_cndmask_b32 indep dst (+1 add)", k_cndmask_indep, 8}, {"v_max_f32", k_max32, 8}, {"cmp + cndmask via vcc", k_cmpsel_vcc, 16}, {"cmp + cndmask via sgpr pair", k_cmpsel_sgpr, 16}, {"v_cmp_gt_f32 vcc", k_cmp32, 8}, {"v_fma_f32", k_fma32, 8}, {"v_rcp_f32", k_rcp32, 8},
    {"cmp + 2 cndmask via vcc (64-bit select)", k_sel2_vcc, 24}, {"cmp + 2 cndmask via sgpr pair", k_sel2_sgpr, 24}, {"cmp, add, cndmask via vcc", k_sel_gap_vcc, 24} };
  for (auto& e : ks) {
    for (int waves = 1; waves <= 4; waves *= 2) {         // 1, 2 or 4 wavefronts per SIMD on one CU (256 / 512 / 1024 threads)
      hipLaunchKernelGGL(e.k, dim3(1), dim3(256 * waves), 0, 0, out, 1.25); hipDeviceSynchronize();
      hipLaunchKernelGGL(e.k, dim3(1), dim3(256 * waves), 0, 0, out, 1.25); hipDeviceSynchronize();
      const double n = 64.0 * REP * e.per;
      printf("%-28s %d wave(s)/SIMD: %6.2f clk per instruction per wave (s_memtime ticks)\n", e.name, waves, (double)out[0] / n);
    }
  }
  return 0;
}
