// Micro-benchmark: issue rate of VALU instruction CLASSES on gfx950 -- what the evaluating kernel's instruction mix costs.
// 16 independent chains per lane, 4 waves per SIMD (the leaf kernel's occupancy) and 8. Cycles per wave-instruction per SIMD are
// REAL shader-clock cycles: every wave brackets its loop with s_memtime (the shader clock counter) and s_memrealtime (a constant
// 100 MHz), the longest wave of the launch is what is reported, and the clock the loop actually ran at is printed beside it
// (round 4's table converted wall time at an assumed 2.4 GHz and read 2.66 "cycles" for v_fma_f32: the part runs VALU-dense
// code below its peak clock, see profiles/r5_valu_class_rates.txt). (leaf_eval_kernel on npt-flange: 47 % of its VALU instructions are f32 add / mul / fma, f64, conversions or integer
// arithmetic by the SQ_INSTS_VALU_* counters; the rest are moves, selects, compares, min / max, logic.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b, unsigned long long* tm) {
  float s[16];
  double d[8];
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p[8];
  for (int i = 0; i < 16; i++) s[i] = a * (float)(threadIdx.x + i + 1);
  for (int i = 0; i < 8; i++) { d[i] = (double)s[i]; p[i] = v2f{s[2 * i], s[2 * i + 1]}; }
  v2f aa = v2f{a, a}, bb = v2f{b, b};
  double da = a, db = b;
  const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it++) {
#define F(i)                                                                                                                  \
  if (OP == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));                                        \
  if (OP == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b));                                                    \
  if (OP == 2) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(a));                                                    \
  if (OP == 3) asm volatile("v_max_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b));                                                    \
  if (OP == 4) asm volatile("v_min_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b));                                                    \
  if (OP == 5) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));                                       \
  if (OP == 6) asm volatile("v_mov_b32 %0, %1" : "+v"(s[i]) : "v"(s[(i + 1) & 15]));                                          \
  if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(b) : "vcc");                                   \
  if (OP == 8) asm volatile("v_cmp_gt_f32 vcc, %0, %1" : : "v"(s[i]), "v"(b) : "vcc");                                        \
  if (OP == 9) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1" : : "v"(s[i]), "v"(b) : "s20", "s21");                            \
  if (OP == 10) asm volatile("v_and_b32 %0, %0, %1" : "+v"(s[i]) : "v"(b));                                                   \
  if (OP == 11) asm volatile("v_add_u32 %0, %0, %1" : "+v"(s[i]) : "v"(b));                                                   \
  if (OP == 12) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(s[i]));                                                         \
  if (OP == 13) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(s[i]) : "v"(b));                                                \
  if (OP == 14) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(s[i]));                                                            \
  if (OP == 15) asm volatile("v_rcp_f32 %0, %0" : "+v"(s[i]));                                                                \
  if (OP == 16) asm volatile("v_sqrt_f32 %0, %0" : "+v"(s[i]));                                                               \
  if (OP == 17) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(s[i]));                       \
  if (OP == 18) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(s[i]) : "s20");                                              \
  if (OP == 19) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));                                      \
  if (OP == 20) asm volatile("v_cmp_class_f32 vcc, %0, %1" : : "v"(s[i]), "v"(b) : "vcc");                                    \
  if (OP == 21) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(s[i]) : "v"(b) : "vcc");                            \
  if (OP == 22) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b) : "vcc");                          \
  if (OP == 23) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));                                 \
  if (OP == 24) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(s[i]));                                                          \
  if (OP == 25) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b));                                                   \
  if (OP == 26) asm volatile("v_fma_f32 %0, %0, s20, %1" : "+v"(s[i]) : "v"(b));       /* SGPR operand */                     \
  if (OP == 27) asm volatile("v_mul_f32 %0, 0x3f8ccccd, %0" : "+v"(s[i]));            /* literal operand */                 \
  if (OP == 28) asm volatile("v_add_f32 %0, |%0|, -%1" : "+v"(s[i]) : "v"(b));          /* VOP3 modifiers */                   \
  if (OP == 29) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(s[i]) : "v"(b));                                     \
  if (OP == 30) asm volatile("v_add_f32 %0, s20, %0" : "+v"(s[i]));                                                            \
  if (OP == 31) asm volatile("v_mul_f32 %0, s20, %0" : "+v"(s[i]));                                                            \
  if (OP == 32) asm volatile("v_max_f32 %0, s20, %0" : "+v"(s[i]));                                                            \
  if (OP == 33) asm volatile("v_max_f32 %0, 0x3f8ccccd, %0" : "+v"(s[i]));                                                     \
  if (OP == 34) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(s[i]) : "v"(a), "v"(b));                                           \
  if (OP == 35) asm volatile("v_fmac_f32 %0, s20, %1" : "+v"(s[i]) : "v"(b));                                                  \
  if (OP == 36) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3f8ccccd" : "+v"(s[i]) : "v"(a));                                      \
  if (OP == 37) asm volatile("v_fmamk_f32 %0, %0, 0x3f8ccccd, %1" : "+v"(s[i]) : "v"(b));                                      \
  if (OP == 38) asm volatile("v_fma_f32 %0, %0, 1.0, %1" : "+v"(s[i]) : "v"(b));                                               \
  if (OP == 39) asm volatile("v_cmp_gt_f32 vcc, s20, %0" : : "v"(s[i]) : "vcc");                                               \
  if (OP == 50) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(s[i]));                                                            \
  if (OP == 51) asm volatile("v_sub_f32 %0, %0, %1\n v_mul_f32 %0, %0, %1" : "+v"(s[i]) : "v"(b));                             \
  if (OP == 52) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(b));                                           \
  if (OP == 53) asm volatile("v_max_f32 %0, %0, %0" : "+v"(s[i]));                                                             \
  if (OP == 54) asm volatile("v_mul_f32 %0, 1.0, %0" : "+v"(s[i]));                                                            \
  if (OP == 55) asm volatile("v_xor_b32 %0, 0x80000000, %0" : "+v"(s[i]));                                                     \
  if (OP == 57) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(b) : "vcc");        \
  if (OP == 58) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(s[i]) : "v"(b) : "s20", "s21"); \
  if (OP == 59) asm volatile("v_cmp_gt_f32 s[20:21], %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(s[i]) : "v"(b) : "s20", "s21"); \
  if (OP == 60) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cmp_gt_f32 s[20:21], %0, %1 \n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(s[i]) : "v"(b) : "vcc", "s20", "s21"); \
  if (OP == 56) asm volatile("v_and_b32 %0, 0x7fffffff, %0" : "+v"(s[i]));
    REP16(F)
    REP16(F)
#undef F
#define G(i)                                                                                             \
  if (OP == 40) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[i & 7]) : "v"(da), "v"(db));            \
  if (OP == 41) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"(db));                         \
  if (OP == 42) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"(da));                         \
  if (OP == 43) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i & 7]) : "v"(aa), "v"(bb));         \
  if (OP == 44) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(aa));                      \
  if (OP == 45) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(bb));
    REP16(G)
    REP16(G)
#undef G
  }
  const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
  if ((threadIdx.x & 63) == 0) { atomicMax(&tm[0], c1 - c0); atomicMax(&tm[1], r1 - r0); }
  float r = 0;
  for (int i = 0; i < 16; i++) r += s[i];
  for (int i = 0; i < 8; i++) r += (float)d[i] + p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = r;
}

int main() {
  hipDeviceProp_t pr;
  (void)hipGetDeviceProperties(&pr, 0);
  const int cus = pr.multiProcessorCount, iters = 4000;
  float* dmem;
  (void)hipMalloc(&dmem, (size_t)cus * 8 * 256 * 4);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  struct Row { const char* name; void (*fn)(float*, int, float, float, unsigned long long*); };
  unsigned long long* dtm;
  (void)hipMalloc(&dtm, 16);
#define ROW(n, op) Row{n, k<op>}
  std::vector<Row> rows = {ROW("v_fma_f32", 0), ROW("v_add_f32", 1), ROW("v_sub_f32", 25), ROW("v_mul_f32", 2), ROW("v_max_f32", 3), ROW("v_min_f32", 4), ROW("v_med3_f32", 5),
                           ROW("v_max3_f32", 19), ROW("v_mov_b32", 6), ROW("v_cndmask_b32 vcc", 7), ROW("v_cndmask_b32 sgpr pair", 29), ROW("v_cmp_gt_f32 -> vcc", 8),
                           ROW("v_cmp_gt_f32 -> sgpr pair", 9), ROW("v_cmp_class_f32", 20), ROW("v_and_b32", 10), ROW("v_add_u32", 11), ROW("v_lshlrev_b32", 12),
                           ROW("v_bfe_u32", 24), ROW("v_mul_lo_u32", 13), ROW("v_cvt_f32_u32", 14), ROW("v_rcp_f32", 15), ROW("v_sqrt_f32", 16), ROW("v_mov_b32_dpp", 17),
                           ROW("v_readlane_b32", 18), ROW("v_div_scale_f32", 21), ROW("v_div_fmas_f32", 22), ROW("v_div_fixup_f32", 23), ROW("v_fma_f32 sgpr operand", 26),
                           ROW("v_mul_f32 literal operand", 27), ROW("v_add_f32 sgpr src0", 30), ROW("v_mul_f32 sgpr src0", 31), ROW("v_max_f32 sgpr src0", 32), ROW("v_max_f32 literal", 33),
                           ROW("v_fmac_f32 vgprs", 34), ROW("v_fmac_f32 sgpr src0", 35), ROW("v_fmaak_f32 literal", 36), ROW("v_fmamk_f32 literal", 37), ROW("v_fma_f32 inline 1.0", 38),
                           ROW("v_cmp_gt_f32 sgpr src0", 39), ROW("v_add_f32 inline 1.0", 50), ROW("v_sub+v_mul pair (x2)", 51), ROW("v_cndmask_b32 vcc (no clobber)", 52),
                           ROW("v_max_f32 x,x (canonicalise)", 53), ROW("v_mul_f32 1.0 (canonicalise)", 54), ROW("v_xor_b32 sign", 55), ROW("cmp->vcc + cndmask vcc (x2)", 57), ROW("cmp->sgpr + cndmask sgpr (x2)", 58), ROW("cmp->sgpr, s_nop 1, cndmask (x2+nop)", 59), ROW("cmp vcc, cmp sgpr, cndmask vcc (x3)", 60), ROW("v_and_b32 abs", 56), ROW("v_add_f32 |x|,-y (VOP3)", 28), ROW("v_fma_f64", 40), ROW("v_add_f64", 41), ROW("v_mul_f64", 42),
                           ROW("v_pk_fma_f32", 43), ROW("v_pk_mul_f32", 44), ROW("v_pk_add_f32", 45)};
  for (int wps : {4, 8}) {
    printf("---- %d waves per SIMD\n", wps);
    for (const Row& r : rows) {
      const int grid = cus * wps;
      float ms = 0;
      unsigned long long tm[2] = {0, 0};
      for (int rep = 0; rep < 2; rep++) {
        (void)hipMemset(dtm, 0, 16);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(r.fn, dim3(grid), dim3(256), 0, 0, dmem, iters, 1.0001f, 0.5f, dtm);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipMemcpy(tm, dtm, 16, hipMemcpyDeviceToHost);
      }
      const double per_simd = (double)wps * iters * 32;  // wave-instructions a SIMD issues: its waves x iterations x 32 per iteration
      const double ghz = tm[1] ? (double)tm[0] / ((double)tm[1] * 10.0) : 0.0;  // s_memrealtime ticks are 10 ns
      printf("%-28s %6.2f cycles per wave-instruction per SIMD (s_memtime)   clock %.2f GHz   [wall time at an assumed 2.4 GHz: %.2f]\n", r.name,
             (double)tm[0] / per_simd, ghz, ms * 1e6 / per_simd * 2.4);
    }
  }
  return 0;
}
