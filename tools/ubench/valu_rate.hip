// Issue cost of individual VALU instructions on gfx950 (development micro-benchmark for the advection kernels,
// which are VALU-issue bound: SQ_INSTS_VALU x 4 cycles x waves/SIMD accounts for ~70 % of k_vel_bwd's duration).
// Each kernel runs 8 independent dependency chains of ONE instruction, 4 waves per SIMD, whole chip; the printed
// number is cycles per wave64 instruction per SIMD at the clock measured with s_memtime-free arithmetic:
// cycles = elapsed * f_clk / (instructions per SIMD), with f_clk calibrated from v_add_u32 = 4 cycles.
// build: hipcc --offload-arch=gfx950 -O3 valu_rate.hip -o valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>

#define CHAIN8(ASM32)                                                                   \
  for (int it = 0; it < iters; it++) {                                                  \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                     \
      _Pragma("unroll") for (int q = 0; q < 8; q++) { ASM32; }                          \
    }                                                                                   \
  }

enum Op { ADD_U32, MUL_LO_U32, MUL_U24, MAD_U24, LSHL_ADD_U64, CVT_I32_F32, CVT_F32_I32, MIN3_F32, MIN_I32, FMA_F32,
          PK_FMA_F32, MUL_F32, FMA_F64, MUL_F64, ADD_F64, CVT_F64_F32, CVT_F32_F64, RCP_F32, SQRT_F32, DIV_SCALE,
          DIV_FMAS, DIV_FIXUP, CNDMASK, CMP_F32, ADD3_U32, ASHR, LSHL_ADD_U32, ADD_F32, ADD_F32_E64, PK_ADD_F32, PK_MUL_F32, MED3_F32, MAX_F32, FMAC_F32, TRUNC_F32, FLOOR_F32, RSQ_F32, MAD_U64_U32, CMP_F32_E64, CNDMASK_E64, SUB_F32_SGPR, MUL_F32_LIT, AND_B32, MAD_I32_I24, FMA_F32_SGPR2, MIX_ADD_CVT, MIX_MUL_FMA, MIX_ADD_MUL_AND, MIX_PKMUL_CVT, MIX_4, MOV_B32, NOPS };

template <int OP>
__global__ __launch_bounds__(256) void k(float* out, int iters, float fa, int ia) {
  float f[8]; int i[8]; double dd[8]; unsigned long long l[8]; float2 p[8];
  for (int q = 0; q < 8; q++) { f[q] = threadIdx.x + q + 1.5f; i[q] = threadIdx.x + q; dd[q] = f[q]; l[q] = i[q]; p[q] = make_float2(f[q], f[q]); }
  const double da = fa;
  const unsigned long long la = ia;
  if (OP == ADD_U32) CHAIN8(asm volatile("v_add_u32 %0, %0, %1" : "+v"(i[q]) : "v"(ia)))
  if (OP == MUL_LO_U32) CHAIN8(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(i[q]) : "v"(ia)))
  if (OP == MUL_U24) CHAIN8(asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(i[q]) : "v"(ia)))
  if (OP == MAD_U24) CHAIN8(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(i[q]) : "v"(ia)))
  if (OP == LSHL_ADD_U64) CHAIN8(asm volatile("v_lshl_add_u64 %0, %0, 2, %1" : "+v"(l[q]) : "v"(la)))
  if (OP == CVT_I32_F32) CHAIN8(asm volatile("v_cvt_i32_f32 %0, %1" : "=v"(i[q]) : "v"(f[q])))
  if (OP == CVT_F32_I32) CHAIN8(asm volatile("v_cvt_f32_i32 %0, %1" : "=v"(f[q]) : "v"(i[q])))
  if (OP == MIN3_F32) CHAIN8(asm volatile("v_min3_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(fa)))
  if (OP == MIN_I32) CHAIN8(asm volatile("v_min_i32 %0, %0, %1" : "+v"(i[q]) : "v"(ia)))
  if (OP == FMA_F32) CHAIN8(asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(fa)))
  if (OP == PK_FMA_F32) CHAIN8(asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[q]) : "v"(p[(q + 1) & 7])))
  if (OP == MUL_F32) CHAIN8(asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[q]) : "v"(fa)))
  if (OP == FMA_F64) CHAIN8(asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(dd[q]) : "v"(da)))
  if (OP == MUL_F64) CHAIN8(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(dd[q]) : "v"(da)))
  if (OP == ADD_F64) CHAIN8(asm volatile("v_add_f64 %0, %0, %1" : "+v"(dd[q]) : "v"(da)))
  if (OP == CVT_F64_F32) CHAIN8(asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(dd[q]) : "v"(f[q])))
  if (OP == CVT_F32_F64) CHAIN8(asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[q]) : "v"(dd[q])))
  if (OP == RCP_F32) CHAIN8(asm volatile("v_rcp_f32 %0, %0" : "+v"(f[q])))
  if (OP == SQRT_F32) CHAIN8(asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[q])))
  if (OP == DIV_SCALE) CHAIN8(asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(f[q]) : "v"(fa) : "vcc"))
  if (OP == DIV_FMAS) CHAIN8(asm volatile("v_div_fmas_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(fa) : "vcc"))
  if (OP == DIV_FIXUP) CHAIN8(asm volatile("v_div_fixup_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(fa)))
  if (OP == CNDMASK) CHAIN8(asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[q]) : "v"(fa) : "vcc"))
  if (OP == CMP_F32) CHAIN8(asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(f[q]), "v"(fa) : "vcc"))
  if (OP == ADD3_U32) CHAIN8(asm volatile("v_add3_u32 %0, %0, %1, %1" : "+v"(i[q]) : "v"(ia)))
  if (OP == ASHR) CHAIN8(asm volatile("v_ashrrev_i32 %0, 31, %0" : "+v"(i[q])))
  if (OP == LSHL_ADD_U32) CHAIN8(asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(i[q]) : "v"(ia)))
  if (OP == ADD_F32) CHAIN8(asm volatile("v_add_f32_e32 %0, %0, %1" : "+v"(f[q]) : "v"(fa)))
  if (OP == ADD_F32_E64) CHAIN8(asm volatile("v_add_f32_e64 %0, %0, %1" : "+v"(f[q]) : "v"(fa)))
  if (OP == PK_ADD_F32) CHAIN8(asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[q]) : "v"(p[(q + 1) & 7])))
  if (OP == PK_MUL_F32) CHAIN8(asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[q]) : "v"(p[(q + 1) & 7])))
  if (OP == MED3_F32) CHAIN8(asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(f[q]) : "v"(fa)))
  if (OP == MAX_F32) CHAIN8(asm volatile("v_max_f32_e32 %0, %0, %1" : "+v"(f[q]) : "v"(fa)))
  if (OP == FMAC_F32) CHAIN8(asm volatile("v_fmac_f32_e32 %0, %1, %1" : "+v"(f[q]) : "v"(fa)))
  if (OP == TRUNC_F32) CHAIN8(asm volatile("v_trunc_f32_e32 %0, %0" : "+v"(f[q])))
  if (OP == FLOOR_F32) CHAIN8(asm volatile("v_floor_f32_e32 %0, %0" : "+v"(f[q])))
  if (OP == RSQ_F32) CHAIN8(asm volatile("v_rsq_f32_e32 %0, %0" : "+v"(f[q])))
  if (OP == MAD_U64_U32) CHAIN8(asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(l[q]) : "v"(ia) : "vcc"))
  if (OP == CMP_F32_E64) CHAIN8(asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(f[q]), "v"(fa) : "s20", "s21"))
  if (OP == CNDMASK_E64) CHAIN8(asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(f[q]) : "v"(fa)))
  if (OP == SUB_F32_SGPR) CHAIN8(asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(f[q]) : "s"(fa)))
  if (OP == MUL_F32_LIT) CHAIN8(asm volatile("v_mul_f32_e32 %0, 0x3e800000, %0" : "+v"(f[q])))
  if (OP == AND_B32) CHAIN8(asm volatile("v_and_b32_e32 %0, %0, %1" : "+v"(i[q]) : "v"(ia)))
  if (OP == MAD_I32_I24) CHAIN8(asm volatile("v_mad_i32_i24 %0, %0, %1, %1" : "+v"(i[q]) : "v"(ia)))
  if (OP == FMA_F32_SGPR2) CHAIN8(asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[q]) : "s"(fa)))
  if (OP == MIX_ADD_CVT) CHAIN8(asm volatile("v_add_f32_e32 %0, %0, %2\n v_cvt_i32_f32_e32 %1, %0" : "+v"(f[q]), "=v"(i[q]) : "v"(fa)))
  if (OP == MIX_MUL_FMA) CHAIN8(asm volatile("v_mul_f32_e32 %0, %0, %2\n v_fma_f32 %1, %1, %2, %2" : "+v"(f[q]), "+v"(p[q].x) : "v"(fa)))
  if (OP == MIX_ADD_MUL_AND) CHAIN8(asm volatile("v_add_f32_e32 %0, %0, %2\n v_mul_f32_e32 %1, %1, %2\n v_and_b32_e32 %3, %3, %4" : "+v"(f[q]), "+v"(p[q].x), "+v"(fa), "+v"(i[q]) : "v"(ia)))
  if (OP == MIX_PKMUL_CVT) CHAIN8(asm volatile("v_pk_mul_f32 %0, %0, %2\n v_cvt_i32_f32_e32 %1, %3" : "+v"(p[q]), "=v"(i[q]) : "v"(p[(q + 1) & 7]), "v"(f[q])))
  if (OP == MIX_4) CHAIN8(asm volatile("v_add_f32_e32 %0, %0, %2\n v_cvt_i32_f32_e32 %1, %0\n v_mul_f32_e32 %3, %3, %2\n v_fma_f32 %4, %4, %2, %2" : "+v"(f[q]), "=v"(i[q]), "+v"(fa), "+v"(p[q].x), "+v"(p[q].y)))
  if (OP == MOV_B32) CHAIN8(asm volatile("v_mov_b32_e32 %0, %1" : "=v"(f[q]) : "v"(p[q].x)))
  float s = 0;
  for (int q = 0; q < 8; q++) s += f[q] + i[q] + (float)dd[q] + (float)l[q] + p[q].x + p[q].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static double g_ns_add = 0;   // ns per v_add_u32 per SIMD -> defines "4 cycles"

template <int OP>
void run(float* d, const char* name) {
  const int blocks = 256 * 4, iters = 4000;          // 4 waves per SIMD
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<blocks, 256>>>(d, 10, 1.0f, 3);
  hipEventRecord(e0);
  k<OP><<<blocks, 256>>>(d, iters, 1.0f, 3);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double per_simd = 4.0 * iters * 32;          // wave-instructions issued by one SIMD
  const double ns = ms * 1e6 / per_simd;
  if (OP == ADD_U32) g_ns_add = ns;
  printf("%-16s %7.3f ms  %6.3f ns/instr/SIMD  = %5.2f cycles (v_add_u32 := 4)\n", name, ms, ns, 4.0 * ns / g_ns_add);
}

int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
#define R(x) run<x>(d, #x);
  R(ADD_U32) R(ADD_U32) R(MUL_LO_U32) R(MUL_U24) R(MAD_U24) R(LSHL_ADD_U64) R(LSHL_ADD_U32) R(ADD3_U32) R(ASHR) R(CVT_I32_F32) R(CVT_F32_I32)
  R(MIN3_F32) R(MIN_I32) R(MUL_F32) R(FMA_F32) R(PK_FMA_F32) R(FMA_F64) R(MUL_F64) R(ADD_F64) R(CVT_F64_F32) R(CVT_F32_F64)
  R(RCP_F32) R(SQRT_F32) R(DIV_SCALE) R(DIV_FMAS) R(DIV_FIXUP) R(CNDMASK) R(CMP_F32)
  R(ADD_F32) R(ADD_F32_E64) R(PK_ADD_F32) R(PK_MUL_F32) R(MED3_F32) R(MAX_F32) R(FMAC_F32) R(TRUNC_F32) R(FLOOR_F32) R(RSQ_F32) R(MAD_U64_U32) R(CMP_F32_E64) R(CNDMASK_E64) R(SUB_F32_SGPR) R(MUL_F32_LIT) R(AND_B32) R(MAD_I32_I24) R(FMA_F32_SGPR2)
  printf("mixes: cycles per GROUP (2, 2, 3, 2, 4 instructions)\n");
  R(MIX_ADD_CVT) R(MIX_MUL_FMA) R(MIX_ADD_MUL_AND) R(MIX_PKMUL_CVT) R(MIX_4) R(MOV_B32)
  return 0;
}
