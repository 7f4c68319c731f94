// Does the fp32 matrix pipe run beside the fp32 vector pipe on gfx950? (development micro-benchmark for the conv stack:
// fp32 MFMA and v_pk_fma_f32 both peak at 256 flop/clk/CU, so a kernel that could keep both busy would double the conv
// ceiling.) Three kernels, whole chip, 4 waves per SIMD: N x v_pk_fma_f32 (8 chains), N x MFMA (4 accumulators), and both
// in one loop. If the mixed loop takes max(a, b) the pipes overlap; if it takes a + b they share the issue slot.
// build: hipcc --offload-arch=gfx950 -O3 mfma_coissue.hip -o mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));

enum Mode { PK_ONLY, MFMA4_ONLY, MFMA32_ONLY, MFMA16_ONLY, MIX4_1, MIX4_2, MIX32_8, MIX32_16, MIX16_4, FMA_ONLY, MIX4_FMA2 };

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float fa) {
  f2 p[8];
  float s1[8];
  for (int q = 0; q < 8; q++) { p[q] = f2{threadIdx.x + q + 1.5f, threadIdx.x * 0.5f + q}; s1[q] = threadIdx.x + q; }
  f4 acc4[4];
  f16v acc32[2];
  for (int q = 0; q < 4; q++) acc4[q] = f4{fa, fa, fa, fa};
  for (int q = 0; q < 2; q++) for (int e = 0; e < 16; e++) acc32[q][e] = fa + e;
  const float a = fa + threadIdx.x, b = fa * 0.5f;
  for (int it = 0; it < iters; it++) {
    if (MODE == PK_ONLY) {
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int q = 0; q < 8; q++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[q]) : "v"(p[(q + 1) & 7]));
    }
    if (MODE == FMA_ONLY) {
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int q = 0; q < 8; q++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s1[q]) : "v"(fa));
    }
    if (MODE == MFMA4_ONLY || MODE == MIX4_1 || MODE == MIX4_2 || MODE == MIX4_FMA2) {
#pragma unroll
      for (int u = 0; u < 8; u++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          acc4[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc4[q], 0, 0, 0);
          if (MODE == MIX4_1) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[q]) : "v"(p[(q + 1) & 7]));
          if (MODE == MIX4_2) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[q]) : "v"(p[(q + 1) & 7]));
            asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[q + 4]) : "v"(p[(q + 5) & 7]));
          }
          if (MODE == MIX4_FMA2) {
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s1[q]) : "v"(fa));
            asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s1[q + 4]) : "v"(fa));
          }
        }
    }
    if (MODE == MFMA32_ONLY || MODE == MIX32_8 || MODE == MIX32_16) {
#pragma unroll
      for (int u = 0; u < 2; u++)
#pragma unroll
        for (int q = 0; q < 2; q++) {
          acc32[q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc32[q], 0, 0, 0);
          if (MODE == MIX32_8 || MODE == MIX32_16) {
#pragma unroll
            for (int r = 0; r < (MODE == MIX32_8 ? 1 : 2); r++)
#pragma unroll
              for (int e = 0; e < 8; e++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[e]) : "v"(p[(e + 1) & 7]));
          }
        }
    }
    if (MODE == MFMA16_ONLY || MODE == MIX16_4) {
#pragma unroll
      for (int u = 0; u < 4; u++)
#pragma unroll
        for (int q = 0; q < 4; q++) {
          acc4[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[q], 0, 0, 0);
          if (MODE == MIX16_4) {
#pragma unroll
            for (int e = 0; e < 4; e++) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[e + 4 * (q & 1)]) : "v"(p[(e + 1) & 7]));
          }
        }
    }
  }
  float s = 0;
  for (int q = 0; q < 8; q++) s += p[q].x + p[q].y + s1[q];
  for (int q = 0; q < 4; q++) s += acc4[q].x + acc4[q].y + acc4[q].z + acc4[q].w;
  for (int q = 0; q < 2; q++) for (int e = 0; e < 16; e++) s += acc32[q][e];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(float* d, const char* name, double mfma_per_iter, double mfma_flops, double pk_per_iter, double fma_per_iter = 0) {
  const int iters = 4096, blocks = 256 * 4;   // 4 blocks of 4 waves per CU -> 4 waves per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(d, 64, 1.0f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(d, iters, 1.0f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double waves = blocks * 4.0;
  const double flops = waves * iters * (mfma_per_iter * mfma_flops + pk_per_iter * 256 + fma_per_iter * 128);
  printf("%-34s %8.3f ms  %7.1f TFLOP/s   (per wave-iteration: %g mfma, %g pk_fma, %g fma)\n", name, ms, flops / ms * 1e-9,
         mfma_per_iter, pk_per_iter, fma_per_iter);
}

int main() {
  float* d; hipMalloc(&d, 256 * 4 * 256 * 4);
  run<PK_ONLY>(d, "v_pk_fma_f32 only", 0, 0, 32);
  run<FMA_ONLY>(d, "v_fma_f32 only", 0, 0, 0, 32);
  run<MFMA4_ONLY>(d, "mfma 4x4x1 only", 32, 512, 0);
  run<MIX4_1>(d, "mfma 4x4x1 + 1 pk_fma each", 32, 512, 32);
  run<MIX4_2>(d, "mfma 4x4x1 + 2 pk_fma each", 32, 512, 64);
  run<MIX4_FMA2>(d, "mfma 4x4x1 + 2 v_fma each", 32, 512, 0, 64);
  run<MFMA16_ONLY>(d, "mfma 16x16x4 only", 16, 2048, 0);
  run<MIX16_4>(d, "mfma 16x16x4 + 4 pk_fma each", 16, 2048, 64);
  run<MFMA32_ONLY>(d, "mfma 32x32x2 only", 4, 4096, 0);
  run<MIX32_8>(d, "mfma 32x32x2 + 8 pk_fma each", 4, 4096, 32);
  run<MIX32_16>(d, "mfma 32x32x2 + 16 pk_fma each", 4, 4096, 64);
  return 0;
}
