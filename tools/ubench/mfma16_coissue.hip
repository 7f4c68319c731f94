// Does the 16-bit-input matrix pipe run beside the vector ALUs on gfx950 -- inside one wave, and between the waves of one SIMD?
// (round 5: the conv kernels of conv_mfma16.hip behave as if their MFMAs, vector instructions and LDS reads simply ADD.)
// Whole chip, W waves per SIMD; per mode the time of `iters` loop bodies:
//   M     : 16 x v_mfma_f32_16x16x32_f16 on 4 accumulators                 (the matrix pipe alone)
//   V n   : n x v_fma_f32 on 8 independent chains                          (the vector pipe alone)
//   MV n  : the 16 MFMAs with n v_fma interleaved evenly, SAME wave
//   SPLIT n: waves alternate roles by wave id: even waves run M, odd waves run V n   (matrix || vector between waves of a SIMD)
// build: hipcc --offload-arch=gfx950 -O3 mfma16_coissue.hip -o mfma16_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NV>
__device__ __forceinline__ void valu_block(float (&s)[8], float fa) {
#pragma unroll
  for (int q = 0; q < NV; q++) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(s[q & 7]) : "v"(fa));
}

template <int MODE, int NV>     // MODE 0 = M, 1 = V, 2 = MV (same wave), 3 = SPLIT
__global__ __launch_bounds__(256) void k(float* out, int iters, float fa, int wps) {
  float s[8];
  for (int q = 0; q < 8; q++) s[q] = threadIdx.x + q;
  f4 acc[4];
  for (int q = 0; q < 4; q++) acc[q] = f4{fa, fa, fa, fa};
  h8 a, b;
  for (int e = 0; e < 8; e++) { a[e] = (_Float16)(fa + e); b[e] = (_Float16)(fa * 0.5f + threadIdx.x); }
  // SPLIT: roles by ROUND of resident blocks (blockIdx / CUs): a block's four waves sit on the four SIMDs of one CU, and a CU
  // holds one block of each round -- so every SIMD hosts matrix waves and vector waves side by side
  const int round = (int)(blockIdx.x / (gridDim.x / wps));
  const bool do_m = MODE == 0 || MODE == 2 || (MODE == 3 && (round & 1) == 0);
  const bool do_v = MODE == 1 || (MODE == 3 && (round & 1) == 1);
  for (int it = 0; it < iters; it++) {
    if (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 16; u++) {
        acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
        valu_block<NV / 16>(s, fa);
      }
    } else {
      if (do_m) {
#pragma unroll
        for (int u = 0; u < 16; u++) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
      }
      if (do_v) valu_block<NV>(s, fa);
    }
  }
  float r = 0.0f;
  for (int q = 0; q < 8; q++) r += s[q];
  for (int q = 0; q < 4; q++) r += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <int MODE, int NV>
static float run(float* d_out, int blocks, int iters, int wps) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE, NV><<<blocks, 256>>>(d_out, 10, 1.0f, wps);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE, NV><<<blocks, 256>>>(d_out, iters, 1.0f, wps);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main(int argc, char** argv) {
  int cus = 256; hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
  const int iters = 20000;
  float* d_out; hipMalloc(&d_out, sizeof(float) * 256 * cus * 8);
  for (int wps : {1, 2, 4}) {                       // waves per SIMD (a block = 4 waves = one per SIMD)
    const int blocks = cus * wps;
    printf("== %d wave(s) per SIMD, %d blocks, %d iterations, clock attr %.2f GHz\n", wps, blocks, iters, clk_khz / 1e6);
    auto rep = [&](const char* name, float ms, int n_mfma_waves, int nv) {
      printf("  %-10s %8.3f ms   = %7.1f ns per body  (%s)\n", name, ms, ms * 1e6 / iters, "");
      (void)n_mfma_waves; (void)nv;
    };
    rep("M", run<0, 0>(d_out, blocks, iters, wps), 0, 0);
    rep("V 32", run<1, 32>(d_out, blocks, iters, wps), 0, 0);
    rep("V 64", run<1, 64>(d_out, blocks, iters, wps), 0, 0);
    rep("V 128", run<1, 128>(d_out, blocks, iters, wps), 0, 0);
    rep("MV 32", run<2, 32>(d_out, blocks, iters, wps), 0, 0);
    rep("MV 64", run<2, 64>(d_out, blocks, iters, wps), 0, 0);
    rep("MV 128", run<2, 128>(d_out, blocks, iters, wps), 0, 0);
    if (wps >= 2) {
      rep("SPLIT 32", run<3, 32>(d_out, blocks, iters, wps), 0, 0);
      rep("SPLIT 64", run<3, 64>(d_out, blocks, iters, wps), 0, 0);
      rep("SPLIT 128", run<3, 128>(d_out, blocks, iters, wps), 0, 0);
    }
  }
  return 0;
}
