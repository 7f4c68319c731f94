// How many independent v_fma_f32 can ride between back-to-back v_mfma_f32_16x16x4_f32 on gfx950 before the MFMA rate
// drops? (development micro-benchmark for the conv kernels; build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  f32x4 acc[8];
  for (int r = 0; r < 8; r++) acc[r] = f32x4{0, 0, 0, 0};
  float v[16];
  for (int q = 0; q < 16; q++) v[q] = threadIdx.x + q;
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int r = 0; r < 8; r++) {
      acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[r], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV; q++) v[(r * NV + q) % 16] = __builtin_fmaf(v[(r * NV + q) % 16], a, b);
    }
  }
  float s = 0;
  for (int r = 0; r < 8; r++) s += acc[r][0] + acc[r][1] + acc[r][2] + acc[r][3];
  for (int q = 0; q < 16; q++) s += v[q];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV>
void run(float* d, int waves_per_simd) {
  const int blocks = 256 * waves_per_simd, iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NV><<<blocks, 256>>>(d, 10, 1.0f, 0.5f);
  hipEventRecord(e0);
  k<NV><<<blocks, 256>>>(d, iters, 1.0f, 0.5f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double mfma = (double)blocks * 4 * iters * 8;           // wave-level MFMAs
  const double tf = mfma * 2048 / (ms * 1e-3) / 1e12, vtf = mfma * NV * 128 / (ms * 1e-3) / 1e12;
  printf("waves/SIMD %d  v_fma per MFMA %2d: %.3f ms  MFMA %.1f TF  VALU %.1f TF  sum %.1f TF\n", waves_per_simd, NV, ms, tf, vtf, tf + vtf);
}

int main() {
  float* d; hipMalloc(&d, 256 * 8 * 256 * 4);
  for (int w : {1, 2, 4}) { run<0>(d, w); run<2>(d, w); run<4>(d, w); run<6>(d, w); run<8>(d, w); run<12>(d, w); run<16>(d, w); }
  return 0;
}
