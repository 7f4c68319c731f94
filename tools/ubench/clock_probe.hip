// Which clock does the chip sustain under a full-chip v_pk_fma_f32 stream? s_memtime (clock64) against the 100 MHz
// s_memrealtime (wall_clock64) inside the kernel, plus hipEvent time for the instruction-rate view.
// build: hipcc --offload-arch=gfx950 -O3 -w clock_probe.hip -o clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* clk, int iters, float fa) {
  float2 p[8];
  for (int q = 0; q < 8; q++) p[q] = make_float2(threadIdx.x + q, threadIdx.x - q);
  float2 w = make_float2(fa, fa * 0.5f);
  const float2 sw = make_float2(__builtin_amdgcn_readfirstlane(fa), __builtin_amdgcn_readfirstlane(fa * 0.25f));
  const long long c0 = clock64(), r0 = wall_clock64();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 4; u++)
#pragma unroll
      for (int q = 0; q < 8; q++) {
        if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[q]) : "v"(w));
        if (MODE == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(p[q].x) : "v"(w.x));
        if (MODE == 2) asm volatile("s_nop 3");
        if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[q]) : "v"(w), "s"(sw));
        if (MODE == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p[q]) : "v"(w), "s"(sw));
        if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(p[q]) : "v"(w), "v"(w));
      }
  }
  const long long c1 = clock64(), r1 = wall_clock64();
  float s = 0;
  for (int q = 0; q < 8; q++) s += p[q].x + p[q].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = r1 - r0; }
}
template <int MODE>
void run(const char* name, float* d, long long* dc, int wps) {
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<256 * wps, 256>>>(d, dc, 10, 1.0f);
  hipEventRecord(e0);
  k<MODE><<<256 * wps, 256>>>(d, dc, iters, 1.0f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, dc, 16, hipMemcpyDeviceToHost);
  const double per_simd = (double)wps * iters * 32;
  printf("%-10s waves/SIMD %d: %.3f ms; s_memtime/s_memrealtime = %.3f (x100 MHz = %.0f MHz if s_memtime is the core clock); %.3f ns per instr per SIMD\n",
         name, wps, ms, (double)h[0] / h[1], 100.0 * h[0] / h[1], ms * 1e6 / per_simd);
}
int main() {
  float* d; long long* dc; hipMalloc(&d, 256 * 8 * 256 * 4); hipMalloc(&dc, 16);
  for (int w : {1, 4, 8}) { run<0>("pk_fma", d, dc, w); run<1>("add_u32", d, dc, w); run<2>("s_nop", d, dc, w); run<3>("pk_sgpr_bc", d, dc, w); run<4>("pk_sgpr", d, dc, w); run<5>("pk_vgpr_bc", d, dc, w); }
  return 0;
}
