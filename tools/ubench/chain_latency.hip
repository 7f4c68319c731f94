// Latency of one DEPENDENT instruction chain on gfx950 (development micro-benchmark for the PCG sweeps, whose step is a chain
// DPP shift -> add -> add -> mul with an LDS exchange + barrier every few steps). One workgroup, W waves, each wave runs
// `iters` rounds of the chain; time / (iters * ops per round) = clocks per dependent instruction as one wave sees them.
// build: hipcc --offload-arch=gfx950 -O3 chain_latency.hip -o chain_latency
#include <hip/hip_runtime.h>
#include <cstdio>

enum Mode { ADD4, DPP_WAVE_ADD3, DPP_ROW_ADD3, STEP, STEP_LDS2, STEP_LDS2_BAR, LDS_ONLY, BAR_ONLY, STEP_BPERM, VMEM_LD, VMEM_LDST, VMEM_ST };

template <int MODE>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float fa, const float4* gin = nullptr, float4* gout = nullptr) {
  __shared__ float ring[16][2][64][2];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float q = fa + lane, c = 1.0f + fa * 1e-6f, r = fa * 0.5f;
  const int nb = w > 0 ? w - 1 : 0;
  ring[w][0][lane][0] = ring[w][0][lane][1] = ring[w][1][lane][0] = ring[w][1][lane][1] = 0.0f;
  __syncthreads();
  for (int it = 0; it < iters; it++) {
    if (MODE == ADD4) {
#pragma unroll
      for (int u = 0; u < 8; u++) asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_mul_f32 %0, %0, %2" : "+v"(q) : "v"(r), "v"(c));
    }
    if (MODE == DPP_WAVE_ADD3 || MODE == DPP_ROW_ADD3 || MODE == STEP || MODE == STEP_BPERM) {
#pragma unroll
      for (int u = 0; u < 8; u++) {
        float n;
        if (MODE == DPP_ROW_ADD3) n = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x111, 0xf, 0xf, true));
        else if (MODE == STEP_BPERM) n = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + 63) & 63) * 4, __builtin_bit_cast(int, q)));
        else n = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x138, 0xf, 0xf, false));
        asm volatile("" : "+v"(n));
        q = ((r + n) + q) * c;
        asm volatile("" : "+v"(q));
      }
    }
    if (MODE == STEP_LDS2 || MODE == STEP_LDS2_BAR) {
#pragma unroll
      for (int e = 0; e < 4; e++) {
        const float2 below = *reinterpret_cast<const float2*>(&ring[nb][(e & 1) ^ 1][lane][0]);
        float h[2];
#pragma unroll
        for (int v = 0; v < 2; v++) {
          float n = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x138, 0xf, 0xf, false));
          asm volatile("" : "+v"(n));
          q = (((r + (v ? below.y : below.x)) + n) + q) * c;
          h[v] = q;
        }
        *reinterpret_cast<float2*>(&ring[w][e & 1][lane][0]) = make_float2(h[0], h[1]);
        if (MODE == STEP_LDS2_BAR) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
    if (MODE == VMEM_LD || MODE == VMEM_LDST || MODE == VMEM_ST) {
      // the sweep's group of four steps: 2 exchanges of 2 steps, then one 16-byte store and two 16-byte loads (prefetched 4 groups ahead)
      static_assert(true, "");
      float4 pa[4], pb[4];
      if (it == 0) {
#pragma unroll
        for (int g2 = 0; g2 < 4; g2++) { pa[g2] = gin[(long long)(w * 4096 + g2) * 64 + lane]; pb[g2] = gin[(long long)((w + 16) * 4096 + g2) * 64 + lane]; }
      }
#pragma unroll
      for (int g2 = 0; g2 < 4; g2++) {
        const float rr[4] = {pa[g2].x, pa[g2].y, pa[g2].z, pa[g2].w}, cc[4] = {pb[g2].x, pb[g2].y, pb[g2].z, pb[g2].w};
        float h[4];
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const float2 below = *reinterpret_cast<const float2*>(&ring[nb][(e & 1) ^ 1][lane][0]);
#pragma unroll
          for (int v = 0; v < 2; v++) {
            float n = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, q), 0x138, 0xf, 0xf, false));
            asm volatile("" : "+v"(n));
            q = (((rr[e * 2 + v] + (v ? below.y : below.x)) + n) + q) * cc[e * 2 + v];
            h[e * 2 + v] = q;
          }
          *reinterpret_cast<float2*>(&ring[w][e & 1][lane][0]) = make_float2(h[e * 2], h[e * 2 + 1]);
          asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        const long long gi = (long long)(w * 4096 + ((it * 4 + g2) & 4095)) * 64 + lane;
        if (MODE != VMEM_LD) gout[gi] = make_float4(h[0], h[1], h[2], h[3]);
        if (MODE != VMEM_ST) {
          const long long gn = (long long)(w * 4096 + ((it * 4 + g2 + 4) & 4095)) * 64 + lane;
          pa[g2] = gin[gn]; pb[g2] = gin[gn + 16ll * 4096 * 64];
        }
      }
    }
    if (MODE == LDS_ONLY) {
#pragma unroll
      for (int e = 0; e < 8; e++) {
        ring[w][0][lane][0] = q;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        q = ring[nb][0][lane][0] + r;
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q) :: "memory");
      }
    }
    if (MODE == BAR_ONLY) {
#pragma unroll
      for (int e = 0; e < 8; e++) { asm volatile("s_barrier" ::: "memory"); q += r; }
    }
  }
  out[threadIdx.x] = q + nw;
}

template <int MODE>
void run(float* d, const char* name, int waves, double per_iter, const char* unit) {
  const int iters = 20000;
  static float4 *gin = nullptr, *gout = nullptr;
  if (!gin) { hipMalloc(&gin, 32ll * 4096 * 64 * 16); hipMalloc(&gout, 16ll * 4096 * 64 * 16); hipMemset(gin, 0, 32ll * 4096 * 64 * 16); }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<MODE><<<1, waves * 64>>>(d, 100, 1.0f, gin, gout);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<1, waves * 64>>>(d, iters, 1.0f, gin, gout);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("%-44s %2d waves: %7.1f ns per %s\n", name, waves, ms * 1e6 / (iters * per_iter), unit);
}

int main() {
  float* d; hipMalloc(&d, 4096 * 4);
  run<ADD4>(d, "add, add, add, mul (dependent)", 1, 32, "instruction");
  run<DPP_ROW_ADD3>(d, "step with dpp row_shr:1", 1, 8, "step (dpp, add, add, mul)");
  run<DPP_WAVE_ADD3>(d, "step with dpp wave_shr:1", 1, 8, "step (dpp, add, add, mul)");
  run<STEP_BPERM>(d, "step with ds_bpermute", 1, 8, "step");
  run<DPP_WAVE_ADD3>(d, "step with dpp wave_shr:1", 8, 8, "step (dpp, add, add, mul)");
  run<LDS_ONLY>(d, "LDS write -> wait -> read -> wait", 1, 8, "round trip");
  run<LDS_ONLY>(d, "LDS write -> wait -> read -> wait", 8, 8, "round trip");
  run<BAR_ONLY>(d, "s_barrier", 2, 8, "barrier");
  run<BAR_ONLY>(d, "s_barrier", 8, 8, "barrier");
  run<BAR_ONLY>(d, "s_barrier", 16, 8, "barrier");
  run<STEP_LDS2>(d, "2 steps + LDS exchange, no barrier", 8, 8, "step");
  run<STEP_LDS2_BAR>(d, "2 steps + LDS exchange + barrier", 2, 8, "step");
  run<STEP_LDS2_BAR>(d, "2 steps + LDS exchange + barrier", 8, 8, "step");
  run<STEP_LDS2_BAR>(d, "2 steps + LDS exchange + barrier", 16, 8, "step");
  run<VMEM_ST>(d, "sweep group, + 16-byte store / group", 8, 16, "step");
  run<VMEM_LD>(d, "sweep group, + two 16-byte loads / group", 8, 16, "step");
  run<VMEM_LDST>(d, "sweep group, + loads and store", 2, 16, "step");
  run<VMEM_LDST>(d, "sweep group, + loads and store", 8, 16, "step");
  run<VMEM_LDST>(d, "sweep group, + loads and store", 16, 16, "step");
  return 0;
}
