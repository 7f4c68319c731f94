// Can a VALU stream of v_fma_f32 with a scalar (SGPR) weight operand and LDS-fed data reach the fp32 vector peak on
// gfx950? (development micro-benchmark for the direct conv kernel; hipcc --offload-arch=gfx950 -O3 valu_conv.hip)
#include <hip/hip_runtime.h>
#include <cstdio>

// per wave: 64 lanes = 64 voxels, V rows each, 8 output channels: acc[V][8]; per (c, tap-row): V+2 row values from LDS
template <int V>
__global__ __launch_bounds__(256) void k(float* out, const float* __restrict__ w, int iters) {
  __shared__ float tile[8 * 10 * 72];
  for (int t = threadIdx.x; t < 8 * 10 * 72; t += 256) tile[t] = t * 1e-4f;
  __syncthreads();
  float acc[V][8];
  for (int v = 0; v < V; v++) for (int co = 0; co < 8; co++) acc[v][co] = 0.f;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int it = 0; it < iters; it++) {
    const float* wp = w;
#pragma unroll 1
    for (int c = 0; c < 8; c++) {
#pragma unroll
      for (int dz = 0; dz < 3; dz++) {
        float row[V + 2][3];
#pragma unroll
        for (int r = 0; r < V + 2; r++)
#pragma unroll
          for (int dx = 0; dx < 3; dx++) row[r][dx] = tile[(c * 10 + ((wave + dz + r) % 10)) * 72 + lane + dx];
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
          for (int dx = 0; dx < 3; dx++) {
#pragma unroll
            for (int co = 0; co < 8; co++) {
              const float wv = wp[((dz * 3 + dy) * 3 + dx) * 8 + co];     // uniform address -> s_load
#pragma unroll
              for (int v = 0; v < V; v++) acc[v][co] = __builtin_fmaf(row[v + dy][dx], wv, acc[v][co]);
            }
          }
      }
      wp += 27 * 8;
    }
  }
  float s = 0;
  for (int v = 0; v < V; v++) for (int co = 0; co < 8; co++) s += acc[v][co];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int V>
void run(float* d, const float* w, int blocks_per_cu) {
  const int blocks = 256 * blocks_per_cu, iters = 40;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  k<V><<<blocks, 256>>>(d, w, 2);
  (void)hipEventRecord(e0);
  k<V><<<blocks, 256>>>(d, w, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  const double fma = (double)blocks * 256 * iters * 8.0 * 27 * 8 * V;
  printf("V %d  blocks/CU %d: %.3f ms  %.1f TFLOP/s\n", V, blocks_per_cu, ms, 2 * fma / (ms * 1e-3) / 1e12);
}

int main() {
  float *d, *w; (void)hipMalloc(&d, 256 * 8 * 256 * 4); (void)hipMalloc(&w, 8 * 27 * 8 * 4); (void)hipMemset(w, 0, 8 * 27 * 8 * 4);
  for (int b : {2, 4, 8}) { run<2>(d, w, b); run<4>(d, w, b); run<6>(d, w, b); }
  return 0;
}
