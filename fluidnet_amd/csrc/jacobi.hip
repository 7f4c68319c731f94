// jacobi.hip -- one Jacobi sweep of the pressure Poisson problem (gfx950).
// Replaces generic/tfluids.cu:1765-1821 (kernel) -- the reference has no CPU version
// (generic/tfluids.cc:836-839). The iteration loop lives in abi.cpp (:1853-1921 semantics).
// Algorithmic bytes: p_prev, flags, div -> p = 16 B/cell/iteration; HBM/L2-bound.
#include "tfl_device.hpp"
#include "tfl_host.hpp"

namespace tfl {

template <bool IS3D, bool RESID>
__global__ __launch_bounds__(256) void k_jacobi(Dom d, const float* __restrict__ pp, const float* __restrict__ flags,
                                                const float* __restrict__ div, float* __restrict__ p,
                                                double* __restrict__ resid_sq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  const long long cells = d.sc;
  double e2 = 0.0;
  if (i < d.X && j < d.Y) {
    pp += b * cells; flags += b * cells; div += b * cells; p += b * cells;
    const int o = TFL_AT(d, i, j, k);
    float out = 0.0f;
    const int fc = (int)flags[o];
    if (!on_border<IS3D>(d, i, j, k) && !(fc & kObstacle)) {
      const float c = pp[o];
      float p1 = pp[o - 1], p2 = pp[o + 1], p3 = pp[o - d.sy], p4 = pp[o + d.sy];
      float p5 = IS3D ? pp[o - d.sz] : 0.0f, p6 = IS3D ? pp[o + d.sz] : 0.0f;
      if (((int)flags[o - 1]) & kObstacle) p1 = c;
      if (((int)flags[o + 1]) & kObstacle) p2 = c;
      if (((int)flags[o - d.sy]) & kObstacle) p3 = c;
      if (((int)flags[o + d.sy]) & kObstacle) p4 = c;
      if (IS3D) {
        if (((int)flags[o - d.sz]) & kObstacle) p5 = c;
        if (((int)flags[o + d.sz]) & kObstacle) p6 = c;
      }
      out = (p1 + p2 + p3 + p4 + p5 + p6 + div[o]) / (IS3D ? 6.0f : 4.0f);
    }
    p[o] = out;
    if (RESID) { const double e = (double)out - (double)pp[o]; e2 = e * e; }
  }
  if (RESID) {
    // wave64 shuffle reduce, then one LDS slot per wave, then one atomic per block
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e2 += __shfl_down(e2, off, 64);
    __shared__ double part[4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if ((tid & 63) == 0) part[tid >> 6] = e2;
    __syncthreads();
    if (tid == 0) atomicAdd(&resid_sq[b], part[0] + part[1] + part[2] + part[3]);
  }
}

void jacobi_iteration(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* p_prev, const float* flags,
                      const float* div, float* p, double* resid_sq) {
  const Dom d = make_dom(Z, Y, X);
  const dim3 blk(64, 4, 1), grd((X + 63) / 64, (Y + 3) / 4, (unsigned)(zwin_planes(Z) * B));
  if (is3d) {
    if (resid_sq) { TFL_TIMED("k_jacobi", st); k_jacobi<true, true><<<grd, blk, 0, st>>>(d, p_prev, flags, div, p, resid_sq); }
    else { TFL_TIMED("k_jacobi", st); k_jacobi<true, false><<<grd, blk, 0, st>>>(d, p_prev, flags, div, p, nullptr); }
  } else {
    if (resid_sq) { TFL_TIMED("k_jacobi", st); k_jacobi<false, true><<<grd, blk, 0, st>>>(d, p_prev, flags, div, p, resid_sq); }
    else { TFL_TIMED("k_jacobi", st); k_jacobi<false, false><<<grd, blk, 0, st>>>(d, p_prev, flags, div, p, nullptr); }
  }
}

}  // namespace tfl
