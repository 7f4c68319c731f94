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
    // all loads in ONE round trip (round 4: flags[o] first and the rest behind its test were two; a border cell reads its
    // own index instead of neighbours that do not exist -- 20 dependent launches per step make this a latency kernel)
    const bool inner = !on_border<IS3D>(d, i, j, k);
    const int sx = inner ? 1 : 0, sy = inner ? d.sy : 0, sz = (inner && IS3D) ? d.sz : 0;
    const int fc = (int)flags[o];
    const float c = pp[o], dv = div[o];
    float p1 = pp[o - sx], p2 = pp[o + sx], p3 = pp[o - sy], p4 = pp[o + sy];
    float p5 = IS3D ? pp[o - sz] : 0.0f, p6 = IS3D ? pp[o + sz] : 0.0f;
    const int f1 = (int)flags[o - sx], f2 = (int)flags[o + sx], f3 = (int)flags[o - sy], f4 = (int)flags[o + sy];
    const int f5 = IS3D ? (int)flags[o - sz] : 0, f6 = IS3D ? (int)flags[o + sz] : 0;
    // (the update itself is evaluated for every cell and selected at the end: behind a branch, hipcc sinks the loads into it)
    p1 = (f1 & kObstacle) ? c : p1;
    p2 = (f2 & kObstacle) ? c : p2;
    p3 = (f3 & kObstacle) ? c : p3;
    p4 = (f4 & kObstacle) ? c : p4;
    if (IS3D) {
      p5 = (f5 & kObstacle) ? c : p5;
      p6 = (f6 & kObstacle) ? c : p6;
    }
    const float upd = (p1 + p2 + p3 + p4 + p5 + p6 + dv) / (IS3D ? 6.0f : 4.0f);
    out = (inner && !(fc & kObstacle)) ? upd : 0.0f;
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
