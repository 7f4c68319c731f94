// jacobi.hip -- one Jacobi sweep of the pressure Poisson problem (gfx950).
// Replaces generic/tfluids.cu:1765-1821 (kernel) -- the reference has no CPU version
// (generic/tfluids.cc:836-839). The iteration loop lives in abi.cpp (:1853-1921 semantics).
// Algorithmic bytes: p_prev, flags, div -> p = 16 B/cell/iteration; HBM/L2-bound.
#include <atomic>
#include <cstdlib>

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

// ---- small 2-D grids: the WHOLE solve in one launch (round 4, late) -------------------------------------------------------
// BASELINE config 1 (64 x 64, 20 iterations) spent its step in 20 dependent launches of a kernel that moves 64 KB. Up to
// 16 K cells the two pressure buffers fit the LDS of one CU (2 x 64 KB): one block of 1024 threads per batch item keeps the
// ping-pong there, a barrier stands where a launch stood, and the flags / divergence of a thread's cells are read ONCE into
// registers. Per-cell arithmetic and iteration order are those of k_jacobi (bit-equal pressure; the residual of the last
// iteration is summed in fp64 by one block instead of by atomics over many: equal to rounding, as before).
constexpr int kJLdsThreads = 1024, kJLdsMaxCells = kJLdsThreads * 16;
template <int kJLdsCpt>      // cells per thread: 1, 4 (64 x 64) or 16 (128 x 128)
__global__ __launch_bounds__(kJLdsThreads) void k_jacobi_lds(int Y, int X, const float* __restrict__ flags,
                                                              const float* __restrict__ div, float* __restrict__ p,
                                                              float* __restrict__ p_prev, int iters,
                                                              double* __restrict__ resid_sq) {
  extern __shared__ float jbuf[];            // [2][N]
  const int N = Y * X, b = blockIdx.x, tid = threadIdx.x;
  flags += (long long)b * N; div += (long long)b * N; p += (long long)b * N; p_prev += (long long)b * N;
  float dv[kJLdsCpt];
  int bits[kJLdsCpt];    // bit 0-3: the -x / +x / -y / +y neighbour is an obstacle (reads the cell itself); bit 4: the cell is updated
#pragma unroll
  for (int q = 0; q < kJLdsCpt; q++) {
    const int c = tid + q * kJLdsThreads;
    const bool in = c < N;
    const int cc = in ? c : 0;
    const int i = cc % X, j = cc / X;
    const bool inner = in && i >= 1 && i <= X - 2 && j >= 1 && j <= Y - 2;
    const int sx = inner ? 1 : 0, sy = inner ? X : 0;
    const int fc = (int)flags[cc];
    const int f1 = (int)flags[cc - sx], f2 = (int)flags[cc + sx], f3 = (int)flags[cc - sy], f4 = (int)flags[cc + sy];
    dv[q] = div[cc];
    // an obstacle neighbour contributes the cell's own previous pressure (generic/tfluids.cu:1797-1810)
    bits[q] = ((f1 & kObstacle) ? 1 : 0) | ((f2 & kObstacle) ? 2 : 0) | ((f3 & kObstacle) ? 4 : 0) | ((f4 & kObstacle) ? 8 : 0) |
              ((inner && !(fc & kObstacle)) ? 16 : 0);
    if (in) { jbuf[c] = 0.0f; jbuf[N + c] = 0.0f; }     // generic/tfluids.cu:1869-1872: both buffers start at zero
  }
  __syncthreads();
  double e2 = 0.0;
  for (int it = 0; it < iters; it++) {
    const float* prev = jbuf + (it & 1) * N;
    float* cur = jbuf + ((it & 1) ^ 1) * N;
    const bool last = it + 1 == iters;
#pragma unroll
    for (int q = 0; q < kJLdsCpt; q++) {
      const int c = tid + q * kJLdsThreads;
      if (c >= N) continue;
      const int m = bits[q];
      const bool upd_cell = (m & 16) != 0;            // only updated cells have all four neighbours
      const int sx = upd_cell ? 1 : 0, sy = upd_cell ? X : 0;
      const float p1 = prev[(m & 1) ? c : c - sx], p2 = prev[(m & 2) ? c : c + sx];
      const float p3 = prev[(m & 4) ? c : c - sy], p4 = prev[(m & 8) ? c : c + sy];
      const float upd = (p1 + p2 + p3 + p4 + 0.0f + 0.0f + dv[q]) / 4.0f;
      const float out = upd_cell ? upd : 0.0f;
      cur[c] = out;
      if (last) {
        const float pv = prev[c];
        p[c] = out; p_prev[c] = pv;
        if (resid_sq) { const double e = (double)out - (double)pv; e2 += e * e; }
      }
    }
    __syncthreads();
  }
  if (resid_sq) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) e2 += __shfl_down(e2, off, 64);
    __shared__ double part[kJLdsThreads / 64];
    if ((tid & 63) == 0) part[tid >> 6] = e2;
    __syncthreads();
    if (tid == 0) {
      double s = 0.0;
      for (int w = 0; w < kJLdsThreads / 64; w++) s += part[w];
      resid_sq[b] = s;
    }
  }
}

// true = the solve ran (p and resid_sq[b] written); false = shape / settings outside this path (caller iterates launches)
bool jacobi_solve_lds(hipStream_t st, int B, int Y, int X, const float* flags, const float* div, float* p, float* p_prev,
                      int iters, double* resid_sq) {
  static const bool off = getenv("TFL_JACOBI_LDS") && atoi(getenv("TFL_JACOBI_LDS")) == 0;
  const long long N = (long long)Y * X;
  if (off || N > kJLdsMaxCells || N < 1 || iters < 1) return false;
  const size_t lds = sizeof(float) * 2 * (size_t)N;
  if (lds > 48 * 1024) {
    static std::atomic<int> attr_dev_mask[2];          // per-device: the dynamic-LDS limit of the kernel has been raised
    int dev = 0; (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64) return false;
    const int word = dev >> 5, bit = 1 << (dev & 31);
    if (!(attr_dev_mask[word].load() & bit)) {
      if (hipFuncSetAttribute((const void*)k_jacobi_lds<16>, hipFuncAttributeMaxDynamicSharedMemorySize, kJLdsMaxCells * 8) != hipSuccess) return false;
      attr_dev_mask[word].fetch_or(bit);
    }
  }
  TFL_TIMED("k_jacobi_lds", st);
  if (N <= kJLdsThreads) k_jacobi_lds<1><<<B, kJLdsThreads, lds, st>>>(Y, X, flags, div, p, p_prev, iters, resid_sq);
  else if (N <= 4 * kJLdsThreads) k_jacobi_lds<4><<<B, kJLdsThreads, lds, st>>>(Y, X, flags, div, p, p_prev, iters, resid_sq);
  else k_jacobi_lds<16><<<B, kJLdsThreads, lds, st>>>(Y, X, flags, div, p, p_prev, iters, resid_sq);
  return true;
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
