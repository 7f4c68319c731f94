// pcg.hip -- the reference's baseline pressure solver, solveLinearSystemPCG (gfx950).
//
// Replaces tfluids_CudaMain_solveLinearSystemPCG (generic/tfluids.cu:864-1759) and its helpers
// findConnectedFluidComponents (generic/find_connected_fluid_components.cc), createReducedSystemIndices /
// setupLaplacian (generic/tfluids.cu:864-1093), which build a CSR matrix on the CPU for every call and run CG
// through cuSPARSE / cuBLAS with a host read-back after every dot product.
//
// Here the solver is matrix-free and stays on the device:
//   * connected fluid components: union-find label propagation on the grid (hook + compress sweeps); a
//     component is named by its smallest cell index, so components come out in the reference's scan order;
//   * A = the 5/7-point Laplacian of setupLaplacian, applied straight from the flag grid: diagonal = number of
//     non-obstacle neighbours, -1 for every fluid neighbour; vectors are grid-shaped, masked by the component;
//   * CG is Golub & Van Loan alg. 10.3.1 exactly as the reference codes it (same update order, same
//     clampToEpsilon guards, same `r.r > tol^2 && iter <= maxIter` loop), with alpha / beta / the residual living
//     in a device struct: the host enqueues a chunk of iterations and reads one flag back per chunk; kernels of
//     iterations past convergence see `done` and return;
//   * preconditioners: for this stencil ILU(0) and IC(0) touch only the diagonal (every fill-in position is
//     outside the pattern): d(n) = a(n,n) - sum over lower fluid neighbours m of 1/d(m), in lexicographic order.
//     ilu0: L = unit lower with l(n,m) = -1/d(m), U = upper part of A with d on the diagonal.
//     ic0:  R upper with R(n,n) = sqrt(d(n)), R(n,q) = -1/R(n,n); M = R^T R.
//     Cells of one hyperplane i+j+k = const are independent (what cusparse's csrsv level analysis discovers at run
//     time). The factorisation (once per solve) runs one launch per hyperplane; the two triangular solves of every CG
//     iteration run as PIPELINED WAVEFRONTS on 3-D grids -- two launches instead of ~760 at 128^3 (k_wf_sweep below). 2-D grids keep the
//     launch-per-hyperplane sweeps; 3-D grids with more than 240 sub-boxes run the wavefronts one range of slabs per launch.
// Dot products are two-stage fp64 reductions with a fixed order (bit-reproducible run to run).
#include "tfl_device.hpp"
#include "tfl_host.hpp"
#include "tfl_vec4.hpp"

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <limits>
#include <type_traits>
#include <vector>

namespace tfl {

namespace {

constexpr int kRedBlocks = 512;    // partial sums per dot product
constexpr int kMaxComponents = 4096;

// Device-resident solver state of one component solve.
struct PcgState {
  double rr1, rr0;         // current / previous ||r||^2
  double num, prev_num;    // r.z of this / the previous iteration (preconditioned form)
  double den;              // s.A s
  float alpha, beta;
  float tol2;
  int iter, max_iter, done, first, bad;   // bad: a NaN residual was seen
  int pending;                            // an update's r.r partials have not been folded into rr1 yet
  double sum_x;
};

__device__ __forceinline__ float clamp_to_epsilon(float v) {   // generic/tfluids.cu:1203-1214
  const float eps = 1.17549435e-38f;
  if (fabsf(v) < eps) return v < 0.0f ? fminf(v, -eps) : fmaxf(v, eps);
  return v;
}

__device__ __forceinline__ bool cell_fluid(const float* __restrict__ flags, int o) { return (((int)flags[o]) & kFluid) != 0; }
__device__ __forceinline__ bool cell_obstacle(const float* __restrict__ flags, int o) { return (((int)flags[o]) & kObstacle) != 0; }

// ---- connected components -------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cc_init(Dom d, bool is3d, const float* __restrict__ flags, int* __restrict__ label,
                                                 int* __restrict__ border_fluid) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= d.sc) return;
  const bool f = cell_fluid(flags, o);
  label[o] = f ? o : -1;
  if (f) {
    const int i = o % d.X, j = (o / d.X) % d.Y, k = o / (d.X * d.Y);
    if (i < 1 || i > d.X - 2 || j < 1 || j > d.Y - 2 || (is3d && (k < 1 || k > d.Z - 2))) atomicAdd(border_fluid, 1);
  }
}

// hook: a cell that sees a smaller label next door pulls its own label AND its current root down to it
__global__ __launch_bounds__(256) void k_cc_hook(Dom d, bool is3d, int* __restrict__ label, int* __restrict__ changed) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= d.sc) return;
  const int l = label[o];
  if (l < 0) return;
  const int i = o % d.X, j = (o / d.X) % d.Y, k = o / (d.X * d.Y);
  int m = l;
  auto look = [&](int n) { const int ln = label[n]; if (ln >= 0 && ln < m) m = ln; };
  if (i > 0) look(o - 1);
  if (i < d.X - 1) look(o + 1);
  if (j > 0) look(o - d.sy);
  if (j < d.Y - 1) look(o + d.sy);
  if (is3d && k > 0) look(o - d.sz);
  if (is3d && k < d.Z - 1) look(o + d.sz);
  if (m < l) {
    atomicMin(&label[l], m);
    atomicMin(&label[o], m);
    *changed = 1;
  }
}
// compress: point every cell at the root of its tree
__global__ __launch_bounds__(256) void k_cc_compress(Dom d, int* __restrict__ label) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= d.sc) return;
  int l = label[o];
  if (l < 0) return;
  while (true) { const int p = label[l]; if (p == l) break; l = p; }
  label[o] = l;
}
// roots[0..count) = the cells that name a component; size_at[root] = cells in it
__global__ __launch_bounds__(256) void k_cc_count(Dom d, const int* __restrict__ label, int* __restrict__ size_at,
                                                  int* __restrict__ roots, int* __restrict__ count) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= d.sc) return;
  const int l = label[o];
  if (l < 0) return;
  // wave-aggregated: in the common case (one big component) all lanes carry the same label -> one atomic per wave
  const unsigned long long act = __ballot(1);
  const int leader = __ffsll((long long)act) - 1;
  const int l0 = __shfl(l, leader, 64);
  if (__ballot(l == l0) == act) { if ((int)(threadIdx.x & 63) == leader) atomicAdd(&size_at[l0], __popcll(act)); }
  else atomicAdd(&size_at[l], 1);
  if (l == o) { const int pos = atomicAdd(count, 1); if (pos < kMaxComponents) roots[pos] = o; }
}

// ---- reductions -----------------------------------------------------------------------------------------
__device__ __forceinline__ void block_partial(double v, double* __restrict__ partials) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (part[0] + part[1]) + (part[2] + part[3]);
}
__device__ __forceinline__ double reduce_partials(const double* __restrict__ partials) {   // one block of 256
  double v = 0.0;
  for (int t = threadIdx.x; t < kRedBlocks; t += 256) v += partials[t];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __shared__ double part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = v;
  __syncthreads();
  return (part[0] + part[1]) + (part[2] + part[3]);   // valid in thread 0 (and everywhere: all read part[])
}

// ---- CG pieces ------------------------------------------------------------------------------------------
// x = 0, r = rhs on the component (0 elsewhere), partial r.r
__global__ __launch_bounds__(256) void k_pcg_setup(long long n, const int* __restrict__ label, int root,
                                                   const float* __restrict__ div, float* __restrict__ x,
                                                   float* __restrict__ r, float* __restrict__ s, double* __restrict__ partials) {
  double acc = 0.0;
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < n; o += (long long)gridDim.x * 256) {
    const float v = (label[o] == root) ? div[o] : 0.0f;
    x[o] = 0.0f; r[o] = v; s[o] = 0.0f;
    acc += (double)v * v;
  }
  block_partial(acc, partials);
}
__global__ __launch_bounds__(256) void k_pcg_begin(PcgState* __restrict__ S, const double* __restrict__ partials, float tol,
                                                   int max_iter) {
  const double rr = reduce_partials(partials);
  if (threadIdx.x == 0) {
    S->rr1 = rr; S->rr0 = 0.0; S->num = 0.0; S->prev_num = 0.0; S->den = 0.0; S->alpha = 0.0f; S->beta = 0.0f;
    S->tol2 = tol * tol; S->iter = 0; S->max_iter = max_iter; S->first = 1; S->sum_x = 0.0; S->pending = 0;
    S->bad = (rr != rr) ? 1 : 0;
    S->done = (!((float)rr > S->tol2) || S->bad) ? 1 : 0;   // while (r_norm_sq1 > tol * tol && iter <= max_iter)
  }
}
// The scalar steps of an iteration (alpha, beta, the residual bookkeeping) ride at the head of the chip-wide kernel that
// needs them: every block folds the 512 partial sums itself, in the same fixed order, and block 0 records the result
// -- one launch per iteration for scalars (k_pcg_top) instead of four, which at 128^3 were a third of its time.
// Each reduction has its own partials buffer (a block may still be reading the previous one).
__device__ __forceinline__ void pcg_fold_residual(PcgState* __restrict__ S, double rr) {   // the end of an iteration
  S->prev_num = S->num;
  S->rr0 = S->rr1;
  S->rr1 = rr;
  S->first = 0;
  S->pending = 0;
  if (rr != rr) { S->bad = 1; S->done = 1; }
}
// end of the previous iteration (if one is pending), then the loop condition and iter++
__global__ __launch_bounds__(256) void k_pcg_top(PcgState* __restrict__ S, const double* __restrict__ p_rr, int test) {
  if (S->done) return;
  const int pending = S->pending;
  double rr = 0.0;
  if (pending) rr = reduce_partials(p_rr);
  if (threadIdx.x != 0) return;
  if (pending) pcg_fold_residual(S, rr);
  if (S->done || !test) return;
  if (!((float)S->rr1 > S->tol2) || S->iter > S->max_iter) { S->done = 1; return; }
  S->iter++;
}
// partial a.b over the grid (vectors are 0 off the component)
__global__ __launch_bounds__(256) void k_pcg_dot(const PcgState* __restrict__ S, long long n, const float* __restrict__ a,
                                                 const float* __restrict__ b, double* __restrict__ partials) {
  if (S->done) return;
  double acc = 0.0;
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < n; o += (long long)gridDim.x * 256) acc += (double)a[o] * b[o];
  block_partial(acc, partials);
}
// beta (num = r.z from `p_num` when preconditioned, else r.r), then s = z + beta * s
// (iteration 1: s = z; the reference's scal-then-axpy order: (beta*s) + z)
__global__ __launch_bounds__(256) void k_pcg_dir(PcgState* __restrict__ S, long long n, const double* __restrict__ p_num, int precond,
                                                 const float* __restrict__ z, float* __restrict__ s) {
  if (S->done) return;
  const bool first = S->first != 0;
  double num = S->rr1;
  if (precond) num = reduce_partials(p_num);
  // beta_k = r_{k-1}.z_{k-1} / (r_{k-2}.z_{k-2}); the second is last iteration's numerator
  const float beta = first ? 0.0f : (float)num / clamp_to_epsilon((float)(precond ? S->prev_num : S->rr0));
  if (blockIdx.x == 0 && threadIdx.x == 0) { S->num = num; S->beta = beta; }
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < n; o += (long long)gridDim.x * 256)
    s[o] = first ? z[o] : (beta * s[o] + 1.0f * z[o]);
}
// w = A s on the component, partial s.w
template <bool IS3D>
__global__ __launch_bounds__(256) void k_pcg_apply(const PcgState* __restrict__ S, Dom d, const float* __restrict__ flags,
                                                   const int* __restrict__ label, int root, const float* __restrict__ s,
                                                   float* __restrict__ w, double* __restrict__ partials) {
  if (S->done) return;
  double acc = 0.0;
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < d.sc; o += (long long)gridDim.x * 256) {
    float v = 0.0f;
    // setupLaplacian, generic/tfluids.cu:938-1075: the row of an interior fluid cell. Every load is UNCONDITIONAL (round 4:
    // `if (f & kFluid) off += s[nb]` made each neighbour a branch whose join drains the load queue -- 14 memory round trips in
    // a row per cell); a cell outside the component reads its own index instead of its neighbours' and drops the values.
    const bool in = label[o] == root;
    const int nb[6] = {(int)o - d.sz, (int)o - d.sy, (int)o - 1, (int)o + 1, (int)o + d.sy, (int)o + d.sz};
    float fn[6], sn[6];
#pragma unroll
    for (int q = 0; q < 6; q++) {
      if (!IS3D && (q == 0 || q == 5)) { fn[q] = 0.0f; sn[q] = 0.0f; continue; }
      const long long e = in ? (long long)nb[q] : o;
      fn[q] = flags[e]; sn[q] = s[e];
    }
    const float so = s[o];
    if (in) {
      float diag = 0.0f, off = 0.0f;
#pragma unroll
      for (int q = 0; q < 6; q++) {
        if (!IS3D && (q == 0 || q == 5)) continue;
        const int f = (int)fn[q];
        if (!(f & kObstacle)) diag += 1.0f;
        if (f & kFluid) off += sn[q];
      }
      v = diag * so - off;
    }
    w[o] = v;
    acc += (double)so * v;
  }
  block_partial(acc, partials);
}
// alpha = num / s.w (from `p_den`), then x += alpha s; r -= alpha w; partial r.r
__global__ __launch_bounds__(256) void k_pcg_update(PcgState* __restrict__ S, long long n, const double* __restrict__ p_den,
                                                    const float* __restrict__ s, const float* __restrict__ w, float* __restrict__ x,
                                                    float* __restrict__ r, double* __restrict__ p_rr) {
  if (S->done) return;
  const double den = reduce_partials(p_den);
  const float alpha = (float)S->num / clamp_to_epsilon((float)den), nalpha = -alpha;
  if (blockIdx.x == 0 && threadIdx.x == 0) { S->den = den; S->alpha = alpha; S->pending = 1; }
  __syncthreads();      // reduce_partials' shared scratch is reused by block_partial below
  double acc = 0.0;
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < n; o += (long long)gridDim.x * 256) {
    x[o] = alpha * s[o] + x[o];
    const float rv = nalpha * w[o] + r[o];
    r[o] = rv;
    acc += (double)rv * rv;
  }
  block_partial(acc, p_rr);
}
// sum of x over the component (for the mean), then p = x - mean on the component
__global__ __launch_bounds__(256) void k_pcg_sum(long long n, const float* __restrict__ x, double* __restrict__ partials) {
  double acc = 0.0;
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < n; o += (long long)gridDim.x * 256) acc += (double)x[o];
  block_partial(acc, partials);
}
__global__ __launch_bounds__(256) void k_pcg_sum_end(PcgState* __restrict__ S, const double* __restrict__ partials) {
  const double v = reduce_partials(partials);
  if (threadIdx.x == 0) S->sum_x = v;
}
__global__ __launch_bounds__(256) void k_pcg_write(const PcgState* __restrict__ S, long long n, const int* __restrict__ label,
                                                   int root, int size, const float* __restrict__ x, float* __restrict__ p) {
  const float mean = (float)(S->sum_x / (double)size);
  for (long long o = (long long)blockIdx.x * 256 + threadIdx.x; o < n; o += (long long)gridDim.x * 256)
    if (label[o] == root) p[o] = x[o] - mean;      // copyPressureFromSystem, generic/tfluids.cu:1216-1240
}

// ---- ILU(0) / IC(0) on the 5/7-point pattern: wavefront sweeps over hyperplanes h = i + j + k -------------
// thread -> (j, k) of the hyperplane, i = h - j - k
struct Wave { int i, j, k, o; bool ok; };
template <bool IS3D>
__device__ __forceinline__ Wave wave_cell(const Dom& d, int h, const int* __restrict__ label, int root) {
  Wave c;
  const int t = blockIdx.x * 256 + threadIdx.x;
  c.j = 1 + t % (d.Y - 2);
  c.k = IS3D ? 1 + t / (d.Y - 2) : 0;
  c.i = h - c.j - c.k;
  c.ok = c.i >= 1 && c.i <= d.X - 2 && (!IS3D || c.k <= d.Z - 2) && t < (d.Y - 2) * (IS3D ? d.Z - 2 : 1);
  c.o = c.ok ? TFL_AT(d, c.i, c.j, c.k) : 0;
  if (c.ok && label[c.o] != root) c.ok = false;
  return c;
}
// dg(n) = a(n,n) - sum_{lower fluid nbrs m} 1/dg(m)
template <bool IS3D>
__global__ __launch_bounds__(256) void k_ilu_factor(Dom d, int h, const float* __restrict__ flags, const int* __restrict__ label,
                                                    int root, float* __restrict__ dg) {
  const Wave c = wave_cell<IS3D>(d, h, label, root);
  if (!c.ok) return;
  const int nb[6] = {c.o - d.sz, c.o - d.sy, c.o - 1, c.o + 1, c.o + d.sy, c.o + d.sz};
  float diag = 0.0f;
#pragma unroll
  for (int q = 0; q < 6; q++) {
    if (!IS3D && (q == 0 || q == 5)) continue;
    if (!(((int)flags[nb[q]]) & kObstacle)) diag += 1.0f;
  }
#pragma unroll
  for (int q = 0; q < 3; q++) {       // lower neighbours in column order: z-1, y-1, x-1
    if (!IS3D && q == 0) continue;
    if (((int)flags[nb[q]]) & kFluid) diag -= 1.0f / dg[nb[q]];
  }
  dg[c.o] = diag;
}
// forward solve.  ilu0: y(n) = r(n) + sum_m y(m)/dg(m).   ic0: y(n) = (r(n) + sum_m y(m)/R(m,m)) / R(n,n)
template <bool IS3D, bool IC>
__global__ __launch_bounds__(256) void k_ilu_forward(const PcgState* __restrict__ S, Dom d, int h, const float* __restrict__ flags,
                                                     const int* __restrict__ label, int root, const float* __restrict__ dg,
                                                     const float* __restrict__ r, float* __restrict__ y) {
  if (S->done) return;
  const Wave c = wave_cell<IS3D>(d, h, label, root);
  if (!c.ok) return;
  const int nb[3] = {c.o - d.sz, c.o - d.sy, c.o - 1};
  float v = r[c.o];
#pragma unroll
  for (int q = 0; q < 3; q++) {
    if (!IS3D && q == 0) continue;
    if (((int)flags[nb[q]]) & kFluid) v += IC ? y[nb[q]] / sqrtf(dg[nb[q]]) : y[nb[q]] / dg[nb[q]];
  }
  y[c.o] = IC ? v / sqrtf(dg[c.o]) : v;
}
// backward solve. ilu0: z(n) = (y(n) + sum_q z(q)) / dg(n).   ic0: z(n) = (y(n) + (sum_q z(q)) / R(n,n)) / R(n,n)
template <bool IS3D, bool IC>
__global__ __launch_bounds__(256) void k_ilu_backward(const PcgState* __restrict__ S, Dom d, int h, const float* __restrict__ flags,
                                                      const int* __restrict__ label, int root, const float* __restrict__ dg,
                                                      const float* __restrict__ y, float* __restrict__ z) {
  if (S->done) return;
  const Wave c = wave_cell<IS3D>(d, h, label, root);
  if (!c.ok) return;
  const int nb[3] = {c.o + 1, c.o + d.sy, c.o + d.sz};
  float acc = 0.0f;
#pragma unroll
  for (int q = 0; q < 3; q++) {
    if (!IS3D && q == 2) continue;
    if (((int)flags[nb[q]]) & kFluid) acc += z[nb[q]];
  }
  if (IC) { const float rnn = sqrtf(dg[c.o]); z[c.o] = (y[c.o] + acc / rnn) / rnn; }
  else z[c.o] = (y[c.o] + acc) / dg[c.o];
}

// ---- the triangular solves as TWO launches: pipelined wavefronts (3-D) ------------------------------------------
// A lexicographic IC(0) / ILU(0) solve is a 3-D recurrence: cell (i, j, k) needs (i-1, j, k), (i, j-1, k), (i, j, k-1).
// One launch per hyperplane (above) is X+Y+Z-6 launches per solve, ~760 per CG iteration at 128^3: 2.7 ms of launch
// latency around ~20 us of arithmetic. Here a sweep is ONE launch:
//   * the interior is cut into sub-boxes of 64 rows (j) x kWfPlanes (8) planes (k) x all of x; a block of 8 COMPUTE waves
//     + 2 HELPER waves owns one: compute wave w <-> plane, lane l <-> row, and at step t the thread works on the cell
//     with (i-1) + l + kWfLag w = t. Its three predecessors were computed one step earlier by itself (x: a register),
//     one step earlier by the lane below (y: one DPP wave shift) and kWfLag (2) steps earlier by the wave below (z).
//     That lag lets the waves of a block exchange through LDS once per kWfLag steps ("interval"): one 8-byte read, one
//     8-byte write and one barrier. A step is a dependent chain of ~60 ns (tools/ubench/chain_latency.hip);
//   * a block's steps are bound by its CU: four waves per SIMD would take turns at ~12 instructions per step, and every
//     operand and result of the sub-box passes through one CU's 64 B/clk vector-memory pipe. Hence two waves per SIMD
//     (8 planes) and SKEWED arrays laid out [sub-box][plane][t / 4][row][t % 4]: per group of four steps a wave reads
//     cc and r and writes its results with ONE 16-byte access each. Two chip-wide kernels per solve copy r into, and
//     z out of, that layout (k_wf_skew);
//   * sub-boxes depend on their lower j / k neighbours through global memory, with no flags and no fences: the edge
//     plane and the edge lanes of a block are also stored as 8-byte {value, tag} pairs (tag = the launch's sequence
//     number) into hand-off arrays, and the consumer re-reads a pair until it carries this launch's tag (bounded; an
//     error word ends the solve instead of hanging the GPU). All of that is the HELPER waves' job (one for the slab
//     direction, one for the strips): they prefetch the predecessors' pairs a few intervals ahead, hand the values to
//     the compute waves through the same LDS slots a neighbouring plane would use, and publish the block's own edge
//     results out of those slots. The compute waves run a branch-free loop with nothing but their operand prefetch in
//     flight -- a wave's loads complete in issue order, and with the short-fused pair loads in the same queue every
//     operand prefetch had to arrive within their two groups (measured: memory latency / 8 per step, whatever the
//     block's size). A consumer settles by itself at the smallest lag behind its predecessor that memory latency allows.
// With c = 1/sqrt(d) (IC) or 1/d (ILU) and cc = c*c (IC) or c (ILU), both factorisations run the same recurrences:
//   forward    q = (((r + q_z) + q_y) + q_x) * cc          (q = y * c of the reference's forward solve)
//   backward   z = q + ((z_x + z_y) + z_z) * cc
// (the reference divides by R(m,m) / d(m) where this multiplies by reciprocals: inside the solver's tolerance).
#ifndef TFL_WF_PLANES
#define TFL_WF_PLANES 8
#endif
#ifndef TFL_WF_DEPTH
#define TFL_WF_DEPTH 16
#endif
// kWfDepth: steps the operands are prefetched ahead; kWfHelperAhead: exchange intervals the helper wave prefetches the
// predecessors' pairs ahead (a block can only run that far + the memory round trip behind its predecessor)
#ifndef TFL_WF_LAG
#define TFL_WF_LAG 2
#endif
#ifndef TFL_WF_HELPER_AHEAD
#define TFL_WF_HELPER_AHEAD 4
#endif
constexpr int kWfRows = 64, kWfPlanes = TFL_WF_PLANES, kWfLag = TFL_WF_LAG, kWfDepth = TFL_WF_DEPTH, kWfHelperAhead = TFL_WF_HELPER_AHEAD, kWfMaxBlocks = 240;
static_assert(kWfDepth % 4 == 0 && (kWfLag == 1 || kWfLag == 2 || kWfLag == 4), "operands in groups of four steps");
static_assert(kWfHelperAhead >= 1 && (kWfDepth / kWfLag) % kWfHelperAhead == 0, "the helper's rotating slots repeat with the unrolled loop");
static_assert(kWfPlanes * kWfLag <= 64, "one helper lane per (plane, step of the interval)");

struct WfGeom {
  int X, Y, Z, ns, nb, NT;     // strips, slabs, steps per sub-box (a multiple of kWfDepth)
  long long sub;               // cells of one sub-box in the skewed arrays = kWfPlanes * NT * kWfRows
  int b_lo, b_cnt;             // the slabs [b_lo, b_lo + b_cnt) this launch works on (all strips of them)
};
// the sub-box of a workgroup: blocks of one launch are numbered strip-major over the launch's slabs
__device__ __forceinline__ int wf_block(const WfGeom& g, int& sidx, int& bidx) {
  sidx = (int)blockIdx.x / g.b_cnt;
  bidx = g.b_lo + ((int)blockIdx.x - sidx * g.b_cnt);
  return sidx * g.nb + bidx;
}
inline WfGeom wf_geom(int Z, int Y, int X) {
  WfGeom g;
  g.X = X; g.Y = Y; g.Z = Z;
  g.ns = (Y - 2 + kWfRows - 1) / kWfRows; g.nb = (Z - 2 + kWfPlanes - 1) / kWfPlanes;
  g.NT = (((X - 2) + (kWfRows - 1) + kWfLag * (kWfPlanes - 1)) + kWfDepth - 1) / kWfDepth * kWfDepth;
  g.sub = (long long)kWfPlanes * g.NT * kWfRows;
  g.b_lo = 0; g.b_cnt = g.nb;
  return g;
}
inline bool wf_usable(bool is3d, int Z, int Y, int X) {
  if (!is3d || X < 3 || Y < 3 || Z < 3) return false;
  static const bool off = getenv("TFL_PCG_HYPERPLANES") != nullptr;     // A/B switch: the one-launch-per-hyperplane sweeps
  const WfGeom g = wf_geom(Z, Y, X);
  return !off && g.ns <= kWfMaxBlocks;      // (grids with more sub-boxes than fit the GPU at once run slab range by slab range)
}
// floats of: cc, r, q, z (skewed) | the slab hand-off pairs [block][t / 4][row][t % 4] | the strip hand-off pairs
// [block][plane][t] | the error word
// (floats, including kWfGuard pairs of padding at either end: the helper wave reads up to kWfLag * (kWfPlanes - 1) resp. 63 steps
// past a block's rows without clamping)
constexpr long long kWfGuard = 4 * kWfRows * (kWfLag * (kWfPlanes - 1) / 4 + 2);        // pairs; >= 64 too
inline long long wf_handoff_k(const WfGeom& g) { return 2ll * g.ns * g.nb * g.NT * kWfRows + 4 * kWfGuard; }
inline long long wf_handoff_s(const WfGeom& g) { return 2ll * g.ns * g.nb * kWfPlanes * g.NT + 4 * kWfGuard; }
inline long long wf_floats(int Z, int Y, int X) {
  const WfGeom g = wf_geom(Z, Y, X);
  return 4 * g.sub * g.ns * g.nb + wf_handoff_k(g) + wf_handoff_s(g) + 64;
}
__device__ __forceinline__ long long wf_at(const WfGeom& g, int i, int j, int k) {
  const int s = (j - 1) / kWfRows, l = (j - 1) % kWfRows, b = (k - 1) / kWfPlanes, w = (k - 1) % kWfPlanes;
  const int t = (i - 1) + l + kWfLag * w;
  return ((((long long)(s * g.nb + b) * kWfPlanes + w) * (g.NT / 4) + (t >> 2)) * kWfRows + l) * 4 + (t & 3);
}

// cc (skewed) from the factor's diagonal: one thread per interior cell
template <bool IC>
__global__ __launch_bounds__(256) void k_wf_build(WfGeom g, Dom d, const int* __restrict__ label, int root, const float* __restrict__ dg,
                                                  float* __restrict__ cs) {
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int nx = g.X - 2, ny = g.Y - 2, nz = g.Z - 2;
  if (t >= (long long)nx * ny * nz) return;
  const int i = 1 + (int)(t % nx), j = 1 + (int)((t / nx) % ny), k = 1 + (int)(t / ((long long)nx * ny));
  const int o = TFL_AT(d, i, j, k);
  float cc = 0.0f;
  if (label[o] == root) {
    if (IC) { const float c = 1.0f / sqrtf(dg[o]); cc = c * c; }
    else cc = 1.0f / dg[o];
  }
  cs[wf_at(g, i, j, k)] = cc;
}

// natural layout <-> skewed layout of a vector on the interior cells (chip-wide, one thread per cell: the sweeps
// themselves must not gather / scatter -- a wave's 64 rows are 64 cache lines per instruction)
template <bool TO_SKEWED>
__global__ __launch_bounds__(256) void k_wf_skew(const PcgState* __restrict__ S, WfGeom g, Dom d, float* __restrict__ nat, float* __restrict__ skw) {
  if (S->done) return;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  const int nx = g.X - 2, ny = g.Y - 2, nz = g.Z - 2;
  if (t >= (long long)nx * ny * nz) return;
  const int i = 1 + (int)(t % nx), j = 1 + (int)((t / nx) % ny), k = 1 + (int)(t / ((long long)nx * ny));
  const int o = TFL_AT(d, i, j, k);
  const long long a = wf_at(g, i, j, k);
  if (TO_SKEWED) skw[a] = nat[o];
  else nat[o] = skw[a];
}

// {value, tag} pairs: one 8-byte access, coherent across the chip (sc1), so a tag never arrives before its value
__device__ __forceinline__ unsigned long long wf_load_pair(const float2* p) {
  return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void wf_store_pair(float2* p, float v, int tag) {
  const unsigned long long u = (unsigned long long)__float_as_uint(v) | ((unsigned long long)(unsigned)tag << 32);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// the same loads for the retry path, opaque to the compiler's wait-count bookkeeping (they complete before they return)
__device__ __forceinline__ unsigned long long wf_reload_pair(const float2* p) {
  unsigned long long v;
  asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ int wf_reload_word(const int* p) {
  int v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ float wf_value(unsigned long long u) { return __uint_as_float((unsigned)u); }
__device__ __forceinline__ int wf_tag(unsigned long long u) { return (int)(u >> 32); }

// kWfLag consecutive floats of one lane with one LDS access
__device__ __forceinline__ void wf_lds_read(const float* p, float* v) {
  if (kWfLag == 4) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2 % kWfLag] = t.z; v[3 % kWfLag] = t.w; }
  else if (kWfLag == 2) { const float2 t = *reinterpret_cast<const float2*>(p); v[0] = t.x; v[1 % kWfLag] = t.y; }
  else v[0] = *p;
}
__device__ __forceinline__ void wf_lds_write(float* p, const float* v) {
  if (kWfLag == 4) *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1 % kWfLag], v[2 % kWfLag], v[3 % kWfLag]);
  else if (kWfLag == 2) *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1 % kWfLag]);
  else *p = v[0];
}

struct WfArrays {
  const float* cs;      // cc
  const float* in;      // forward: r; backward: q
  float* out;           // forward: q; backward: z
  float2* hk;           // slab hand-off pairs of this launch's results
  float2* hs;           // strip hand-off pairs
  int* err;
  long long* trace;     // TFL_WF_TRACE: [block][2] = wall clock (10 ns units) at a block's first and last step, else null
};

// A COMPUTE wave's share of a sweep (wave w < kWfPlanes <-> plane). DIR = +1 forward (lower neighbours), -1 backward
// (upper neighbours, steps run from the last to the first). It touches global memory only for its own operands and
// results: what it needs from other sub-boxes arrives through LDS from the block's HELPER wave (below), in the same
// slots and with the same lag as what it needs from the plane next door.
template <int DIR>
__device__ __forceinline__ void wf_compute(const WfGeom& g, const WfArrays& A, int tag, float (*ring)[2][kWfRows][kWfLag],
                                           float (*edge_s)[kWfPlanes + 1][kWfLag]) {
  constexpr int D = kWfDepth, G = D / 4;
  int sidx, bidx;
  const int blk = wf_block(g, sidx, bidx);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int NT = g.NT, NG = NT / 4;
  const long long base4 = ((long long)blk * kWfPlanes + w) * NG * kWfRows + lane;       // in float4 units; + group * 64
  const float4* c4 = reinterpret_cast<const float4*>(A.cs) + base4;
  const float4* i4 = reinterpret_cast<const float4*>(A.in) + base4;
  float4* o4 = reinterpret_cast<float4*>(A.out) + base4;
  // the wave whose results this one consumes with a lag: the plane below / above, or the helper (index kWfPlanes) for
  // the plane that borders the slab predecessor
  const int nbw = DIR > 0 ? (w > 0 ? w - 1 : kWfPlanes) : (w + 1 < kWfPlanes ? w + 1 : kWfPlanes);
  auto grp = [&](int ig) { return DIR > 0 ? ig : NG - 1 - ig; };           // memory group of the ig-th group in time

  // operands, prefetched G groups ahead into rotating registers (slot = group mod G; the loop below is unrolled by G)
  float4 cv[G], rv[G];
  auto issue = [&](int ig, int slot) {
    const int gm = grp(ig);
    cv[slot] = c4[(long long)gm * kWfRows];
    rv[slot] = i4[(long long)gm * kWfRows];
  };
  float q_prev = 0.0f;      // this thread's previous result (forward: q, backward: z)
  auto body = [&](int t0, auto more) {
#pragma unroll
    for (int c = 0; c < G; c++) {
      const int ig = t0 / 4 + c, gm = grp(ig);
      // in memory the group is ordered by t; backward walks it from its last element
      const float ccs[4] = {DIR > 0 ? cv[c].x : cv[c].w, DIR > 0 ? cv[c].y : cv[c].z, DIR > 0 ? cv[c].z : cv[c].y, DIR > 0 ? cv[c].w : cv[c].x};
      const float ins[4] = {DIR > 0 ? rv[c].x : rv[c].w, DIR > 0 ? rv[c].y : rv[c].z, DIR > 0 ? rv[c].z : rv[c].y, DIR > 0 ? rv[c].w : rv[c].x};
      float h4[4];
#pragma unroll
      for (int e = 0; e < 4 / kWfLag; e++) {       // LDS exchanges of this group: kWfLag steps each
        const int par = (c * (4 / kWfLag) + e) & 1;
        float below[kWfLag], edge[kWfLag];
        wf_lds_read(&ring[nbw][par ^ 1][lane][0], below);      // what the neighbour wave computed during the previous interval
        wf_lds_read(&edge_s[par ^ 1][w][0], edge);             // the strip predecessor's values for the edge lane (every lane reads them)
#pragma unroll
        for (int v = 0; v < kWfLag; v++) {
          const int u4 = e * kWfLag + v;
          // y neighbour: the lane below / above; the edge lane (0 forward, 63 backward) has no source lane and keeps `old`
          float nb_y = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge[v]), __builtin_bit_cast(int, q_prev),
                                                                               DIR > 0 ? 0x138 /*wave_shr:1*/ : 0x130 /*wave_shl:1*/, 0xf, 0xf, false));
          asm volatile("" : "+v"(nb_y));
          float res;
          if (DIR > 0) res = (((ins[u4] + below[v]) + nb_y) + q_prev) * ccs[u4];
          else res = __builtin_fmaf((q_prev + nb_y) + below[v], ccs[u4], ins[u4]);
          h4[u4] = res;
          q_prev = res;
        }
        wf_lds_write(&ring[w][par][lane][0], &h4[e * kWfLag]);
        // workgroup barrier that waits for the LDS write only (__syncthreads() would also drain the operand prefetch)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      const float m4[4] = {DIR > 0 ? h4[0] : h4[3], DIR > 0 ? h4[1] : h4[2], DIR > 0 ? h4[2] : h4[1], DIR > 0 ? h4[3] : h4[0]};   // by t
      o4[(long long)gm * kWfRows] = make_float4(m4[0], m4[1], m4[2], m4[3]);
      if constexpr (decltype(more)::value) issue(ig + G, c);
    }
  };
#pragma unroll
  for (int c = 0; c < G; c++) issue(c, c);
  if (A.trace && threadIdx.x == 0) A.trace[blk * 2] = wall_clock64();
  // the helper has to deliver the first interval's edge values before anyone starts (it is the "+1" of this barrier)
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  int t0 = 0;
  for (; t0 + D < NT; t0 += D) body(t0, std::true_type{});
  body(t0, std::false_type{});
  if (A.trace && threadIdx.x == 0) A.trace[blk * 2 + 1] = wall_clock64();
}

// The HELPER wave of a block (wave index kWfPlanes): it alone reads what other sub-boxes hand over, and passes it on
// through LDS one exchange interval ahead of its use, in step with the block's barriers:
//   ring[kWfPlanes][.][lane][.]  the slab predecessor's edge-plane values  -> read by the bordering plane as its "plane next door"
//   edge_s[.][w][.]              the strip predecessor's edge-lane values of plane w -> the DPP `old` of every compute wave
// Its loads are the only ones in the block with a short fuse (the pairs exist only a few steps before they are needed),
// and a wave's loads complete in issue order: in a compute wave they made every operand prefetch issued before them
// wait in line (measured: each step of every block then cost memory latency / 8, whatever the block's size). Here they
// have a wave -- and a vmcnt -- of their own. Pairs are prefetched kWfHelperAhead intervals ahead, checked against this
// launch's tag and re-read until they carry it (bounded; an error word ends the solve instead of hanging the GPU).
template <int DIR, int ROLE>
__device__ __forceinline__ void wf_helper(const WfGeom& g, const WfArrays& A, int tag, float (*ring)[2][kWfRows][kWfLag],
                                          float (*edge_s)[kWfPlanes + 1][kWfLag]) {
  constexpr int P = kWfHelperAhead, NI = kWfDepth / kWfLag;     // the loop is unrolled by NI intervals (one compute body); NI % P == 0
  constexpr bool SLAB = ROLE == 0;
  int sidx, bidx;
  const int blk = wf_block(g, sidx, bidx);
  const int lane = threadIdx.x & 63;
  const int NT = g.NT, NG = NT / 4;
  const int ps = DIR > 0 ? (sidx > 0 ? blk - g.nb : -1) : (sidx + 1 < g.ns ? blk + g.nb : -1);
  const int pk = DIR > 0 ? (bidx > 0 ? blk - 1 : -1) : (bidx + 1 < g.nb ? blk + 1 : -1);
  const bool has_pred = SLAB ? pk >= 0 : ps >= 0;
  const bool has_succ = SLAB ? (DIR > 0 ? bidx + 1 < g.nb : bidx > 0) : (DIR > 0 ? sidx + 1 < g.ns : sidx > 0);     // someone reads what this block publishes
  // SLAB: lane <-> row of the edge plane, kWfLag pairs per interval. STRIP: lane = w * kWfLag + v <-> (plane w, step v in
  // memory order of the interval) of the edge lane; the lanes beyond that work on a spare row / the guard padding
  const int sw = min(lane / kWfLag, kWfPlanes), sv = lane % kWfLag;
  const bool live = SLAB || lane < kWfPlanes * kWfLag;
  const float2* from = SLAB ? A.hk + ((long long)(has_pred ? pk : blk) * NG * kWfRows + lane) * 4
                            : A.hs + ((long long)(has_pred ? ps : blk) * kWfPlanes + min(sw, kWfPlanes - 1)) * NT;
  float2* mine = SLAB ? A.hk + ((long long)blk * NG * kWfRows + lane) * 4 : A.hs + ((long long)blk * kWfPlanes + min(sw, kWfPlanes - 1)) * NT;
  constexpr int kOff = SLAB ? DIR * kWfLag * (kWfPlanes - 1) : DIR * (kWfRows - 1);
  constexpr int NV = SLAB ? kWfLag : 1;       // pairs per lane and interval
  // the first step IN MEMORY ORDER of interval iv (an interval's kWfLag steps are consecutive in t, and so are their pairs
  // in both hand-off arrays: kOff of the slab is a multiple of kWfLag and a group of four steps never splits an interval)
  auto t_lo = [&](int iv) { return DIR > 0 ? iv * kWfLag : NT - (iv + 1) * kWfLag; };
  // addresses are NOT clamped: both arrays carry kWfGuard pairs of padding at either end (wf_floats), a pair outside
  // [0, NT) is loaded from a neighbouring block's rows or the padding and never looked at
  auto at = [&](const float2* base, int t) { return SLAB ? base + (long long)(t >> 2) * (4 * kWfRows) + (t & 3) : base + (t + sv); };

  unsigned long long pv[P][NV];      // rotating: the pairs of interval (index mod P)
  bool dead = false;
  auto issue = [&](int iv, int slot) {
    const float2* src = at(from, t_lo(iv) + kOff);
#pragma unroll
    for (int v = 0; v < NV; v++) pv[slot][v] = wf_load_pair(src + v);
  };
  // deliver interval iv's values into the slots the compute waves read during interval iv (parity of iv, inverted). A
  // helper is one wave running a serial instruction stream in step with eight that do ~25 instructions per interval: its
  // fast path is straight-line (one test and branch for all the interval's pairs); re-reading is the rare path
  auto deliver = [&](int iv, int slot) {
    const int par = iv & 1;
    const int tp = t_lo(iv) + kOff + (SLAB ? 0 : sv);
    const bool want = has_pred && live && tp >= 0 && tp < NT;       // SLAB: wave-uniform, the same for the interval's steps
    bool stale = false;
#pragma unroll
    for (int v = 0; v < NV; v++) stale |= want && wf_tag(pv[slot][v]) != tag;
    if (__builtin_expect(__ballot(stale) != 0ull, 0) && !dead) {
      for (int spin = 0;; spin++) {
        bool again = false;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          if (want && wf_tag(pv[slot][v]) != tag) pv[slot][v] = wf_reload_pair(at(from, t_lo(iv) + kOff) + v);
          again |= want && wf_tag(pv[slot][v]) != tag;
        }
        if (__ballot(again) == 0ull) break;
        // never hang the GPU: give up after ~1 s, or when another block already has (looked at every 64th retry only)
        if (spin > (1 << 17) || ((spin & 63) == 63 && wf_reload_word(A.err))) { atomicExch(A.err, 1); dead = true; break; }
      }
    }
    if (SLAB) {
      float kval[kWfLag];     // in TIME order of the interval's steps, like a compute wave's results
#pragma unroll
      for (int v = 0; v < kWfLag; v++) kval[v] = want ? wf_value(pv[slot][DIR > 0 ? v : kWfLag - 1 - v]) : 0.0f;
      wf_lds_write(&ring[kWfPlanes][par ^ 1][lane][0], kval);
    } else {
      edge_s[par ^ 1][sw][DIR > 0 ? sv : kWfLag - 1 - sv] = want ? wf_value(pv[slot][0]) : 0.0f;
    }
  };
  // publish interval iv's results of the block's edge plane (-> the slab successor) resp. of every plane's edge lane (->
  // the strip successor) as {value, tag} pairs, out of the LDS slots the compute waves wrote them to during interval iv
  // (stable until the barrier that ends interval iv + 1): the compute waves carry no hand-off code at all
  auto publish = [&](int iv, int ptag) {
    if (!has_succ) return;        // block-uniform
    const int par = iv & 1;
    float res[kWfLag];            // time order
    wf_lds_read(SLAB ? &ring[DIR > 0 ? kWfPlanes - 1 : 0][par][lane][0] : &ring[min(sw, kWfPlanes - 1)][par][DIR > 0 ? kWfRows - 1 : 0][0], res);
    float2* dst = live ? const_cast<float2*>(at(mine, t_lo(iv))) : A.hs - 1 - lane;       // idle lanes: into the guard padding
    if (SLAB) {
#pragma unroll
      for (int v = 0; v < kWfLag; v++) wf_store_pair(dst + v, res[DIR > 0 ? v : kWfLag - 1 - v], ptag);
    } else {
      float one = res[0];
#pragma unroll
      for (int v = 1; v < kWfLag; v++) one = (DIR > 0 ? sv : kWfLag - 1 - sv) == v ? res[v] : one;
      wf_store_pair(dst, one, live ? ptag : 0);
    }
  };
  const int n_iv = NT / kWfLag;
#pragma unroll
  for (int i = 0; i < P; i++) issue(i, i);
  deliver(0, 0);
  issue(P, 0);
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // the compute waves' start barrier
  // during interval iv a helper publishes interval iv - 1, delivers interval iv + 1 and prefetches interval iv + 1 + P
  for (int iv0 = 0; iv0 < n_iv; iv0 += NI) {
#pragma unroll
    for (int i = 0; i < NI; i++) {
      // no branch around the loads (hipcc sizes the loop's vmcnt waits for the path that issues fewest): past the end the
      // last interval is delivered again, into slots nobody reads any more
      const int iv = iv0 + i;
      publish(max(iv - 1, 0), iv > 0 ? tag : 0);          // (interval 0 has nothing behind it: a pair nobody will accept)
      deliver(min(iv + 1, n_iv - 1), (i + 1) % P);
      issue(min(iv + 1 + P, n_iv - 1), (i + 1) % P);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
  }
  publish(n_iv - 1, tag);
}

template <int DIR>
__global__ __launch_bounds__((kWfPlanes + 2) * 64) void k_wf_sweep(const PcgState* __restrict__ S, WfGeom g, WfArrays A, int tag) {
  if (S->done) return;
  __shared__ __attribute__((aligned(16))) float ring[kWfPlanes + 1][2][kWfRows][kWfLag];   // results of a wave's last two exchange intervals
  __shared__ __attribute__((aligned(16))) float edge_s[2][kWfPlanes + 1][kWfLag];     // [.][kWfPlanes]: spare row for the helper's idle lanes
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (w <= kWfPlanes) {
#pragma unroll
    for (int v = 0; v < kWfLag; v++) ring[w][0][lane][v] = ring[w][1][lane][v] = 0.0f;
  }
  if (threadIdx.x < 2 * (kWfPlanes + 1) * kWfLag) (&edge_s[0][0][0])[threadIdx.x] = 0.0f;
  __syncthreads();
  if (w < kWfPlanes) wf_compute<DIR>(g, A, tag, ring, edge_s);
  else if (w == kWfPlanes) wf_helper<DIR, 0>(g, A, tag, ring, edge_s);
  else wf_helper<DIR, 1>(g, A, tag, ring, edge_s);
}

// ---- normalizePressureMean (generic/tfluids.cc:845-925): p -= mean of p over the cell's fluid component ----
__global__ __launch_bounds__(256) void k_npm_sum(Dom d, const int* __restrict__ label, const float* __restrict__ p,
                                                 double* __restrict__ sum_at) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  const int l = o < d.sc ? label[o] : -1;
  const bool valid = l >= 0;
  const unsigned long long vm = __ballot(valid);
  if (!vm) return;                                   // wave-uniform
  // one fp64 atomic per wave when every fluid lane of the wave sits in the same component (the common case)
  const int l0 = __shfl(l, __ffsll((long long)vm) - 1, 64);
  double v = valid ? (double)p[o] : 0.0;
  if (__ballot(valid && l != l0) == 0ull) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(&sum_at[l0], v);
  } else if (valid) {
    atomicAdd(&sum_at[l], v);
  }
}
__global__ __launch_bounds__(256) void k_npm_sub(Dom d, const int* __restrict__ label, const int* __restrict__ size_at,
                                                 const double* __restrict__ sum_at, float* __restrict__ p) {
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= d.sc) return;
  const int l = label[o];
  if (l < 0) return;
  p[o] = p[o] - (float)(sum_at[l] / (double)size_at[l]);
}

inline int cdiv(long long a, int b) { return (int)((a + b - 1) / b); }

}  // namespace

long long pcg_workspace_floats(int Z, int Y, int X) {
  const long long n = (long long)Z * Y * X;
  // label, size_at (int32) + x, r, z, s, w, dg, y (fp32) + roots + partials/state (fp64, kept 8-byte aligned first)
  return 2 * (3 * kRedBlocks + 64) + 9 * n + kMaxComponents + 64 + (wf_usable(Z > 1, Z, Y, X) ? wf_floats(Z, Y, X) + 4 : 0);
}

// Solves every component of every batch element. Returns 0, or a negative code with `msg` filled:
// -1 invalid flags (fluid on the border), -2 NaN residual, -3 too many components, -4 HIP error, -5 the pipelined
// triangular solves timed out (only with allow_wavefronts: repeat without).
int pcg_solve(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* p, const float* flags, const float* div,
              int precond /*0 none, 1 ilu0, 2 ic0*/, float tol, int max_iter, int verbose, float* workspace, float* residual,
              char* msg, size_t msg_len, bool allow_wavefronts) {
  const Dom d = make_dom(Z, Y, X);
  const long long n = d.sc;
  double* partials = reinterpret_cast<double*>(workspace);
  double* p_den = partials + kRedBlocks;      // s.w
  double* p_rr = partials + 2 * kRedBlocks;   // r.r of the update
  PcgState* S = reinterpret_cast<PcgState*>(partials + 3 * kRedBlocks);
  float* base = workspace + 2 * (3 * kRedBlocks + 64);
  int* label = reinterpret_cast<int*>(base);
  int* size_at = reinterpret_cast<int*>(base + n);
  float* x = base + 2 * n; float* r = base + 3 * n; float* z = base + 4 * n; float* s = base + 5 * n;
  float* w = base + 6 * n; float* dg = base + 7 * n; float* y = base + 8 * n;
  int* roots = reinterpret_cast<int*>(base + 9 * n);
  int* counters = roots + kMaxComponents;   // [0] changed, [1] count, [2] border fluid
  // pipelined wavefront sweeps (3-D): the skewed arrays (cc, r, q, z), the two hand-off arrays of {value, tag} pairs, an error word
  const bool wf = allow_wavefronts && wf_usable(is3d, Z, Y, X);
  const WfGeom wg = wf_geom(Z > 2 ? Z : 3, Y > 2 ? Y : 3, X > 2 ? X : 3);
  float* wfbase = base + 9 * n + kMaxComponents + 64;
  wfbase += (4 - (((uintptr_t)wfbase >> 2) & 3)) & 3;                  // 16-byte aligned (float4 rows, {value, tag} pairs)
  float* cs = wfbase;
  const long long wtot = wg.sub * wg.ns * wg.nb;
  float* rsk = wf ? cs + wtot : nullptr;
  float* qsk = wf ? cs + 2 * wtot : nullptr;
  float* zsk = wf ? cs + 3 * wtot : nullptr;
  float2* wf_hk = wf ? reinterpret_cast<float2*>(cs + 4 * wtot) + kWfGuard : nullptr;       // 16-byte aligned: wtot is a multiple of 1024
  float2* wf_hs = wf ? reinterpret_cast<float2*>(cs + 4 * wtot + wf_handoff_k(wg)) + kWfGuard : nullptr;
  int* wferr = wf ? reinterpret_cast<int*>(cs + 4 * wtot + wf_handoff_k(wg) + wf_handoff_s(wg)) : nullptr;
  int epoch = 0;                                // tag of the next sweep; the skewed arrays are zeroed with it
  static const int wf_cap = exp_env("TFL_WF_MAX_BLOCKS") ? std::max(1, atoi(exp_env("TFL_WF_MAX_BLOCKS"))) : kWfMaxBlocks;     // (tests: small launches)
  const int wf_slabs = std::max(1, std::min(wf_cap, kWfMaxBlocks) / std::max(wg.ns, 1));      // slabs per launch
  // development aid: TFL_WF_TRACE=1 prints when every sub-box of the last forward / backward sweep started and finished
  static const bool wf_trace_on = exp_env("TFL_WF_TRACE") != nullptr;
  long long* wf_trace = nullptr;
  if (wf && wf_trace_on && precond && hipMalloc(&wf_trace, sizeof(long long) * 4 * wg.ns * wg.nb) != hipSuccess) wf_trace = nullptr;
  struct TraceGuard { long long*& p; ~TraceGuard() { if (p) { (void)hipFree(p); p = nullptr; } } } trace_guard{wf_trace};   // every exit frees it
  auto hip_ok = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    snprintf(msg, msg_len, "solveLinearSystemPCG: %s: %s", what, hipGetErrorString(e));
    return false;
  };
  if (!hip_ok(hipMemsetAsync(p, 0, sizeof(float) * (size_t)B * n, st), "memset p")) return -4;   // :1341
  const int gcell = cdiv(n, 256);
  float max_res = -std::numeric_limits<float>::infinity();
  const int nlev = (X - 2) + (Y - 2) + (is3d ? Z - 2 : 0);        // hyperplanes h = hmin .. hmin + nlev - dims
  const int hmin = is3d ? 3 : 2, hmax = (X - 2) + (Y - 2) + (is3d ? Z - 2 : 0);
  (void)nlev;
  const int gwave = cdiv((long long)(Y - 2) * (is3d ? Z - 2 : 1), 256);
  for (int b = 0; b < B; b++) {
    const float* fl = flags + (long long)b * n;
    const float* dv = div + (long long)b * n;
    float* pb = p + (long long)b * n;
    // ---- components -----------------------------------------------------------------------------------
    int h_cnt[3] = {0, 0, 0};
    if (!hip_ok(hipMemsetAsync(counters, 0, 3 * sizeof(int), st), "memset")) return -4;
    if (!hip_ok(hipMemsetAsync(size_at, 0, sizeof(int) * (size_t)n, st), "memset")) return -4;
    { TFL_TIMED("k_cc", st); k_cc_init<<<gcell, 256, 0, st>>>(d, is3d, fl, label, counters + 2); }
    for (int round = 0;; round++) {
      if (!hip_ok(hipMemsetAsync(counters, 0, sizeof(int), st), "memset")) return -4;
      { TFL_TIMED("k_cc", st);
        k_cc_hook<<<gcell, 256, 0, st>>>(d, is3d, label, counters);
        k_cc_compress<<<gcell, 256, 0, st>>>(d, label); }
      if (!hip_ok(hipMemcpyAsync(h_cnt, counters, 3 * sizeof(int), hipMemcpyDeviceToHost, st), "memcpy")) return -4;
      if (!hip_ok(hipStreamSynchronize(st), "sync")) return -4;
      if (!h_cnt[0]) break;
      if (round > 100000) { snprintf(msg, msg_len, "solveLinearSystemPCG: component labelling did not converge"); return -4; }
    }
    if (h_cnt[2] > 0) {   // generic/tfluids.cu:1083-1091 raises for a fluid cell on the border
      snprintf(msg, msg_len, "solveLinearSystemPCG: Non fluid cell found in a connected component or fluid cell found on the "
               "domain border (%d fluid border cells)", h_cnt[2]);
      return -1;
    }
    { TFL_TIMED("k_cc", st); k_cc_count<<<gcell, 256, 0, st>>>(d, label, size_at, roots, counters + 1); }
    if (!hip_ok(hipMemcpyAsync(h_cnt, counters, 3 * sizeof(int), hipMemcpyDeviceToHost, st), "memcpy")) return -4;
    if (!hip_ok(hipStreamSynchronize(st), "sync")) return -4;
    const int ncomp = h_cnt[1];
    if (ncomp > kMaxComponents) {
      snprintf(msg, msg_len, "solveLinearSystemPCG: %d fluid components (the solver handles %d)", ncomp, kMaxComponents);
      return -3;
    }
    std::vector<int> h_roots(ncomp), h_sizes(ncomp);
    if (ncomp) {
      if (!hip_ok(hipMemcpy(h_roots.data(), roots, sizeof(int) * ncomp, hipMemcpyDeviceToHost), "memcpy roots")) return -4;
      std::sort(h_roots.begin(), h_roots.end());   // scan order of the first cell = the reference's numbering
      for (int cidx = 0; cidx < ncomp; cidx++)
        if (!hip_ok(hipMemcpy(&h_sizes[cidx], size_at + h_roots[cidx], sizeof(int), hipMemcpyDeviceToHost), "memcpy size")) return -4;
    }
    // ---- one CG solve per component -------------------------------------------------------------------
    for (int cidx = 0; cidx < ncomp; cidx++) {
      const int root = h_roots[cidx], size = h_sizes[cidx];
      if (size == 1) {   // :1375-1383: no valid solution, pressure stays 0
        if (verbose) printf("PCG batch %d component %d has size 1, skipping.\n", b + 1, cidx + 1);
        continue;
      }
      int pc = precond;
      if (size < 5) pc = 0;   // :1390-1393
      if (verbose) printf("PCG batch %d component %d has size %d.\nPCG: %d component %d using precond type %s\n", b + 1, cidx + 1,
                          size, b + 1, cidx + 1, pc == 0 ? "none" : (pc == 1 ? "ilu0" : "ic0"));
      // z is only ever written on the component; its other cells must read as 0 in s = z and r.z
      if (pc && !hip_ok(hipMemsetAsync(z, 0, sizeof(float) * (size_t)n, st), "memset z")) return -4;
      { TFL_TIMED("k_pcg", st);
        k_pcg_setup<<<kRedBlocks, 256, 0, st>>>(n, label, root, dv, x, r, s, partials);
        k_pcg_begin<<<1, 256, 0, st>>>(S, partials, tol, max_iter); }
      if (pc) {
        TFL_TIMED("k_pcg_precond", st);
        for (int h = hmin; h <= hmax; h++) {
          if (is3d) k_ilu_factor<true><<<gwave, 256, 0, st>>>(d, h, fl, label, root, dg);
          else k_ilu_factor<false><<<gwave, 256, 0, st>>>(d, h, fl, label, root, dg);
        }
        if (wf) {
          const long long tot = wg.sub * wg.ns * wg.nb;
          if (!hip_ok(hipMemsetAsync(cs, 0, sizeof(float) * (size_t)(tot * 4 + wf_handoff_k(wg) + wf_handoff_s(wg) + 16), st), "memset skewed arrays")) return -4;
          epoch = 0;
          const int gb = cdiv((long long)(X - 2) * (Y - 2) * (Z - 2), 256);
          if (pc == 2) k_wf_build<true><<<gb, 256, 0, st>>>(wg, d, label, root, dg, cs);
          else k_wf_build<false><<<gb, 256, 0, st>>>(wg, d, label, root, dg, cs);
        }
      }
      PcgState hs;
      const int chunk = verbose ? 1 : (pc ? (wf ? 16 : 4) : 32);
      for (;;) {
        for (int it = 0; it < chunk; it++) {
          k_pcg_top<<<1, 256, 0, st>>>(S, p_rr, 1);
          const float* dir_src = r;
          if (pc && wf) {
            // M^-1 r as two launches (pipelined wavefronts)
            const int nblk = wg.ns * wg.nb;
            const int gb = cdiv((long long)(X - 2) * (Y - 2) * (Z - 2), 256);
            { TFL_TIMED("k_pcg_precond", st);
              k_wf_skew<true><<<gb, 256, 0, st>>>(S, wg, d, r, rsk);
              // every sub-box of a launch must be resident (they wait for each other): at most wf_cap of them per launch,
              // all strips of a range of slabs, the ranges in sweep order -- a later launch finds its predecessors' pairs
              // waiting under the same tag
              ++epoch;
              for (int b0 = 0; b0 < wg.nb; b0 += wf_slabs) {
                WfGeom c = wg; c.b_lo = b0; c.b_cnt = std::min(wf_slabs, wg.nb - b0);
                k_wf_sweep<1><<<wg.ns * c.b_cnt, (kWfPlanes + 2) * 64, 0, st>>>(S, c, WfArrays{cs, rsk, qsk, wf_hk, wf_hs, wferr, wf_trace}, epoch);
              }
              ++epoch;
              for (int b1 = wg.nb; b1 > 0; b1 -= wf_slabs) {
                WfGeom c = wg; c.b_lo = std::max(b1 - wf_slabs, 0); c.b_cnt = b1 - c.b_lo;
                k_wf_sweep<-1><<<wg.ns * c.b_cnt, (kWfPlanes + 2) * 64, 0, st>>>(S, c, WfArrays{cs, qsk, zsk, wf_hk, wf_hs, wferr, wf_trace ? wf_trace + 2 * nblk : nullptr}, epoch);
              }
              k_wf_skew<false><<<gb, 256, 0, st>>>(S, wg, d, z, zsk); }
            dir_src = z;
          } else if (pc) {
            TFL_TIMED("k_pcg_precond", st);
            for (int h = hmin; h <= hmax; h++) {
              if (is3d) { if (pc == 2) k_ilu_forward<true, true><<<gwave, 256, 0, st>>>(S, d, h, fl, label, root, dg, r, y);
                          else k_ilu_forward<true, false><<<gwave, 256, 0, st>>>(S, d, h, fl, label, root, dg, r, y); }
              else { if (pc == 2) k_ilu_forward<false, true><<<gwave, 256, 0, st>>>(S, d, h, fl, label, root, dg, r, y);
                     else k_ilu_forward<false, false><<<gwave, 256, 0, st>>>(S, d, h, fl, label, root, dg, r, y); }
            }
            for (int h = hmax; h >= hmin; h--) {
              if (is3d) { if (pc == 2) k_ilu_backward<true, true><<<gwave, 256, 0, st>>>(S, d, h, fl, label, root, dg, y, z);
                          else k_ilu_backward<true, false><<<gwave, 256, 0, st>>>(S, d, h, fl, label, root, dg, y, z); }
              else { if (pc == 2) k_ilu_backward<false, true><<<gwave, 256, 0, st>>>(S, d, h, fl, label, root, dg, y, z);
                     else k_ilu_backward<false, false><<<gwave, 256, 0, st>>>(S, d, h, fl, label, root, dg, y, z); }
            }
            dir_src = z;
          }
          TFL_TIMED("k_pcg", st);
          if (pc) k_pcg_dot<<<kRedBlocks, 256, 0, st>>>(S, n, r, z, partials);
          k_pcg_dir<<<kRedBlocks, 256, 0, st>>>(S, n, partials, pc ? 1 : 0, dir_src, s);
          if (is3d) k_pcg_apply<true><<<kRedBlocks, 256, 0, st>>>(S, d, fl, label, root, s, w, p_den);
          else k_pcg_apply<false><<<kRedBlocks, 256, 0, st>>>(S, d, fl, label, root, s, w, p_den);
          k_pcg_update<<<kRedBlocks, 256, 0, st>>>(S, n, p_den, s, w, x, r, p_rr);
        }
        { TFL_TIMED("k_pcg", st); k_pcg_top<<<1, 256, 0, st>>>(S, p_rr, 0); }     // fold the last update's residual for the host
        if (!hip_ok(hipMemcpyAsync(&hs, S, sizeof(PcgState), hipMemcpyDeviceToHost, st), "memcpy state")) return -4;
        if (!hip_ok(hipStreamSynchronize(st), "sync")) return -4;
        if (verbose && !hs.done)
          printf("PCG batch %d comp %d iter %d: residual %g (tol %g)\n", b + 1, cidx + 1, hs.iter, std::sqrt(hs.rr1), (double)tol);
        if (wf && pc) {
          int e = 0;
          if (!hip_ok(hipMemcpy(&e, wferr, sizeof(int), hipMemcpyDeviceToHost), "memcpy")) return -4;
          static const bool pretend = exp_env("TFL_WF_TEST_TIMEOUT") != nullptr;     // tests: exercise the caller's fallback
          if (pretend) e = 1;
          // (its sub-boxes wait for each other, so all of them must be resident at once: not the case when something else holds
          // part of the GPU. The caller repeats the solve with one launch per hyperplane.)
          if (e) { snprintf(msg, msg_len, "solveLinearSystemPCG: a block of the pipelined triangular solve never saw its predecessor"); return -5; }
        }
        if (hs.bad) { snprintf(msg, msg_len, "solveLinearSystemPCG: ERROR: r_norm_sq1 is nan!"); return -2; }
        // the loop ends when the NEXT top-of-loop test fails
        if (hs.done || !((float)hs.rr1 > tol * tol) || hs.iter > max_iter) break;
      }
      max_res = std::max(max_res, (float)std::sqrt((float)hs.rr1));
      { TFL_TIMED("k_pcg", st);
        k_pcg_sum<<<kRedBlocks, 256, 0, st>>>(n, x, partials);
        k_pcg_sum_end<<<1, 256, 0, st>>>(S, partials);
        k_pcg_write<<<kRedBlocks, 256, 0, st>>>(S, n, label, root, size, x, pb); }
    }
  }
  if (wf_trace) {
    const int nblk = wg.ns * wg.nb;
    std::vector<long long> t(4 * nblk);
    if (hipMemcpy(t.data(), wf_trace, sizeof(long long) * 4 * nblk, hipMemcpyDeviceToHost) == hipSuccess) {
      long long t0 = t[0];
      for (int i = 0; i < 4 * nblk; i++) if (t[i] && t[i] < t0) t0 = t[i];
      for (int dir = 0; dir < 2; dir++)
        for (int b2 = 0; b2 < nblk; b2++)
          fprintf(stderr, "[tfl] wf %s strip %d slab %2d: start %8.2f us  end %8.2f us\n", dir ? "bwd" : "fwd", b2 / wg.nb, b2 % wg.nb,
                  (t[dir * 2 * nblk + 2 * b2] - t0) * 0.01, (t[dir * 2 * nblk + 2 * b2 + 1] - t0) * 0.01);
    }
  }
  if (residual) *residual = max_res;
  return 0;
}

long long npm_workspace_floats(int Z, int Y, int X) { return 4ll * Z * Y * X + 16 + kMaxComponents; }

// normalizePressureMean for every batch element. workspace: sum_at (fp64, first for alignment), label, size_at, counters.
int normalize_pressure_mean(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* p, const float* flags,
                            float* workspace, char* msg, size_t msg_len) {
  const Dom d = make_dom(Z, Y, X);
  const long long n = d.sc;
  double* sum_at = reinterpret_cast<double*>(workspace);
  int* label = reinterpret_cast<int*>(workspace + 2 * n);
  int* size_at = reinterpret_cast<int*>(workspace + 3 * n);
  int* counters = reinterpret_cast<int*>(workspace + 4 * n);    // [0] changed, [1] count, [2] border fluid, [3..] roots sink
  auto hip_ok = [&](hipError_t e, const char* what) {
    if (e == hipSuccess) return true;
    snprintf(msg, msg_len, "normalizePressureMean: %s: %s", what, hipGetErrorString(e));
    return false;
  };
  const int gcell = cdiv(n, 256);
  for (int b = 0; b < B; b++) {
    const float* fl = flags + (long long)b * n;
    float* pb = p + (long long)b * n;
    int h_cnt[3] = {0, 0, 0};
    if (!hip_ok(hipMemsetAsync(counters, 0, 3 * sizeof(int), st), "memset")) return -4;
    if (!hip_ok(hipMemsetAsync(size_at, 0, sizeof(int) * (size_t)n, st), "memset")) return -4;
    if (!hip_ok(hipMemsetAsync(sum_at, 0, sizeof(double) * (size_t)n, st), "memset")) return -4;
    TFL_TIMED("k_cc", st);
    k_cc_init<<<gcell, 256, 0, st>>>(d, is3d, fl, label, counters + 2);
    for (int round = 0;; round++) {
      if (!hip_ok(hipMemsetAsync(counters, 0, sizeof(int), st), "memset")) return -4;
      k_cc_hook<<<gcell, 256, 0, st>>>(d, is3d, label, counters);
      k_cc_compress<<<gcell, 256, 0, st>>>(d, label);
      if (!hip_ok(hipMemcpyAsync(h_cnt, counters, 3 * sizeof(int), hipMemcpyDeviceToHost, st), "memcpy")) return -4;
      if (!hip_ok(hipStreamSynchronize(st), "sync")) return -4;
      if (!h_cnt[0]) break;
      if (round > 100000) { snprintf(msg, msg_len, "normalizePressureMean: component labelling did not converge"); return -4; }
    }
    // sizes only (the roots list is not needed: means are looked up through the label)
    k_cc_count<<<gcell, 256, 0, st>>>(d, label, size_at, counters + 3, counters + 1);
    k_npm_sum<<<gcell, 256, 0, st>>>(d, label, pb, sum_at);
    k_npm_sub<<<gcell, 256, 0, st>>>(d, label, size_at, sum_at, pb);
  }
  return 0;
}

}  // namespace tfl
