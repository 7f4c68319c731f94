// model.hip -- the ConvNet pressure projection of FluidNet's `default` model (gfx950), everything
// except the convolution layers themselves (conv.hip).
//
// Replaces the nngraph of torch/lib/model.lua:27-401 (forward only):
//   SetWallBcs(UDiv) -> VelocityDivergence -> scale = std(UDiv_bc) -> {pDiv, div}/scale, occupancy
//   -> conv stack -> VelocityUpdate(pPred, UDiv_bc/scale) -> p, U *= scale -> SetWallBcs(U)
// The reference runs this as ~25 cuDNN / THC / tfluids launches with dense temporaries per node
// (mask tensors, JoinTable copies, ...). Here the non-conv nodes are three HBM-bound kernels:
//   k_bcs_div_stats : U, flags -> U_bc (into the caller's U output buffer), div, and the per-sample
//                     sum(u), sum(u^2) of U_bc accumulated in fp64 (one atomic pair per block)
//   k_net_input     : pDiv, div, flags, stats -> the 3 input planes {pDiv/scale, div/scale, occ}
//   k_project       : pPred, U_bc, flags, stats -> p = pPred*scale, U = SetWallBcs(VelocityUpdate(
//                     U_bc/scale, pPred) * scale), optionally fused with simulate()'s trailing
//                     setConstVals + clamp (lib/simulate.lua:321-326)
#include "tfl_device.hpp"
#include "tfl_fastmath.hpp"
#include "tfl_host.hpp"
#include "tfl_vec4.hpp"

#include <algorithm>
#include <cstdlib>

// minimum waves per SIMD the register allocator has to leave room for (tools/ab_build.sh -DTFL_LB_...=n for A/B runs)
#ifndef TFL_LB_BCS
#define TFL_LB_BCS 1
#endif
#ifndef TFL_LB_PROJECT
#define TFL_LB_PROJECT 1
#endif

namespace tfl {

// setWallBcs decision for the cell's own three face components, third_party/tfluids.cc:926-1002, as a pure
// function of the cell's flag word and its six neighbours' (0 where the neighbour is outside the grid).
template <bool IS3D>
__device__ __forceinline__ void wall_mask_from(int fc, int fxm, int fxp, int fym, int fyp, int fzm, int fzp, bool& zx,
                                               bool& zy, bool& zz) {
  zx = zy = zz = false;
  const bool cf = fc & kFluid, co = fc & kObstacle;
  if (!cf && !co) return;
  zx = (fxm & kObstacle) || (co && (fxm & kFluid));
  zy = (fym & kObstacle) || (co && (fym & kFluid));
  zz = IS3D && ((fzm & kObstacle) || (co && (fzm & kFluid)));
  if (cf) {
    if ((fxm & kStick) || (fxp & kStick)) { zy = true; zz = IS3D; }
    if ((fym & kStick) || (fyp & kStick)) { zx = true; zz = IS3D; }
    if (IS3D && ((fzm & kStick) || (fzp & kStick))) { zx = true; zy = true; }
  }
}

template <bool IS3D>
__device__ __forceinline__ void wall_zero_mask(const Dom& d, const float* __restrict__ flags, int i, int j, int k, int o,
                                               bool& zx, bool& zy, bool& zz) {
  const int fc = (int)flags[o];
  if (!(fc & (kFluid | kObstacle))) { zx = zy = zz = false; return; }
  const int fxm = i > 0 ? (int)flags[o - 1] : 0;
  const int fym = j > 0 ? (int)flags[o - d.sy] : 0;
  const int fzm = (IS3D && k > 0) ? (int)flags[o - d.sz] : 0;
  int fxp = 0, fyp = 0, fzp = 0;
  if (fc & kFluid) {
    fxp = i < d.X - 1 ? (int)flags[o + 1] : 0;
    fyp = j < d.Y - 1 ? (int)flags[o + d.sy] : 0;
    fzp = (IS3D && k < d.Z - 1) ? (int)flags[o + d.sz] : 0;
  }
  wall_mask_from<IS3D>(fc, fxm, fxp, fym, fyp, fzm, fzp, zx, zy, zz);
}

// U_bc component AXIS of cell (i,j,k): the input velocity with the wall BCs applied on the fly
template <bool IS3D, int AXIS>
__device__ __forceinline__ float u_bc_at(const Dom& d, const float* __restrict__ U, const float* __restrict__ flags,
                                         int i, int j, int k) {
  const int o = TFL_AT(d, i, j, k);
  bool zx, zy, zz;
  wall_zero_mask<IS3D>(d, flags, i, j, k, o, zx, zy, zz);
  const bool z = AXIS == 0 ? zx : (AXIS == 1 ? zy : zz);
  return z ? 0.0f : U[o + AXIS * d.sc];
}

// stats[b] = the sum of `count` partial pairs at p, by the 256 threads of a block in a FIXED order (strided accumulate, then a
// shared-memory tree): the scale is bit-reproducible run to run and independent of how the launches were split. COHERENT:
// the partials were written by other blocks of the SAME launch (k_bcs_div_stats' fused tail below) -- loads that are
// coherent across the chip (sc1), as their stores were.
template <bool COHERENT>
__device__ __forceinline__ void reduce_partials(const double* __restrict__ p, long long count, double* __restrict__ out, int tid) {
  __shared__ double sh[512];
  if (!COHERENT) {      // the plain form shares its summation order with the first conv layer's own reduction (tfl_device.hpp)
    double r1, r2;
    block_sum_pairs(p, count, sh, tid, r1, r2);
    if (tid == 0) { out[0] = r1; out[1] = r2; }
    return;
  }
  double s1 = 0.0, s2 = 0.0;
  for (long long t = tid; t < count; t += 256) {
    s1 += __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p + t * 2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    s2 += __builtin_bit_cast(double, __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p + t * 2 + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  }
  double* sh1 = sh; double* sh2 = sh + 256;
  sh1[tid] = s1; sh2[tid] = s2;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) { sh1[tid] += sh1[tid + w]; sh2[tid] += sh2[tid + w]; }
    __syncthreads();
  }
  if (tid == 0) { out[0] = sh1[0]; out[1] = sh2[0]; }
  __syncthreads();
}

// The reduction of the partials folded into the launch that produces them (round 5): every block
// publishes its pair with chip-coherent stores, waits for their acknowledgement and takes a ticket; the block that draws the
// last one reduces all pairs in k_reduce_stats' order (the same bits) and re-arms the counter. Only for a launch that covers
// the whole array (tfl_model_forward); a z-slab rank's windowed launches keep the separate k_reduce_stats.
// Two levels of tickets: ticket[1 + plane] counts the blocks of a (batch item, plane), ticket[0] the planes that are complete
// -- one counter for all blocks serialised 2048 same-address atomics at 128^3 (+16 us on a 19 us kernel, measured); with
// one counter per plane the atomics of different planes proceed in parallel and only 128 meet on the last one.
struct StatTail { unsigned* ticket; double* stats; long long per_sample; unsigned per_plane, planes; int B; };
[[maybe_unused]] constexpr int kStatTickets = 1 << 16;       // ticket words a model owns (abi.cpp): B * Z + 1 of them are used
// (FOLD is a template flag of the kernels: the default instantiation carries none of this -- its loads stay one batch,
// tests/test_isa_cpu.py)
template <bool FOLD>
__device__ __forceinline__ void publish_and_maybe_reduce(const StatTail& tl, double* __restrict__ partials, long long blk, double p1,
                                                         double p2, int tid) {
  if (!FOLD) {
    if (tid == 0) { partials[blk * 2] = p1; partials[blk * 2 + 1] = p2; }
    return;
  }
  __shared__ int last;
  if (tid == 0) {
    if (tl.ticket) {
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(partials + blk * 2), __builtin_bit_cast(unsigned long long, p1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(reinterpret_cast<unsigned long long*>(partials + blk * 2 + 1), __builtin_bit_cast(unsigned long long, p2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // both pairs have reached the coherence point before the ticket is drawn
      unsigned* mine = tl.ticket + 1 + (unsigned)(blk / tl.per_plane);
      last = 0;
      if (atomicAdd(mine, 1u) == tl.per_plane - 1u) {       // this plane is complete (its counter re-armed by the one block that sees that)
        *mine = 0u;
        last = atomicAdd(tl.ticket, 1u) == tl.planes - 1u;
      }
    } else {
      partials[blk * 2] = p1; partials[blk * 2 + 1] = p2;
      last = 0;
    }
  }
  if (!tl.ticket) return;             // uniform
  __syncthreads();
  if (!last) return;                  // block-uniform
  for (int b = 0; b < tl.B; b++) reduce_partials<true>(partials + (long long)b * tl.per_sample * 2, tl.per_sample, tl.stats + b * 2, tid);
  if (tid == 0) *tl.ticket = 0u;      // re-armed for the next launch (which starts behind this one on the stream)
}

// partials[block*2 + {0,1}] = this block's sum u, sum u^2 of U_bc (fp64). A second tiny kernel
// (k_reduce_stats) adds the partials of each sample in a fixed order, so the scale is bit-reproducible
// run to run and independent of how the grid is sharded -- no atomics (8192 same-address fp64 atomics
// cost 0.2 ms at 128^3, 10x the kernel itself).
template <bool IS3D, bool FOLD = false>
__global__ __launch_bounds__(256) void k_bcs_div_stats(Dom d, const float* __restrict__ U, const float* __restrict__ flags,
                                                       float* __restrict__ Ubc, float* __restrict__ div,
                                                       double* __restrict__ partials, StatTail tl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  double s1 = 0.0, s2 = 0.0;
  if (i < d.X && j < d.Y) {
    U += b * cells * C; Ubc += b * cells * C; flags += b * cells; div += b * cells;
    const int o = TFL_AT(d, i, j, k);
    bool zx, zy, zz;
    wall_zero_mask<IS3D>(d, flags, i, j, k, o, zx, zy, zz);
    const float ux = zx ? 0.0f : U[o];
    const float uy = zy ? 0.0f : U[o + d.sc];
    const float uz = IS3D ? (zz ? 0.0f : U[o + 2 * d.sc]) : 0.0f;
    Ubc[o] = ux; Ubc[o + d.sc] = uy; if (IS3D) Ubc[o + 2 * d.sc] = uz;
    s1 = (double)ux + (double)uy + (double)uz;
    s2 = (double)ux * ux + (double)uy * uy + (double)uz * uz;
    float dv = 0.0f;  // velocityDivergenceForward on U_bc, tfluids.cc:1008-1066
    if (!on_border<IS3D>(d, i, j, k) && (((int)flags[o]) & kFluid)) {
      dv = ux - u_bc_at<IS3D, 0>(d, U, flags, i + 1, j, k) + uy - u_bc_at<IS3D, 1>(d, U, flags, i, j + 1, k);
      if (IS3D) dv += (uz - u_bc_at<IS3D, 2>(d, U, flags, i, j, k + 1));
    }
    div[o] = dv;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
  __shared__ double part[8];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if ((tid & 63) == 0) { part[(tid >> 6) * 2] = s1; part[(tid >> 6) * 2 + 1] = s2; }
  __syncthreads();
  const long long blk = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * ((long long)b * d.Z + k));
  publish_and_maybe_reduce<FOLD>(tl, partials, blk, (part[0] + part[2]) + (part[4] + part[6]), (part[1] + part[3]) + (part[5] + part[7]), tid);
}

// k_bcs_div_stats with four x-cells per thread (tfl_vec4.hpp). The wall-BC masks of the cell's +x / +y / +z
// neighbours (needed for the divergence of U_bc) are rebuilt in registers from the flag rows already
// loaded for the cell's own mask plus four more rows (the stick test of the +y / +z neighbour looks two rows
// away); U_bc.x of cell i0+4 comes from the next lane. ~26 vector loads per 4 cells instead of ~120 dword loads.
// ---- round 6: the wall code of a cell -- the setWallBcs decision for its three face components and "is fluid" in ONE byte, a
// pure function of the flags (wall_mask_from of the cell's word and its six neighbours'): bits 0 / 1 / 2 = zero u_x / u_y / u_z,
// bit 3 = fluid (and, for k_project, what velocityUpdateForward asks of the cell and its three minus-neighbours: 16 bits in all). The flags of a scene are set once; k_bcs_div_stats evaluates wall_mask_from three times per cell per step from ten
// rows of flag words (and is, with 650 vector instructions per wave, a flag-decoding kernel more than a streaming one): with a
// tfl_wall_plan (tfl_wall_plan_create: this kernel, once) it reads three rows of bytes instead. Same decisions: same bits.
template <bool IS3D>
__global__ __launch_bounds__(256) void k_wall_code(Dom d, const float* __restrict__ flags, unsigned short* __restrict__ code) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  if (i >= d.X || j >= d.Y) return;
  flags += b * d.sc; code += b * d.sc;
  const int o = TFL_AT(d, i, j, k);
  const int fc = (int)flags[o];
  const int fxm = i > 0 ? (int)flags[o - 1] : 0, fxp = i < d.X - 1 ? (int)flags[o + 1] : 0;
  const int fym = j > 0 ? (int)flags[o - d.sy] : 0, fyp = j < d.Y - 1 ? (int)flags[o + d.sy] : 0;
  const int fzm = (IS3D && k > 0) ? (int)flags[o - d.sz] : 0, fzp = (IS3D && k < d.Z - 1) ? (int)flags[o + d.sz] : 0;
  bool zx, zy, zz;
  wall_mask_from<IS3D>(fc, fxm, fxp, fym, fyp, fzm, fzp, zx, zy, zz);
  // bits 0-2: zero u_x / u_y / u_z (setWallBcs); 3: fluid; 4: empty and not outflow; 5-7: the -x / -y / -z neighbour is fluid;
  // 8-10: it is empty (what velocityUpdateForward asks of the flags, tfluids.cc:1072-1156)
  unsigned m = (zx ? 1u : 0u) | (zy ? 2u : 0u) | (zz ? 4u : 0u) | ((fc & kFluid) ? 8u : 0u) | (((fc & kEmpty) && !(fc & kOutflow)) ? 16u : 0u);
  m |= ((fxm & kFluid) ? 32u : 0u) | ((fym & kFluid) ? 64u : 0u) | ((fzm & kFluid) ? 128u : 0u);
  m |= ((fxm & kEmpty) ? 256u : 0u) | ((fym & kEmpty) ? 512u : 0u) | ((fzm & kEmpty) ? 1024u : 0u);
  code[o] = (unsigned short)m;
}

// k_bcs_div_stats_v4 on wall codes: the same loads of U, the same arithmetic and summation order, the same stores -- the flag
// rows replaced by the code bytes of the cell's row, the row above (y + 1) and the plane above (z + 1)
template <bool IS3D>
__global__ __launch_bounds__(256, TFL_LB_BCS) void k_bcs_div_stats_code(Dom d, const float* __restrict__ U, const unsigned short* __restrict__ code,
                                                                       float* __restrict__ Ubc, float* __restrict__ div,
                                                                       double* __restrict__ partials, StatTail tl) {
  const V4Ctx c = v4_ctx(d);
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  const bool live = c.i0 < d.X && j < d.Y;
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  U += b * cells * C; Ubc += b * cells * C; code += b * cells; div += b * cells;
  const int o = TFL_AT(d, c.i0, j, k);
  const bool yp = live && j < d.Y - 1, zp = live && IS3D && k < d.Z - 1;
  // unconditional loads (tfl_vec4.hpp): a lane that must not read takes word 0 and drops it
  const unsigned long long cc_v = *reinterpret_cast<const unsigned long long*>(code + (live ? o : 0));      // four 16-bit codes
  const unsigned long long cy_v = *reinterpret_cast<const unsigned long long*>(code + (yp ? o + d.sy : 0));
  const unsigned long long cz_v = *reinterpret_cast<const unsigned long long*>(code + (zp ? o + d.sz : 0));
  const unsigned long long cc = live ? cc_v : 0ull, cy = yp ? cy_v : 0ull, cz = zp ? cz_v : 0ull;
  float u[3][4], uyp[4], uzp[4];
#pragma unroll
  for (int a = 0; a < 3; a++) v4_load(U, o + a * d.sc, live && a < C, 0.0f, u[a]);
  v4_load(U, o + d.sc + d.sy, yp, 0.0f, uyp);
  v4_load(U, o + 2 * d.sc + d.sz, zp, 0.0f, uzp);
  const bool need = c.last && live && c.has_r;
  const int oo = o + 4;
  const unsigned gcode = code[need ? oo : 0];
  const float gu = U[need ? oo : 0];
  double s1 = 0.0, s2 = 0.0;
  float ubx[5];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const unsigned m = (unsigned)(cc >> (16 * q));
    if (m & 1u) u[0][q] = 0.0f;
    if (m & 2u) u[1][q] = 0.0f;
    if (!IS3D || (m & 4u)) u[2][q] = 0.0f;
    ubx[q] = u[0][q];
    if (live) {
      s1 += (double)u[0][q] + (double)u[1][q] + (double)u[2][q];
      s2 += (double)u[0][q] * u[0][q] + (double)u[1][q] * u[1][q] + (double)u[2][q] * u[2][q];
    }
  }
  ubx[4] = from_lane_above(ubx[0]);
  if (c.last) ubx[4] = (need && !(gcode & 1u)) ? gu : 0.0f;
  float dv[4];
  const bool row_inner = live && j >= 1 && j <= d.Y - 2 && (!IS3D || (k >= 1 && k <= d.Z - 2));
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = c.i0 + q;
    dv[q] = 0.0f;
    if (row_inner && i >= 1 && i <= d.X - 2 && ((unsigned)(cc >> (16 * q)) & 8u)) {   // tfluids.cc:1008-1066 on U_bc
      const float by = ((unsigned)(cy >> (16 * q)) & 2u) ? 0.0f : uyp[q];
      float t = u[0][q] - ubx[q + 1] + u[1][q] - by;
      if (IS3D) {
        const float bz = ((unsigned)(cz >> (16 * q)) & 4u) ? 0.0f : uzp[q];
        t += (u[2][q] - bz);
      }
      dv[q] = t;
    }
  }
  if (live) {
#pragma unroll
    for (int a = 0; a < 3; a++)
      if (a < C) v4_store(Ubc, o + a * d.sc, u[a]);
    v4_store(div, o, dv);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
  __shared__ double part[8];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if ((tid & 63) == 0) { part[(tid >> 6) * 2] = s1; part[(tid >> 6) * 2 + 1] = s2; }
  __syncthreads();
  const long long blk = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * ((long long)b * d.Z + k));
  publish_and_maybe_reduce<false>(tl, partials, blk, (part[0] + part[2]) + (part[4] + part[6]), (part[1] + part[3]) + (part[5] + part[7]), tid);
}

template <bool IS3D, bool FOLD = false>
__global__ __launch_bounds__(256, TFL_LB_BCS) void k_bcs_div_stats_v4(Dom d, const float* __restrict__ U, const float* __restrict__ flags,
                                                          float* __restrict__ Ubc, float* __restrict__ div,
                                                          double* __restrict__ partials, StatTail tl) {
  const V4Ctx c = v4_ctx(d);
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  const bool live = c.i0 < d.X && j < d.Y;
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  U += b * cells * C; Ubc += b * cells * C; flags += b * cells; div += b * cells;
  const int o = TFL_AT(d, c.i0, j, k);
  const bool ym = live && j > 0, yp = live && j < d.Y - 1, yp2 = live && j < d.Y - 2;
  const bool zm = live && IS3D && k > 0, zp = live && IS3D && k < d.Z - 1, zp2 = live && IS3D && k < d.Z - 2;
  // flag rows: own (6 wide), y-1, y+1 (6 wide), z-1, z+1 (6 wide); for the +y / +z neighbours' stick tests:
  // (y+2,z), (y+1,z-1), (y+1,z+1), (y,z+2), (y-1,z+1)
  float fc[6], fym[4], fyp[6], fzm[4], fzp[6], fyp2[4], fypzm[4], fypzp[4], fzp2[4], fymzp[4];
  v4_load6<true, true>(c, flags, o, live, 0.0f, fc);
  v4_load(flags, o - d.sy, ym, 0.0f, fym);
  v4_load6<true, true>(c, flags, o + d.sy, yp, 0.0f, fyp);
  v4_load(flags, o - d.sz, zm, 0.0f, fzm);
  v4_load6<true, true>(c, flags, o + d.sz, zp, 0.0f, fzp);
  v4_load(flags, o + 2 * d.sy, yp2, 0.0f, fyp2);
  v4_load(flags, o + d.sy - d.sz, yp && zm, 0.0f, fypzm);
  v4_load(flags, o + d.sy + d.sz, yp && zp, 0.0f, fypzp);
  v4_load(flags, o + 2 * d.sz, zp2, 0.0f, fzp2);
  v4_load(flags, o - d.sy + d.sz, ym && zp, 0.0f, fymzp);
  float u[3][4], uyp[4], uzp[4];
#pragma unroll
  for (int a = 0; a < 3; a++) v4_load(U, o + a * d.sc, live && a < C, 0.0f, u[a]);
  v4_load(U, o + d.sc + d.sy, yp, 0.0f, uyp);
  v4_load(U, o + 2 * d.sc + d.sz, zp, 0.0f, uzp);
  // what the last lane of a row segment needs of cell i0 + 4 (its wall-BC mask and U.x): issued here, with the row loads
  // (loads unconditional, tfl_vec4.hpp v4_load: every lane reads -- the lanes that need nothing, cell 0 of the field)
  const bool need = c.last && live && c.has_r;
  const int oo = o + 4;
  const float gym = flags[need && ym ? oo - d.sy : 0], gyp = flags[need && yp ? oo + d.sy : 0];
  const float gzm = flags[need && zm ? oo - d.sz : 0], gzp = flags[need && zp ? oo + d.sz : 0];
  const float gu = U[need ? oo : 0];
  // own cells
  double s1 = 0.0, s2 = 0.0;
  float ubx[5];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    bool zx, zy, zz;
    wall_mask_from<IS3D>((int)fc[q + 1], (int)fc[q], (int)fc[q + 2], (int)fym[q], (int)fyp[q + 1], (int)fzm[q], (int)fzp[q + 1],
                         zx, zy, zz);
    if (zx) u[0][q] = 0.0f;
    if (zy) u[1][q] = 0.0f;
    if (!IS3D || zz) u[2][q] = 0.0f;
    ubx[q] = u[0][q];
    if (live) {
      s1 += (double)u[0][q] + (double)u[1][q] + (double)u[2][q];
      s2 += (double)u[0][q] * u[0][q] + (double)u[1][q] * u[1][q] + (double)u[2][q] * u[2][q];
    }
  }
  // U_bc.x of cell i0+4: the next lane's first cell, or (segment end) rebuilt from memory
  ubx[4] = from_lane_above(ubx[0]);
  {
    if (c.last) {
      float v = 0.0f;
      if (need) {
        const int f = (int)fc[5];
        bool zx, zy, zz;
        wall_mask_from<IS3D>(f, (int)fc[4], 0, ym ? (int)gym : 0, yp ? (int)gyp : 0, zm ? (int)gzm : 0, zp ? (int)gzp : 0, zx, zy, zz);
        v = zx ? 0.0f : gu;
      }
      ubx[4] = v;
    }
  }
  float dv[4];
  const bool row_inner = live && j >= 1 && j <= d.Y - 2 && (!IS3D || (k >= 1 && k <= d.Z - 2));
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = c.i0 + q;
    dv[q] = 0.0f;
    if (row_inner && i >= 1 && i <= d.X - 2 && (((int)fc[q + 1]) & kFluid)) {   // tfluids.cc:1008-1066 on U_bc
      bool zx, zy, zz;
      // +y neighbour (i, j+1, k): its -y neighbour is this cell
      wall_mask_from<IS3D>((int)fyp[q + 1], (int)fyp[q], (int)fyp[q + 2], (int)fc[q + 1], (int)fyp2[q], (int)fypzm[q],
                           (int)fypzp[q], zx, zy, zz);
      const float by = zy ? 0.0f : uyp[q];
      float t = u[0][q] - ubx[q + 1] + u[1][q] - by;
      if (IS3D) {
        // +z neighbour (i, j, k+1): its -z neighbour is this cell
        wall_mask_from<IS3D>((int)fzp[q + 1], (int)fzp[q], (int)fzp[q + 2], (int)fymzp[q], (int)fypzp[q], (int)fc[q + 1],
                             (int)fzp2[q], zx, zy, zz);
        const float bz = zz ? 0.0f : uzp[q];
        t += (u[2][q] - bz);
      }
      dv[q] = t;
    }
  }
  if (live) {
#pragma unroll
    for (int a = 0; a < 3; a++)
      if (a < C) v4_store(Ubc, o + a * d.sc, u[a]);
    v4_store(div, o, dv);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_down(s1, off, 64); s2 += __shfl_down(s2, off, 64); }
  __shared__ double part[8];
  const int tid = threadIdx.y * blockDim.x + threadIdx.x;
  if ((tid & 63) == 0) { part[(tid >> 6) * 2] = s1; part[(tid >> 6) * 2 + 1] = s2; }
  __syncthreads();
  const long long blk = blockIdx.x + (long long)gridDim.x * (blockIdx.y + (long long)gridDim.y * ((long long)b * d.Z + k));
  publish_and_maybe_reduce<FOLD>(tl, partials, blk, (part[0] + part[2]) + (part[4] + part[6]), (part[1] + part[3]) + (part[5] + part[7]), tid);
}

// stats[b*2 + 0] = sum u, stats[b*2 + 1] = sum u^2 over all C*Z*Y*X values of U_bc[b]; one block per
// sample, fixed summation order (strided accumulate, then a shared-memory tree).
__global__ __launch_bounds__(256) void k_reduce_stats(const double* __restrict__ partials, long long per_sample,
                                                      long long first, long long count, double* __restrict__ stats) {
  // [first, first + count) = the blocks of the z-planes that take part (a z-slab rank reduces only the
  // planes it owns; the halo planes belong to its neighbours)
  const int b = blockIdx.x;
  reduce_partials<false>(partials + ((long long)b * per_sample + first) * 2, count, stats + b * 2, (int)threadIdx.x);
}

// lib/modules/variance.lua:44-76 (n-1) + Sqrt; the Clamp after it is a no-op (model.lua:106 typo). One-pass form
// n*s2 - s1^2 in fp64 (the reference: two passes in fp32 THC reductions); the numerator is clamped at 0 because
// rounding can take it slightly negative for a near-constant U, where the two-pass variance is >= 0.
__device__ __forceinline__ float scale_from_stats(const double* __restrict__ stats, int b, double n) {
  const double s1 = stats[b * 2], s2 = stats[b * 2 + 1];
  return (float)sqrt(fmax(n * s2 - s1 * s1, 0.0) / (n * (n - 1.0)));
}

template <bool IS3D>
__global__ __launch_bounds__(256) void k_net_input(Dom d, const float* __restrict__ pDiv, const float* __restrict__ div,
                                                   const float* __restrict__ flags, const double* __restrict__ stats,
                                                   double count, float* __restrict__ x3) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  if (i >= d.X || j >= d.Y) return;
  const long long cells = d.sc;
  const float scale = scale_from_stats(stats, b, count);
  const int o = TFL_AT(d, i, j, k);
  const long long bo = b * cells;
  x3 += bo * 3;
  x3[o] = pDiv[bo + o] / scale;               // nn.ApplyScale(true) = CDivTable, apply_scale.lua:24-30
  x3[o + d.sc] = div[bo + o] / scale;
  const int f = (int)flags[bo + o];           // tfluids.FlagsToOccupancy, generic/tfluids.cu:355-371
  x3[o + 2 * d.sc] = (f == kFluid) ? 0.0f : ((f == kObstacle) ? 1.0f : -1.0f);
}

// The general net input (tfl_model_opts): model.lua:130-148's JoinTable of the selected fields.
template <bool IS3D>
__global__ __launch_bounds__(256) void k_net_input_gen(Dom d, int in_pDiv, int in_UDiv, int in_div, const float* __restrict__ pDiv,
                                                       const float* __restrict__ Ubc, const float* __restrict__ div,
                                                       const float* __restrict__ flags, const double* __restrict__ stats,
                                                       double count, float* __restrict__ x) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  if (i >= d.X || j >= d.Y) return;
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  const float scale = scale_from_stats(stats, b, count);
  const int o = TFL_AT(d, i, j, k);
  const long long bo = b * cells;
  x += bo * (in_pDiv + in_UDiv * C + in_div + 1);
  int ch = 0;
  if (in_pDiv) x[o + (ch++) * cells] = pDiv[bo + o] / scale;               // nn.ApplyScale(true) = CDivTable
  if (in_UDiv)
    for (int c = 0; c < C; c++) x[o + (ch++) * cells] = Ubc[bo * C + o + c * cells] / scale;
  if (in_div) x[o + (ch++) * cells] = div[bo + o] / scale;
  const int f = (int)flags[bo + o];           // tfluids.FlagsToOccupancy, generic/tfluids.cu:355-371
  x[o + ch * cells] = (f == kFluid) ? 0.0f : ((f == kObstacle) ? 1.0f : -1.0f);
}

__global__ __launch_bounds__(1024) void k_field_stats(long long n, const float* __restrict__ field, int mode,
                                                      double* __restrict__ stats) {
  const int b = blockIdx.x;
  const float* p = field + (long long)b * n;
  double s1 = 0.0, s2 = 0.0;
  if (mode != 2)
    for (long long t = threadIdx.x; t < n; t += 1024) { const double v = (double)p[t]; s1 += v; s2 += v * v; }
  __shared__ double sh1[1024], sh2[1024];
  sh1[threadIdx.x] = s1; sh2[threadIdx.x] = s2;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) { sh1[threadIdx.x] += sh1[threadIdx.x + w]; sh2[threadIdx.x] += sh2[threadIdx.x + w]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    stats[b * 2] = mode == 0 ? sh1[0] : 0.0;
    stats[b * 2 + 1] = mode == 2 ? 1.0 : sh2[0];
  }
}

__global__ __launch_bounds__(256) void k_skip_channel(long long cells, const float* __restrict__ pDiv,
                                                      const double* __restrict__ stats, double count,
                                                      float* __restrict__ dst, int och, int ch) {
  const int b = blockIdx.y;
  const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
  if (t >= cells) return;
  dst[((long long)b * och + ch) * cells + t] = pDiv[(long long)b * cells + t] / scale_from_stats(stats, b, count);
}

struct BcArgs {  // optional fused tail of simulate(): setConstVals + clamp, simulate.lua:321-326
  const float* UBC; const float* UInvMask;   // a dense pair: acts on every cell
  BcFoldArg fold;                            // or tfl_simulate_step's sparse pair with its box (tfl_host.hpp)
  int enable_clamp; float lo, hi;
  // the fp16 conv path's range-error count (device word, final once the conv kernels are done) and the pinned host word a
  // non-zero count is copied to, so that the host sees it at its next call without a stream sync (abi.cpp range_gate)
  const unsigned long long* range_src; unsigned long long* range_dst;
  // the z-slab step's reach word (sticky max|u_z|, device) and its mapped pinned mirror: the same thread copies it, every step
  // (round 6: an async 4-byte D2H copy on the stream blocks the host until the stream has drained on this stack)
  const float* reach_src; float* reach_dst;
  // round 6: non-null = every block also folds max |u_z| of the cells it WRITES into that word (one atomic per block, and only
  // where the block's maximum exceeds the word) -- the slab step's next reach check then needs no k_absmax launch of its own.
  // Only in launches whose blocks are all full (model_project decides): the block reduction has no dead lanes to care for.
  float* reach_acc;
  // round 6: the flags' tfl_wall_plan (16-bit codes: k_wall_code), or null -- k_project_v4<., true> reads ONE row of codes where
  // the plain form loads five rows of flag words and two single cells
  const unsigned short* wall_code;
  // round 6: the device count of reach publications; the publishing thread increments it and leaves the new value beside the
  // maximum in the pinned mirror (reach_dst[1]) -- what the host waits on instead of an event
  unsigned* reach_tick;
};
// one thread of the launch forwards a non-zero count (the host word is only written when something went wrong)
__device__ __forceinline__ void forward_range_count(const BcArgs& bc, bool first_thread) {
  if (bc.range_src && first_thread) {
    const unsigned long long v = *bc.range_src;
    if (v) *bc.range_dst = v;
  }
}
// ... and the z-slab reach word with its publication count -- at the END of the kernel (its loads and the fence must not sit
// between the kernel's own loads: tests/test_isa_cpu.py)
__device__ __forceinline__ void publish_reach(const BcArgs& bc, bool first_thread) {
  if (bc.reach_src && first_thread) {
    // (read through atomics: the word as every block's atomicMax has left it so far, past any cache line this CU may hold)
    const unsigned v = atomicOr(reinterpret_cast<unsigned*>(const_cast<float*>(bc.reach_src)), 0u);
    *reinterpret_cast<unsigned*>(bc.reach_dst) = v;
    if (bc.reach_tick) {
      const unsigned t = atomicAdd(bc.reach_tick, 1u) + 1u;       // (one launch at a time: the launches of a stream run in order)
      __threadfence_system();                       // the maximum is visible to the host before the count that announces it
      reinterpret_cast<volatile unsigned*>(bc.reach_dst)[1] = t;
    }
  }
}

template <bool IS3D>
__global__ __launch_bounds__(256) void k_project(Dom d, const float* __restrict__ pPred, const float* __restrict__ flags,
                                                 const double* __restrict__ stats, double count,
                                                 float* __restrict__ Uio, float* __restrict__ pOut, BcArgs bc) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  forward_range_count(bc, (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x | threadIdx.y) == 0);
  if (i >= d.X || j >= d.Y) return;
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  const float scale = scale_from_stats(stats, b, count);
  pPred += b * cells; flags += b * cells; pOut += b * cells; Uio += b * cells * C;
  const int o = TFL_AT(d, i, j, k);
  const float pc = pPred[o];
  float u[3];
  u[0] = Uio[o] / scale; u[1] = Uio[o + d.sc] / scale; u[2] = IS3D ? Uio[o + 2 * d.sc] / scale : 0.0f;
  if (!on_border<IS3D>(d, i, j, k)) {  // velocityUpdateForward, tfluids.cc:1072-1156
    const int fc = (int)flags[o];
    const int fn[3] = {(int)flags[o - 1], (int)flags[o - d.sy], IS3D ? (int)flags[o - d.sz] : 0};
    const int st[3] = {1, d.sy, d.sz};
    if (fc & kFluid) {
#pragma unroll
      for (int c = 0; c < C; c++) {
        if (fn[c] & kFluid) u[c] -= (pc - pPred[o - st[c]]);
        if (fn[c] & kEmpty) u[c] -= pc;
      }
    } else if ((fc & kEmpty) && !(fc & kOutflow)) {
#pragma unroll
      for (int c = 0; c < C; c++) u[c] = (fn[c] & kFluid) ? u[c] + pPred[o - st[c]] : 0.0f;
    }
  }
  bool z[3];
  wall_zero_mask<IS3D>(d, flags, i, j, k, o, z[0], z[1], z[2]);
  pOut[o] = pc * scale;  // nn.ApplyScale(false) = CMulTable, model.lua:383-387
#pragma unroll
  for (int c = 0; c < C; c++) {
    float v = z[c] ? 0.0f : u[c] * scale;
    const long long g = b * cells * C + o + c * d.sc;
    if (bc.UBC) v = v * bc.UInvMask[g] + bc.UBC[g];
    else if (bc.fold.dev) {
      const BcFold f = *bc.fold.dev;
      if (fold_row(f, j, k) && fold_col(f, i)) v = v * f.inv[g] + f.bc[g];
    }
    if (bc.enable_clamp) v = fminf(fmaxf(v, bc.lo), bc.hi);
    Uio[o + c * d.sc] = v;
  }
  publish_reach(bc, (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x | threadIdx.y) == 0);
}

// k_project with four consecutive x cells per thread (X % 4 == 0, 16-byte aligned rows): every access is
// a 16-byte vector, the x-1 / x+4 neighbours are single scalar loads. Same per-cell arithmetic (bit-exact).
__device__ __forceinline__ void unpack4(const float4 v, float* o) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; }

template <bool IS3D, bool CODE = false>
__global__ __launch_bounds__(256, TFL_LB_PROJECT) void k_project_v4(Dom d, const float* __restrict__ pPred, const float* __restrict__ flags,
                                                    const double* __restrict__ stats, double count,
                                                    float* __restrict__ Uio, float* __restrict__ pOut, BcArgs bc) {
  const int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  forward_range_count(bc, (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x | threadIdx.y) == 0);
  if (i0 >= d.X || j >= d.Y) return;
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  const float scale = scale_from_stats(stats, b, count);
  // the sticky reach word as it stands when the block starts (a uniform load, issued with the kernel's other loads: see the end)
  const float reach_seen = (IS3D && bc.reach_acc) ? *bc.reach_acc : 0.0f;
  pPred += b * cells; flags += b * cells; pOut += b * cells; Uio += b * cells * C;
  const int o = TFL_AT(d, i0, j, k);
  const float4 z4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  // every load is unconditional (tfl_vec4.hpp v4_load: a predicated load costs a drained load queue at its join -- 16 in a
  // row here before round 4): a lane that must not read takes the thread's own vector / cell instead and drops it
  auto ld4 = [&](const float* p, int off, bool ok) {
    const float4 v = *reinterpret_cast<const float4*>(p + (ok ? off : o));
    return ok ? v : z4;
  };
  // flags: the row itself (+ one cell either side) and the four neighbouring rows
  float fc[4], fym[4], fyp[4], fzm[4], fzp[4], pc[4], pym[4], pzm[4];
  float f_left = 0.0f, f_right = 0.0f;
  unsigned long long cw = 0ull;      // CODE: the four cells' 16-bit wall codes
  if (CODE) {
    cw = *reinterpret_cast<const unsigned long long*>(bc.wall_code + b * cells + o);
  } else {
    unpack4(ld4(flags, o, true), fc);
    unpack4(ld4(flags, o - d.sy, j > 0), fym);
    unpack4(ld4(flags, o + d.sy, j < d.Y - 1), fyp);
    unpack4(ld4(flags, o - d.sz, IS3D && k > 0), fzm);
    unpack4(ld4(flags, o + d.sz, IS3D && k < d.Z - 1), fzp);
    const float f_left_v = flags[i0 > 0 ? o - 1 : o], f_right_v = flags[i0 + 4 < d.X ? o + 4 : o];
    f_left = i0 > 0 ? f_left_v : 0.0f; f_right = i0 + 4 < d.X ? f_right_v : 0.0f;
  }
  unpack4(ld4(pPred, o, true), pc);
  unpack4(ld4(pPred, o - d.sy, j > 0), pym);
  unpack4(ld4(pPred, o - d.sz, IS3D && k > 0), pzm);
  const float p_left_v = pPred[i0 > 0 ? o - 1 : o];
  const float p_left = i0 > 0 ? p_left_v : 0.0f;
  float u[3][4];
#pragma unroll
  for (int c = 0; c < 3; c++) unpack4(ld4(Uio, o + c * d.sc, c < C), u[c]);
  float ubc[3][4], umk[3][4];
  bool bc_row = bc.UBC != nullptr;            // does a pair act on this thread's cells, and on which columns
  const float *pb = bc.UBC, *pm = bc.UInvMask;
  int bx0 = 0, bx1 = 0x7fffffff;
  if (!bc_row && fold_block(bc.fold, (int)(blockIdx.y * blockDim.y), (int)(blockIdx.y * blockDim.y + blockDim.y - 1), k, k)) {
    const BcFold f = *bc.fold.dev;
    bc_row = fold_row(f, j, k) && i0 <= f.x1 && i0 + 3 >= f.x0;
    pb = f.bc; pm = f.inv; bx0 = f.x0; bx1 = f.x1;
  }
  if (bc_row) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
      unpack4(ld4(pb + b * cells * C, o + c * d.sc, c < C), ubc[c]);
      unpack4(ld4(pm + b * cells * C, o + c * d.sc, c < C), umk[c]);
    }
  }
  float po[4];
  const bool row_border = j < 1 || j > d.Y - 2 || (IS3D && (k < 1 || k > d.Z - 2));
  const bool scale_ok = scale >= 0x1p-12f && scale <= 0x1p21f;      // (one batch item per block: uniform)
  const float inv_scale = scale_ok ? rcp_refined(scale) : 0.0f;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = i0 + q;
    const unsigned m = CODE ? (unsigned)(cw >> (16 * q)) : 0u;
    const int f = CODE ? 0 : (int)fc[q];
    const int fxm = CODE ? 0 : (q > 0 ? (int)fc[q - 1] : (int)f_left), fxp = CODE ? 0 : (q < 3 ? (int)fc[q + 1] : (int)f_right);
    const float pxm = q > 0 ? pc[q - 1] : p_left;
    // nn.ApplyScale(true) = CDivTable (apply_scale.lua:24-30). The twelve quotients of a thread share their denominator: one
    // refined reciprocal + an exact-remainder step each (tfl_fastmath.hpp div_by<1>: bit-equal to `/` for a scale in
    // [2^-12, 2^21] and |u| in [2^-40, 2^40], profiles/r03_exact_math.txt; u = -0 gives +0; round 6: the IEEE divisions were a
    // fifth of this kernel's 674 vector instructions per wave) -- `/` itself for a scale outside that range (block-uniform)
    float v[3];
    if (scale_ok) { v[0] = div_by<1>(u[0][q], scale, inv_scale); v[1] = div_by<1>(u[1][q], scale, inv_scale); v[2] = IS3D ? div_by<1>(u[2][q], scale, inv_scale) : 0.0f; }
    else { v[0] = u[0][q] / scale; v[1] = u[1][q] / scale; v[2] = IS3D ? u[2][q] / scale : 0.0f; }
    if (!(row_border || i < 1 || i > d.X - 2)) {   // velocityUpdateForward, tfluids.cc:1072-1156
      const int fn[3] = {fxm, CODE ? 0 : (int)fym[q], (!CODE && IS3D) ? (int)fzm[q] : 0};
      const float pn[3] = {pxm, pym[q], pzm[q]};
      // (CODE: the same questions answered by the cell's code -- bit 3 fluid, 4 empty and not outflow, 5 + c / 8 + c the
      // minus-neighbour along c is fluid / empty)
      const bool cell_fluid = CODE ? (m & 8u) != 0 : (f & kFluid) != 0;
      const bool cell_open = CODE ? (m & 16u) != 0 : ((f & kEmpty) && !(f & kOutflow));
      if (cell_fluid) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          const bool nf = CODE ? (m & (32u << c)) != 0 : (fn[c] & kFluid) != 0, ne = CODE ? (m & (256u << c)) != 0 : (fn[c] & kEmpty) != 0;
          if (nf) v[c] -= (pc[q] - pn[c]);
          if (ne) v[c] -= pc[q];
        }
      } else if (cell_open) {
#pragma unroll
        for (int c = 0; c < C; c++) {
          const bool nf = CODE ? (m & (32u << c)) != 0 : (fn[c] & kFluid) != 0;
          v[c] = nf ? v[c] + pn[c] : 0.0f;
        }
      }
    }
    bool z[3];
    if (CODE) { z[0] = m & 1u; z[1] = m & 2u; z[2] = m & 4u; }
    else wall_mask_from<IS3D>(f, fxm, fxp, (int)fym[q], (int)fyp[q], (int)fzm[q], (int)fzp[q], z[0], z[1], z[2]);
    po[q] = pc[q] * scale;
#pragma unroll
    for (int c = 0; c < C; c++) {
      float w = z[c] ? 0.0f : v[c] * scale;
      if (bc_row && i >= bx0 && i <= bx1) w = w * umk[c][q] + ubc[c][q];
      if (bc.enable_clamp) w = fminf(fmaxf(w, bc.lo), bc.hi);
      u[c][q] = w;
    }
  }
  *reinterpret_cast<float4*>(pOut + o) = make_float4(po[0], po[1], po[2], po[3]);
#pragma unroll
  for (int c = 0; c < 3; c++)
    if (c < C) *reinterpret_cast<float4*>(Uio + o + c * d.sc) = make_float4(u[c][0], u[c][1], u[c][2], u[c][3]);
  if (IS3D && bc.reach_acc) {       // (kernel-uniform; every thread of the block is here: model_project)
    float m = fmaxf(fmaxf(fabsf(u[2][0]), fabsf(u[2][1])), fmaxf(fabsf(u[2][2]), fabsf(u[2][3])));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float wm[4];
    const int tid = threadIdx.y * blockDim.x + threadIdx.x;
    if ((tid & 63) == 0) wm[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) {
      const float bm = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
      // (the word was read when the block started and can only have GROWN since: one atomic more, never one less; non-negative
      // floats order like their bits)
      if (bm > reach_seen) atomicMax(reinterpret_cast<unsigned int*>(bc.reach_acc), __float_as_uint(bm));
    }
  }
  publish_reach(bc, (blockIdx.x | blockIdx.y | blockIdx.z | threadIdx.x | threadIdx.y) == 0);
}

// x = clamp(x * invMask + bc): setConstVals (+ the final U:clamp) of lib/simulate.lua:130-160,326
__global__ __launch_bounds__(256) void k_apply_bcs(long long n, float* __restrict__ x, const float* __restrict__ bcv,
                                                   const float* __restrict__ inv, int do_clamp, float lo, float hi) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    float v = x[t];
    if (bcv) v = v * inv[t] + bcv[t];
    if (do_clamp) v = fminf(fmaxf(v, lo), hi);
    x[t] = v;
  }
}

// The same on an index list: BC tensors are dense in the reference API but almost everywhere the
// identity (invMask = 1, bc = 0: the plume touches 4 of 128 y-rows); the host caches the indices where
// they are not and only those elements are touched. Skipped elements satisfy x*1+0 == x.
__global__ __launch_bounds__(256) void k_apply_bcs_indexed(long long n, const int* __restrict__ idx, float* __restrict__ x,
                                                           const float* __restrict__ bcv, const float* __restrict__ inv) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int e = idx[t];
    x[e] = x[e] * inv[e] + bcv[e];
  }
}

// Several index lists in one launch: blockIdx.y picks the (x, bc, invMask, idx) tuple.
struct BcMultiArgs { long long n[8]; const int* idx[8]; float* x[8]; const float* bc[8]; const float* inv[8]; };
__global__ __launch_bounds__(256) void k_apply_bcs_indexed_multi(BcMultiArgs a) {
  const int f = blockIdx.y;
  const long long n = a.n[f];
  const int* __restrict__ idx = a.idx[f];
  float* __restrict__ x = a.x[f];
  const float* __restrict__ bcv = a.bc[f];
  const float* __restrict__ inv = a.inv[f];
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int e = idx[t];
    x[e] = x[e] * inv[e] + bcv[e];
  }
}

// tfl_bc_plan_create: the cells where a (bc, invMask) pair is not the identity. Pass 1 (idx == nullptr) counts,
// pass 2 fills; counters[0] = count, counters[1] = 1 if some listed cell has invMask != 0 or |bc| > 1e6 (the pair is
// then not idempotent / does not commute with the step's final clamp). The order of the list is irrelevant.
__global__ __launch_bounds__(256) void k_bc_scan(long long n, const float* __restrict__ bcv, const float* __restrict__ inv,
                                                 int* __restrict__ counters, int* __restrict__ idx) {
  for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (long long)gridDim.x * blockDim.x) {
    const float m = inv[e], b = bcv[e];
    if (m != 1.0f || b != 0.0f) {
      const int pos = atomicAdd(&counters[0], 1);
      if (idx) idx[pos] = (int)e;
      if (m != 0.0f || !(fabsf(b) <= 1e6f)) counters[1] = 1;
    }
  }
}

// Halo planes of several fields <-> one contiguous message buffer (z-slab decomposition): buffer layout
// [field][b][channel][that field's planes][Y][X]; every field has its own plane range (the halo depth a phase needs
// differs per field). One launch per message instead of a dozen strided copies.
struct PackArgs {
  float* ptr[8];
  float* buf[8];        // where the field's elements live in the message (one message may have several buffers: the
                        // lower and the upper neighbour's go out in ONE launch)
  long long start[9];   // prefix sums of elements per field over the launch
  long long per_row[8]; // elements per (b, channel) row of the field in the buffer = its planes * Y * X
  long long zoff[8];    // element offset of the field's first plane inside a (b, channel) row of the array
  int n;
};
__global__ __launch_bounds__(256) void k_pack_planes(PackArgs a, long long zstride, int unpack) {
  const long long total = a.start[a.n];
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (long long)gridDim.x * blockDim.x) {
    int f = 0;
#pragma unroll
    for (int q = 1; q < 8; q++) if (q < a.n && t >= a.start[q]) f = q;
    const long long r = t - a.start[f];
    const long long row = r / a.per_row[f], within = r - row * a.per_row[f];   // row = b*C + c
    float* g = a.ptr[f] + row * zstride + a.zoff[f] + within;
    if (unpack) *g = a.buf[f][r];
    else a.buf[f][r] = *g;
  }
}

#define TFL_GRID3(d, B) dim3(((d).X + 63) / 64, ((d).Y + 3) / 4, (unsigned)((d).nw * (B)))

long long model_stat_blocks(int B, int Z, int Y, int X) {
  return (long long)((X + 63) / 64) * ((Y + 3) / 4) * Z * B;
}

bool model_stats_fold_requested() {       // EXPERIMENTS flavour + TFL_STATS_FOLD=1: k_reduce_stats folded into the last block of k_bcs_div_stats
#ifdef TFL_EXPERIMENTS
  const char* ef = getenv("TFL_STATS_FOLD");
  return ef && atoi(ef) == 1;
#else
  return false;
#endif
}

long long model_stat_pairs_per_plane(int B, int Z, int Y, int X, const float* U, const float* flags, const float* Ubc, const float* div) {
  const Dom d = make_dom(Z, Y, X);
  const dim3 grd = TFL_GRID3(d, B);
  const Vec4Launch v = vec4_launch(B, Z, Y, X, {U, flags, Ubc, div});
  return v.ok ? (long long)v.grd.x * v.grd.y : (long long)grd.x * grd.y;
}

// stages: bit 0 = k_bcs_div_stats on the current z-window (per-plane partial sums land in absolute slots, so the
// launch may be split into boundary / interior windows), bit 1 = reduce the partials of planes [zlo, zhi) into stats
// the wall codes of a whole flags array (tfl_wall_plan_create; no z-window: the caller clears it)
void wall_code(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* flags, unsigned short* code) {
  const Dom d = make_dom(Z, Y, X);
  const dim3 blk(64, 4, 1), grd = TFL_GRID3(d, B);
  if (is3d) k_wall_code<true><<<grd, blk, 0, st>>>(d, flags, code);
  else k_wall_code<false><<<grd, blk, 0, st>>>(d, flags, code);
}

void model_pre(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* U, const float* flags, float* Ubc,
               float* div, double* partials, double* stats, int zlo, int zhi, int stages, unsigned* ticket, const unsigned short* code) {
  const Dom d = make_dom(Z, Y, X);
  const dim3 blk(64, 4, 1), grd = TFL_GRID3(d, B);
  const Vec4Launch v = vec4_launch(B, Z, Y, X, {U, flags, Ubc, div});
  const long long per_plane = v.ok ? (long long)v.grd.x * v.grd.y : (long long)grd.x * grd.y;   // <= model_stat_blocks / (Z*B)
  // both stages over the whole array in one call (tfl_model_forward): the launch CAN reduce its own partials
  // (publish_and_maybe_reduce) -- opt-in, TFL_STATS_FOLD=1: bit-identical, but measured slower than the 4-wave k_reduce_stats
  // launch it saves (profiles/r05_step_experiments.txt: k_bcs_div_stats 15.4 -> 23.5 us with two-level tickets, 35.6 with one
  // counter, against 4.2 us for the launch: the chip-coherent stores, their acknowledgement and the ticket sit at the end of
  // every one of the 2048 short blocks of a streaming kernel)
#ifdef TFL_EXPERIMENTS
  const bool fused = ticket && (stages & 3) == 3 && d.nw == Z && d.n0 == Z && zlo == 0 && zhi == Z && model_stats_fold_requested() &&
                     (long long)Z * B + 1 <= kStatTickets && per_plane < (1ll << 31);
#else
  const bool fused = false;      // (the producer-side fold exists only in the EXPERIMENTS flavour: measured slower, round 5)
  (void)ticket; (void)zlo; (void)zhi;
#endif
  StatTail tl = {nullptr, stats, per_plane * Z, (unsigned)per_plane, (unsigned)(Z * B), B};
  if (fused) tl.ticket = ticket;
  if (stages & 1) {
    if (v.ok && code && !fused) {      // round 6: the scene's wall codes instead of its flag words (tfl_wall_plan)
      TFL_TIMED_EXT("k_bcs_div_stats", st);
      if (is3d) TFL_LAUNCH_EXT((k_bcs_div_stats_code<true>), v.grd, v.blk, 0, st, d, U, code, Ubc, div, partials, tl);
      else TFL_LAUNCH_EXT((k_bcs_div_stats_code<false>), v.grd, v.blk, 0, st, d, U, code, Ubc, div, partials, tl);
    } else if (v.ok) {
      TFL_TIMED_EXT("k_bcs_div_stats", st);
#ifdef TFL_EXPERIMENTS
      if (fused) {
        if (is3d) TFL_LAUNCH_EXT((k_bcs_div_stats_v4<true, true>), v.grd, v.blk, 0, st, d, U, flags, Ubc, div, partials, tl);
        else TFL_LAUNCH_EXT((k_bcs_div_stats_v4<false, true>), v.grd, v.blk, 0, st, d, U, flags, Ubc, div, partials, tl);
      } else
#endif
      {
        if (is3d) TFL_LAUNCH_EXT((k_bcs_div_stats_v4<true, false>), v.grd, v.blk, 0, st, d, U, flags, Ubc, div, partials, tl);
        else TFL_LAUNCH_EXT((k_bcs_div_stats_v4<false, false>), v.grd, v.blk, 0, st, d, U, flags, Ubc, div, partials, tl);
      }
    } else {
      TFL_TIMED("k_bcs_div_stats", st);
#ifdef TFL_EXPERIMENTS
      if (fused) {
        if (is3d) k_bcs_div_stats<true, true><<<grd, blk, 0, st>>>(d, U, flags, Ubc, div, partials, tl);
        else k_bcs_div_stats<false, true><<<grd, blk, 0, st>>>(d, U, flags, Ubc, div, partials, tl);
      } else
#endif
      {
        if (is3d) k_bcs_div_stats<true, false><<<grd, blk, 0, st>>>(d, U, flags, Ubc, div, partials, tl);
        else k_bcs_div_stats<false, false><<<grd, blk, 0, st>>>(d, U, flags, Ubc, div, partials, tl);
      }
    }
  }
  if ((stages & 2) && !fused) { TFL_TIMED_EXT("k_reduce_stats", st); TFL_LAUNCH_EXT(k_reduce_stats, B, 256, 0, st, (const double*)partials, per_plane * Z, per_plane * zlo, per_plane * (zhi - zlo), stats); }
}

void model_net_input(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* pDiv, const float* div,
                     const float* flags, const double* stats, double count, float* x3) {
  const Dom d = make_dom(Z, Y, X);
  const dim3 blk(64, 4, 1), grd = TFL_GRID3(d, B);
  if (is3d) { TFL_TIMED("k_net_input", st); k_net_input<true><<<grd, blk, 0, st>>>(d, pDiv, div, flags, stats, count, x3); }
  else { TFL_TIMED("k_net_input", st); k_net_input<false><<<grd, blk, 0, st>>>(d, pDiv, div, flags, stats, count, x3); }
}

void model_net_input_gen(hipStream_t st, bool is3d, int B, int Z, int Y, int X, int in_pDiv, int in_UDiv, int in_div,
                         const float* pDiv, const float* Ubc, const float* div, const float* flags, const double* stats,
                         double count, float* x) {
  const Dom d = make_dom(Z, Y, X);
  const dim3 blk(64, 4, 1), grd = TFL_GRID3(d, B);
  TFL_TIMED("k_net_input", st);
  if (is3d) k_net_input_gen<true><<<grd, blk, 0, st>>>(d, in_pDiv, in_UDiv, in_div, pDiv, Ubc, div, flags, stats, count, x);
  else k_net_input_gen<false><<<grd, blk, 0, st>>>(d, in_pDiv, in_UDiv, in_div, pDiv, Ubc, div, flags, stats, count, x);
}

void model_field_stats(hipStream_t st, int B, long long n, const float* field, int mode, double* stats) {
  TFL_TIMED("k_field_stats", st);
  k_field_stats<<<B, 1024, 0, st>>>(n, field, mode, stats);
}

void model_skip_channel(hipStream_t st, int B, long long cells, const float* pDiv, const double* stats, double count,
                        float* dst, int och, int ch) {
  TFL_TIMED("k_skip_channel", st);
  k_skip_channel<<<dim3((unsigned)((cells + 255) / 256), (unsigned)B), 256, 0, st>>>(cells, pDiv, stats, count, dst, och, ch);
}

bool model_project(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* pPred, const float* flags,
                   const double* stats, double count, float* Uio, float* pOut, const float* UBC, const float* UInvMask,
                   int do_clamp, float lo, float hi, const unsigned long long* range_src, unsigned long long* range_dst,
                   const float* reach_src, float* reach_dst, float* reach_acc, const unsigned short* wall_code, unsigned* reach_tick) {
  const Dom d = make_dom(Z, Y, X);
  const dim3 blk(64, 4, 1), grd = TFL_GRID3(d, B);
  // a dense pair acts everywhere; without one, tfl_simulate_step's sparse pair (if it asked: tfl_host.hpp BcFold) in its box
  BcArgs bc; bc.enable_clamp = do_clamp; bc.lo = lo; bc.hi = hi;
  bc.range_src = range_dst ? range_src : nullptr; bc.range_dst = range_dst;
  bc.reach_src = reach_dst ? reach_src : nullptr; bc.reach_dst = reach_dst;
  bc.reach_acc = nullptr; bc.wall_code = nullptr; bc.reach_tick = bc.reach_src ? reach_tick : nullptr;
  bc.UBC = UBC; bc.UInvMask = UInvMask; bc.fold = UBC ? no_fold() : take_fold();
  const uintptr_t al = (uintptr_t)pPred | (uintptr_t)flags | (uintptr_t)Uio | (uintptr_t)pOut | (uintptr_t)UBC |
                       (uintptr_t)UInvMask;
  if (X % 4 == 0 && (al & 15) == 0 && !exp_env("TFL_NO_VEC4")) {
    const dim3 vb(32, 8, 1), vg((X / 4 + 31) / 32, (Y + 7) / 8, (unsigned)(d.nw * B));
    // the reach maximum rides along where every block of the launch is full (128 x 8 cells: no thread leaves the kernel early)
    const bool acc = is3d && reach_acc && X % 128 == 0 && Y % 8 == 0;
    if (acc) bc.reach_acc = reach_acc;
    TFL_TIMED_EXT("k_project", st);
    if (wall_code) {
      bc.wall_code = wall_code;
      if (is3d) TFL_LAUNCH_EXT((k_project_v4<true, true>), vg, vb, 0, st, d, pPred, flags, stats, count, Uio, pOut, bc);
      else TFL_LAUNCH_EXT((k_project_v4<false, true>), vg, vb, 0, st, d, pPred, flags, stats, count, Uio, pOut, bc);
    } else if (is3d) TFL_LAUNCH_EXT((k_project_v4<true>), vg, vb, 0, st, d, pPred, flags, stats, count, Uio, pOut, bc);
    else TFL_LAUNCH_EXT((k_project_v4<false>), vg, vb, 0, st, d, pPred, flags, stats, count, Uio, pOut, bc);
    return acc;
  }
  if (is3d) { TFL_TIMED("k_project", st); k_project<true><<<grd, blk, 0, st>>>(d, pPred, flags, stats, count, Uio, pOut, bc); }
  else { TFL_TIMED("k_project", st); k_project<false><<<grd, blk, 0, st>>>(d, pPred, flags, stats, count, Uio, pOut, bc); }
  return false;
}

void apply_bcs(hipStream_t st, long long n, float* x, const float* bcv, const float* inv, int do_clamp, float lo,
               float hi) {
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  { TFL_TIMED("k_apply_bcs", st); k_apply_bcs<<<(int)(blocks > 0 ? blocks : 1), 256, 0, st>>>(n, x, bcv, inv, do_clamp, lo, hi); }
}

long long pack_planes(hipStream_t st, int n, float* const* ptrs, const int* rows, const int* zlo, const int* nplanes,
                      long long zstride, long long yx, float* buf, int unpack, float* const* bufs) {
  PackArgs a;
  a.n = n;
  a.start[0] = 0;
  for (int i = 0; i < 8; i++) {
    a.ptr[i] = i < n ? ptrs[i] : nullptr;
    a.buf[i] = i < n ? (bufs ? bufs[i] : buf + a.start[i]) : nullptr;   // bufs = NULL: one contiguous buffer, field after field
    a.per_row[i] = i < n ? (long long)nplanes[i] * yx : 1;
    a.zoff[i] = i < n ? (long long)zlo[i] * yx : 0;
    a.start[i + 1] = a.start[i] + (i < n ? (long long)rows[i] * a.per_row[i] : 0);
  }
  long long blocks = (a.start[n] + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) return 0;
  { TFL_TIMED(unpack ? "k_unpack_planes" : "k_pack_planes", st); k_pack_planes<<<(int)blocks, 256, 0, st>>>(a, zstride, unpack); }
  return a.start[n];
}

void bc_scan(hipStream_t st, long long n, const float* bcv, const float* inv, int* counters, int* idx) {
  const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
  k_bc_scan<<<blocks > 0 ? blocks : 1, 256, 0, st>>>(n, bcv, inv, counters, idx);
}

void apply_bcs_indexed_multi(hipStream_t st, int count, const long long* n, const int* const* idx, float* const* x,
                             const float* const* bcv, const float* const* inv) {
  BcMultiArgs a;
  long long nmax = 0;
  for (int i = 0; i < 8; i++) {
    a.n[i] = i < count ? n[i] : 0; a.idx[i] = i < count ? idx[i] : nullptr; a.x[i] = i < count ? x[i] : nullptr;
    a.bc[i] = i < count ? bcv[i] : nullptr; a.inv[i] = i < count ? inv[i] : nullptr;
    if (a.n[i] > nmax) nmax = a.n[i];
  }
  if (nmax == 0) return;
  const unsigned bx = (unsigned)std::min<long long>((nmax + 255) / 256, 4096);
  TFL_TIMED_EXT("k_apply_bcs_indexed", st);
  TFL_LAUNCH_EXT(k_apply_bcs_indexed_multi, dim3(bx, (unsigned)count, 1), 256, 0, st, a);
}

void apply_bcs_indexed(hipStream_t st, long long n, const int* idx, float* x, const float* bcv, const float* inv) {
  if (n <= 0) return;
  long long blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  { TFL_TIMED_EXT("k_apply_bcs_indexed", st); TFL_LAUNCH_EXT(k_apply_bcs_indexed, (int)blocks, 256, 0, st, n, idx, x, bcv, inv); }
}

}  // namespace tfl
