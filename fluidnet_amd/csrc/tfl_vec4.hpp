// tfl_vec4.hpp -- "four x-cells per thread" building blocks for the streaming stencil kernels (gfx950).
//
// A thread owns cells i0..i0+3 of one grid row and moves them as one 16-byte vector (global_load_dwordx4:
// a wave moves 1 KB per instruction instead of 256 B). Stencil taps at i0-1 / i0+4 live in the float4 of
// the neighbouring LANE: they are fetched with a DPP wave shift (one v_mov_b32_dpp, no memory traffic);
// only the first / last lane of a row segment goes to memory for them.
//
// Launch contract (vec4_launch): blockDim = (BX, 256/BX, 1) with BX a power of two <= 32, so that the BX
// threads of a row segment are BX consecutive lanes of one wave; X % 4 == 0 and every pointer 16-byte
// aligned. Kernels built on these helpers must not return before their last DPP (a lane that has left
// delivers garbage to its neighbour): they predicate loads and stores instead.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <initializer_list>

#include "tfl_device.hpp"
#include "tfl_host.hpp"

namespace tfl {

// value held by lane-1 / lane+1 of the wave (undefined for lane 0 / 63: callers patch segment ends)
__device__ __forceinline__ float from_lane_below(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x138 /*wave_shr:1*/, 0xf, 0xf, false));
}
__device__ __forceinline__ float from_lane_above(float x) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x130 /*wave_shl:1*/, 0xf, 0xf, false));
}

struct V4Ctx {
  int i0;        // first of the thread's four cells
  bool first;    // first lane of the row segment (its i0-1 neighbour is not in a lane)
  bool last;     // last lane of the row segment, or the last float4 of the grid row
  bool has_l;    // cell i0-1 exists
  bool has_r;    // cell i0+4 exists
};
__device__ __forceinline__ V4Ctx v4_ctx(const Dom& d, int bx) {     // bx: the block's tile column (blockIdx.x, or block_tile's)
  V4Ctx c;
  c.i0 = (bx * blockDim.x + threadIdx.x) * 4;
  c.first = threadIdx.x == 0;
  c.last = threadIdx.x == blockDim.x - 1 || c.i0 + 4 >= d.X;
  c.has_l = c.i0 > 0;
  c.has_r = c.i0 + 4 < d.X;
  return c;
}
__device__ __forceinline__ V4Ctx v4_ctx(const Dom& d) { return v4_ctx(d, (int)blockIdx.x); }
// (batch item, plane) of tile row bz of the launch (dom_bk with an explicit index)
__device__ __forceinline__ void dom_bk_of(const Dom& d, int bz, int& b, int& k) {
  b = bz / d.nw;
  const int r = bz - b * d.nw;
  k = r < d.n0 ? d.w0 + r : d.w1 + (r - d.n0);
}

// r[0..3] = p[o..o+3] (pad when !ok). The load itself is UNCONDITIONAL: a predicated load compiles to a branch, and hipcc
// drains the whole load queue (s_waitcnt vmcnt(0)) at the join of every such branch -- k_confine_v4 had 20 of those between
// its 30 loads, i.e. 20 memory round trips in a row per wave (round 4; profiles/r04_stream_loads.txt). A lane that must not
// read takes the first vector of the array instead (p is a field's base pointer: valid, 16-byte aligned) and drops it.
__device__ __forceinline__ void v4_load(const float* __restrict__ p, int o, bool ok, float pad, float* r) {
  const float4 v = *reinterpret_cast<const float4*>(p + (ok ? o : 0));
  r[0] = ok ? v.x : pad; r[1] = ok ? v.y : pad; r[2] = ok ? v.z : pad; r[3] = ok ? v.w : pad;
}
__device__ __forceinline__ void v4_store(float* __restrict__ p, int o, const float* r) {
  *reinterpret_cast<float4*>(p + o) = make_float4(r[0], r[1], r[2], r[3]);
}
// e[0..5] = cells i0-1 .. i0+4 given e[1..4] already in registers (pad outside the grid / when !ok).
// EVERY lane of the wave must call this (DPP).
template <bool LEFT, bool RIGHT>
__device__ __forceinline__ void v4_edges(const V4Ctx& c, const float* __restrict__ p, int o, bool ok, float pad, float* e) {
  // (the segment-end loads are unconditional too, see v4_load: every lane loads, the lanes that need no value read p[0])
  if (LEFT) {
    const bool need = c.first && ok && c.has_l;
    const float el = p[need ? o - 1 : 0];
    e[0] = from_lane_below(e[4]);
    if (c.first) e[0] = need ? el : pad;
  }
  if (RIGHT) {
    const bool need = c.last && ok && c.has_r;
    const float er = p[need ? o + 4 : 0];
    e[5] = from_lane_above(e[1]);
    if (c.last) e[5] = need ? er : pad;
  }
}
// e[0..5] = p[o-1 .. o+4]
template <bool LEFT, bool RIGHT>
__device__ __forceinline__ void v4_load6(const V4Ctx& c, const float* __restrict__ p, int o, bool ok, float pad, float* e) {
  v4_load(p, o, ok, pad, e + 1);
  v4_edges<LEFT, RIGHT>(c, p, o, ok, pad, e);
}

// Host side: launch geometry for a vec4 kernel, or ok=false when the contract does not hold (odd X, a
// misaligned view, or TFL_NO_VEC4 set: callers then fall back to their one-cell-per-thread kernel).
struct Vec4Launch { bool ok; dim3 blk, grd; };
inline Vec4Launch vec4_launch(int B, int Z, int Y, int X, std::initializer_list<const void*> ptrs) {
  Vec4Launch l; l.ok = false;
  static const bool disabled = exp_env("TFL_NO_VEC4") != nullptr;
  uintptr_t al = 0;
  for (const void* q : ptrs) al |= (uintptr_t)q;
  if (disabled || X % 4 != 0 || (al & 15) != 0) return l;
  const int nx = X / 4, bx = nx <= 8 ? 8 : (nx <= 16 ? 16 : 32), by = 256 / bx;
  l.blk = dim3(bx, by, 1);
  l.grd = dim3((nx + bx - 1) / bx, (Y + by - 1) / by, (unsigned)(zwin_planes(Z) * B));
  l.ok = true;
  return l;
}

}  // namespace tfl
