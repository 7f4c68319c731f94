// advect_vel3.hip -- advectVel on a 3-D MAC grid, trace-based methods (eulerOurs, maccormackOurs): the
// north-star's "advection kernel" (third_party/tfluids.cc:594-632 SemiLagrangeEulerOursMAC, :660-699
// MacCormackCorrectMAC, :748-774 MacCormackClampMAC; generic/calc_line_trace.cc:313-503).
//
// The gather kernels of advect.hip spend their time in two places (profiles/r02_pmc_sq.txt): the
// vector ALU (650 instructions per wave: three IEEE divisions and a full-range sqrt per trace, branchy
// out-of-domain / obstacle handling whose phi copies are 17 % of the stream, clamped index arithmetic, 64-bit
// addresses) and the texture addresser (123 gather instructions per wave). Here:
//
//  * FAST PATH / SLOW PATH. A cell whose three back-traces are "ordinary" -- fluid cell that is not a border cell, displacement shorter than 0.99 cell, end point in a fluid cell -- needs none of the reference's
//    special cases: ONE trace step, no wall clipping, no ray/box test, no index clamps, every tap inside the 3^3
//    neighbourhood of the cell. That path is straight-line code with the reference's operation order; the
//    division by the trace length uses one refined reciprocal for the three components and the norm a one-step
//    corrected v_sqrt: the root checked exhaustively, the quotient SAMPLE-verified bit-equal to sqrtf() / `/` on the operand range (tfl_fastmath.hpp,
//    tools/ubench/exact_math.hip). Every other lane (walls, obstacles' neighbours, fast flow) runs the generic
//    functions of tfl_device.hpp / tfl_advect.hpp afterwards, from scratch: same result as before by construction.
//  * LDS TILE. Per 64x4x1-cell block the 66x6x3 halo tile of U (3 components) and flags is staged once with
//    coalesced row loads (one field per wave; scalar row pointers, no per-lane address arithmetic); the 18 MAC taps,
//    the trace's flag look-ups, the forward pass's 24 interpolation taps and the backward pass's 48 clamp corners
//    are ds_reads with immediate offsets from one address VGPR. Only the backward pass's 24 interpolation taps of
//    the forward field remain global gathers, issued for the three components together (one L2 round trip).
//  * What bounds these kernels now (profiles/r03_pmc_adv.txt): instruction ISSUE -- a SIMD retires one
//    instruction of any kind (VALU, SALU, LDS, VMEM) per ~2.7 clocks here, so the count of ALL instructions per
//    wave is the budget; packed fp32 (v_pk_mul/add) issues at half rate and buys nothing (SLP vectorisation is off
//    for this file). A z-marched variant (4-slot plane ring, 1.55x instead of 4.6x staging) issued 24 % fewer
//    instructions per cell but lost more to block-count quantisation at 128^3 (2048 long blocks on 256 CUs).
//
// Algorithmic HBM bytes per cell are unchanged: pass A 28 B (U3, flags -> fwd3), pass B 40 B (fwd3, U3, flags -> dst3).
#include "tfl_advect.hpp"
#include "tfl_fastmath.hpp"

#include <cstdlib>

namespace tfl {
namespace {

// A block covers 64 x 4 x KZ cells: one wave per grid row, KZ consecutive planes one after the other on ONE staged tile
// (66 x 6 x (KZ + 2) words per field). KZ = 1 (round 3) stages 4.6 words per cell and field, KZ = 2 3.1: what the four
// fields' staging moves through L2 and LDS per cell was the larger part of these kernels' time (round 4).
#ifndef TFL_VEL3_KZ
#define TFL_VEL3_KZ 2
#endif
// timing ablations (tools/ab_build.sh -DTFL_VEL3_ABL=..): 1 = the tile is filled with constants instead of staged (no staging
// loads; every lane on the fast path), 2 = pass B's 24 gathers of the forward field replaced by the cell's own value,
// 4 = no stores
#ifndef TFL_VEL3_ABL
#define TFL_VEL3_ABL 0
#endif
constexpr int KZ = TFL_VEL3_KZ, LZ = KZ + 2, NR = LZ * 6;            // planes of the tile; rows of one field of the tile
static_assert(2 * NR <= 64, "the halo columns of a field are staged by one wave: 2 * 6 * (KZ + 2) lanes");
constexpr int TX = 64, TY = 4;
constexpr int LX = TX + 2, LY = TY + 2, LP = LX * LY, LN = LZ * LP;  // 66, 6, 396 (plane), field
constexpr int FL = 3 * LN;                                           // tile offset of the flags field
constexpr float kFastLen = 0.99f;                                    // longest displacement the fast path takes

// "surely a plain fluid cell": the flag word is exactly TypeFluid. Any other word (obstacle, or fluid with further
// bits set) sends the lane to the generic path, which decodes the bits as the reference does.
__device__ __forceinline__ bool plain_fluid(float f) { return f == 1.0f; }

__device__ __forceinline__ float ldg(const float* __restrict__ base, unsigned byte_off) {   // uniform base + 32-bit lane offset
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}

// Wave w stages field w: NR rows of 64 (one coalesced 256-B load each) + the two halo columns (2 NR lanes).
// EDGE = false (block whose halo rows and planes lie inside the array): row pointers advance by scalar adds.
// EDGE = true: rows / columns outside the array are loaded from the nearest inside one (fast lanes never read them).
template <bool EDGE>
__device__ __forceinline__ void stage_tile(float* __restrict__ tile, const float* __restrict__ g, const Dom& d, int x0,
                                           int y0, int k, int lane) {
  float v[NR], h;
  const int hr = min(lane >> 1, NR - 1), hz = hr / 6, hy = hr - hz * 6, side = lane & 1;
  if (TFL_VEL3_ABL & 1) {                 // constants instead of loads: 1.0 = a fluid flag word, a velocity of 1 cell per unit time
#pragma unroll
    for (int r = 0; r < NR; r++) v[r] = 1.0f;
    h = 1.0f;
  } else if (EDGE) {
    const unsigned xl4 = (unsigned)min(x0 + lane, d.X - 1) * 4u;
#pragma unroll
    for (int r = 0; r < NR; r++) {
      const int z = min(max(k - 1 + r / 6, 0), d.Z - 1), y = min(max(y0 - 1 + r % 6, 0), d.Y - 1);   // wave-uniform
      v[r] = ldg(g + ((long long)z * d.sz + (long long)y * d.sy), xl4);
    }
    const int gz = min(max(k - 1 + hz, 0), d.Z - 1), gy = min(max(y0 - 1 + hy, 0), d.Y - 1);
    const int gx = side ? min(x0 + TX, d.X - 1) : max(x0 - 1, 0);
    h = ldg(g, (unsigned)(gz * d.sz + gy * d.sy + gx) * 4u);
  } else {
    // rows and planes all inside the array; only the columns may stick out (x0 = 0, or the last block of a row)
    const int sy4 = d.sy * 4, sz4 = d.sz * 4;
    const char* row = reinterpret_cast<const char*>(g + ((long long)(k - 1) * d.sz + (long long)(y0 - 1) * d.sy));
    const unsigned l4 = (unsigned)min(x0 + lane, d.X - 1) * 4u;
    const int gx = side ? min(x0 + TX, d.X - 1) : max(x0 - 1, 0);
    h = *reinterpret_cast<const float*>(row + (unsigned)(__mul24(hz, sz4) + __mul24(hy, sy4) + gx * 4));
#pragma unroll
    for (int z = 0; z < LZ; z++) {
      const char* rp = row;
#pragma unroll
      for (int y = 0; y < 6; y++) { v[z * 6 + y] = *reinterpret_cast<const float*>(rp + l4); rp += sy4; }
      row += sz4;
    }
  }
#pragma unroll
  for (int r = 0; r < NR; r++) tile[(r / 6) * LP + (r % 6) * LX + 1 + lane] = v[r];
  if (lane < 2 * NR) tile[hz * LP + hy * LX + (side ? LX - 1 : 0)] = h;
}

// tile index of global cell (x, y, zg) = x + y*LX + zg*LP + cbias  (cbias: per lane, see the kernels)
__device__ __forceinline__ int tidx(int x, int y, int zg, int cbias) {
  return __mul24(zg, LP) + (__mul24(y, LX) + x) + cbias;
}

// get_at_mac (third_party/grid.cc:379-417) of the three faces of the cell at tile index c, from the tile
__device__ __forceinline__ void mac_from_tile(const float* __restrict__ t, int c, v3& u0, v3& u1, v3& u2) {
  const float* ux = t + c;
  const float* uy = t + LN + c;
  const float* uz = t + 2 * LN + c;
  u0.x = ux[0];
  u0.y = 0.25f * (uy[0] + uy[-1] + uy[LX] + uy[-1 + LX]);
  u0.z = 0.25f * (uz[0] + uz[-1] + uz[LP] + uz[-1 + LP]);
  u1.x = 0.25f * (ux[0] + ux[-LX] + ux[1] + ux[1 - LX]);
  u1.y = uy[0];
  u1.z = 0.25f * (uz[0] + uz[-LX] + uz[LP] + uz[-LX + LP]);
  u2.x = 0.25f * (ux[0] + ux[-LP] + ux[1] + ux[1 - LP]);
  u2.y = 0.25f * (uy[0] + uy[-LP] + uy[LX] + uy[LX - LP]);
  u2.z = uz[0];
}

// One ordinary back-trace (calcLineTrace with length <= kFastLen: a single step, calc_line_trace.cc:313-503).
// Returns false when the lane needs the generic trace (long displacement, NaN, end point not in a fluid cell).
// `p` is the traced position (global z); valid only when true is returned.
// FAST (the tolerance mode, tfl_set_advect_mode): direction x length IS the displacement, so the end point is ctr + d --
// no root, no reciprocal, no quotients (the reference's normalise-then-rescale differs from it by an ulp or two of the
// position); the thresholds (|d|^2 > 1e-6, |d| <= 0.99) are applied to the squared length.
template <bool FAST>
__device__ __forceinline__ bool trace_fast(const float* __restrict__ tile, int cbias, v3 ctr, v3 u, float ndt, v3& p) {
  const float dx = u.x * ndt, dy = u.y * ndt, dz = u.z * ndt;     // scale3(u, -dt)
  const float l2 = dx * dx + dy * dy + dz * dz;                   // vec3::norm, vec3.h:119-127
  const bool nz = l2 > 1e-6f;
  if (FAST) {
    p.x = nz ? ctr.x + dx : ctr.x; p.y = nz ? ctr.y + dy : ctr.y; p.z = nz ? ctr.z + dz : ctr.z;
    return (l2 <= kFastLen * kFastLen) & plain_fluid(tile[FL + tidx((int)p.x, (int)p.y, (int)p.z, cbias)]);
  }
  const float len = nz ? sqrt_exact(l2) : 0.0f;
  const float r = nz ? rcp_refined(len) : 0.0f;                   // len == 0: direction 0, p = ctr (the reference returns pos)
  const float qx = div_by<1>(dx, len, r), qy = div_by<1>(dy, len, r), qz = div_by<1>(dz, len, r);
  p.x = ctr.x + qx * len;                                         // next = pos + dt * step, step = min(length - 0, 1) = length
  p.y = ctr.y + qy * len;
  p.z = ctr.z + qz * len;
  return (len <= kFastLen) & plain_fluid(tile[FL + tidx((int)p.x, (int)p.y, (int)p.z, cbias)]);
}

// interpol (grid.cc:182-202) of one tile field at p, for a position the fast trace produced: p - 0.5 lies in
// [i - 1, i + 1) on every axis, so buildIndex's clamps cannot act; pc - float(int(pc)) == fract(pc) for pc >= 0.
struct FastLerp { int x, y, z; float s0, s1, t0, t1, f0, f1; };
__device__ __forceinline__ FastLerp lerp_fast(v3 p) {
  FastLerp L;
  const float px = p.x - 0.5f, py = p.y - 0.5f, pz = p.z - 0.5f;
  L.x = (int)px; L.y = (int)py; L.z = (int)pz;
  L.s1 = __builtin_amdgcn_fractf(px); L.t1 = __builtin_amdgcn_fractf(py); L.f1 = __builtin_amdgcn_fractf(pz);
  L.s0 = 1.0f - L.s1; L.t0 = 1.0f - L.t1; L.f0 = 1.0f - L.f1;
  return L;
}
template <bool FAST>
__device__ __forceinline__ float lerp8(const FastLerp& L, float g000, float g010, float g100, float g110, float g001,
                                       float g011, float g101, float g111) {   // g[x][y][z]
  if (FAST) {   // a + t (b - a): 14 instead of 21 operations, contracted
    const float a0 = __builtin_fmaf(L.t1, g010 - g000, g000), a1 = __builtin_fmaf(L.t1, g110 - g100, g100);
    const float b0 = __builtin_fmaf(L.t1, g011 - g001, g001), b1 = __builtin_fmaf(L.t1, g111 - g101, g101);
    const float lo = __builtin_fmaf(L.s1, a1 - a0, a0), hi = __builtin_fmaf(L.s1, b1 - b0, b0);
    return __builtin_fmaf(L.f1, hi - lo, lo);
  }
  const float lo = (g000 * L.t0 + g010 * L.t1) * L.s0 + (g100 * L.t0 + g110 * L.t1) * L.s1;
  const float hi = (g001 * L.t0 + g011 * L.t1) * L.s0 + (g101 * L.t0 + g111 * L.t1) * L.s1;
  return lo * L.f0 + hi * L.f1;
}
template <bool FAST>
__device__ __forceinline__ float sample_tile(const float* __restrict__ g, int cbias, v3 p) {
  const FastLerp L = lerp_fast(p);
  const float* q = g + tidx(L.x, L.y, L.z, cbias);
  return lerp8<FAST>(L, q[0], q[LX], q[1], q[1 + LX], q[LP], q[LP + LX], q[LP + 1], q[LP + 1 + LX]);
}

// min/max of the 2^3 corner box at tile index b, accumulated as manta_clamp_bounds does (tfl_advect.hpp)
__device__ __forceinline__ void box_minmax(const float* __restrict__ q, float& lo, float& hi) {
  lo = __builtin_fminf(__builtin_fminf(lo, q[0]), q[1]);
  hi = __builtin_fmaxf(__builtin_fmaxf(hi, q[0]), q[1]);
  lo = __builtin_fminf(__builtin_fminf(lo, q[LX]), q[1 + LX]);
  hi = __builtin_fmaxf(__builtin_fmaxf(hi, q[LX]), q[1 + LX]);
  lo = __builtin_fminf(__builtin_fminf(lo, q[LP]), q[LP + 1]);
  hi = __builtin_fmaxf(__builtin_fmaxf(hi, q[LP]), q[LP + 1]);
  lo = __builtin_fminf(__builtin_fminf(lo, q[LP + LX]), q[LP + 1 + LX]);
  hi = __builtin_fmaxf(__builtin_fmaxf(hi, q[LP + LX]), q[LP + 1 + LX]);
}
// MacCormackClampMAC bounds (tfluids.cc:701-746) of one component for |vel| < 1 at a cell >= 2 inside the domain:
// int(pos -+ vel) lies in [i - 1, i] on every axis, the index clamps and isInBounds cannot act.
__device__ __forceinline__ void clamp_bounds_tile(const float* __restrict__ g, int cbias, v3 ijk, v3 vel, float& lo, float& hi) {
  lo = 3.402823466e+38f; hi = -3.402823466e+38f;
  box_minmax(g + tidx((int)(ijk.x - vel.x), (int)(ijk.y - vel.y), (int)(ijk.z - vel.z), cbias), lo, hi);
  box_minmax(g + tidx((int)(ijk.x + vel.x), (int)(ijk.y + vel.y), (int)(ijk.z + vel.z), cbias), lo, hi);
}

// common prologue: block -> plane/batch item, tile staged, cell geometry. `deep` = not a border cell of the whole grid and
// the 3^3 neighbourhood inside the local array. With a displacement <= 0.99 from the centre of such a cell the trace
// cannot leave the domain (p > 0.51, p < N - 0.51), p - 0.5 lies in (i - 1, i + 1) so buildIndex's clamps and the clamp
// boxes' index clamps cannot act, and every tap lies in [i - 1, i + 1]: inside the grid and inside the tile.
// (One batch item and the whole window in grid.z is the common launch: it skips dom_bk's integer division.)
#define TFL_VEL3_STAGE()                                                                           \
  __shared__ float tile[4 * LN];                                                                   \
  const Dom& d = a.d;                                                                              \
  /* groups of KZ planes tile the window's two plane runs */                                       \
  const int ga_ = (d.n0 + KZ - 1) / KZ, gn_ = ga_ + (d.nw - d.n0 + KZ - 1) / KZ;                   \
  int b = 0, g_ = (int)blockIdx.z;                                                                 \
  if ((int)gridDim.z != gn_) { b = g_ / gn_; g_ -= b * gn_; }                                      \
  const int k0 = g_ < ga_ ? d.w0 + g_ * KZ : d.w1 + (g_ - ga_) * KZ;                               \
  const int kend = g_ < ga_ ? d.w0 + d.n0 : d.w1 + (d.nw - d.n0);                                  \
  const long long cells = (long long)d.sc;                                                         \
  flags += b * cells; U += b * cells * 3;                                                          \
  const int lane = threadIdx.x, w = __builtin_amdgcn_readfirstlane(threadIdx.y);                   \
  const int x0 = blockIdx.x * TX, y0 = blockIdx.y * TY;                                            \
  const bool inner = y0 >= 1 && y0 + TY < d.Y && k0 >= 1 && k0 + KZ < d.Z;                         \
  {                                                                                                \
    const float* sf = w < 3 ? U + w * cells : flags;                                               \
    if (inner) stage_tile<false>(tile + w * LN, sf, d, x0, y0, k0, lane);                          \
    else stage_tile<true>(tile + w * LN, sf, d, x0, y0, k0, lane);                                 \
  }                                                                                                \
  __syncthreads();                                                                                 \
  const int i = x0 + lane, j = y0 + w;                                                             \
  if (i >= d.X || j >= d.Y) return

// the cell of plane k0 + tz (inside the loop over the block's planes)
#define TFL_VEL3_CELL(tz)                                                                          \
  const int k = k0 + (tz);                                                                         \
  if (k >= kend) break;                                                                            \
  const int kg = k + d.zg;                                                                         \
  const int c0 = ((tz) + 1) * LP + (w + 1) * LX + lane + 1;                                        \
  const int cbias = c0 - (i + j * LX + kg * LP);                                                   \
  const bool deep = i >= 1 && i <= d.X - 2 && j >= 1 && j <= d.Y - 2 && kg >= 1 && kg <= d.Zg - 2 && k >= 1 && k <= d.Z - 2; \
  const v3 ctr = mk3((float)i + 0.5f, (float)j + 0.5f, (float)kg + 0.5f);                          \
  const int o = TFL_AT(d, i, j, k);                                                                \
  const unsigned o4 = (unsigned)o * 4u, sc4 = (unsigned)d.sc * 4u

// store through a uniform base + 32-bit lane offset
__device__ __forceinline__ void stg(float* __restrict__ base, unsigned byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// ---- pass A / the single-pass method: SemiLagrangeEulerOursMAC ------------------------------------------------
template <bool FAST>
__global__ __launch_bounds__(256) void k_vel3_fwd(AdvArgs a, const float* __restrict__ U, const float* __restrict__ flags,
                                                  float* __restrict__ out) {
  TFL_VEL3_STAGE();
  out += b * cells * 3;
#pragma unroll 1
  for (int tz = 0; tz < KZ; tz++) {
  TFL_VEL3_CELL(tz);
  float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
  unsigned slow = 0;
  const float cf = tile[FL + c0];
  if (deep && plain_fluid(cf)) {
    v3 u0, u1, u2, p0, p1, p2;
    mac_from_tile(tile, c0, u0, u1, u2);
    const bool k0 = trace_fast<FAST>(tile, cbias, ctr, u0, -a.dt, p0);
    const bool k1 = trace_fast<FAST>(tile, cbias, ctr, u1, -a.dt, p1);
    const bool k2 = trace_fast<FAST>(tile, cbias, ctr, u2, -a.dt, p2);
    v0 = sample_tile<FAST>(tile, cbias, p0);
    v1 = sample_tile<FAST>(tile + LN, cbias, p1);
    v2 = sample_tile<FAST>(tile + 2 * LN, cbias, p2);
    slow = (k0 ? 0u : 1u) | (k1 ? 0u : 2u) | (k2 ? 0u : 4u);
  } else if (!on_border<true>(d, i, j, k)) {
    if ((((int)cf) & kFluid) == 0) { v0 = tile[c0]; v1 = tile[LN + c0]; v2 = tile[2 * LN + c0]; }   // tfluids.cc:598-601
    else slow = 7u;
  }
  if (slow) {   // rare lanes: the generic trace + sampler on global memory
    if (slow & 1u) v0 = sl_mac_from_u<true, true, 0>(a, flags, U, get_at_mac<true, 0>(d, U, i, j, k), a.dt, i, j, k);
    if (slow & 2u) v1 = sl_mac_from_u<true, true, 1>(a, flags, U, get_at_mac<true, 1>(d, U, i, j, k), a.dt, i, j, k);
    if (slow & 4u) v2 = sl_mac_from_u<true, true, 2>(a, flags, U, get_at_mac<true, 2>(d, U, i, j, k), a.dt, i, j, k);
  }
  if (!(TFL_VEL3_ABL & 4) || a.dt == 12345.0f) { stg(out, o4, v0); stg(out, o4 + sc4, v1); stg(out, o4 + 2u * sc4, v2); }
  }
}

// ---- pass B: backward trace on fwd + MacCormackCorrectMAC + MacCormackClampMAC --------------------------------
// the 8 interpolation corners of a global channel plane at a fast-trace position (the forward field is not in the
// tile). Issued for all three components BEFORE anything consumes them: one L2 round trip instead of three.
__device__ __forceinline__ void gather8_global(const float* __restrict__ g, const Dom& d, unsigned safe_off4, bool ok,
                                               const FastLerp& L, float* __restrict__ c) {
  // local plane = global plane - zg; a lane whose trace failed reads its own cell (any valid address) and is redone later
  unsigned q4 = (unsigned)(__mul24(L.z - d.zg, d.sz * 4) + (__mul24(L.y, d.sy * 4) + L.x * 4));
  q4 = ok ? q4 : safe_off4;
  const unsigned sy4 = (unsigned)d.sy * 4u, sz4 = (unsigned)d.sz * 4u, one4 = (unsigned)d.one * 4u;
  const unsigned a00 = q4, a01 = q4 + sy4, a10 = q4 + sz4, a11 = q4 + sz4 + sy4;
  c[0] = ldg(g, a00); c[1] = ldg(g, a01); c[2] = ldg(g, a00 + one4); c[3] = ldg(g, a01 + one4);
  c[4] = ldg(g, a10); c[5] = ldg(g, a11); c[6] = ldg(g, a10 + one4); c[7] = ldg(g, a11 + one4);
}

template <bool FAST>
__global__ __launch_bounds__(256) void k_vel3_bwd(AdvArgs a, double half_strength, const float* __restrict__ U,
                                                  const float* __restrict__ flags, const float* __restrict__ fwd,
                                                  float* __restrict__ dst) {
  TFL_VEL3_STAGE();
  fwd += b * cells * 3; dst += b * cells * 3;
#pragma unroll 1
  for (int tz = 0; tz < KZ; tz++) {
  TFL_VEL3_CELL(tz);
  const float f0 = ldg(fwd, o4), f1 = ldg(fwd, o4 + sc4), f2 = ldg(fwd, o4 + 2u * sc4);
  float r0 = f0, r1 = f1, r2 = f2;
  const float cf = tile[FL + c0];
  unsigned slow = 7u;
  if (deep && plain_fluid(cf)) {
    // MacCormackCorrectMAC skips a face whose other cell is not fluid (tfluids.cc:672-690)
    const bool s0 = (((int)tile[FL + c0 - 1]) & kFluid) == 0, s1 = (((int)tile[FL + c0 - LX]) & kFluid) == 0,
               s2 = (((int)tile[FL + c0 - LP]) & kFluid) == 0;
    v3 u0, u1, u2, p0, p1, p2;
    mac_from_tile(tile, c0, u0, u1, u2);
    const bool k0 = trace_fast<FAST>(tile, cbias, ctr, u0, a.dt, p0);
    const bool k1 = trace_fast<FAST>(tile, cbias, ctr, u1, a.dt, p1);
    const bool k2 = trace_fast<FAST>(tile, cbias, ctr, u2, a.dt, p2);
    const FastLerp L0 = lerp_fast(p0), L1 = lerp_fast(p1), L2 = lerp_fast(p2);
    float g0[8], g1[8], g2[8];
    if (TFL_VEL3_ABL & 2) {
#pragma unroll
      for (int q = 0; q < 8; q++) { g0[q] = f0; g1[q] = f1; g2[q] = f2; }
    } else {
      gather8_global(fwd, d, o4, k0, L0, g0);
      gather8_global(fwd + d.sc, d, o4, k1, L1, g1);
      gather8_global(fwd + 2 * d.sc, d, o4, k2, L2, g2);
    }
    const v3 ijk = mk3((float)i, (float)j, (float)kg);
    float lo0, hi0, lo1, hi1, lo2, hi2;
    clamp_bounds_tile(tile, cbias, ijk, scale3(u0, a.dt), lo0, hi0);
    clamp_bounds_tile(tile + LN, cbias, ijk, scale3(u1, a.dt), lo1, hi1);
    clamp_bounds_tile(tile + 2 * LN, cbias, ijk, scale3(u2, a.dt), lo2, hi2);
    const float uo0 = tile[c0], uo1 = tile[LN + c0], uo2 = tile[2 * LN + c0];
    const float b0 = lerp8<FAST>(L0, g0[0], g0[1], g0[2], g0[3], g0[4], g0[5], g0[6], g0[7]);
    const float b1 = lerp8<FAST>(L1, g1[0], g1[1], g1[2], g1[3], g1[4], g1[5], g1[6], g1[7]);
    const float b2 = lerp8<FAST>(L2, g2[0], g2[1], g2[2], g2[3], g2[4], g2[5], g2[6], g2[7]);
    // the reference evaluates f + strength * 0.5 * (orig - bwd) in double (unsuffixed 0.5, tfluids.cc:693)
    if (FAST) {   // the correction in fp32, contracted
      const float hs = (float)half_strength;
      if (!s0) r0 = __builtin_fmaf(hs, uo0 - b0, f0);
      if (!s1) r1 = __builtin_fmaf(hs, uo1 - b1, f1);
      if (!s2) r2 = __builtin_fmaf(hs, uo2 - b2, f2);
    } else {
      if (!s0) r0 = (float)((double)f0 + half_strength * (double)(uo0 - b0));
      if (!s1) r1 = (float)((double)f1 + half_strength * (double)(uo1 - b1));
      if (!s2) r2 = (float)((double)f2 + half_strength * (double)(uo2 - b2));
    }
    // std::min(hi, std::max(lo, v)) with lo <= hi (extrema of one set): the median of the three
    r0 = __builtin_amdgcn_fmed3f(r0, lo0, hi0); r1 = __builtin_amdgcn_fmed3f(r1, lo1, hi1);
    r2 = __builtin_amdgcn_fmed3f(r2, lo2, hi2);
    // a failed trace also invalidates the clamp corners (|vel| may exceed the tile): the whole component is redone
    slow = (k0 ? 0u : 1u) | (k1 ? 0u : 2u) | (k2 ? 0u : 4u);
  }
  if (slow) {
    const bool fl = (((int)cf) & kFluid) != 0;
    const bool border = on_border<true>(d, i, j, k);
    const bool sk0 = !fl || (i > 0 && !fluid_at(d, flags, i - 1, j, k));
    const bool sk1 = !fl || (j > 0 && !fluid_at(d, flags, i, j - 1, k));
    const bool sk2 = !fl || (k > 0 && !fluid_at(d, flags, i, j, k - 1));
    const v3 ijk = mk3((float)i, (float)j, (float)kg);
#define TFL_VEL3_SLOW(C, BIT, F, SK, R)                                                                              \
    if (slow & BIT) {                                                                                                \
      float v = F;                                                                                                   \
      if (!border) {                                                                                                 \
        const v3 u = get_at_mac<true, C>(d, U, i, j, k);                                                             \
        float lo, hi;                                                                                                \
        const bool ok = manta_clamp_bounds<true>(d, U + C * d.sc, ijk, scale3(u, a.dt), lo, hi);                     \
        const float bw = fl ? sl_mac_from_u<true, true, C>(a, flags, fwd, u, -a.dt, i, j, k) : F;                    \
        if (!SK) v = (float)((double)F + half_strength * (double)(U[o + C * d.sc] - bw));                            \
        v = ok ? fclampf(v, lo, hi) : F;                                                                             \
      } else if (!SK) {                                                                                              \
        v = (float)((double)F + half_strength * (double)(U[o + C * d.sc] - 0.0f));                                   \
      }                                                                                                              \
      R = v;                                                                                                         \
    }
    TFL_VEL3_SLOW(0, 1u, f0, sk0, r0)
    TFL_VEL3_SLOW(1, 2u, f1, sk1, r1)
    TFL_VEL3_SLOW(2, 4u, f2, sk2, r2)
#undef TFL_VEL3_SLOW
  }
  if (!(TFL_VEL3_ABL & 4) || a.dt == 12345.0f) { stg(dst, o4, r0); stg(dst, o4 + sc4, r1); stg(dst, o4 + 2u * sc4, r2); }
  }
}

}  // namespace

bool advect_vel3(hipStream_t st, bool two_pass, const AdvArgs& a, int B, const float* U, const float* flags, float* fwd,
                 float* dst, int stages) {
  static const bool off = getenv("TFL_ADVECT_GATHER") != nullptr;   // A/B switch: the round-2 gather kernels
  const Dom& d = a.d;
  // 24-bit multiplies address the planes (4*X*Y < 2^24); 32-bit BYTE offsets address the cells of all three channels
  // (o4 + 2*sc4 in the loads / stores of U, fwd, dst): 12*Z*Y*X < 2^32. Larger grids take the gather kernels.
  if (off || d.Z < 3 || (long long)d.X * d.Y * 4 >= (1 << 24) || 12ll * d.sc >= (1ll << 32)) return false;
  const int groups = (d.n0 + KZ - 1) / KZ + (d.nw - d.n0 + KZ - 1) / KZ;
  const dim3 blk(TX, TY, 1), grd((d.X + TX - 1) / TX, (d.Y + TY - 1) / TY, (unsigned)(groups * B));
  const bool pa = stages & 2, pb = stages & 4;
  float* outA = two_pass ? fwd : dst;
  if (pa) {
    TFL_TIMED_EXT("k_vel_fwd", st);
    if (a.fast) TFL_LAUNCH_EXT(k_vel3_fwd<true>, grd, blk, 0, st, a, U, flags, outA);
    else TFL_LAUNCH_EXT(k_vel3_fwd<false>, grd, blk, 0, st, a, U, flags, outA);
  }
  if (two_pass && pb) {
    TFL_TIMED_EXT("k_vel_bwd", st);
    if (a.fast) TFL_LAUNCH_EXT(k_vel3_bwd<true>, grd, blk, 0, st, a, (double)a.strength * 0.5, U, flags, (const float*)fwd, dst);
    else TFL_LAUNCH_EXT(k_vel3_bwd<false>, grd, blk, 0, st, a, (double)a.strength * 0.5, U, flags, (const float*)fwd, dst);
  }
  return true;
}

}  // namespace tfl
