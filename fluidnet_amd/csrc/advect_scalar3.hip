// advect_scalar3.hip -- advectScalar on a 3-D grid, trace-based methods (eulerOurs, maccormackOurs), on the design of
// advect_vel3.hip: fast path / slow path + LDS tile (third_party/tfluids.cc:152-207 SemiLagrangeEulerOurs[SavePos],
// :220-234 MacCormackCorrect, :331-378 getClampBounds, :380-413 MacCormackClampOurs; generic/calc_line_trace.cc:313-503).
//
// What the round-2 gather kernels of advect.hip cost: a separate 3^3 min/max pass over the whole grid (k_minmax3, two
// more planes written and gathered back), 24 / 26 gather instructions per wave, the generic trace for every lane
// (three IEEE divisions + a full-range sqrt), 112.7 B/cell of HBM traffic at 256^3 against 52 algorithmic. Here:
//
//  * ONE LDS FIELD PER PASS: the advected scalar MASKED by the flags -- s where the cell is fluid, NaN where it is not
//    or lies outside the grid (a NaN with a private payload, told apart from a NaN in the field by its bit pattern). That single tile answers every question the operator asks of its neighbourhood:
//      - "is the end point of the trace in a fluid cell" (calcLineTrace's blocked test): the word there is not NaN;
//      - getInterpolatedWithFluidHi (grid.cc:204-332): with all eight corners fluid it is the plain trilinear form in
//        the same operation order, and with any corner missing the plain form comes out NaN (0 x NaN = NaN): the
//        result itself tells the lane to redo the 1-D lerps the way lerp_fluid drops non-fluid taps (still from the
//        tile: every lane next to a wall does this);
//      - getClampBounds' 3^3 search over the FLUID cells inside the grid around int(forward position): v_min3_f32 /
//        v_max3_f32 skip NaN operands exactly like the reference's `val < minv` comparisons skip non-fluid cells.
//    So the min/max grid, its kernel and its two planes disappear: pass A reads the 27 taps from the tile (halo 2).
//  * FAST PATH / SLOW PATH as in advect_vel3.hip: a fluid cell that is no border cell, displacement <= 0.99 cell, end
//    point in a fluid cell, every interpolation corner fluid. Everything else (obstacle neighbours, fast flow, NaNs)
//    re-runs the generic functions of tfl_device.hpp afterwards from scratch: same result as the gather kernels by
//    construction. Division and square root of the trace: tfl_fastmath.hpp.
//  * A block of 64 x 4 x TZ threads covers TZ x KZ planes (KZ planes per thread) on one staged tile: two planes make the
//    halo-2 tile of pass A 68 x 8 x 6 = 6.4 staged words per cell (10.6 for one plane), the halo-1 tile of pass B 3.1;
//    the shape is chosen per pass and grid size (launch()).
//
// Algorithmic HBM bytes per cell: pass A 24 (s, U3, flags -> fwd) + 8 (the clamp bounds of the forward position, two
// planes of the fwdPos temp), pass B 28 (fwd, s, U3, flags -> dst) + 8. `sampleOutsideFluid`, 2-D grids and the Manta
// methods stay on advect.hip.
#include "tfl_advect.hpp"
#include "tfl_fastmath.hpp"

#include <cstdlib>

namespace tfl {
namespace {

// timing ablations (tools/ab_build.sh -DTFL_SCAL3_ABL=..): 1 = the tile is filled with a constant instead of staged, 2 = no 27-tap
// bounds search (pass A), 4 = no stores, 8 = no trace / lerp (the cell's own value)
#ifndef TFL_SCAL3_ABL
#define TFL_SCAL3_ABL 0
#endif
constexpr int TX = 64, TY = 4;
constexpr float kFastLen = 0.99f;
struct SBlock { int L, gx, gy, gz; };      // L < 0: the block is blockIdx of its own 3-D launch
// the "not a fluid cell" word of the masked tile: a quiet NaN with a payload no arithmetic produces (hardware NaNs are
// 0x7fc00000 or carry an input's payload), recognised by its BIT PATTERN -- so a NaN that really sits in the advected
// field of a fluid cell stays a value, as in the reference
constexpr unsigned kMaskBits = 0x7fc5a5a5u;
__device__ __forceinline__ bool is_mask(float x) { return __builtin_bit_cast(unsigned, x) == kMaskBits; }

template <int TZ, int H>
struct Tile {
  static constexpr int LX = TX + 2 * H, LY = TY + 2 * H, LZ = TZ + 2 * H, LP = LX * LY, N = LP * LZ;
};

// block -> (batch item, first plane, end of its plane run): groups of TZ planes tile the window's two runs
template <int TZ>
__device__ __forceinline__ void group_planes(const Dom& d, int g, int gz, int& b, int& k0, int& kend) {      // gz = plane groups x batch items of the launch
  const int ga = (d.n0 + TZ - 1) / TZ, gb = (d.nw - d.n0 + TZ - 1) / TZ, G = ga + gb;
  b = 0;
  if (gz != G) { b = g / G; g -= b * G; }
  if (g < ga) { k0 = d.w0 + g * TZ; kend = d.w0 + d.n0; }
  else { k0 = d.w1 + (g - ga) * TZ; kend = d.w1 + (d.nw - d.n0); }
}

__device__ __forceinline__ float ldg(const float* __restrict__ base, unsigned byte_off) {   // uniform base + 32-bit lane offset
  return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void stg(float* __restrict__ base, unsigned byte_off, float v) {
  *reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
__device__ __forceinline__ float mask_word(float v, float f, bool in) {
  return (in && ((((int)f) & kFluid) != 0)) ? v : __builtin_bit_cast(float, kMaskBits);
}

// tile <- (flags & fluid) ? g : mask word over the block's halo box; cells outside the array get the mask word.
// A wave stages whole rows (one coalesced 256-B load of g and of flags per row, the row's offset is scalar), a thread one
// word of the 2H halo columns. EDGE = false: the block's halo rows and planes all lie inside the array and its 64 columns
// inside the row (only the halo COLUMNS can stick out): no clamps, no row tests.
template <int TZ, int H, bool EDGE, int NW>       // TZ: planes of the tile; NW: waves of the block
__device__ __forceinline__ void stage_masked(float* __restrict__ tile, const float* __restrict__ g,
                                             const float* __restrict__ flags, const Dom& d, int x0, int y0, int k0, int tid) {
  using T = Tile<TZ, H>;
  constexpr int NT = 64 * NW, ROWS = T::LY * T::LZ;
  if (TFL_SCAL3_ABL & 1) {
    for (int it = tid; it < T::N; it += NT) tile[it] = 0.5f;
    return;
  }
  const int lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int gx = x0 + lane;
  const unsigned xc4 = (unsigned)(EDGE ? min(gx, d.X - 1) : gx) * 4u;
  const bool xin = EDGE ? gx < d.X : true;
  constexpr int PER = (ROWS + NW - 1) / NW;
  float sv[PER], fv[PER];
  bool okr[PER];
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const int r = w + q * NW;                                // wave-uniform
    const int rz = r / T::LY, ry = r - rz * T::LY;
    const int gy = y0 - H + ry, gz = k0 - H + rz;
    okr[q] = EDGE ? (r < ROWS && gy >= 0 && gy < d.Y && gz >= 0 && gz < d.Z) : true;
    const int yc = EDGE ? min(max(gy, 0), d.Y - 1) : gy, zc = EDGE ? min(max(gz, 0), d.Z - 1) : ((ROWS % NW) ? min(gz, d.Z - 1) : gz);
    const unsigned ro4 = (unsigned)__builtin_amdgcn_readfirstlane((yc * d.sy + zc * d.sz) * 4);
    sv[q] = ldg(g, ro4 + xc4); fv[q] = ldg(flags, ro4 + xc4);
  }
  // the 2H halo columns of every row: one item per thread
  constexpr int ITEMS = ROWS * 2 * H, HPER = (ITEMS + NT - 1) / NT;
  float hs[HPER], hf[HPER];
  bool hok[HPER];
  int hdst[HPER];
#pragma unroll
  for (int q = 0; q < HPER; q++) {
    const int it = min(tid + q * NT, ITEMS - 1);
    const int r = it / (2 * H), c = it - r * (2 * H);
    const int rz = r / T::LY, ry = r - rz * T::LY;
    const int hx = c < H ? c : T::LX - 2 * H + c;
    const int hgx = x0 - H + hx, gy = y0 - H + ry, gz = k0 - H + rz;
    hok[q] = tid + q * NT < ITEMS && hgx >= 0 && hgx < d.X && (!EDGE || (gy >= 0 && gy < d.Y && gz >= 0 && gz < d.Z));
    const int yc = EDGE ? min(max(gy, 0), d.Y - 1) : gy, zc = EDGE ? min(max(gz, 0), d.Z - 1) : gz;
    const unsigned o4 = (unsigned)(min(max(hgx, 0), d.X - 1) + yc * d.sy + zc * d.sz) * 4u;
    hs[q] = ldg(g, o4); hf[q] = ldg(flags, o4);
    hdst[q] = rz * T::LP + ry * T::LX + hx;
  }
#pragma unroll
  for (int q = 0; q < PER; q++) {
    const int r = w + q * NW;
    const int rz = r / T::LY, ry = r - rz * T::LY;
    if ((ROWS % NW) == 0 || r < ROWS) tile[rz * T::LP + ry * T::LX + H + lane] = mask_word(sv[q], fv[q], okr[q] && xin);
  }
#pragma unroll
  for (int q = 0; q < HPER; q++)
    if (tid + q * NT < ITEMS) tile[hdst[q]] = mask_word(hs[q], hf[q], hok[q]);
}

// getCentered (third_party/grid.cc:346-377) of a cell whose +1 neighbours exist (not a border cell); o4 = byte offset of the cell
__device__ __forceinline__ v3 centred(const Dom& d, const float* __restrict__ U, unsigned o4) {
  const unsigned sc4 = (unsigned)d.sc * 4u;
  v3 r;
  r.x = 0.5f * (ldg(U, o4) + ldg(U, o4 + 4u));
  r.y = 0.5f * (ldg(U, o4 + sc4) + ldg(U, o4 + sc4 + (unsigned)d.sy * 4u));
  r.z = 0.5f * (ldg(U, o4 + 2u * sc4) + ldg(U, o4 + 2u * sc4 + (unsigned)d.sz * 4u));
  return r;
}

// the same for a cell that may be a border cell: `deep` false -> all six loads read the cell itself (valid), the result is unused
__device__ __forceinline__ v3 centred_if(const Dom& d, const float* __restrict__ U, unsigned o4, bool deep) {
  const unsigned sc4 = (unsigned)d.sc * 4u;
  const unsigned ex = deep ? 4u : 0u, ey = deep ? (unsigned)d.sy * 4u : 0u, ez = deep ? (unsigned)d.sz * 4u : 0u;
  v3 r;
  r.x = 0.5f * (ldg(U, o4) + ldg(U, o4 + ex));
  r.y = 0.5f * (ldg(U, o4 + sc4) + ldg(U, o4 + sc4 + ey));
  r.z = 0.5f * (ldg(U, o4 + 2u * sc4) + ldg(U, o4 + 2u * sc4 + ez));
  return r;
}

// One ordinary back-trace (calcLineTrace with length <= kFastLen: a single step) from the centre of a cell that is not
// a border cell; false = the lane needs the generic trace (long displacement, NaN, end point not in a fluid cell).
// `mt` = the masked tile, `e` = tile index of the end point's cell.
// FAST (the tolerance mode, tfl_set_advect_mode): end point = centre + displacement, thresholds on the squared length.
template <bool FAST>
__device__ __forceinline__ bool trace_fast(const float* __restrict__ mt, int LX, int LP, int cbias, v3 ctr, v3 u, float ndt, v3& p, int& e) {
  const float dx = u.x * ndt, dy = u.y * ndt, dz = u.z * ndt;     // scale3(u, -dt)
  const float l2 = dx * dx + dy * dy + dz * dz;                   // vec3::norm, vec3.h:119-127
  const bool nz = l2 > 1e-6f;
  bool shortd;
  if (FAST) {
    p.x = nz ? ctr.x + dx : ctr.x; p.y = nz ? ctr.y + dy : ctr.y; p.z = nz ? ctr.z + dz : ctr.z;
    shortd = l2 <= kFastLen * kFastLen;
  } else {
    float len_, r_;
    sqrt_rcp_exact(l2, len_, r_);          // ONE transcendental (v_rsq) for the root and the reciprocal (tfl_fastmath.hpp; bit-equal, profiles/r03_exact_math.txt)
    const float len = nz ? len_ : 0.0f, r = nz ? r_ : 0.0f;                 // len == 0: direction 0, p = ctr (the reference returns pos)
    const float qx = div_by<1>(dx, len, r), qy = div_by<1>(dy, len, r), qz = div_by<1>(dz, len, r);
    p.x = ctr.x + qx * len;                                       // next = pos + dt * step, step = min(length, 1) = length
    p.y = ctr.y + qy * len;
    p.z = ctr.z + qz * len;
    shortd = len <= kFastLen;
  }
  // a NaN displacement gives len = 0 and p = NaN: (int)NaN = 0 would index outside the tile, keep the lane's own cell
  const bool fin = shortd && p.x == p.x && p.y == p.y && p.z == p.z;
  e = fin ? __mul24((int)p.z, LP) + (__mul24((int)p.y, LX) + (int)p.x) + cbias : cbias;
  const float w = mt[fin ? e : 0];
  return fin && !is_mask(w);
}

// getInterpolatedWithFluidHi (grid.cc:204-332) on the masked tile at a position the fast trace produced: p - 0.5 lies in
// (i - 1, i + 1) on every axis, so buildIndex's clamps cannot act; pc - float(int(pc)) == fract(pc) for pc >= 0.
// First the plain trilinear form (what the reference computes when all eight corners are fluid, same operation order).
// A NaN result means a masked corner took part (or the field holds a NaN): the 1-D lerps are then redone the way
// lerp_fluid (grid.cc:204-222) drops non-fluid taps -- every lane next to a wall or an obstacle, but still from the tile.
// Returns the mask word when no corner is fluid (the reference then samples without flags: generic path).
template <bool FAST>
__device__ __forceinline__ float lerp1(float a, float b, float t0, float t1) {
  return FAST ? __builtin_fmaf(t1, b - a, a) : a * t0 + b * t1;
}
template <bool FAST>
__device__ __forceinline__ float lerp1_fluid(float a, float b, float t0, float t1) {
  return is_mask(a) ? b : (is_mask(b) ? a : lerp1<FAST>(a, b, t0, t1));
}
template <bool FAST>
__device__ __forceinline__ float lerp_tile(const float* __restrict__ mt, int LX, int LP, int cbias, v3 p) {
  const float px = p.x - 0.5f, py = p.y - 0.5f, pz = p.z - 0.5f;
  const float s1 = __builtin_amdgcn_fractf(px), t1 = __builtin_amdgcn_fractf(py), f1 = __builtin_amdgcn_fractf(pz);
  const float s0 = 1.0f - s1, t0 = 1.0f - t1, f0 = 1.0f - f1;
  const float* q = mt + (__mul24((int)pz, LP) + (__mul24((int)py, LX) + (int)px) + cbias);
  const float g000 = q[0], g010 = q[LX], g100 = q[1], g110 = q[1 + LX];
  const float g001 = q[LP], g011 = q[LP + LX], g101 = q[LP + 1], g111 = q[LP + 1 + LX];
  const float lo = lerp1<FAST>(lerp1<FAST>(g000, g010, t0, t1), lerp1<FAST>(g100, g110, t0, t1), s0, s1);
  const float hi = lerp1<FAST>(lerp1<FAST>(g001, g011, t0, t1), lerp1<FAST>(g101, g111, t0, t1), s0, s1);
  float r = lerp1<FAST>(lo, hi, f0, f1);
  if (r != r) {
    const float lo2 = lerp1_fluid<FAST>(lerp1_fluid<FAST>(g000, g010, t0, t1), lerp1_fluid<FAST>(g100, g110, t0, t1), s0, s1);
    const float hi2 = lerp1_fluid<FAST>(lerp1_fluid<FAST>(g001, g011, t0, t1), lerp1_fluid<FAST>(g101, g111, t0, t1), s0, s1);
    r = lerp1_fluid<FAST>(lo2, hi2, f0, f1);
  }
  return r;
}

// getClampBounds (tfluids.cc:331-378) from global memory for a lane off the fast path: the 3^3 neighbourhood of the
// clamped cell of `pos`, fluid cells inside the grid only. lo = +inf > hi = -inf <=> the reference returns false.
__device__ __forceinline__ void clamp_bounds_global(const Dom& d, const float* __restrict__ s, const float* __restrict__ flags, v3 pos,
                                                    float& lo, float& hi) {
  lo = __builtin_inff(); hi = -__builtin_inff();
  const int i0 = iclampi((int)pos.x, 0, d.X - 1), j0 = iclampi((int)pos.y, 0, d.Y - 1), k0 = iclampi((int)pos.z, 0, d.Zg - 1);
  // all 54 loads first (clamped addresses), then the search: one memory round trip instead of 54 dependent ones
  float fv[27], sv[27];
#pragma unroll
  for (int n = 0; n < 27; n++) {
    const int ii = i0 - 1 + n % 3, jj = j0 - 1 + (n / 3) % 3, kl = k0 - 1 + n / 9 - d.zg;
    const int o = TFL_AT(d, iclampi(ii, 0, d.X - 1), iclampi(jj, 0, d.Y - 1), iclampi(kl, 0, d.Z - 1));
    fv[n] = flags[o]; sv[n] = s[o];
  }
#pragma unroll
  for (int n = 0; n < 27; n++) {
    const int ii = i0 - 1 + n % 3, jj = j0 - 1 + (n / 3) % 3, kk = k0 - 1 + n / 9, kl = kk - d.zg;
    // kl: plane of the local array (a z-slab never consumes cells whose box leaves it)
    const bool in = kk >= 0 && kk < d.Zg && kl >= 0 && kl < d.Z && jj >= 0 && jj < d.Y && ii >= 0 && ii < d.X;
    if (in && ((((int)fv[n]) & kFluid) != 0)) {
      if (sv[n] < lo) lo = sv[n];
      if (sv[n] > hi) hi = sv[n];
    }
  }
}

__device__ __forceinline__ float min3r(float a, float b, float c) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ float max3r(float a, float b, float c) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// Block part of a kernel: block -> plane group / batch item, tile staged (interior blocks without clamps). A block of
// 64 x 4 x TZ threads covers PZ = TZ * KZ planes: thread (lane, ty, tz) owns the cells (i, j, k0 + tz + TZ q), q < KZ --
// with KZ = 2 the same tile is staged by half the waves, twice as many blocks are resident and a launch needs half the
// rounds of them (these kernels are bound by the latency of a block's life -- load, barrier, trace, store -- not by
// throughput: profiles/r04_advect_experiments.txt 9). All global accesses of the fast path are a uniform base + a 32-bit
// byte offset: no 64-bit address arithmetic per lane.
#define TFL_SCAL3_BLOCK(H, SRC) TFL_SCAL3_GEOM(H); TFL_SCAL3_STAGE(H, SRC)
/* the geometry first: a kernel issues its own per-cell loads between GEOM and STAGE, so that they travel with the tile's */ \
/* loads instead of costing a memory round trip of their own behind them (round 4) */
/* (round 6: `sb` = the block's place -- SBlock{-1, ...}: blockIdx as it comes (the kernels below); SBlock{L, gx, gy, gz}: the */ \
/* L-th block of a gx x gy x gz launch, for the pair kernels of advect_pair3.hip -- and `tile` a pointer to T::N floats of LDS) */
#define TFL_SCAL3_GEOM(H)                                                                          \
  constexpr int PZ = TZ * KZ;                                                                      \
  using T = Tile<PZ, H>;                                                                           \
  const Dom& d = a.d;                                                                              \
  int bx_, by_, bz_;                                                                               \
  if (sb.L < 0) block_tile(a.ord, bx_, by_, bz_);                                                  \
  else block_tile_linear(a.ord, (unsigned)sb.L, (unsigned)sb.gx, (unsigned)sb.gy, bx_, by_, bz_);  \
  int b, k0, kend; group_planes<PZ>(d, bz_, sb.L < 0 ? (int)gridDim.z : sb.gz, b, k0, kend);       \
  const long long cells = (long long)d.sc;                                                         \
  s += b * cells; flags += b * cells; U += b * cells * 3;                                          \
  const int lane = threadIdx.x, ty = threadIdx.y, tz = threadIdx.z;                                \
  const int tid = lane + 64 * (ty + TY * tz);                                                      \
  const int x0 = bx_ * TX, y0 = by_ * TY;                                                          \
  const int i = x0 + lane, j = y0 + ty;                                                            \
  const bool inner = y0 >= H && y0 + TY + H <= d.Y && k0 >= H && k0 + PZ + H <= d.Z && x0 + TX <= d.X;   \
  const unsigned sc4 = (unsigned)d.sc * 4u;                                                        \
  const unsigned oxy4 = (unsigned)(min(i, d.X - 1) + __mul24(min(j, d.Y - 1), d.sy)) * 4u;         \
  /* on_border (bnd = 1) without branches: c < 1 || c > N - 2  <=>  unsigned(c - 1) >= unsigned(N - 2) */ \
  const bool border_xy = ((unsigned)(i - 1) >= (unsigned)(d.X - 2)) | ((unsigned)(j - 1) >= (unsigned)(d.Y - 2))
#define TFL_SCAL3_STAGE(H, SRC)                                                                    \
  if (inner) stage_masked<PZ, H, false, TY * TZ>(tile, SRC, flags, d, x0, y0, k0, tid);            \
  else stage_masked<PZ, H, true, TY * TZ>(tile, SRC, flags, d, x0, y0, k0, tid)

// Cell part (inside a loop over q): geometry of the thread's q-th cell
#define TFL_SCAL3_CELL(H, q)                                                                       \
  const int pz = tz + TZ * (q);                       /* plane of the tile's interior */           \
  const int k = k0 + pz;                                                                           \
  const bool live = (i < d.X) & (j < d.Y) & (k < kend);                                            \
  const unsigned o4 = oxy4 + (unsigned)__mul24(min(k, d.Z - 1), d.sz) * 4u;                        \
  const bool border = border_xy | ((unsigned)(k - 1) >= (unsigned)(d.Z - 2));                      \
  const int kg = k + d.zg;                                                                         \
  const bool deep = live & !border & ((unsigned)(kg - 1) < (unsigned)(d.Zg - 2));                  \
  const int c0 = (pz + H) * T::LP + (ty + H) * T::LX + lane + H;                                   \
  const int cbias = c0 - (i + j * T::LX + kg * T::LP);                                             \
  const v3 ctr = mk3((float)i + 0.5f, (float)j + 0.5f, (float)kg + 0.5f)

// ---- pass A / the single-pass method: SemiLagrangeEulerOurs[SavePos] + getClampBounds of the forward position ----------
template <int TZ, int KZ, bool BOUNDS, bool FAST>
__device__ __forceinline__ void scal3_fwd_body(const SBlock sb, float* __restrict__ tile, const AdvArgs& a, const float* __restrict__ s,
                                               const float* __restrict__ U, const float* __restrict__ flags, float* __restrict__ out,
                                               float* __restrict__ bounds) {
  constexpr int HH = BOUNDS ? 2 : 1;
  TFL_SCAL3_GEOM(HH);
  out += b * cells;
  if (BOUNDS) bounds += b * cells * 3;
  // loads that do not depend on the tile -- the cells' own values and the six faces of their centred velocities -- are asked
  // for BEFORE the tile's (unconditionally: a cell that is not `deep` reads itself)
  float svq[KZ];
  v3 uq[KZ];
#pragma unroll
  for (int q = 0; q < KZ; q++) {
    TFL_SCAL3_CELL(HH, q);
    (void)c0; (void)cbias; (void)ctr;
    svq[q] = ldg(s, o4);
    uq[q] = centred_if(d, U, o4, deep);
  }
  TFL_SCAL3_STAGE(HH, s);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < KZ; q++) {
    TFL_SCAL3_CELL(HH, q);
    if (!live) continue;
    if (border) { stg(out, o4, 0.0f); continue; }
    const v3 u = uq[q];
    const bool fl = !is_mask(tile[c0]);
    float v = svq[q];
    int e = c0;                     // cell of the forward position: the cell itself where nothing is advected (tfluids.cc:159-163)
    bool slow = false;
    if (fl) {
      slow = true;
      if (deep) {
        if (TFL_SCAL3_ABL & 8) { v = tile[c0] + u.x; slow = false; }
        else {
          v3 p;
          const bool ok = trace_fast<FAST>(tile, T::LX, T::LP, cbias, ctr, u, -a.dt, p, e);
          const float r = lerp_tile<FAST>(tile, T::LX, T::LP, cbias, ok ? p : ctr);
          if (ok && !is_mask(r)) { v = r; slow = false; }
        }
      }
    }
    float lo = __builtin_inff(), hi = -__builtin_inff();
    if (BOUNDS && !slow && (TFL_SCAL3_ABL & 2)) { lo = tile[e]; hi = lo; }
    if (BOUNDS && !slow && !(TFL_SCAL3_ABL & 2)) {
      const float* qq = tile + (e - 1 - T::LX - T::LP);
      float t[27];
#pragma unroll
      for (int n = 0; n < 27; n++) t[n] = qq[(n / 9) * T::LP + ((n / 3) % 3) * T::LX + (n % 3)];
#pragma unroll
      for (int n = 0; n < 26; n += 2) { lo = min3r(lo, t[n], t[n + 1]); hi = max3r(hi, t[n], t[n + 1]); }
      lo = min3r(lo, t[26], t[26]); hi = max3r(hi, t[26], t[26]);
    }
    if (slow) {   // rare lanes: the generic trace + fluid-aware sampler on global memory
      v3 back;
      v = sl_euler_ours<true>(a, flags, U, s, a.dt, i, j, k, back);
      if (BOUNDS) clamp_bounds_global(d, s, flags, back, lo, hi);
    }
    if ((TFL_SCAL3_ABL & 4) && a.dt != 12345.0f) continue;
    stg(out, o4, v);
    if (BOUNDS) { stg(bounds, o4, lo); stg(bounds, o4 + sc4, hi); }
  }
}
template <int TZ, int KZ, bool BOUNDS, bool FAST>
__global__ __launch_bounds__(256 * TZ) void k_scal3_fwd(AdvArgs a, const float* __restrict__ s, const float* __restrict__ U,
                                                        const float* __restrict__ flags, float* __restrict__ out,
                                                        float* __restrict__ bounds) {
  __shared__ float tile[Tile<TZ * KZ, BOUNDS ? 2 : 1>::N];
  scal3_fwd_body<TZ, KZ, BOUNDS, FAST>(SBlock{-1, 0, 0, 0}, tile, a, s, U, flags, out, bounds);
}

// ---- pass B: backward trace on fwd + MacCormackCorrect + MacCormackClampOurs ---------------------------------------------
template <int TZ, int KZ, bool FAST>
__device__ __forceinline__ void scal3_bwd_body(const SBlock sb, float* __restrict__ tile, const AdvArgs& a, double half_strength,
                                               const float* __restrict__ s, const float* __restrict__ U, const float* __restrict__ flags,
                                               const float* __restrict__ fwd, const float* __restrict__ bounds, float* __restrict__ dst,
                                               const BcFoldArg& folda) {
  TFL_SCAL3_GEOM(1);
  fwd += b * cells; dst += b * cells; bounds += b * cells * 3;
  const bool fold_blk = fold_block(folda, y0, y0 + TY - 1, k0, kend - 1);
  // per-cell loads first (see k_scal3_fwd), then the tile of the forward field
  float svq[KZ], fq[KZ], bloq[KZ], bhiq[KZ];
  v3 uq[KZ];
#pragma unroll
  for (int q = 0; q < KZ; q++) {
    TFL_SCAL3_CELL(1, q);
    (void)c0; (void)cbias; (void)ctr;
    svq[q] = ldg(s, o4); fq[q] = ldg(fwd, o4);
    bloq[q] = ldg(bounds, o4); bhiq[q] = ldg(bounds, o4 + sc4);
    uq[q] = centred_if(d, U, o4, deep);
  }
  TFL_SCAL3_STAGE(1, fwd);
  __syncthreads();
#pragma unroll
  for (int q = 0; q < KZ; q++) {
    TFL_SCAL3_CELL(1, q);
    if (!live) continue;
    const v3 u = uq[q];
    const float sv = svq[q], f = fq[q], blo = bloq[q], bhi = bhiq[q];
    // fluid cell <=> its word of the masked tile is not the mask word (a border cell of the array is never `deep`, and its
    // flags are read directly: the correction below has no border test)
    const bool fl = border ? ((((int)ldg(flags, o4)) & kFluid) != 0) : !is_mask(tile[c0]);
    float bwd = border ? 0.0f : f;
    if (fl && !border) {
      bool slow = true;
      if (deep) {
        if (TFL_SCAL3_ABL & 8) { bwd = tile[c0] + u.x; slow = false; }
        else {
          v3 p; int e;
          const bool ok = trace_fast<FAST>(tile, T::LX, T::LP, cbias, ctr, u, a.dt, p, e);
          const float r = lerp_tile<FAST>(tile, T::LX, T::LP, cbias, ok ? p : ctr);
          if (ok && !is_mask(r)) { bwd = r; slow = false; }
        }
      }
      if (slow) { v3 back; bwd = sl_euler_ours<true>(a, flags, U, fwd, -a.dt, i, j, k, back); }
    }
    // MacCormackCorrect has no border test; the unsuffixed 0.5 makes the reference evaluate the correction in double and
    // round once (tfluids.cc:231)
    float v = f;
    if (fl) v = FAST ? __builtin_fmaf((float)half_strength, sv - bwd, f) : (float)((double)f + half_strength * (double)(sv - bwd));
    if (!border) v = (blo > bhi) ? f : fclampf(v, blo, bhi);
    if ((TFL_SCAL3_ABL & 4) && a.dt != 12345.0f) continue;
    // the setConstVals that follows the advection in simulate() (tfl_host.hpp BcFold): rows inside the pair's box only
    if (fold_blk) {
      const BcFold fold = *folda.dev;
      if (fold_row(fold, j, k) && fold_col(fold, i)) v = v * ldg(fold.inv + b * cells, o4) + ldg(fold.bc + b * cells, o4);
    }
    stg(dst, o4, v);
  }
}
template <int TZ, int KZ, bool FAST>
__global__ __launch_bounds__(256 * TZ) void k_scal3_bwd(AdvArgs a, double half_strength, const float* __restrict__ s,
                                                        const float* __restrict__ U, const float* __restrict__ flags,
                                                        const float* __restrict__ fwd, const float* __restrict__ bounds,
                                                        float* __restrict__ dst, BcFoldArg folda) {
  __shared__ float tile[Tile<TZ * KZ, 1>::N];
  scal3_bwd_body<TZ, KZ, FAST>(SBlock{-1, 0, 0, 0}, tile, a, half_strength, s, U, flags, fwd, bounds, dst, folda);
}

// pass A (or the single-pass method) / pass B with a block of 64 x 4 x TZ threads and KZ planes per thread
template <int TZ, int KZ, bool FAST>
void launch_a(hipStream_t st, bool two_pass, const AdvArgs& a, int B, const float* s, const float* U, const float* flags, float* out,
              float* bounds) {
  const Dom& d = a.d;
  constexpr int PZ = TZ * KZ;
  const int G = (d.n0 + PZ - 1) / PZ + (d.nw - d.n0 + PZ - 1) / PZ;
  const dim3 blk(TX, TY, TZ), grd((d.X + TX - 1) / TX, (d.Y + TY - 1) / TY, (unsigned)(G * B));
  if (grd.x * grd.y * grd.z == 0) return;
  TFL_TIMED_EXT("k_scalar_fwd", st);
  AdvArgs ao = a; ao.ord = make_block_order(grd.x, grd.y, grd.z, xcd_order_enabled(), xcd_run(grd.x, grd.y));
  if (two_pass) TFL_LAUNCH_EXT((k_scal3_fwd<TZ, KZ, true, FAST>), grd, blk, 0, st, ao, s, U, flags, out, bounds);
  else TFL_LAUNCH_EXT((k_scal3_fwd<TZ, KZ, false, FAST>), grd, blk, 0, st, ao, s, U, flags, out, (float*)nullptr);
}
template <int TZ, int KZ, bool FAST>
void launch_b(hipStream_t st, const AdvArgs& a, int B, const float* s, const float* U, const float* flags, const float* fwd,
              const float* bounds, float* dst, const BcFoldArg& fold) {
  const Dom& d = a.d;
  constexpr int PZ = TZ * KZ;
  const int G = (d.n0 + PZ - 1) / PZ + (d.nw - d.n0 + PZ - 1) / PZ;
  const dim3 blk(TX, TY, TZ), grd((d.X + TX - 1) / TX, (d.Y + TY - 1) / TY, (unsigned)(G * B));
  if (grd.x * grd.y * grd.z == 0) return;
  TFL_TIMED_EXT("k_scalar_bwd", st);
  AdvArgs ao = a; ao.ord = make_block_order(grd.x, grd.y, grd.z, xcd_order_enabled(), xcd_run(grd.x, grd.y));
  TFL_LAUNCH_EXT((k_scal3_bwd<TZ, KZ, FAST>), grd, blk, 0, st, ao, (double)a.strength * 0.5, s, U, flags, fwd, bounds, dst, fold);
}


#ifdef TFL_EXPERIMENTS
#include "advect_scalar3_march.inc"      // the z-marched form of the two passes (round 5: bit-identical, slower)
#endif

template <bool FAST>
void launch(hipStream_t st, int shape, bool two_pass, const AdvArgs& a, int B, const float* s, const float* U, const float* flags,
            float* fwd, float* bounds, float* dst, int stages) {
  const bool pa = stages & 2, pb = two_pass && (stages & 4);
  float* outA = two_pass ? fwd : dst;
  // Block shapes (threads in z x planes per thread), measured at 128^3 / 256^3 (profiles/r04_advect_experiments.txt 9):
  // pass A 2 x 1: 26.3 / 209 us, 1 x 2: 28.3 / 220, 1 x 4: 28.0 / 185, 1 x 1: 29.6 / 244; pass B 2 x 1: 23.5 / 191,
  // 1 x 2: 22.0 / 153, 1 x 4: 24.5 / 157. Default: pass A 2 x 1 below 6 M cells per item and 1 x 4 above, pass B 1 x 2.
  const bool big = (long long)a.d.sc >= 6000000ll;
  const int sa = shape ? shape : (big ? 14 : 2), sb = shape ? shape : 12;
  if (pa) switch (sa) {
    case 1:  launch_a<1, 1, FAST>(st, two_pass, a, B, s, U, flags, outA, bounds); break;
    case 12: launch_a<1, 2, FAST>(st, two_pass, a, B, s, U, flags, outA, bounds); break;
    case 14: launch_a<1, 4, FAST>(st, two_pass, a, B, s, U, flags, outA, bounds); break;
    default: launch_a<2, 1, FAST>(st, two_pass, a, B, s, U, flags, outA, bounds); break;
  }
  const BcFoldArg fold = pb ? take_fold() : no_fold();   // pass B writes the operator's result
  if (pb) switch (sb) {
    case 1:  launch_b<1, 1, FAST>(st, a, B, s, U, flags, fwd, bounds, dst, fold); break;
    case 2:  launch_b<2, 1, FAST>(st, a, B, s, U, flags, fwd, bounds, dst, fold); break;
    case 14: launch_b<1, 4, FAST>(st, a, B, s, U, flags, fwd, bounds, dst, fold); break;
    default: launch_b<1, 2, FAST>(st, a, B, s, U, flags, fwd, bounds, dst, fold); break;
  }
}

}  // namespace

#ifndef TFL_SCAL3_NO_ENTRY      // (advect_pair3.hip includes this file for the kernels' bodies only)
bool advect_scalar3(hipStream_t st, bool two_pass, const AdvArgs& a, int B, const float* s, const float* U, const float* flags,
                    float* fwd, float* bounds, float* dst, int stages) {
  static const bool off = exp_env("TFL_ADVECT_GATHER") != nullptr || exp_env("TFL_SCALAR_GATHER") != nullptr;   // A/B switch: the round-2 gather kernels
  static const int tzsel = getenv("TFL_SCAL3_TZ") ? atoi(getenv("TFL_SCAL3_TZ")) : 0;   // 0 = per pass and grid size (launch)
  const Dom& d = a.d;
  // 24-bit multiplies address the tile and the planes; 32-bit BYTE offsets the cells of the three velocity channels
  if (off || a.outside || d.Z < 3 || (long long)d.X * d.Y * 4 >= (1 << 24) || 12ll * d.sc >= (1ll << 32)) return false;
#ifdef TFL_EXPERIMENTS
  // the z-marched kernels (round 5): TFL_SCAL3_MARCH=1 -- bit-identical, one staging pass per plane, and SLOWER than the
  // tile kernels (profiles/r05_advect_experiments.txt: pass A 29.5 vs 25.3 us at 128^3, 176 vs 162 at 256^3; pass B 21.3 vs
  // 20.8, 143 vs 141): these passes run at the length of their instruction streams, and the ring addressing + the per-plane
  // staging bookkeeping make the marched stream LONGER per cell (345 against 274 vector instructions in pass A)
  static const bool march = getenv("TFL_SCAL3_MARCH") && atoi(getenv("TFL_SCAL3_MARCH")) == 1;
  if (march && !tzsel) {
    if (a.fast) zm::launch<true>(st, two_pass, a, B, s, U, flags, fwd, bounds, dst, stages);
    else zm::launch<false>(st, two_pass, a, B, s, U, flags, fwd, bounds, dst, stages);
    return true;
  }
#endif
  if (a.fast) launch<true>(st, tzsel, two_pass, a, B, s, U, flags, fwd, bounds, dst, stages);
  else launch<false>(st, tzsel, two_pass, a, B, s, U, flags, fwd, bounds, dst, stages);
  return true;
}
#endif

}  // namespace tfl
