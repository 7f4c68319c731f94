// advect_pair3.hip -- advectScalar and advectVel of one simulate() step as TWO launches instead of four (round 6; VERDICT r05
// item 1b: "advectScalar || advectVel passes A as one launch"). The two operators are independent -- both read the pre-advection
// U (lib/simulate.lua:183-200), their temporaries are disjoint -- so the passes A of both run as two block ranges of ONE 1-D
// launch, and likewise the passes B (when pass B of advectVel does not add the buoyancy force, which needs the advected density:
// the z-slab step). On a z-slab of 16-40 planes every advection launch sits at its 5-10 us floor (ramp + tail of a launch that
// does not fill the chip for long); a pair costs about what its longer half does. (Two streams do not work for this on this
// stack -- an event hop costs 12-15 us, tools/ubench/host_costs.hip -- and hipExtAnyOrderLaunch is ignored on gfx9.)
// The kernels ARE the tile kernels of advect_scalar3.hip (block shape 64 x 4 threads, two planes per thread) and
// advect_vel3_kz1.inc, called as device functions with the block's place in its launch passed in: same code, same bits
// (tests/test_hip_simulate.py: the z-slab runs against the un-cut step; tests/test_hip_parity.py against the oracle).
#include "tfl_advect.hpp"
#include "tfl_fastmath.hpp"

#include <cstdlib>

#define TFL_SCAL3_NO_ENTRY
#include "advect_scalar3.hip"      // namespace tfl { namespace { scal3_fwd_body, scal3_bwd_body, Tile, SBlock ... } }
#ifndef TFL_VEL3_ABL
#define TFL_VEL3_ABL 0
#endif

namespace tfl {
namespace {
#include "advect_vel3_kz1.inc"     // namespace kz1 { vel3_fwd_body, vel3_bwd_body, VBlock ... }

constexpr int kPairTileA = Tile<2, 2>::N > 4 * kz1::LN ? Tile<2, 2>::N : 4 * kz1::LN;      // floats of LDS: the larger of the two tiles
constexpr int kPairTileB = Tile<2, 1>::N > 4 * kz1::LN ? Tile<2, 1>::N : 4 * kz1::LN;

struct PairGrid { int ns, sgx, sgy, sgz, vgx, vgy, vgz; };      // blocks [0, ns): the scalar pass (sgx x sgy x sgz tiles), the rest: the velocity pass

__device__ __forceinline__ kz1::VBlock vel_block(const PairGrid& g, int m) {
  const int r = m / g.vgx;
  const int z = r / g.vgy;
  return kz1::VBlock{m - r * g.vgx, r - z * g.vgy, z, g.vgz};
}

template <bool FAST>
__global__ __launch_bounds__(256) void k_adv3_fwd_pair(PairGrid g, AdvArgs as, const float* __restrict__ s, const float* __restrict__ U,
                                                       const float* __restrict__ flags, float* __restrict__ sfwd, float* __restrict__ bounds,
                                                       AdvArgs av, float* __restrict__ vfwd) {
  __shared__ float tile[kPairTileA];
  const int L = (int)blockIdx.x;      // (block-uniform branch)
  if (L < g.ns) scal3_fwd_body<1, 2, true, FAST>(SBlock{L, g.sgx, g.sgy, g.sgz}, tile, as, s, U, flags, sfwd, bounds);
  else kz1::vel3_fwd_body<FAST>(vel_block(g, L - g.ns), tile, av, U, flags, vfwd);
}

template <bool FAST>
__global__ __launch_bounds__(256) void k_adv3_bwd_pair(PairGrid g, AdvArgs as, double half_strength, const float* __restrict__ s,
                                                       const float* __restrict__ U, const float* __restrict__ flags,
                                                       const float* __restrict__ sfwd, const float* __restrict__ bounds, float* __restrict__ sdst,
                                                       BcFoldArg fold_s, AdvArgs av, const float* __restrict__ vfwd, float* __restrict__ vdst,
                                                       BcFoldArg fold_v) {
  __shared__ float tile[kPairTileB];
  const int L = (int)blockIdx.x;
  if (L < g.ns) scal3_bwd_body<1, 2, FAST>(SBlock{L, g.sgx, g.sgy, g.sgz}, tile, as, half_strength, s, U, flags, sfwd, bounds, sdst, fold_s);
  else kz1::vel3_bwd_body<FAST, 0>(vel_block(g, L - g.ns), tile, av, half_strength, U, flags, vfwd, vdst, fold_v, BuoyFold{nullptr, 0.0f, 0.0f, 0.0f});
}

}  // namespace

// stages: 2 = the two passes A, 4 = the two passes B (tfl_set_stages as for the operators). false = not taken (shape outside the
// tile kernels' range, a grid big enough for the two-planes-per-block velocity kernels, the gather kernels asked for): the caller
// runs the two operators one after the other. fold_s / fold_v: the setConstVals pairs pass B may apply (tfl_host.hpp BcFold).
bool advect_pair3(hipStream_t st, const AdvArgs& a0, int B, const float* s, const float* U, const float* flags, float* sfwd, float* sbounds,
                  float* sdst, float* vfwd, float* vdst, int stages, const BcFoldArg& fold_s, const BcFoldArg& fold_v) {
  static const bool off = (getenv("TFL_ADV_PAIR") && atoi(getenv("TFL_ADV_PAIR")) == 0) || exp_env("TFL_ADVECT_GATHER") || exp_env("TFL_SCALAR_GATHER") ||
                          getenv("TFL_VEL3_KZ") || getenv("TFL_SCAL3_TZ");      // (a forced block shape means the caller wants THOSE kernels)
  const Dom& d = a0.d;
  if (off || a0.outside || d.Z < 3 || (long long)d.X * d.Y * 4 >= (1 << 24) || 12ll * d.sc >= (1ll << 32) || (long long)d.sc >= 6000000ll) return false;
  const int G = (d.n0 + 1) / 2 + (d.nw - d.n0 + 1) / 2;      // scalar: groups of two planes over the window's two runs
  PairGrid g;
  g.sgx = (d.X + TX - 1) / TX; g.sgy = (d.Y + TY - 1) / TY; g.sgz = G * B; g.ns = g.sgx * g.sgy * g.sgz;
  g.vgx = (d.X + kz1::TX - 1) / kz1::TX; g.vgy = (d.Y + kz1::TY - 1) / kz1::TY; g.vgz = d.nw * B;
  const long long nv = (long long)g.vgx * g.vgy * g.vgz;
  if (g.ns <= 0 || nv <= 0 || g.ns + nv >= (1ll << 31)) return false;
  AdvArgs as = a0, av = a0;
  as.ord = make_block_order(g.sgx, g.sgy, g.sgz, xcd_order_enabled(), xcd_run(g.sgx, g.sgy));      // as the scalar launchers: runs of tiles per XCD
  av.outside = 0;
  const unsigned grid = (unsigned)(g.ns + nv);
  if (stages & 2) {
    TFL_TIMED_EXT("k_adv_fwd_pair", st);
    if (a0.fast) TFL_LAUNCH_EXT((k_adv3_fwd_pair<true>), grid, dim3(64, 4, 1), 0, st, g, as, s, U, flags, sfwd, sbounds, av, vfwd);
    else TFL_LAUNCH_EXT((k_adv3_fwd_pair<false>), grid, dim3(64, 4, 1), 0, st, g, as, s, U, flags, sfwd, sbounds, av, vfwd);
  }
  if (stages & 4) {
    TFL_TIMED_EXT("k_adv_bwd_pair", st);
    const double hs = (double)a0.strength * 0.5;
    if (a0.fast) TFL_LAUNCH_EXT((k_adv3_bwd_pair<true>), grid, dim3(64, 4, 1), 0, st, g, as, hs, s, U, flags, (const float*)sfwd, (const float*)sbounds, sdst, fold_s, av, (const float*)vfwd, vdst, fold_v);
    else TFL_LAUNCH_EXT((k_adv3_bwd_pair<false>), grid, dim3(64, 4, 1), 0, st, g, as, hs, s, U, flags, (const float*)sfwd, (const float*)sbounds, sdst, fold_s, av, (const float*)vfwd, vdst, fold_v);
  }
  return true;
}

}  // namespace tfl
