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
//  * What bounds these kernels now (profiles/r03_pmc_adv.txt, profiles/r01_r04_where_the_time_went.md): instruction ISSUE -- a SIMD retires one
//    instruction of any kind (VALU, SALU, LDS, VMEM) per ~2.7 clocks here, so the count of ALL instructions per
//    wave is the budget; packed fp32 (v_pk_mul/add) issues at half rate and buys nothing (SLP vectorisation is off
//    for this file). A z-marched variant (4-slot plane ring, 1.55x instead of 4.6x staging) issued 24 % fewer
//    instructions per cell but lost more to block-count quantisation at 128^3 (2048 long blocks on 256 CUs).
//
// Algorithmic HBM bytes per cell are unchanged: pass A 28 B (U3, flags -> fwd3), pass B 40 B (fwd3, U3, flags -> dst3).
#include "tfl_advect.hpp"
#include "tfl_fastmath.hpp"

#include <cstdlib>

// timing ablations (tools/ab_build.sh -DTFL_VEL3_ABL=..): 1 = the tile is filled with constants instead of staged (no staging
// loads; every lane on the fast path), 2 = pass B's 24 gathers of the forward field replaced by the cell's own value,
// 4 = no stores
#ifndef TFL_VEL3_ABL
#define TFL_VEL3_ABL 0
#endif

namespace tfl {
namespace {
// block depth 1 (a block = 64 x 4 cells of one plane: the round-3 kernels, no loop, 54 / 72 VGPRs, no SGPR spills in pass A)
#include "advect_vel3_kz1.inc"
// block depth 2 (two planes per block on one staged tile: 3.1 instead of 4.6 staged words per cell and field, at the price of
// registers -- 63 / 94 VGPRs, SGPR spills): the better one once the staging traffic leaves the caches
#define TFL_VEL3_KZ 2
#define TFL_VEL3_NS kz2
#include "advect_vel3.inc"
#undef TFL_VEL3_KZ
#undef TFL_VEL3_NS
}  // namespace

bool advect_vel3(hipStream_t st, bool two_pass, const AdvArgs& a, int B, const float* U, const float* flags, float* fwd,
                 float* dst, int stages) {
  static const bool off = exp_env("TFL_ADVECT_GATHER") != nullptr;   // A/B switch: the round-2 gather kernels
  static const int kz_env = getenv("TFL_VEL3_KZ") ? atoi(getenv("TFL_VEL3_KZ")) : 0;
  const Dom& d = a.d;
  // 24-bit multiplies address the planes (4*X*Y < 2^24); 32-bit BYTE offsets address the cells of all three channels
  // (o4 + 2*sc4 in the loads / stores of U, fwd, dst): 12*Z*Y*X < 2^32. Larger grids take the gather kernels.
  if (off || d.Z < 3 || (long long)d.X * d.Y * 4 >= (1 << 24) || 12ll * d.sc >= (1ll << 32)) return false;
  // block depth: at 128^3 the four fields' staging stays in the caches and the leaner one-plane kernels win (k_vel_bwd
  // 37.5 vs 42 us); from ~6 M cells per batch item on the two-plane kernels do (256^3: 285 vs 316 us, pass A 144 vs 175)
  const bool deep = kz_env ? kz_env >= 2 : (long long)d.sc >= 6000000ll;
  // ... unless pass B has been asked to add the buoyancy force (tfl_host.hpp BuoyFold): only the one-plane pass B carries that
  // fold (and the sparse setConstVals pair's) -- in the two-plane kernel it costs the fifth wave per SIMD -- and at 256^3 the
  // one-plane pass B with the force (~320 us) beats the two-plane one plus k_add_buoyancy and two k_apply_bcs_indexed
  // (291 + 94 + 11 us). Pass A stays two-plane. TFL_VEL3_KZ_B=2 keeps the old split.
  static const int kzb_env = exp_env("TFL_VEL3_KZ_B") ? atoi(exp_env("TFL_VEL3_KZ_B")) : 0;
  const bool b_shallow = deep && two_pass && (stages & 4) && g_buoy.rho && kzb_env != 2;
  if (deep && b_shallow) {
    if (stages & 2) kz2::launch(st, two_pass, a, B, U, flags, fwd, dst, stages & ~4);
    kz1::launch(st, two_pass, a, B, U, flags, fwd, dst, stages & ~2);
  } else if (deep) kz2::launch(st, two_pass, a, B, U, flags, fwd, dst, stages);
  else kz1::launch(st, two_pass, a, B, U, flags, fwd, dst, stages);
  return true;
}

}  // namespace tfl
