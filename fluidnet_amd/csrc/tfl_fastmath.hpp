// tfl_fastmath.hpp -- correctly rounded fp32 division and square root for operands of KNOWN range (gfx950).
//
// hipcc's `/` and sqrtf() are correctly rounded for every operand: v_div_scale x2 + v_rcp + 5 FMA-class ops +
// v_div_fmas + v_div_fixup per quotient (11 VALU, 41 issue clocks measured), v_sqrt + 14 fix-up / range ops per
// root. The line trace (generic/calc_line_trace.cc:313-345, vec3.h:119-127) divides the three components of one
// vector by its own norm: one denominator, in (1e-3, 0.99] on the advection fast path, numerators no larger than it.
// In that range the scaling, the special-case fix-up and two of the three reciprocals are dead weight:
//   r  = refined reciprocal of b (v_rcp + one Newton step)                     3 VALU, shared by the 3 quotients
//   q0 = a * r; e = fma(-b, q0, a) (exact remainder); q = fma(e, r, q0)        3 VALU per quotient (+2 per extra step)
// tools/ubench/exact_math.hip checks bit equality with `/` and sqrtf() on the GPU (the quotient on a SAMPLE: 2^33 hashed pairs + every
// mantissa of b; every float in [2^-40, 2^40) for the root); its verdict is recorded in profiles/.
// (the root exhaustively; for the quotient this is evidence, not a proof: a single Markstein step is only guaranteed
// given a correctly rounded reciprocal -- TFL_DIV_STEPS=2 is the setting for strict parity). Outside the checked range (denormal quotients, b near the ends of the exponent range, inf/NaN) the results may
// differ from `/`: callers use them only where DESIGN 3.2 shows the operands are inside it or the result is discarded.
#pragma once
#include <hip/hip_runtime.h>

namespace tfl {

// number of quotient refinement steps div_by uses on the product path; exact_math.hip prints the number needed
#ifndef TFL_DIV_STEPS
#define TFL_DIV_STEPS 2
#endif

__device__ __forceinline__ float rcp_refined(float b) {
  const float r0 = __builtin_amdgcn_rcpf(b);
  const float e = __builtin_fmaf(-b, r0, 1.0f);
  return __builtin_fmaf(e, r0, r0);
}

// a / b given r = rcp_refined(b)
template <int STEPS = TFL_DIV_STEPS>
__device__ __forceinline__ float div_by(float a, float b, float r) {
  float q = a * r;
#pragma unroll
  for (int s = 0; s < STEPS; s++) q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
  return q;
}

// sqrt(x), x normal and far from the ends of the exponent range: faithful v_sqrt, exact residual, one correction
__device__ __forceinline__ float sqrt_exact(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rsqf(x);
  const float d = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(d, h, s);
}

// len = sqrt(x) and r = 1 / len from ONE transcendental (v_rsq): s0 = x * rsq(x) is within 2 ulp of the root, its
// residual x - s0^2 comes out of one FMA, half the rsq is the correction slope; the rsq is also the seed of the
// reciprocal of len (one Newton step). 7 VALU for both (sqrt_exact + rcp_refined: 8 with three transcendentals).
__device__ __forceinline__ void sqrt_rcp_exact(float x, float& len, float& r) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float s0 = x * y;
  const float h = 0.5f * y;
  const float d = __builtin_fmaf(-s0, s0, x);
  len = __builtin_fmaf(d, h, s0);
  const float e = __builtin_fmaf(-len, y, 1.0f);
  r = __builtin_fmaf(e, y, y);
}

}  // namespace tfl
