// Bit-equality proof runs for the two "cheap but correctly rounded" helpers of the advection fast path
// (fluidnet_amd/csrc/tfl_fastmath.hpp): shared-reciprocal division q = a / b for the three components of
// delta / length (calc_line_trace.cc:343 in the reference), and sqrt for vec3::norm (generic/vec3.h:119-127).
// Both must equal the compiler's correctly rounded `/` and sqrtf() on every operand the fast path can see:
//   b = length in (1e-3, 0.99]   (norm3 returns 0 or > 1e-3; longer traces leave the fast path)
//   |a| <= b (1 + 2^-22)         (a component of the vector whose norm b is) and |a| >= 2^-100 (below that the
//                                 quotient only enters `pos + q * length`, where it is absorbed: DESIGN 3.2)
//   x = l2 in (1e-6, 1]
// A numerator of -0 gives +0 instead of -0 (the refinement adds (+0) + (-0)); the quotient only ever enters
// `pos + q * length`, which absorbs the sign. Those cases are counted separately.
// The run covers much wider ranges: sqrt exhaustively over every float in [2^-40, 2^40]; division over 2^33
// hashed pairs with b in [2^-12, 2^21] plus, for every one of the 2^23 mantissas of b in [0.5, 1), 256 numerators.
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off exact_math.hip -o exact_math
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../fluidnet_amd/csrc/tfl_fastmath.hpp"

__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

struct Counts { unsigned long long n, bad1, bad2, badsqrt, ex_a, ex_b, zsign, badsqrt2, bad3; };

__global__ void k_sqrt(uint32_t lo, uint32_t hi, Counts* c) {
  unsigned long long bad = 0, bad2 = 0, n = 0;
  for (uint64_t u = lo + (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; u < hi; u += (uint64_t)gridDim.x * blockDim.x) {
    const float x = __builtin_bit_cast(float, (uint32_t)u);
    const float want = sqrtf(x), got = tfl::sqrt_exact(x);
    n++;
    if (__builtin_bit_cast(uint32_t, want) != __builtin_bit_cast(uint32_t, got)) { bad++; c->ex_a = (uint32_t)u; }
    float l2, r2; tfl::sqrt_rcp_exact(x, l2, r2);
    if (__builtin_bit_cast(uint32_t, want) != __builtin_bit_cast(uint32_t, l2)) { bad2++; c->ex_b = (uint32_t)u; }
  }
  atomicAdd(&c->n, n); atomicAdd(&c->badsqrt, bad); atomicAdd(&c->badsqrt2, bad2);
}

// mode 0: hashed pairs; mode 1: every mantissa of b in [0.5, 1) x 256 hashed numerators, b scaled into [2^-10, 1);
// mode 2: numerators of any magnitude against the same denominators (the ConvNet input's x / scale)
__global__ void k_div(int mode, uint64_t total, uint32_t seed, Counts* c) {
  unsigned long long bad1 = 0, bad2 = 0, bad3 = 0, n = 0, zs = 0;
  for (uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t h1 = hash32((uint32_t)t ^ seed), h2 = hash32((uint32_t)(t >> 32) * 0x9e3779b9U + h1 + 0x1234567U),
                   h3 = hash32(h2 ^ 0xdeadbeefU);
    float a, b;
    if (mode == 0) {
      // b: random mantissa, exponent 2^-12 .. 2^21
      const int eb = 127 - 12 + (int)(h2 % 34);   // 2^-12 .. 2^21 (the vorticity kernels normalise gradients up to 2^20)
      b = __builtin_bit_cast(float, ((uint32_t)eb << 23) | (h1 & 0x7fffffu));
    } else if (mode == 1) {
      const int eb = 126 - (int)((t >> 23) % 10);
      b = __builtin_bit_cast(float, ((uint32_t)eb << 23) | ((uint32_t)t & 0x7fffffu));
    } else {
      // mode 2: x / scale of the ConvNet's input normalisation: b in [2^-12, 2^21], |a| anywhere in [2^-40, 2^40]
      const int eb = 127 - 12 + (int)(h2 % 34);
      b = __builtin_bit_cast(float, ((uint32_t)eb << 23) | (h1 & 0x7fffffu));
    }
    // a: |a| <= b * (1 + 2^-20) mostly near b's magnitude, sometimes tiny (down to 2^-100), random sign
    const uint32_t kind = h3 >> 28;
    float mag;
    if (kind < 10) {  // ratio in [0, 1]: random mantissa, exponent 0..-8 below b
      const float ratio = __builtin_bit_cast(float, ((uint32_t)(127 - 1 - (h3 >> 8) % 9) << 23) | (h2 & 0x7fffffu)) ;
      mag = b * (2.0f * ratio);  // rounded product; may exceed b slightly
      if (mag > b * 1.000001f) mag = b;
    } else if (kind < 13) {
      const int ea = 27 + (int)((h3 >> 8) % 100);   // 2^-100 .. 2^-1
      mag = __builtin_bit_cast(float, ((uint32_t)ea << 23) | (h2 & 0x7fffffu));
      if (mag > b) mag = b;
    } else if (kind < 15) {
      mag = b;  // quotient exactly 1
      if (kind == 14) mag = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, b) - 1 - (h2 & 3));
    } else {
      mag = 0.0f;
    }
    a = (h3 & 1) ? -mag : mag;
    if (mode == 2) {
      const int ea = 127 - 40 + (int)((h3 >> 4) % 81);
      a = __builtin_bit_cast(float, ((h3 & 1u) << 31) | ((uint32_t)ea << 23) | (h2 & 0x7fffffu));
      const float want2 = a / b, r2 = tfl::rcp_refined(b), q2 = tfl::div_by<1>(a, b, r2);
      n++;
      if (__builtin_bit_cast(uint32_t, want2) != __builtin_bit_cast(uint32_t, q2)) { bad1++; c->ex_a = __builtin_bit_cast(uint32_t, a); c->ex_b = __builtin_bit_cast(uint32_t, b); }
      continue;
    }
    // rsq-seeded reciprocal (sqrt_rcp_exact): b has to be the root of an x; take x = RN(b*b) and its root as b
    float r3; { float b3; tfl::sqrt_rcp_exact(b * b, b3, r3); if (mode == 0) b = b3; else if (b3 != b) r3 = tfl::rcp_refined(b); }
    if (mag > b) { mag = b; a = (h3 & 1) ? -mag : mag; }
    const float want = a / b;
    const float q3 = tfl::div_by<1>(a, b, r3);
    const float r = tfl::rcp_refined(b);
    const float q1 = tfl::div_by<1>(a, b, r), q2 = tfl::div_by<2>(a, b, r);
    n++;
    // a = -0: the quotient is -0 and the refinement returns +0 ((+0) + (-0)); equal as numbers, counted apart
    if (want == 0.0f && q1 == 0.0f && q2 == 0.0f && q3 == 0.0f) { zs += __builtin_bit_cast(uint32_t, want) != __builtin_bit_cast(uint32_t, q1); continue; }
    if (__builtin_bit_cast(uint32_t, want) != __builtin_bit_cast(uint32_t, q1)) { bad1++; c->ex_a = __builtin_bit_cast(uint32_t, a); c->ex_b = __builtin_bit_cast(uint32_t, b); }
    if (__builtin_bit_cast(uint32_t, want) != __builtin_bit_cast(uint32_t, q2)) bad2++;
    if (__builtin_bit_cast(uint32_t, want) != __builtin_bit_cast(uint32_t, q3)) bad3++;
  }
  atomicAdd(&c->n, n); atomicAdd(&c->bad1, bad1); atomicAdd(&c->bad2, bad2); atomicAdd(&c->zsign, zs); atomicAdd(&c->bad3, bad3);
}

static uint32_t fbits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }

int main() {
  Counts* d; Counts h;
  if (hipMalloc(&d, sizeof(Counts)) != hipSuccess) return 1;
  auto reset = [&]() { memset(&h, 0, sizeof h); (void)hipMemcpy(d, &h, sizeof h, hipMemcpyHostToDevice); };
  auto fetch = [&]() { (void)hipDeviceSynchronize(); (void)hipMemcpy(&h, d, sizeof h, hipMemcpyDeviceToHost); };
  reset();
  k_sqrt<<<4096, 256>>>(fbits(ldexpf(1.0f, -40)), fbits(ldexpf(1.0f, 40)), d);
  fetch();
  printf("sqrt_exact vs sqrtf: %llu operands (every float in [2^-40, 2^40)), mismatches %llu; sqrt_rcp_exact (rsq-seeded) mismatches %llu\n",
         h.n, h.badsqrt, h.badsqrt2);
  if (h.badsqrt2) printf("  example x bits 0x%08llx (rsq-seeded)\n", h.ex_b);
  if (h.badsqrt) printf("  example x bits 0x%08llx\n", h.ex_a);
  int rc = h.badsqrt != 0 || h.badsqrt2 != 0;
  reset();
  k_div<<<8192, 256>>>(0, 1ull << 33, 0x1234u, d);
  fetch();
  printf("div_by vs '/': hashed pairs %llu, mismatches 1-step %llu, 2-step %llu (+ %llu zero quotients of sign +0 for -0); rsq-seeded reciprocal 1-step %llu\n", h.n, h.bad1, h.bad2, h.zsign, h.bad3);
  if (h.bad1) printf("  example a bits 0x%08llx b bits 0x%08llx\n", h.ex_a, h.ex_b);
  rc |= h.bad2 != 0 || h.bad3 != 0;
  const unsigned long long b1a = h.bad1;
  reset();
  k_div<<<8192, 256>>>(1, (1ull << 23) * 10 * 256, 0x777u, d);
  fetch();
  printf("div_by vs '/': all mantissas of b x 10 binades x 256 numerators %llu, mismatches 1-step %llu, 2-step %llu (+ %llu zero-sign); rsq-seeded (where b is a root) %llu\n", h.n,
         h.bad1, h.bad2, h.zsign, h.bad3);
  if (h.bad1) printf("  example a bits 0x%08llx b bits 0x%08llx\n", h.ex_a, h.ex_b);
  rc |= h.bad2 != 0;
  const unsigned long long b1b = h.bad1;
  reset();
  k_div<<<8192, 256>>>(2, 1ull << 33, 0x4321u, d);
  fetch();
  printf("div_by vs '/': free ratio (|a| in [2^-40, 2^40], b in [2^-12, 2^21]) %llu pairs, mismatches 1-step %llu\n", h.n, h.bad1);
  if (h.bad1) printf("  example a bits 0x%08llx b bits 0x%08llx\n", h.ex_a, h.ex_b);
  rc |= h.bad1 != 0;
  printf("TFL_DIV_STEPS needed: %d\n", (b1a || b1b) ? 2 : 1);
  return rc;
}
