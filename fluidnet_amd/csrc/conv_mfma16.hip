// conv_mfma16.hip -- the three 3x3x3 layers of the 3-D `default` projection net on the MATRIX cores (gfx950):
// split-operand fp16 implicit GEMM on v_mfma_f32_16x16x32_f16, fp32 accumulate.
//
// Replaces cudnn.VolumetricConvolution forward (torch/lib/model_utils.lua:104-116) for the layers 3->8, 8->8, 8->8 (k=3)
// of lib/model.lua:219-226 with the ReLU fused and, in the last of them, the two trailing 1x1x1 layers (8->8 + ReLU,
// 8->1). Same role as conv_valu.hip (Winograd on the vector ALUs) and conv_mfma.hip (fp32-operand MFMA): the fp32-input
// MFMA runs at the vector rate on gfx950, the 16-bit-input one at 16x that, so the fp32 operands are SPLIT:
//
//   a = a_h + 2^-11 a_l      a_h = fp16(a), a_l = fp16((a - a_h) 2^11)        (activations)
//   w = w_h + 2^-11 w_l      on w 2^e (e per layer: max |w| 2^e in [8, 16))     (weights, split once on the host)
//
// Each half carries 11 significant bits and round-to-nearest leaves |a - a_h| <= ulp/2, so the pair represents 23+ bits:
// the representation error is 0 for half of all fp32 values and <= 2^-24 |a| otherwise -- the size of ONE fp32 rounding,
// where a 216-term fmaf chain commits 216 of them. All four partial products are kept (w_l a_l rides along for free).
// Products of two halves are exact in fp32; the sums are the MFMA's fp32 accumulation.
//
// GEMM shape (one MFMA = D[16 x 16] += A[16 x 32] B[32 x 16]):
//   M = 16 rows      = 8 output channels x {w_h, w_l}: row 2 c_out + t
//   N = 16 columns   = 16 consecutive x-voxels of one output row
//   K = 32           = 4 groups of 8: group g < 3 = the x-tap dx = g of one (dz, dy), its 8 elements = the input channels
//                      (layer 1: {p_h, d_h, occ, p_l, d_l, 0, 0, 0}); group 3 is idle (zero weights)
// The activation terms go through the SAME accumulator with the weights of the a_h pass pre-multiplied by 2^11 (exact):
//   D = 2^11 ( a_h w + a_l 2^-11 w ) so that out = 2^-(11 + e) (D[2c] + 2^-11 D[2c + 1]) + bias.
// With the weights as the A operand a lane ends up with both halves of TWO output channels of ONE voxel in its four
// accumulator registers (row = 4 (lane >> 4) + reg): the recombination needs no cross-lane traffic.
//
// Activations between the layers live in HBM already split ("h2": per (b, z, y) two rows [x][8] of fp16, hi then lo:
// 32 B per voxel like eight fp32). The producing layer splits once per voxel in its epilogue; the consumer's staging is
// a plain 16-byte-per-lane copy of contiguous rows into LDS, and one ds_read_b128 per lane IS the B fragment of an
// (input row, term): lanes 0-15 / 16-31 / 32-47 read voxels x-1.. / x.. / x+1...
// Block = 4 waves, tile 32 x 4 x 4 voxels; a wave owns 16 x 4 x 2 (8 accumulator quads) and walks the 6 x 4 input rows of
// its halo: each fragment is read once and feeds up to 9 MFMAs (3 dy x 3 dz output rows): 6 LDS reads and 18 MFMAs per
// output row. The 18 weight fragments (9 (dz, dy) x 2 scalings) stay in 72 VGPRs.
//
// Range: fp16 ends at 65504. Activations above it are clamped and COUNTED (tfl_model_range_errors); the net's input is
// normalised by the velocity's standard deviation, so this is a blown-up simulation, not a working point.
#include "tfl_device.hpp"
#include "tfl_fastmath.hpp"
#include "tfl_host.hpp"

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <type_traits>
#include <utility>

namespace tfl {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef _Float16 h4v __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;
typedef const __attribute__((address_space(1))) void glb_void;

#ifdef TFL_EXPERIMENTS      // geometry of the tile kernel (conv_mfma16_exp.inc)
constexpr int kTX = 32, kTY = 4, kTZ = 4;                 // block tile (voxels)
constexpr int kNY = 4, kNZ = 2;                           // output rows of a wave: 16 x kNY x kNZ
constexpr int kHX = kTX + 2, kHY = kTY + 2, kHZ = kTZ + 2;
constexpr int kPlane2 = kHY * 2 * kHX;                    // 16-byte slots of one staged h2 plane (two row-terms per row)
constexpr int kDma = (kPlane2 + 63) / 64;                 // LDS-DMA instructions (64 slots each) per plane
#endif
constexpr float kHalfMax = 65504.0f;

enum { kModeIn = 0, kModeMid = 1, kModeTail = 2 };
// timing ablations (tools/ab_build.sh): 1 = no staging loads, 2 = no MFMAs, 4 = no stores, 8 = no LDS fragment reads,
// 32 = no epilogue arithmetic (z-marched kernels: zeros are stored), 64 = no per-plane barrier
#ifndef TFL_M16_LB
#define TFL_M16_LB 3
#endif
#ifndef TFL_M16_ABL
#define TFL_M16_ABL 0
#endif

// tail pack (tfl_model::tail_pack): {bias3[8], w4[8][8] (out, in), b4[8], w5[8], b5[1]}
[[maybe_unused]] constexpr int kTailW4 = 8, kTailB4 = 72, kTailW5 = 80, kTailB5 = 88, kTailPost4 = 89;   // + post4 (conv3_m16_pack_tail)

struct MIn {            // fused network input (first layer): {pDiv/scale, div/scale, occupancy(flags)}
  const float* pDiv;    // [B][1][Z][Y][X]
  const float* div;
  const float* flags;
  const double* stats;  // [B][2] = sum u, sum u^2 (model.hip)
  double count;
  // round 6 (k_conv3_m16p_in only): non-null = the per-block partial pairs of k_bcs_div_stats, `per_sample` of them per batch
  // item -- every block sums its item's pairs itself, in k_reduce_stats' order (block_sum_pairs: the same bits), and the launch
  // of k_reduce_stats between the two kernels is gone; block 0 of each item leaves the sums in stats_out for k_project
  const double* partials;
  long long per_sample;
  double* stats_out;
};

// a -> (fp16(a), fp16((a - fp16(a)) * 2^11)); |a| <= 65504
__device__ __forceinline__ void split_h(float a, _Float16& hi, _Float16& lo) {
  hi = (_Float16)a;
  lo = (_Float16)((a - (float)hi) * 2048.0f);
}

// (a, b) -> the packed hi halves and the packed lo halves of the split a = a_h + 2^-11 a_l (values in [0, 65504]). The lo half
// is fp16(a 2^11 - a_h 2^11): both products and the difference are exact, so this is split_h's result in four instructions
// per pair less (v_cvt_pk_f16_f32 + v_fma_mix instead of convert back / subtract / scale / convert)
__device__ __forceinline__ void split_pair(float a, float b, uint32_t& H, uint32_t& L) {
  const h2v ph = {(_Float16)a, (_Float16)b};
  const float la = __builtin_fmaf((float)ph[0], -2048.0f, a * 2048.0f);
  const float lb = __builtin_fmaf((float)ph[1], -2048.0f, b * 2048.0f);
  const h2v pl = {(_Float16)la, (_Float16)lb};
  H = __builtin_bit_cast(uint32_t, ph); L = __builtin_bit_cast(uint32_t, pl);
}

}  // namespace

// 4 x 4 transpose of dwords across the four 16-lane groups of a wave: in: R[r] of group g = T[r][g]; out: R[c] of group g =
// T[g][c]. v_permlane32_swap exchanges the upper half of its first operand with the lower half of the second,
// v_permlane16_swap the odd 16-lane rows of the first with the even rows of the second.
__device__ __forceinline__ void transpose4(uint32_t (&R)[4]) {
  auto s02 = __builtin_amdgcn_permlane32_swap(R[0], R[2], false, false);
  auto s13 = __builtin_amdgcn_permlane32_swap(R[1], R[3], false, false);
  auto t01 = __builtin_amdgcn_permlane16_swap(s02[0], s13[0], false, false);
  auto t23 = __builtin_amdgcn_permlane16_swap(s02[1], s13[1], false, false);
  R[0] = t01[0]; R[1] = t01[1]; R[2] = t23[0]; R[3] = t23[1];
}

// De-phase the blocks that share a CU (round 5). All blocks of a one-round launch start together and run the same sequence, so the
// waves that share a SIMD -- one of each resident block -- reach their MFMA bursts, their epilogues and their waits TOGETHER: the
// matrix pipe serialises the bursts and idles through everything else (SQ counters, profiles/r04_pmc_sq.txt: pipe busy 0.29 -
// 0.52, waves 0.34 parked + 0.37 issue-stalled). A block delays its start by (hardware wave slot of its wave 0) x units x 64 clocks.
__device__ __forceinline__ void stagger_start(int units, uint4* lds) {
  if (units <= 0) return;             // uniform
  volatile int* w = reinterpret_cast<volatile int*>(lds);
  if (threadIdx.x == 0) w[0] = (int)__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);   // HW_REG_HW_ID bits [3:0]: WAVE_ID
  __syncthreads();
  const int n = __builtin_amdgcn_readfirstlane(w[0]) * units;
  __syncthreads();
  for (int i = 0; i < n; i++) __builtin_amdgcn_s_sleep(1);
}
#ifdef TFL_EXPERIMENTS
static int stagger_units(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
#endif

// a + b after v_permlane32_swap / v_permlane16_swap of the pair: the sum over the two lane halves (32) or over the odd /
// even 16-lane row pairs (16) of a in the lanes that keep their a, of b in the others (the tail's reduce-scatter).
// (The results go through named scalars: clang 19 folds __builtin_bit_cast(float, r[1]) of the builtin's vector result
// to element 0.)
__device__ __forceinline__ float swap_sum32(float a, float b) {
  auto r = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b), false, false);
  const uint32_t u0 = r[0], u1 = r[1];
  return __builtin_bit_cast(float, u0) + __builtin_bit_cast(float, u1);
}
__device__ __forceinline__ float swap_sum16(float a, float b) {
  auto r = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(uint32_t, a), __builtin_bit_cast(uint32_t, b), false, false);
  const uint32_t u0 = r[0], u1 = r[1];
  return __builtin_bit_cast(float, u0) + __builtin_bit_cast(float, u1);
}


// =====================================================================================================================
// z-MARCHED form of the 8 -> 8 layers (h2 in): a block owns a 32 x 8 column of the plane and walks a chunk of z. Input
// planes stream through a ring of kMRing LDS slots by LDS-DMA issued kMAhead planes ahead (inline asm: hipcc drains
// vmcnt(0) in front of every ds_read that follows a DMA it knows about, which would serialise the pipeline), each plane's
// 12 fragments per wave are read ONCE and feed the three output planes in flight (dz = 0, 1, 2 -> accumulators A0, A1,
// A2, rotated by register moves every step): 12 LDS reads per 72 MFMAs, half the tile kernel's, a third less staging
// traffic (halo 1.33 x (cz + 2) / cz against 2.39), and no exposed staging latency.
constexpr int kMX = 32, kMY = 8;                          // block column (voxels); wave = (x half, y half): 16 x 4 rows
constexpr int kMHX = kMX + 2, kMHY = kMY + 2;
constexpr int kMPlane = kMHY * 2 * kMHX;                  // 680 slots of a staged plane: [row][term][x]
constexpr int kMDma = 3;                                  // DMA instructions per wave and plane: 4 x 3 x 64 = 768 slots
constexpr int kMPitch = 4 * kMDma * 64;                   // ring pitch (the 88 slots behind the plane take the idle lanes)
[[maybe_unused]] constexpr int kMRing = 4, kMAhead = 3;      // (k_conv3_m16z's ring, conv_mfma16_exp.inc)

// one LDS-DMA instruction: 64 lanes x 16 bytes from per-lane global addresses to LDS bytes [lds_byte, lds_byte + 1024)
__device__ __forceinline__ void dma16(const void* gsrc, uint32_t lds_byte) {
  uint32_t keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte) : "memory");
}



// =====================================================================================================================
// z-marched, K-PACKED form of the 8 -> 8 layers: the round-4 default. In k_conv3_m16z one MFMA carries the three x-taps of
// one (dz, dy) -- 24 of the 32 K elements -- and an output row costs 9 x 2 MFMAs. Here the 27 taps of an output voxel are
// packed into SEVEN groups of at most four, whatever plane they come from:
//   A(dy) = {(dz 0; dx 0, 1, 2), (dz 1; dx 0)}      B(dy) = {(dz 1; dx 1, 2), (dz 2; dx 0, 1)}      for dy = 0, 1, 2
//   C     = {(dz 2, dx 2) of dy = 0, 1, 2}, one K group idle
// i.e. 7 x 2 = 14 MFMAs per output row of 16 voxels instead of 18 (56 per wave and plane instead of 72). A fragment now
// mixes planes: lane group g of the wave reads its 16-byte slot from the plane / row / x-offset its K group stands for (the
// fragment of input row r still serves the three output rows r - dy). All terms of an output plane are issued in the step
// its last input plane arrives: ONE accumulator set instead of three in flight (16 VGPRs instead of 48), 14 weight
// fragments instead of 18, no partial-plane masks. Ring of four planes, the next one in flight during the step.
constexpr int kPRing = 4;
constexpr int kPFrags = 14;                               // weight fragments: A(dy) x 2 terms, B(dy) x 2 terms, C x 2 terms

#ifndef TFL_M16P_LB
#define TFL_M16P_LB 2
#endif


// =====================================================================================================================
// k_conv3_m16q (round 5): k_conv3_m16p SOFTWARE-PIPELINED over the planes, with the interleaving written out. A wave's own vector
// instructions slip into the gaps between its own MFMAs; another wave's do not (tools/ubench/mfma16_coissue.hip,
// profiles/r05_conv_experiments.txt 6) -- and in k_conv3_m16p a wave issues the 56 MFMAs of a plane in one burst and the
// epilogue that depends on them in another, so the matrix pipe idles through every epilogue (pipe busy 0.43-0.52). Here the
// epilogue of plane q - 1 (second accumulator set) is cut into 28 slots of 3-7 instructions, one behind every PAIR of plane q's
// MFMAs, the order pinned with sched_barrier fences (hipcc's own scheduler and the sched_group_barrier solver both left the
// epilogue in one block behind the MFMAs); a fragment is read from LDS two pairs before its first use. Same MFMAs in the same
// order per accumulator, same epilogue arithmetic: bit-identical to k_conv3_m16p (TAIL: to its matrix-core tail).
namespace q16 {
struct Mop { int frag, w, oy; };
struct Frag { int kind, off, first; };        // kind 0 / 1 / 2 = the A / B / C fragment pointer; off = slot offset inside the plane
struct Tables { Mop m[56]; Frag f[32]; };
constexpr Tables make_tables() {
  Tables t{};
  int im = 0, fr = 0;
  for (int kind = 0; kind < 2; kind++)
    for (int ry = 0; ry < 6; ry++)
      for (int tm = 0; tm < 2; tm++) {
        t.f[fr] = Frag{kind, (ry * 2 + tm) * kMHX, im};
        for (int dy = 0; dy < 3; dy++) {
          const int oy = ry - dy;
          if (oy < 0 || oy > 3) continue;
          t.m[im++] = Mop{fr, kind * 6 + dy * 2 + tm, oy};
        }
        fr++;
      }
  for (int ry = 0; ry < 4; ry++)
    for (int tm = 0; tm < 2; tm++) {
      t.f[fr] = Frag{2, (ry * 2 + tm) * kMHX, im};
      t.m[im++] = Mop{fr, 12 + tm, ry};
      fr++;
    }
  return t;
}
constexpr Tables kT = make_tables();
#ifndef TFL_M16Q_AHEAD
#define TFL_M16Q_AHEAD 4
#endif
constexpr int kAhead = TFL_M16Q_AHEAD;        // pairs of MFMAs between a fragment's LDS read and its first use
template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>()), ...); }
}  // namespace q16

#ifndef TFL_M16Q_LB
#define TFL_M16Q_LB 2
#endif
template <bool TAIL>
__global__ __launch_bounds__(256, TFL_M16Q_LB) void k_conv3_m16q(Dom d, int cols_x, int cols_y, int cz, int chunks_a, int chunks,
                                                                int n_blocks, const uint4* __restrict__ in,
                                                                const uint4* __restrict__ wfrag, const float* __restrict__ bias,
                                                                void* __restrict__ outv, float post,
                                                                unsigned long long* __restrict__ range_err) {
  using namespace q16;
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  const int per_xcd = (n_blocks + 7) / 8;
  const int blk = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
  if (blk >= n_blocks) return;
  int t = blk;
  const int cx = t % cols_x; t /= cols_x;
  const int cy = t % cols_y; t /= cols_y;
  const int ch = t % chunks;
  const int b = t / chunks;
  const int zc0 = ch < chunks_a ? d.w0 + ch * cz : d.w1 + (ch - chunks_a) * cz;
  const int z_end = ch < chunks_a ? d.w0 + d.n0 : d.w1 + (d.nw - d.n0);
  const int nz = min(cz, z_end - zc0);                  // output planes of this block
  const int nsteps = nz + 2;                            // input planes zc0 - 1 .. zc0 + nz
  const int x0 = cx * kMX, y0 = cy * kMY;
  const long long cells = d.sc;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  h8 W[kPFrags];
#pragma unroll
  for (int f = 0; f < kPFrags; f++) W[f] = __builtin_bit_cast(h8, wfrag[f * 64 + lane]);

  // staging (as k_conv3_m16p)
  const uint4* src = in + (long long)b * cells * 2;
  const uint4* zero = wfrag + kPFrags * 64;
  int st_off[kMDma];
  unsigned st_ok = 0;
#pragma unroll
  for (int j = 0; j < kMDma; j++) {
    const int item = (wave * kMDma + j) * 64 + lane;
    const int r = min(item, kMPlane - 1) / kMHX, hx = min(item, kMPlane - 1) - r * kMHX;
    const int hy = r >> 1, tm = r & 1;
    const int gx = x0 - 1 + hx, gy = y0 - 1 + hy;
    st_off[j] = (min(max(gy, 0), d.Y - 1) * 2 + tm) * d.X + min(max(gx, 0), d.X - 1);
    st_ok |= (item < kMPlane && gx >= 0 && gx < d.X && gy >= 0 && gy < d.Y) ? (1u << j) : 0u;
  }
  const uint32_t lds0 = (uint32_t)(uintptr_t)(lds_void*)lds;
  auto issue = [&](int q) {          // input plane q of the chunk (z = zc0 - 1 + q) -> ring slot q % kPRing
    const int gz = zc0 - 1 + q;
    const bool z_ok = q < nsteps && gz >= 0 && gz < d.Z;
    const uint4* psrc = src + (long long)min(max(gz, 0), d.Z - 1) * d.Y * 2 * d.X;
#pragma unroll
    for (int j = 0; j < kMDma; j++) {
      const uint4* gp = (z_ok && ((st_ok >> j) & 1)) ? psrc + st_off[j] : zero;
      dma16(gp, __builtin_amdgcn_readfirstlane(lds0 + (uint32_t)(((q % kPRing) * kMPitch + (wave * kMDma + j) * 64) * 16)));
    }
  };

  const int wx = wave & 1, wy = wave >> 1;
  const int nn = lane & 15, g = lane >> 4;
  const int row0 = (wy * 4 * 2) * kMHX + wx * 16 + nn;
  const int slotA = row0 + (g < 3 ? g : 0);
  const int slotB = row0 + (g == 0 ? 1 : (g == 1 ? 2 : (g == 2 ? 0 : 1)));
  const int slotC = row0 + 2 + (g < 3 ? g : 0) * 2 * kMHX;
  const int x = x0 + wx * 16 + nn;
  const int y = y0 + wy * 4 + g;                          // the row this lane group stores (after the transpose)
  const bool live = x < d.X && y < d.Y;
  const float bias0 = bias[2 * g], bias1 = bias[2 * g + 1];
  float b4a = 0.0f, b4b = 0.0f, w5a = 0.0f, w5b = 0.0f, b5 = 0.0f, post4 = 0.0f;
  h4v A4 = {0, 0, 0, 0};
  if (TAIL) {
    b4a = bias[kTailB4 + 2 * g]; b4b = bias[kTailB4 + 2 * g + 1]; w5a = bias[kTailW5 + 2 * g]; w5b = bias[kTailW5 + 2 * g + 1];
    b5 = bias[kTailB5]; post4 = bias[kTailPost4];
    A4 = __builtin_bit_cast(h4v, reinterpret_cast<const uint2*>(wfrag + (kPFrags * 64 + 1))[lane]);
  }
  float hmax = 0.0f;

  // ---- the epilogue of one plane as 28 slots (state in registers; slot S touches the accumulators `a` of the plane before) ----
  float h0[4], h1[4], pp[4];
  uint32_t H[4], L[4];
  f4 dq[4];
  float r02 = 0.0f, r13 = 0.0f;
  auto epi = [&](auto sc, const f4 (&a)[4], int z) {
    constexpr int S = decltype(sc)::value;
    if constexpr (S < 16) {
      constexpr int r = S / 4, ph = S % 4;
      if constexpr (ph == 0) h0[r] = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(a[r][1], 0x1p-11f, a[r][0]), post, bias0), 0.0f);
      if constexpr (ph == 1) h1[r] = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(a[r][3], 0x1p-11f, a[r][2]), post, bias1), 0.0f);
      if constexpr (ph == 2) {
        hmax = __builtin_fmaxf(__builtin_fmaxf(hmax, h0[r]), h1[r]);
        split_pair(__builtin_fminf(h0[r], kHalfMax), __builtin_fminf(h1[r], kHalfMax), H[r], L[r]);
      }
      if constexpr (ph == 3 && TAIL) {
        const uint2 bq = make_uint2(H[r], L[r]);
        dq[r] = __builtin_amdgcn_mfma_f32_16x16x16f16(A4, __builtin_bit_cast(h4v, bq), (f4){0.0f, 0.0f, 0.0f, 0.0f}, 0, 0, 0);
      }
    } else if constexpr (!TAIL) {
      if constexpr (S == 16) transpose4(H);
      if constexpr (S == 18) transpose4(L);
      if constexpr (S == 20) {
        if (live) {
          uint4* orow = reinterpret_cast<uint4*>(outv) + (((long long)b * d.Z + z) * d.Y + y) * 2 * d.X + x;
          orow[0] = make_uint4(H[0], H[1], H[2], H[3]);
          orow[d.X] = make_uint4(L[0], L[1], L[2], L[3]);
        }
      }
    } else {
      if constexpr (S >= 17 && S <= 20) {
        constexpr int r = S - 17;
        const float ha = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(dq[r][1], 0x1p-11f, dq[r][0]), post4, b4a), 0.0f);
        const float hb = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(dq[r][3], 0x1p-11f, dq[r][2]), post4, b4b), 0.0f);
        pp[r] = __builtin_fmaf(w5a, ha, w5b * hb);
      }
      if constexpr (S == 21) r02 = swap_sum32(pp[0], pp[2]);
      if constexpr (S == 22) r13 = swap_sum32(pp[1], pp[3]);
      if constexpr (S == 23) {
        const float psel = swap_sum16(r02, r13);
        if (live) reinterpret_cast<float*>(outv)[(long long)b * cells + TFL_AT(d, x, y, z)] = psel + b5;
      }
    }
  };

  // ---- one plane step: the 56 MFMAs of plane q - 2 in 28 pairs, the epilogue slots of the plane before between them ----------
  auto step = [&](int q, f4 (&acc)[4], const f4 (&prev)[4], auto with_epi) {
    constexpr bool EPI = decltype(with_epi)::value;
    const int s0 = ((q - 2) % kPRing) * kMPitch, s1 = ((q - 1) % kPRing) * kMPitch, s2 = (q % kPRing) * kMPitch;
    const uint4* fa = lds + ((g == 3 ? s1 : s0) + slotA);
    const uint4* fb = lds + ((g < 2 ? s1 : s2) + slotB);
    const uint4* fc = lds + (s2 + slotC);
    h8 F[32];
    auto rd = [&](auto fcst) {
      constexpr int f = decltype(fcst)::value;
      const uint4* base = kT.f[f].kind == 0 ? fa : (kT.f[f].kind == 1 ? fb : fc);
      F[f] = __builtin_bit_cast(h8, base[kT.f[f].off]);
    };
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
    // the fragments of the first kAhead pairs
    static_for([&](auto fcst) { constexpr int f = decltype(fcst)::value; if constexpr (kT.f[f].first / 2 < kAhead) rd(fcst); },
               std::make_integer_sequence<int, 32>());
    static_for([&](auto sc) {
      constexpr int S = decltype(sc)::value;
      static_for([&](auto fcst) { constexpr int f = decltype(fcst)::value; if constexpr (kT.f[f].first / 2 == S + kAhead) rd(fcst); },
                 std::make_integer_sequence<int, 32>());
      constexpr Mop m0 = kT.m[2 * S], m1 = kT.m[2 * S + 1];
      acc[m0.oy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[m0.w], F[m0.frag], acc[m0.oy], 0, 0, 0);
      acc[m1.oy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[m1.w], F[m1.frag], acc[m1.oy], 0, 0, 0);
      if constexpr (EPI) {
        __builtin_amdgcn_sched_barrier(0);
        epi(sc, prev, zc0 + q - 3);
        __builtin_amdgcn_sched_barrier(0);
      }
    }, std::make_integer_sequence<int, 28>());
  };
  auto head = [&](int q) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // planes q - 2, q - 1, q have landed (plane q went out a whole step ago)
    __syncthreads();                                    // ... for every wave; every wave is done reading plane q - 3
    issue(q + 1);                                       // into the slot of plane q - 3 (past the chunk: the zero page)
  };

#pragma unroll
  for (int q = 0; q < 3; q++) issue(q);
  f4 accA[4], accB[4];
  head(2);
  step(2, accA, accB, std::false_type());
  // two steps per trip so that the accumulator sets swap roles without register moves
  int q = 3;
#pragma unroll 1
  for (; q + 1 < nsteps; q += 2) {
    head(q);
    step(q, accB, accA, std::true_type());
    head(q + 1);
    step(q + 1, accA, accB, std::true_type());
  }
  if (q < nsteps) {
    head(q);
    step(q, accB, accA, std::true_type());
    static_for([&](auto sc) { epi(sc, accB, zc0 + nsteps - 3); }, std::make_integer_sequence<int, 28>());
  } else {
    static_for([&](auto sc) { epi(sc, accA, zc0 + nsteps - 3); }, std::make_integer_sequence<int, 28>());
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the trailing zero-page DMA must not outlive the block's LDS
  if (!(hmax <= kHalfMax)) atomicAdd(range_err, 1ull);
}


// The FIRST layer (3 -> 8) in the same z-marched, K-packed form: the net input {pDiv / scale, div / scale, occupancy} is
// built plane by plane while the planes before it are multiplied -- each thread loads the raw words of its one or two
// slots of plane q + 1 at the top of step q and converts / splits / writes them to the LDS ring behind the step's MFMAs.
// One row-term per input row ({p_h, d_h, occ, p_l, d_l, 0, 0, 0}: 16 bytes per voxel), 7 fragments, 28 MFMAs per wave
// and plane; against the tile kernel k_conv3_m16<kModeIn> 1.33 instead of 2.4 staged voxels per output voxel and 7
// instead of 9 MFMAs per output row.
constexpr int kIPlane = kMHY * kMHX;                     // 340 16-byte slots of a staged plane: [row][x]
constexpr int kIRing = 4;
constexpr int kIFrags = 7;                               // A(dy), B(dy), C

#ifndef TFL_M16PI_LB
#define TFL_M16PI_LB 4
#endif
__global__ __launch_bounds__(256, TFL_M16PI_LB) void k_conv3_m16p_in(Dom d, int cols_x, int cols_y, int cz, int chunks_a, int chunks,
                                                                    int n_blocks, MIn cin, const uint4* __restrict__ wfrag,
                                                                    const float* __restrict__ bias, void* __restrict__ outv,
                                                                    float post, unsigned long long* __restrict__ range_err, int stag) {
  __shared__ __attribute__((aligned(16))) uint4 lds[kIRing * kIPlane];
  const int per_xcd = (n_blocks + 7) / 8;
  const int blk = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
  if (blk >= n_blocks) return;
  stagger_start(stag, lds);
  int t = blk;
  const int cx = t % cols_x; t /= cols_x;
  const int cy = t % cols_y; t /= cols_y;
  const int ch = t % chunks;
  const int b = t / chunks;
  const int zc0 = ch < chunks_a ? d.w0 + ch * cz : d.w1 + (ch - chunks_a) * cz;
  const int z_end = ch < chunks_a ? d.w0 + d.n0 : d.w1 + (d.nw - d.n0);
  const int nz = min(cz, z_end - zc0);
  const int nsteps = nz + 2;
  const int x0 = cx * kMX, y0 = cy * kMY;
  const long long cells = d.sc;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bool clipped = false;

  h8 W[kIFrags];
#pragma unroll
  for (int f = 0; f < kIFrags; f++) W[f] = __builtin_bit_cast(h8, wfrag[f * 64 + lane]);

  // lib/modules/variance.lua:44-76 (n-1) + Sqrt, as model.hip scale_from_stats
  double s1, s2;
  if (cin.partials) {     // (uniform) the block's own reduction of the partial sums; the ring is not in use yet
    block_sum_pairs(cin.partials + (long long)b * cin.per_sample * 2, cin.per_sample, reinterpret_cast<double*>(lds), tid, s1, s2);
    if (tid == 0 && cx == 0 && cy == 0 && ch == 0) { cin.stats_out[b * 2] = s1; cin.stats_out[b * 2 + 1] = s2; }
  } else {
    s1 = cin.stats[b * 2]; s2 = cin.stats[b * 2 + 1];
  }
  const double n = cin.count;
  const float in_scale = (float)sqrt(fmax(n * s2 - s1 * s1, 0.0) / (n * (n - 1.0)));
  const bool scale_in_range = in_scale >= 0x1p-12f && in_scale <= 0x1p21f;     // wave-uniform
  const float inv_scale = scale_in_range ? rcp_refined(in_scale) : 0.0f;

  // the thread's two slots of a plane (the second one only for tid < kIPlane - 256)
  int st_off[2], st_slot[2];
  bool st_in[2];
#pragma unroll
  for (int j = 0; j < 2; j++) {
    const int item = min(tid + 256 * j, kIPlane - 1);
    const int hy = item / kMHX, hx = item - hy * kMHX;
    const int gx = x0 - 1 + hx, gy = y0 - 1 + hy;
    st_off[j] = min(max(gy, 0), d.Y - 1) * d.sy + min(max(gx, 0), d.X - 1);
    st_in[j] = gx >= 0 && gx < d.X && gy >= 0 && gy < d.Y && tid + 256 * j < kIPlane;
    st_slot[j] = item;
  }
  const float* pP = cin.pDiv + (long long)b * cells;
  const float* pD = cin.div + (long long)b * cells;
  const float* pF = cin.flags + (long long)b * cells;
  float raw[2][3];
  auto load_plane = [&](int q) {      // raw words of input plane q of the chunk (z = zc0 - 1 + q)
    const int gz = min(max(zc0 - 1 + q, 0), d.Z - 1);
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (j == 1 && tid >= kIPlane - 256) continue;
      const int o = gz * d.sz + st_off[j];
      raw[j][0] = pP[o]; raw[j][1] = pD[o]; raw[j][2] = pF[o];
    }
  };
  auto write_plane = [&](int q) {     // -> ring slot q % kIRing; the net input is built here: ApplyScale(true) = CDivTable
    const int gz = zc0 - 1 + q;       // (apply_scale.lua:24-30), FlagsToOccupancy (generic/tfluids.cu:355-371)
    const bool z_ok = q < nsteps && gz >= 0 && gz < d.Z;
#pragma unroll
    for (int j = 0; j < 2; j++) {
      if (j == 1 && tid >= kIPlane - 256) continue;
      float v0, v1;
      if (scale_in_range) { v0 = div_by<1>(raw[j][0], in_scale, inv_scale); v1 = div_by<1>(raw[j][1], in_scale, inv_scale); }
      else { v0 = raw[j][0] / in_scale; v1 = raw[j][1] / in_scale; }
      const int f = (int)raw[j][2];
      const float occ = (f == kFluid) ? 0.0f : ((f == kObstacle) ? 1.0f : -1.0f);
      const float k0 = __builtin_fminf(__builtin_fmaxf(v0, -kHalfMax), kHalfMax), k1 = __builtin_fminf(__builtin_fmaxf(v1, -kHalfMax), kHalfMax);
      const bool ok = z_ok && st_in[j];
      clipped = clipped || (ok && (k0 != v0 || k1 != v1));
      h8 sv = {0, 0, 0, 0, 0, 0, 0, 0};
      if (ok) {
        _Float16 ph, pl, dh, dl;
        split_h(k0, ph, pl); split_h(k1, dh, dl);
        sv[0] = ph; sv[1] = dh; sv[2] = (_Float16)occ; sv[3] = pl; sv[4] = dl;
      }
      lds[(q % kIRing) * kIPlane + st_slot[j]] = __builtin_bit_cast(uint4, sv);
    }
  };

  const int wx = wave & 1, wy = wave >> 1;
  const int nn = lane & 15, g = lane >> 4;
  const int row0 = (wy * 4) * kMHX + wx * 16 + nn;
  const int slotA = row0 + (g < 3 ? g : 0);
  const int slotB = row0 + (g == 0 ? 1 : (g == 1 ? 2 : (g == 2 ? 0 : 1)));
  const int slotC = row0 + 2 + (g < 3 ? g : 0) * kMHX;
  const int x = x0 + wx * 16 + nn;
  const int y = y0 + wy * 4 + g;                          // the row this lane group stores (after the transpose)
  const float bias0 = bias[2 * g], bias1 = bias[2 * g + 1];

#pragma unroll 1
  for (int q = 0; q < 3; q++) { load_plane(q); write_plane(q); }
#pragma unroll 1
  for (int q = 2; q < nsteps; q++) {
    __syncthreads();                  // plane q is in LDS; every wave is done reading plane q - 3
    load_plane(q + 1);                // in flight during the MFMAs
    const int s0 = ((q - 2) % kIRing) * kIPlane, s1 = ((q - 1) % kIRing) * kIPlane, s2 = (q % kIRing) * kIPlane;
    const uint4* fa = lds + ((g == 3 ? s1 : s0) + slotA);
    const uint4* fb = lds + ((g < 2 ? s1 : s2) + slotB);
    const uint4* fc = lds + (s2 + slotC);
    f4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; r++) acc[r] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int kind = 0; kind < 2; kind++)
#pragma unroll
      for (int ry = 0; ry < 6; ry++) {
        const h8 v = __builtin_bit_cast(h8, (kind == 0 ? fa : fb)[ry * kMHX]);
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
          const int oy = ry - dy;
          if (oy < 0 || oy > 3) continue;
          acc[oy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[kind * 3 + dy], v, acc[oy], 0, 0, 0);
        }
      }
#pragma unroll
    for (int ry = 0; ry < 4; ry++) {
      const h8 vc = __builtin_bit_cast(h8, fc[ry * kMHX]);
      acc[ry] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[6], vc, acc[ry], 0, 0, 0);
    }
    // epilogue: recombine, bias, ReLU, split, transposed 16-byte stores (as the 8 -> 8 layers)
    {
      const int z = zc0 + q - 2;
      uint32_t H[4], L[4];
      bool over = false;
#pragma unroll
      for (int oy = 0; oy < 4; oy++) {
        const float h0 = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(acc[oy][1], 0x1p-11f, acc[oy][0]), post, bias0), 0.0f);
        const float h1 = __builtin_fmaxf(__builtin_fmaf(__builtin_fmaf(acc[oy][3], 0x1p-11f, acc[oy][2]), post, bias1), 0.0f);
        const float k0 = __builtin_fminf(h0, kHalfMax), k1 = __builtin_fminf(h1, kHalfMax);
        over = over || ((k0 != h0 || k1 != h1) && x < d.X && y0 + wy * 4 + oy < d.Y);
        _Float16 hh0, hl0, hh1, hl1;
        split_h(k0, hh0, hl0); split_h(k1, hh1, hl1);
        const h2v ph = {hh0, hh1}, pl = {hl0, hl1};
        H[oy] = __builtin_bit_cast(uint32_t, ph); L[oy] = __builtin_bit_cast(uint32_t, pl);
      }
      clipped = clipped || over;      // (one running maximum instead costs this kernel 13 VGPRs and its fifth wave per SIMD)
      transpose4(H); transpose4(L);
      if (x < d.X && y < d.Y) {
        uint4* orow = reinterpret_cast<uint4*>(outv) + (((long long)b * d.Z + z) * d.Y + y) * 2 * d.X + x;
        orow[0] = make_uint4(H[0], H[1], H[2], H[3]);
        orow[d.X] = make_uint4(L[0], L[1], L[2], L[3]);
      }
    }
    write_plane(q + 1);               // into the slot of plane q - 3 (every wave passed this step's barrier)
  }
  if (clipped) atomicAdd(range_err, 1ull);
}


#ifdef TFL_EXPERIMENTS
// the tile kernel, the un-packed and the un-pipelined z-marched kernels, layers 1 + 2 in one launch: conv_mfma16_exp.inc
#define TFL_M16_EXP_SECTION 1
#include "conv_mfma16_exp.inc"
#undef TFL_M16_EXP_SECTION
#endif
// compute units of the current device (cached per device)
static int device_cus() {
  static std::atomic<int> cus[64];
  int dev = 0; (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  int n = cus[dev].load();
  if (!n) {
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cus[dev].store(n);
  }
  return n;
}

// resident blocks per CU of a kernel with `lds` bytes of dynamic LDS (asked once per kernel; `fallback` if the runtime cannot say)
static int blocks_per_cu(const void* fn, size_t lds, int fallback) {
  struct Rec { const void* fn; int dev, n; };  // per kernel AND device: the dynamic-LDS attribute is a per-device setting
  static Rec recs[64];
  static int nrec = 0;
  static std::mutex mu;                       // virtual ranks launch from several host threads
  int dev = 0; (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lock(mu);
  for (int i = 0; i < nrec; i++) if (recs[i].fn == fn && recs[i].dev == dev) return recs[i].n;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  int n = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, 256, lds) != hipSuccess || n <= 0) n = fallback;
  if (nrec < 64) { recs[nrec].fn = fn; recs[nrec].dev = dev; recs[nrec].n = n; nrec++; }
  return n;
}

// Chunk length of the K-packed z-marched kernels (planes a block walks). A block of c output planes takes c + 4 plane steps
// (two halo planes, ~two of weight fetch and pipeline fill); the blocks of a launch occupy the chip in rounds of `slots`
// (= CUs x resident blocks per CU), and a plane step of a CU that holds n blocks costs L + W n: the n blocks hide each
// other's latencies only in part. Fitted on tools/conv_cz_sweep.py (profiles/r04_conv_cz_sweep.txt; 128^2 and 256^2
// columns, 24 .. 136 planes, chunks of 2 .. 64): 8 -> 8 layers L = 0.70, W = 0.35 us, first layer 0.95 / 0.24. On a full
// 128^3 / 256^3 grid the choice is the one the round counting of launch_m16z makes (11 / 8 planes at 128^3); on the thin
// windows of a z-slab rank it picks short chunks (24 planes of 128^2: 3 instead of 8, -4 us per layer).
static int pick_chunk(long long cols, int na, int nb, int slots, int cus, float L, float W, int fill = 4, int cmax = 32) {
  int cz = 8;
  float best = -1.0f;
  for (int c = 2; c <= cmax; c++) {
    long long blocks = cols * ((na + c - 1) / c + (nb + c - 1) / c);
    float per_step = 0.0f;
    while (blocks > 0) {
      const long long r = blocks < slots ? blocks : slots;
      per_step += L + W * (float)((r + cus - 1) / cus);
      blocks -= r;
    }
    const float cost = (float)(c + fill) * per_step;
    if (best < 0.0f || cost < best) { best = cost; cz = c; }
  }
  return cz;
}



// A development switch of the EXPERIMENTS flavour (chunk lengths, the de-phased block starts): the default library reads none
static int exp_int(const char* name, int dflt) {
#ifdef TFL_EXPERIMENTS
  if (const char* e = getenv(name)) return atoi(e);
#else
  (void)name;
#endif
  return dflt;
}

template <bool TAIL>
static void launch_m16p(hipStream_t st, const Dom& d, int B, const void* in, const void* wfrag, const float* bias, void* out,
                        float post, unsigned long long* range_err) {
  const int cxn = (d.X + kMX - 1) / kMX, cyn = (d.Y + kMY - 1) / kMY;
  const int na = d.n0, nb = d.nw - d.n0;
  if (cxn * cyn * (na + nb) * B <= 0) return;
  const size_t lds_bytes = (size_t)16 * kPRing * kMPitch;
  static bool attr_done[2] = {false, false};
  if (!attr_done[TAIL]) { (void)hipFuncSetAttribute((const void*)k_conv3_m16q<TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); attr_done[TAIL] = true; }
  const int slots = device_cus() * blocks_per_cu((const void*)k_conv3_m16q<TAIL>, lds_bytes, TFL_M16Q_LB);
  int cz = pick_chunk((long long)cxn * cyn * B, na, nb, slots, device_cus(), 0.70f, 0.35f);
  if (const int e = exp_int("TFL_M16_CZ", 0); e > 0) cz = e;
  const int chunks_a = (na + cz - 1) / cz, chunks = chunks_a + (nb + cz - 1) / cz;
  const int n_blocks = cxn * cyn * chunks * B;
  const int grid = ((n_blocks + 7) / 8) * 8;
  if (getenv("TFL_DEBUG")) {
    static bool said[2] = {false, false};
    if (!said[TAIL]) {
      said[TAIL] = true;
      fprintf(stderr, "[tfl] k_conv3_m16q<%d>: dynamic LDS %zu B, %d block slots, grid %d, chunks of %d planes\n", (int)TAIL, lds_bytes, slots, grid, cz);
    }
  }
  // the fragments of this kernel lie behind those of k_conv3_m16z in the layer's buffer (conv3_m16_pack_weights)
  const uint4* wp = (const uint4*)wfrag + (9 * 2 * 64 + 1);
  TFL_TIMED_EXT(TAIL ? "k_conv3_tail" : "k_conv3_mid", st);
#ifdef TFL_EXPERIMENTS
  // the tail's 1 x 1 x 1 layers on the vector ALUs (TFL_M16_TAIL_MFMA=0) / the un-pipelined kernel (TFL_M16_PIPE=0): k_conv3_m16p
  const char* etm = getenv("TFL_M16_TAIL_MFMA");       // (read per call: the tests switch it inside one process)
  const bool tmf = !(etm && atoi(etm) == 0);
  const char* epp = getenv("TFL_M16_PIPE");
  if ((epp && atoi(epp) == 0) || (TAIL && !tmf)) {
    if (TAIL && tmf)
      TFL_LAUNCH_EXT((k_conv3_m16p<TAIL, TAIL>), grid, 256, lds_bytes, st, d, cxn, cyn, cz, chunks_a, chunks, n_blocks, (const uint4*)in,
                     wp, bias, out, post, range_err, stagger_units("TFL_M16_STAGGER", 0));
    else
      TFL_LAUNCH_EXT((k_conv3_m16p<TAIL, false>), grid, 256, lds_bytes, st, d, cxn, cyn, cz, chunks_a, chunks, n_blocks, (const uint4*)in,
                     wp, bias, out, post, range_err, stagger_units("TFL_M16_STAGGER", 0));
    return;
  }
#endif
  // the 8 -> 8 layers software-pipelined over the planes, the tail's 1 x 1 x 1 layers on the matrix cores (round 5)
  TFL_LAUNCH_EXT((k_conv3_m16q<TAIL>), grid, 256, lds_bytes, st, d, cxn, cyn, cz, chunks_a, chunks, n_blocks, (const uint4*)in, wp, bias, out, post, range_err);
}

static void launch_m16p_in(hipStream_t st, const Dom& d, int B, MIn cin, const void* wfrag, const float* bias, void* out, float post,
                           unsigned long long* range_err) {
  const int cxn = (d.X + kMX - 1) / kMX, cyn = (d.Y + kMY - 1) / kMY;
  const int na = d.n0, nb = d.nw - d.n0;
  if (cxn * cyn * (na + nb) * B <= 0) return;
  const int slots = device_cus() * blocks_per_cu((const void*)k_conv3_m16p_in, 0, TFL_M16PI_LB);
  int cz = pick_chunk((long long)cxn * cyn * B, na, nb, slots, device_cus(), 0.95f, 0.24f);
  if (const int e = exp_int("TFL_M16_CZ_IN", 0); e > 0) cz = e;
  const int chunks_a = (na + cz - 1) / cz, chunks = chunks_a + (nb + cz - 1) / cz;
  const int n_blocks = cxn * cyn * chunks * B;
  const int grid = ((n_blocks + 7) / 8) * 8;
  if (getenv("TFL_DEBUG")) {
    static bool said = false;
    if (!said) { said = true; fprintf(stderr, "[tfl] k_conv3_m16p_in: %d block slots, grid %d, chunks of %d planes\n", slots, grid, cz); }
  }
  const uint4* wp = (const uint4*)wfrag + (9 * 64 + 1);     // behind the tile kernel's fragments (conv3_m16_pack_weights)
  TFL_TIMED_EXT("k_conv3_in", st);
  TFL_LAUNCH_EXT(k_conv3_m16p_in, grid, 256, 0, st, d, cxn, cyn, cz, chunks_a, chunks, n_blocks, cin, wp, bias, out, post, range_err,
                 exp_int("TFL_M16_STAGGER_IN", 0));
}

#ifdef TFL_EXPERIMENTS
#define TFL_M16_EXP_SECTION 2
#include "conv_mfma16_exp.inc"
#undef TFL_M16_EXP_SECTION
// which of the earlier forms the switches of this flavour ask for: TFL_M16_KPACK = 0 (k_conv3_m16z; first layer: the tile kernel) or
// 2 (first layer only), TFL_M16_TILED bit 0 / 1 = the tile kernel for the mid / tail layer
static int exp_kpack() { static const int v = getenv("TFL_M16_KPACK") ? atoi(getenv("TFL_M16_KPACK")) : 1; return v; }
static int exp_tiled() { const char* e = getenv("TFL_M16_TILED"); return e ? atoi(e) : 0; }
#endif

bool conv3_m16_first_sums_partials() {
#ifdef TFL_EXPERIMENTS
  if (exp_kpack() == 0 || exp_kpack() == 2) return false;      // the tile kernel reads the reduced sums
#endif
  const char* e = getenv("TFL_STATS_CONSUMER");     // A/B switch (read per call: the parity test flips it inside one process): 0 = k_reduce_stats as its own launch
  return !(e && atoi(e) == 0);
}
void conv3_m16_first_fused(hipStream_t st, int B, int Z, int Y, int X, const float* pDiv, const float* div, const float* flags,
                           const double* stats, double count, const void* wfrag, const float* bias, float post, void* out_h2,
                           unsigned long long* range_err, const double* partials, long long per_sample, double* stats_out) {
  MIn ci = {pDiv, div, flags, stats, count, partials, per_sample, stats_out};
#ifdef TFL_EXPERIMENTS
  if (exp_kpack() == 0 || exp_kpack() == 2) { launch_m16<kModeIn>(st, make_dom(Z, Y, X), B, nullptr, wfrag, bias, out_h2, post, ci, range_err); return; }
#endif
  launch_m16p_in(st, make_dom(Z, Y, X), B, ci, wfrag, bias, out_h2, post, range_err);
}
// layers 1 + 2 in ONE launch (k_conv3_m16p_f2, conv_mfma16_exp.inc): measured slower than the two launches (round 5), so only
// the EXPERIMENTS flavour carries it (TFL_M16_FUSE12=1); false = not taken, the caller runs the two layers
bool conv3_m16_first2_fused(hipStream_t st, int B, int Z, int Y, int X, const float* pDiv, const float* div, const float* flags,
                            const double* stats, double count, const void* wfrag1, const float* bias1, float post1,
                            const void* wfrag2, const float* bias2, float post2, void* out_h2, unsigned long long* range_err) {
#ifdef TFL_EXPERIMENTS
  if (exp_kpack() != 1 || exp_tiled() != 0) return false;
  MIn ci = {pDiv, div, flags, stats, count, nullptr, 0, nullptr};
  return launch_m16p_f2(st, make_dom(Z, Y, X), B, ci, wfrag1, bias1, post1, wfrag2, bias2, post2, out_h2, range_err);
#else
  (void)st; (void)B; (void)Z; (void)Y; (void)X; (void)pDiv; (void)div; (void)flags; (void)stats; (void)count; (void)wfrag1; (void)bias1; (void)post1;
  (void)wfrag2; (void)bias2; (void)post2; (void)out_h2; (void)range_err;
  return false;
#endif
}
bool conv3_m16_fuse12_requested() {
#ifdef TFL_EXPERIMENTS
  const char* ef = getenv("TFL_M16_FUSE12");
  return ef && atoi(ef) == 1;
#else
  return false;
#endif
}
void conv3_m16_mid(hipStream_t st, int B, int Z, int Y, int X, const void* in_h2, const void* wfrag, const float* bias, float post,
                   void* out_h2, unsigned long long* range_err) {
#ifdef TFL_EXPERIMENTS
  if (exp_tiled() & 1) { MIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0, nullptr, 0, nullptr}; launch_m16<kModeMid>(st, make_dom(Z, Y, X), B, in_h2, wfrag, bias, out_h2, post, noin, range_err); return; }
  if (exp_kpack() == 0) { launch_m16z<false>(st, make_dom(Z, Y, X), B, in_h2, wfrag, bias, out_h2, post, range_err); return; }
#endif
  launch_m16p<false>(st, make_dom(Z, Y, X), B, in_h2, wfrag, bias, out_h2, post, range_err);
}
void conv3_m16_tail(hipStream_t st, int B, int Z, int Y, int X, const void* in_h2, const void* wfrag, const float* tail_pack,
                    float post, float* p_out, unsigned long long* range_err) {
#ifdef TFL_EXPERIMENTS
  if (exp_tiled() & 2) { MIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0, nullptr, 0, nullptr}; launch_m16<kModeTail>(st, make_dom(Z, Y, X), B, in_h2, wfrag, tail_pack, p_out, post, noin, range_err); return; }
  if (exp_kpack() == 0) { launch_m16z<true>(st, make_dom(Z, Y, X), B, in_h2, wfrag, tail_pack, p_out, post, range_err); return; }
#endif
  launch_m16p<true>(st, make_dom(Z, Y, X), B, in_h2, wfrag, tail_pack, p_out, post, range_err);
}

// ---- host: weights [8][cin][3][3][3] (cudnn order) -> A fragments ----------------------------------------------------
namespace {
// float -> IEEE binary16 bits, round to nearest even (host side; finite inputs)
uint16_t f2h(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  const uint32_t sign = (u >> 16) & 0x8000u;
  u &= 0x7fffffffu;
  if (u >= 0x47800000u) return (uint16_t)(sign | 0x7c00u);            // >= 65536 (callers stay below 65520)
  if (u < 0x38800000u) {                                              // below 2^-14: subnormal half
    if (u < 0x33000000u) return (uint16_t)sign;                       // below 2^-25
    const int e = (int)(u >> 23);                                     // biased exponent, 102..112
    const uint32_t m = (u & 0x7fffffu) | 0x800000u;
    const int shift = 126 - e;                                        // 14..24
    uint32_t r = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1))) r++;
    return (uint16_t)(sign | r);
  }
  uint32_t r = ((u - 0x38000000u) >> 13);
  const uint32_t rem = u & 0x1fffu;
  if (rem > 0x1000u || (rem == 0x1000u && (r & 1))) r++;
  return (uint16_t)(sign | r);
}
float h2f(uint16_t h) {
  const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
  uint32_t e = (h >> 10) & 31, m = h & 0x3ffu;
  float v;
  if (e == 0) v = ldexpf((float)m, -24);
  else if (e == 31) v = INFINITY;
  else v = ldexpf((float)(m | 0x400u), (int)e - 25);
  uint32_t u; memcpy(&u, &v, 4); u |= sign; memcpy(&v, &u, 4);
  return v;
}
}  // namespace

// out: conv3_m16_frag_halves(cin) halves: (9 * RT * 64 + 1) * 8 (RT = 1 for cin == 3, 2 for cin == 8; the last 16 bytes
// are zero: the source of the kernels' out-of-grid staging slots) and, for cin == 8, the K-packed fragments of
// k_conv3_m16p behind them ((14 * 64 + 1) * 8); returns the post-scale 2^-(11 + e)
size_t conv3_m16_frag_halves(int cin) {
  return cin == 3 ? ((size_t)9 * 64 + 1) * 8 + ((size_t)7 * 64 + 1) * 8
                  : ((size_t)9 * 2 * 64 + 1) * 8 + ((size_t)14 * 64 + 1) * 8 + (size_t)64 * 4;   // + the tail's 1 x 1 x 1 A fragment
}
// The A fragment of the tail's 8 -> 8 (k = 1) layer for v_mfma_f32_16x16x16_f16 (k_conv3_m16p<true, true>), written behind the
// K-packed fragments of a cin == 8 layer buffer: lane (m = lane & 15, g = lane >> 4) holds A[m][4g .. 4g + 3]; row m = 2 j + t
// = output channel j as {w_h, w_l} of w4[j][c] 2^e4; column 4g + i = input channel c = 2g + (i & 1), activation term i >> 1
// (0: the high halves, their weights pre-multiplied by 2^11). Returns the post-scale 2^-(11 + e4).
float conv3_m16_pack_tail(const float* w4, uint16_t* frag_buf) {
  float mx = 0.0f;
  for (int i = 0; i < 64; i++) mx = fmaxf(mx, fabsf(w4[i]));
  int e = 0;
  if (mx > 0.0f && std::isfinite(mx)) { int ex; (void)frexpf(mx, &ex); e = 4 - ex; }       // mx 2^e in [8, 16)
  uint16_t* o = frag_buf + ((size_t)9 * 2 * 64 + 1) * 8 + ((size_t)14 * 64 + 1) * 8;
  for (int lane = 0; lane < 64; lane++)
    for (int i = 0; i < 4; i++) {
      const int m = lane & 15, g = lane >> 4, j = m >> 1, wt = m & 1, c = 2 * g + (i & 1);
      const float ws = ldexpf(w4[j * 8 + c], e);
      const float wh = h2f(f2h(ws));
      const float base = wt ? h2f(f2h((ws - wh) * 2048.0f)) : wh;
      o[lane * 4 + i] = f2h((i >> 1) == 0 ? base * 2048.0f : base);
    }
  return ldexpf(1.0f, -(11 + e));
}
float conv3_m16_pack_weights(const float* w, int cin, uint16_t* out) {
  const int RT = cin == 3 ? 1 : 2;
  float mx = 0.0f;
  for (int i = 0; i < 8 * cin * 27; i++) mx = fmaxf(mx, fabsf(w[i]));
  int e = 0;
  if (mx > 0.0f && std::isfinite(mx)) { int ex; (void)frexpf(mx, &ex); e = 4 - ex; }       // mx 2^e in [8, 16)
  for (int p = 0; p < 9; p++)
    for (int r = 0; r < RT; r++)
      for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 8; j++) {
          const int m = lane & 15, g = lane >> 4, co = m >> 1, wt = m & 1;
          int ci = -1; bool act_hi = true;
          if (g < 3) {
            if (cin == 8) { ci = j; act_hi = r == 0; }
            else if (j < 3) { ci = j; act_hi = true; }
            else if (j < 5) { ci = j - 3; act_hi = false; }
          }
          float v = 0.0f;
          if (ci >= 0) {
            const float ws = ldexpf(w[((size_t)(co * cin + ci) * 9 + p) * 3 + g], e);
            const float wh = h2f(f2h(ws));
            const float base = wt ? h2f(f2h((ws - wh) * 2048.0f)) : wh;
            v = act_hi ? base * 2048.0f : base;
          }
          out[(((size_t)p * RT + r) * 64 + lane) * 8 + j] = f2h(v);
        }
  for (int j = 0; j < 8; j++) out[(size_t)9 * RT * 64 * 8 + j] = 0;
  if (cin == 3) {
    // the K-packed fragments of k_conv3_m16p_in behind them: A(dy), B(dy), C (f = 0..6), + 16 zero bytes; K group g of a
    // fragment = one tap as below, its 8 elements = {p_h, d_h, occ, p_l, d_l, 0, 0, 0}
    uint16_t* o2 = out + ((size_t)9 * RT * 64 + 1) * 8;
    for (int f = 0; f < 7; f++)
      for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 8; j++) {
          const int m = lane & 15, g = lane >> 4, co = m >> 1, wt = m & 1;
          const int kind = f < 3 ? 0 : (f < 6 ? 1 : 2), dy = kind == 2 ? 0 : f % 3;
          int kz = -1, ky = 0, kx = 0;
          if (kind == 0) { kz = g == 3 ? 1 : 0; ky = dy; kx = g == 3 ? 0 : g; }
          else if (kind == 1) { kz = g < 2 ? 1 : 2; ky = dy; kx = g == 0 ? 1 : (g == 1 ? 2 : (g == 2 ? 0 : 1)); }
          else if (g < 3) { kz = 2; ky = g; kx = 2; }
          const int ci = j < 3 ? j : (j < 5 ? j - 3 : -1);
          const bool act_hi = j < 3;
          float v = 0.0f;
          if (kz >= 0 && ci >= 0) {
            const float ws = ldexpf(w[(size_t)(co * cin + ci) * 27 + kz * 9 + ky * 3 + kx], e);
            const float wh = h2f(f2h(ws));
            const float base = wt ? h2f(f2h((ws - wh) * 2048.0f)) : wh;
            v = act_hi ? base * 2048.0f : base;
          }
          o2[((size_t)f * 64 + lane) * 8 + j] = f2h(v);
        }
    for (int j = 0; j < 8; j++) o2[(size_t)7 * 64 * 8 + j] = 0;
  }
  if (cin == 8) {
    // the K-packed fragments of k_conv3_m16p behind them: A(dy) x term, B(dy) x term, C x term (f = 0..13), + 16 zero bytes.
    // K group g of a fragment = one tap (kz, ky, kx) or none:
    //   A(dy): (0, dy, 0) (0, dy, 1) (0, dy, 2) (1, dy, 0)    B(dy): (1, dy, 1) (1, dy, 2) (2, dy, 0) (2, dy, 1)
    //   C:     (2, 0, 2) (2, 1, 2) (2, 2, 2) -
    uint16_t* o2 = out + ((size_t)9 * RT * 64 + 1) * 8;
    for (int f = 0; f < 14; f++)
      for (int lane = 0; lane < 64; lane++)
        for (int j = 0; j < 8; j++) {
          const int m = lane & 15, g = lane >> 4, co = m >> 1, wt = m & 1;
          const int kind = f < 6 ? 0 : (f < 12 ? 1 : 2), dy = kind == 2 ? 0 : ((f % 6) >> 1), tm = f & 1;
          int kz = -1, ky = 0, kx = 0;
          if (kind == 0) { kz = g == 3 ? 1 : 0; ky = dy; kx = g == 3 ? 0 : g; }
          else if (kind == 1) { kz = g < 2 ? 1 : 2; ky = dy; kx = g == 0 ? 1 : (g == 1 ? 2 : (g == 2 ? 0 : 1)); }
          else if (g < 3) { kz = 2; ky = g; kx = 2; }
          float v = 0.0f;
          if (kz >= 0) {
            const float ws = ldexpf(w[(size_t)(co * cin + j) * 27 + kz * 9 + ky * 3 + kx], e);
            const float wh = h2f(f2h(ws));
            const float base = wt ? h2f(f2h((ws - wh) * 2048.0f)) : wh;
            v = tm == 0 ? base * 2048.0f : base;       // term 0 multiplies the activations' high halves
          }
          o2[((size_t)f * 64 + lane) * 8 + j] = f2h(v);
        }
    for (int j = 0; j < 8; j++) o2[(size_t)14 * 64 * 8 + j] = 0;
  }
  return ldexpf(1.0f, -(11 + e));
}

}  // namespace tfl
