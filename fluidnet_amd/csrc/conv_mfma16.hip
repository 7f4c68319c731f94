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

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace tfl {

namespace {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int kTX = 32, kTY = 4, kTZ = 4;                 // block tile (voxels)
constexpr int kNY = 4, kNZ = 2;                           // output rows of a wave: 16 x kNY x kNZ
constexpr int kHX = kTX + 2, kHY = kTY + 2, kHZ = kTZ + 2;
constexpr float kHalfMax = 65504.0f;

enum { kModeIn = 0, kModeMid = 1, kModeTail = 2 };

// tail pack (tfl_model::tail_pack): {bias3[8], w4[8][8] (out, in), b4[8], w5[8], b5[1]}
constexpr int kTailW4 = 8, kTailB4 = 72, kTailW5 = 80, kTailB5 = 88;

struct MIn {            // fused network input (first layer): {pDiv/scale, div/scale, occupancy(flags)}
  const float* pDiv;    // [B][1][Z][Y][X]
  const float* div;
  const float* flags;
  const double* stats;  // [B][2] = sum u, sum u^2 (model.hip)
  double count;
};

// a -> (fp16(a), fp16((a - fp16(a)) * 2^11)); |a| <= 65504
__device__ __forceinline__ void split_h(float a, _Float16& hi, _Float16& lo) {
  hi = (_Float16)a;
  lo = (_Float16)((a - (float)hi) * 2048.0f);
}

}  // namespace

// MODE: kModeIn (inputs built from pDiv / div / flags, 1 row-term per input row), kModeMid (h2 in, h2 out),
// kModeTail (h2 in, + the two 1x1x1 layers, planar fp32 pressure out).
// wfrag: [9 (dz, dy)][RT][64 lanes] x 16 B -- the A fragments (weights), built by conv3_m16_pack_weights.
template <int MODE>
__global__ __launch_bounds__(256, 3) void k_conv3_m16(Dom d, int tiles_x, int tiles_y, int tiles_z, int n_tiles,
                                                      const uint4* __restrict__ in, const uint4* __restrict__ wfrag,
                                                      const float* __restrict__ bias, void* __restrict__ outv, float post,
                                                      MIn cin, unsigned long long* __restrict__ range_err) {
  extern __shared__ __attribute__((aligned(16))) uint4 lds[];
  constexpr bool FIRST = MODE == kModeIn, TAIL = MODE == kModeTail;
  constexpr int RT = FIRST ? 1 : 2;                       // row-terms per input row
  constexpr int kRows = kHZ * kHY * RT, kItems = kRows * kHX;
  // XCD-aware tile order: consecutive block ids go round-robin over the 8 XCDs; give each XCD a contiguous tile run
  const int per_xcd = (n_tiles + 7) / 8;
  const int tile = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
  if (tile >= n_tiles) return;
  int t = tile;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y; t /= tiles_y;
  const int tz = t % tiles_z;
  const int b = t / tiles_z;
  // z-window (tfl_device.hpp Dom): the z-tiles cover the plane run [w0, w0 + n0) and then [w1, w1 + nw - n0)
  const int tz_a = (d.n0 + kTZ - 1) / kTZ;
  const int z0 = tz < tz_a ? d.w0 + tz * kTZ : d.w1 + (tz - tz_a) * kTZ;
  const int z_end = tz < tz_a ? d.w0 + d.n0 : d.w1 + (d.nw - d.n0);
  const int x0 = tx * kTX, y0 = ty * kTY;
  const long long cells = d.sc;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  bool clipped = false;

  // ---- weights: 9 * RT A fragments per lane -----------------------------------------------------------------------------
  h8 W[9][RT];
#pragma unroll
  for (int p = 0; p < 9; p++)
#pragma unroll
    for (int r = 0; r < RT; r++) {
      const uint4 v = wfrag[(p * RT + r) * 64 + lane];
      W[p][r] = __builtin_bit_cast(h8, v);
    }

  // ---- staging: halo tile [kHZ][kHY][RT][kHX] of 16-byte slots, zero outside the grid ------------------------------------
  if (FIRST) {
    // lib/modules/variance.lua:44-76 (n-1) + Sqrt, as model.hip scale_from_stats
    const double s1 = cin.stats[b * 2], s2 = cin.stats[b * 2 + 1], n = cin.count;
    const float in_scale = (float)sqrt(fmax(n * s2 - s1 * s1, 0.0) / (n * (n - 1.0)));
    const bool scale_in_range = in_scale >= 0x1p-12f && in_scale <= 0x1p21f;     // wave-uniform
    const float inv_scale = scale_in_range ? rcp_refined(in_scale) : 0.0f;
    constexpr int kIter = (kItems + 255) / 256;
    float ld[kIter][3];
    bool okv[kIter];
#pragma unroll
    for (int i = 0; i < kIter; i++) {
      const int item = min(tid + 256 * i, kItems - 1);
      const int r = item / kHX, hx = item - r * kHX;
      const int hz = r / kHY, hy = r - hz * kHY;
      const int gx = x0 - 1 + hx, gy = y0 - 1 + hy, gz = z0 - 1 + hz;
      okv[i] = gx >= 0 && gx < d.X && gy >= 0 && gy < d.Y && gz >= 0 && gz < d.Z;
      const long long o = (long long)b * cells + TFL_AT(d, min(max(gx, 0), d.X - 1), min(max(gy, 0), d.Y - 1), min(max(gz, 0), d.Z - 1));
      ld[i][0] = cin.pDiv[o]; ld[i][1] = cin.div[o]; ld[i][2] = cin.flags[o];
    }
#pragma unroll
    for (int i = 0; i < kIter; i++) {
      const int item = tid + 256 * i;
      // the net input is built here: ApplyScale(true) = CDivTable (apply_scale.lua:24-30), FlagsToOccupancy
      // (generic/tfluids.cu:355-371); x / scale bit-equal to `/` as in conv_valu.hip
      float v0, v1;
      if (scale_in_range) { v0 = div_by<1>(ld[i][0], in_scale, inv_scale); v1 = div_by<1>(ld[i][1], in_scale, inv_scale); }
      else { v0 = ld[i][0] / in_scale; v1 = ld[i][1] / in_scale; }
      const int f = (int)ld[i][2];
      const float occ = (f == kFluid) ? 0.0f : ((f == kObstacle) ? 1.0f : -1.0f);
      const float c0 = __builtin_fminf(__builtin_fmaxf(v0, -kHalfMax), kHalfMax), c1 = __builtin_fminf(__builtin_fmaxf(v1, -kHalfMax), kHalfMax);
      clipped = clipped || (okv[i] && (c0 != v0 || c1 != v1));
      h8 s = {0, 0, 0, 0, 0, 0, 0, 0};
      if (okv[i]) {
        _Float16 ph, pl, dh, dl;
        split_h(c0, ph, pl); split_h(c1, dh, dl);
        s[0] = ph; s[1] = dh; s[2] = (_Float16)occ; s[3] = pl; s[4] = dl;
      }
      if (item < kItems) lds[item] = __builtin_bit_cast(uint4, s);
    }
  } else {
    constexpr int kIter = (kItems + 255) / 256;
    uint4 ld[kIter];
    const uint4* src = in + (long long)b * cells * 2;
#pragma unroll
    for (int i = 0; i < kIter; i++) {
      const int item = min(tid + 256 * i, kItems - 1);
      const int r = item / kHX, hx = item - r * kHX;
      const int rr = r >> 1, tm = r & 1;
      const int hz = rr / kHY, hy = rr - hz * kHY;
      const int gx = x0 - 1 + hx, gy = y0 - 1 + hy, gz = z0 - 1 + hz;
      const bool ok = gx >= 0 && gx < d.X && gy >= 0 && gy < d.Y && gz >= 0 && gz < d.Z;
      const long long o = ((long long)(min(max(gz, 0), d.Z - 1) * d.Y + min(max(gy, 0), d.Y - 1)) * 2 + tm) * d.X + min(max(gx, 0), d.X - 1);
      ld[i] = src[o];
      if (!ok) ld[i] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int i = 0; i < kIter; i++) {
      const int item = tid + 256 * i;
      if (item < kItems) lds[item] = ld[i];
    }
  }
  __syncthreads();

  // ---- main loop: wave = (x half, z pair); every (input row, term) fragment feeds up to 9 MFMAs ------------------------
  const int wx = wave & 1, wz = wave >> 1;
  const int nn = lane & 15, g = lane >> 4;
  const uint4* fbase = lds + (wz * kNZ * kHY * RT) * kHX + wx * 16 + nn + (g < 3 ? g : 0);
  f4 acc[kNZ][kNY];
#pragma unroll
  for (int a = 0; a < kNZ; a++)
#pragma unroll
    for (int c = 0; c < kNY; c++) acc[a][c] = (f4){0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int rz = 0; rz < kNZ + 2; rz++)
#pragma unroll
    for (int ry = 0; ry < kNY + 2; ry++)
#pragma unroll
      for (int tm = 0; tm < RT; tm++) {
        const h8 f = __builtin_bit_cast(h8, fbase[((rz * kHY + ry) * RT + tm) * kHX]);
#pragma unroll
        for (int dz = 0; dz < 3; dz++)
#pragma unroll
          for (int dy = 0; dy < 3; dy++) {
            const int oz = rz - dz, oy = ry - dy;
            if (oz >= 0 && oz < kNZ && oy >= 0 && oy < kNY)
              acc[oz][oy] = __builtin_amdgcn_mfma_f32_16x16x32_f16(W[dz * 3 + dy][tm], f, acc[oz][oy], 0, 0, 0);
          }
      }

  // ---- epilogue: recombine, bias, ReLU; split again (h2 out) or run the 1x1x1 tail ------------------------------------
  const int x = x0 + wx * 16 + nn;
  const int c0 = 2 * g, c1 = 2 * g + 1;
  const float bias0 = bias[c0], bias1 = bias[c1];
  float w4a[8], w4b[8], b4j[2], w5j[2], b5 = 0.0f;
  // reduce-scatter of the 8 partial sums q_j over the four lane groups: after the exchange with lane ^ 32 a lane keeps
  // j in {4 (g >> 1) .. +3}, after lane ^ 16 the two j = 4 (g >> 1) + 2 (g & 1) + {0, 1}
  const int j0 = 4 * (g >> 1) + 2 * (g & 1);
  if (TAIL) {
#pragma unroll
    for (int j = 0; j < 8; j++) { w4a[j] = bias[kTailW4 + j * 8 + c0]; w4b[j] = bias[kTailW4 + j * 8 + c1]; }
    b4j[0] = bias[kTailB4 + j0]; b4j[1] = bias[kTailB4 + j0 + 1];
    w5j[0] = bias[kTailW5 + j0]; w5j[1] = bias[kTailW5 + j0 + 1];
    b5 = bias[kTailB5];
  }
#pragma unroll
  for (int oz = 0; oz < kNZ; oz++) {
    const int z = z0 + wz * kNZ + oz;
#pragma unroll
    for (int oy = 0; oy < kNY; oy++) {
      const int y = y0 + oy;
      const bool live = x < d.X && y < d.Y && z < z_end;
      const f4 a = acc[oz][oy];
      float h0 = __builtin_fmaxf((a[0] + a[1] * 0x1p-11f) * post + bias0, 0.0f);
      float h1 = __builtin_fmaxf((a[2] + a[3] * 0x1p-11f) * post + bias1, 0.0f);
      if (!TAIL) {
        const float k0 = __builtin_fminf(h0, kHalfMax), k1 = __builtin_fminf(h1, kHalfMax);
        clipped = clipped || (live && (k0 != h0 || k1 != h1));
        _Float16 hh0, hl0, hh1, hl1;
        split_h(k0, hh0, hl0); split_h(k1, hh1, hl1);
        if (live) {
          uint32_t* orow = reinterpret_cast<uint32_t*>(outv) + ((((long long)b * d.Z + z) * d.Y + y) * 2 * d.X + x) * 4 + g;
          const h2v ph = {hh0, hh1}, pl = {hl0, hl1};
          orow[0] = __builtin_bit_cast(uint32_t, ph);
          orow[(long long)d.X * 4] = __builtin_bit_cast(uint32_t, pl);
        }
      } else {
        float q[8];
#pragma unroll
        for (int j = 0; j < 8; j++) q[j] = w4a[j] * h0 + w4b[j] * h1;
        float r4[4];
#pragma unroll
        for (int j = 0; j < 4; j++) {       // lanes g >> 1 == 0 keep j 0..3 and send 4..7; the others the reverse
          const float send = (g >> 1) ? q[j] : q[4 + j];
          const float keep = (g >> 1) ? q[4 + j] : q[j];
          r4[j] = keep + __shfl_xor(send, 32);
        }
        float r2[2];
#pragma unroll
        for (int j = 0; j < 2; j++) {
          const float send = (g & 1) ? r4[j] : r4[2 + j];
          const float keep = (g & 1) ? r4[2 + j] : r4[j];
          r2[j] = keep + __shfl_xor(send, 16);
        }
        float pp = w5j[0] * __builtin_fmaxf(r2[0] + b4j[0], 0.0f) + w5j[1] * __builtin_fmaxf(r2[1] + b4j[1], 0.0f);
        pp += __shfl_xor(pp, 16);
        pp += __shfl_xor(pp, 32);
        if (live && g == 0) reinterpret_cast<float*>(outv)[(long long)b * cells + TFL_AT(d, x, y, z)] = pp + b5;
      }
    }
  }
  if (clipped) atomicAdd(range_err, 1ull);
}

template <int MODE>
static void launch_m16(hipStream_t st, const Dom& d, int B, const void* in, const void* wfrag, const float* bias, void* out,
                       float post, MIn cin, unsigned long long* range_err) {
  const int tx = (d.X + kTX - 1) / kTX, ty = (d.Y + kTY - 1) / kTY;
  const int tz = (d.n0 + kTZ - 1) / kTZ + (d.nw - d.n0 + kTZ - 1) / kTZ;   // z-tiles of the compute window's two plane runs
  const int n_tiles = tx * ty * tz * B;
  if (n_tiles <= 0) return;
  const int grid = ((n_tiles + 7) / 8) * 8;
  const size_t lds_bytes = (size_t)16 * kHZ * kHY * (MODE == kModeIn ? 1 : 2) * kHX;
  static int attr_dev = -1;                    // the attribute is per device
  int dev = 0; (void)hipGetDevice(&dev);
  if (attr_dev != dev) {
    (void)hipFuncSetAttribute((const void*)k_conv3_m16<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_dev = dev;
    if (getenv("TFL_DEBUG")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_conv3_m16<MODE>, 256, lds_bytes);
      fprintf(stderr, "[tfl] k_conv3_m16<%d>: dynamic LDS %zu B, occupancy %d blocks/CU, grid %d\n", MODE, lds_bytes, nb, grid);
    }
  }
  TFL_TIMED_EXT(MODE == kModeTail ? "k_conv3_tail" : (MODE == kModeIn ? "k_conv3_in" : "k_conv3_mid"), st);
  TFL_LAUNCH_EXT((k_conv3_m16<MODE>), grid, 256, lds_bytes, st, d, tx, ty, tz, n_tiles, (const uint4*)in, (const uint4*)wfrag,
                 bias, out, post, cin, range_err);
}

void conv3_m16_first_fused(hipStream_t st, int B, int Z, int Y, int X, const float* pDiv, const float* div, const float* flags,
                           const double* stats, double count, const void* wfrag, const float* bias, float post, void* out_h2,
                           unsigned long long* range_err) {
  MIn ci = {pDiv, div, flags, stats, count};
  launch_m16<kModeIn>(st, make_dom(Z, Y, X), B, nullptr, wfrag, bias, out_h2, post, ci, range_err);
}
void conv3_m16_mid(hipStream_t st, int B, int Z, int Y, int X, const void* in_h2, const void* wfrag, const float* bias, float post,
                   void* out_h2, unsigned long long* range_err) {
  MIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  launch_m16<kModeMid>(st, make_dom(Z, Y, X), B, in_h2, wfrag, bias, out_h2, post, noin, range_err);
}
void conv3_m16_tail(hipStream_t st, int B, int Z, int Y, int X, const void* in_h2, const void* wfrag, const float* tail_pack,
                    float post, float* p_out, unsigned long long* range_err) {
  MIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  launch_m16<kModeTail>(st, make_dom(Z, Y, X), B, in_h2, wfrag, tail_pack, p_out, post, noin, range_err);
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

// out: 9 * RT * 64 * 8 halves (RT = 1 for cin == 3, 2 for cin == 8); returns the post-scale 2^-(11 + e)
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
  return ldexpf(1.0f, -(11 + e));
}

}  // namespace tfl
