// conv2d_mfma.hip -- the 2-D `default` projection net (3->16, 16->16 x3 k=3, 16->1 k=1; the topology of
// the shipped data/models/myModel2D) as an fp32 MFMA implicit GEMM on gfx950 (v_mfma_f32_16x16x4_f32).
//
// Replaces cudnn.SpatialConvolution forward + ReLU (torch/lib/model_utils.lua:80-99). With N = 16 output
// channels the 16x16 tile is filled natively:
//   D[m][n]  m = 16 consecutive x pixels, n = output channel
//   A[m][k] = in[c0 + k][y + dy][x0 + m + dx]   k = 0..3: FOUR INPUT CHANNELS of one tap per MFMA
//   B[k][n] = w[n][c0 + k][dy][dx]
// i.e. 9 * C_in/4 MFMAs per 16 pixels (36 for the 16->16 layers; the first layer pads 3 channels to 4 with
// zero weights). The block stages a channel-planar halo tile [C_in][6][34] in LDS whose plane pitch is
// == 16 (mod 32) dwords, so the two channel planes a 32-lane half reads land on disjoint bank halves.
// Block = 4 waves, tile 32 x 4 pixels, one row (two 16-pixel MFMA tiles = two independent accumulators) per
// wave. 128^2 -> 128 blocks: the whole net is ~120k MFMAs, launch-bound by construction, so layers are as fused
// as the halo allows: network input built while staging (layer 1), 1x1 output layer folded into layer 4.
// Activations stay channel-planar [16][Y][X]: in the D layout a lane owns 4 consecutive pixels of one channel
// = one float4 store.
#include "tfl_device.hpp"
#include "tfl_host.hpp"

namespace tfl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int k2TX = 32, k2TY = 4, k2LW = 34, k2Rows = k2TY + 2;
constexpr int k2Plane = 208;   // >= 6*34 = 204 and == 16 (mod 32)

struct Conv2In {   // fused network input (first layer): {pDiv/scale, div/scale, occupancy(flags)}
  const float* pDiv; const float* div; const float* flags; const double* stats; double count;
};
struct Conv2Tail { const float* w5; const float* b5; };   // fused 16 -> 1, k = 1 output layer

// CIN4 = input channels padded to a multiple of 4 (4 or 16). FUSED: build the net input while staging.
// TAIL: fold the 16 -> 1 (k=1) layer into the epilogue and write the planar pressure.
template <int CIN4, bool FUSED, bool TAIL>
__global__ __launch_bounds__(256) void k_conv2_mfma(int B, int Y, int X, const float* __restrict__ in,
                                                    const float* __restrict__ bfrag, const float* __restrict__ bias,
                                                    float* __restrict__ out, Conv2In ci, Conv2Tail tail) {
  __shared__ float lds[CIN4 * k2Plane];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_x = (X + k2TX - 1) / k2TX, tiles_y = (Y + k2TY - 1) / k2TY;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y;
  const int b = t / tiles_y;
  const int x0 = tx * k2TX, y0 = ty * k2TY;
  const long long cells = (long long)X * Y;
  float in_scale = 1.0f;
  if (FUSED) {   // lib/modules/variance.lua:44-76 (n-1) + Sqrt
    const double s1 = ci.stats[b * 2], s2 = ci.stats[b * 2 + 1], n = ci.count;
    in_scale = (float)sqrt(fmax(n * s2 - s1 * s1, 0.0) / (n * (n - 1.0)));
  }
  // ---- B fragments (per-lane, host-arranged): [tap][c4 group]; asked for first, they travel with the tile's loads ------
  constexpr int NB = 9 * (CIN4 / 4);
  float bf[NB];
#pragma unroll
  for (int q = 0; q < NB; q++) bf[q] = bfrag[q * 64 + lane];
  const float bv = bias[lane & 15];
  // ---- stage the halo tile (zero outside the image = the convolution's zero padding) -----------------
  constexpr int CREAL = FUSED ? 3 : CIN4;
  for (int idx = tid; idx < k2Rows * k2LW; idx += 256) {
    const int xx = idx % k2LW, yy = idx / k2LW;
    const int gx = x0 - 1 + xx, gy = y0 - 1 + yy;
    const bool ok = gx >= 0 && gx < X && gy >= 0 && gy < Y;
    const long long o = (long long)gy * X + gx;
    if (FUSED) {
      float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f;
      if (ok) {
        const long long bo = b * cells + o;
        v0 = ci.pDiv[bo] / in_scale;   // ApplyScale(true) = CDivTable, apply_scale.lua:24-30
        v1 = ci.div[bo] / in_scale;
        const int f = (int)ci.flags[bo];   // FlagsToOccupancy, generic/tfluids.cu:355-371
        v2 = (f == kFluid) ? 0.0f : ((f == kObstacle) ? 1.0f : -1.0f);
      }
      lds[0 * k2Plane + idx] = v0; lds[1 * k2Plane + idx] = v1; lds[2 * k2Plane + idx] = v2;
      lds[3 * k2Plane + idx] = 0.0f;
    } else {
      // unconditional loads (a predicated load is a branch whose join drains the load queue: the 16 channels went two per
      // memory round trip); a halo cell outside the image reads the image's first pixel and stores the padding zero
      const float* ip = in + b * cells * CREAL + (ok ? o : 0);
      float v[CIN4];
#pragma unroll
      for (int c = 0; c < CIN4; c++) v[c] = ip[c * cells];
#pragma unroll
      for (int c = 0; c < CIN4; c++) lds[c * k2Plane + idx] = ok ? v[c] : 0.0f;
    }
  }
  const int n = lane & 15, k4 = lane >> 4;
  f32x4 acc[2] = {(f32x4){bv, bv, bv, bv}, (f32x4){bv, bv, bv, bv}};
  __syncthreads();
  // ---- implicit GEMM: one MFMA = one tap x four input channels ------------------------------------------
  const float* abase = lds + k4 * k2Plane + wave * k2LW + (lane & 15);
#pragma unroll
  for (int c4 = 0; c4 < CIN4 / 4; c4++) {
#pragma unroll
    for (int dy = 0; dy < 3; dy++) {
#pragma unroll
      for (int dx = 0; dx < 3; dx++) {
        const float bval = bf[(dy * 3 + dx) * (CIN4 / 4) + c4];
        const float* ap = abase + c4 * 4 * k2Plane + dy * k2LW + dx;
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[0], bval, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[16], bval, acc[1], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: lane holds pixels 4g..4g+3 (g = lane>>4) of channel n of each 16-pixel tile -------------
  const int y = y0 + wave, g = lane >> 4;
  if (!TAIL) {
    float* op = out + b * cells * 16 + (long long)n * cells + (long long)y * X;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      const int x = x0 + h * 16 + 4 * g;
      if (y < Y) {
        if (x + 3 < X && ((reinterpret_cast<uintptr_t>(op + x) & 15) == 0)) {
          float4 v = make_float4(fmaxf(acc[h][0], 0.0f), fmaxf(acc[h][1], 0.0f), fmaxf(acc[h][2], 0.0f), fmaxf(acc[h][3], 0.0f));
          *reinterpret_cast<float4*>(op + x) = v;
        } else {
#pragma unroll
          for (int i = 0; i < 4; i++) if (x + i < X) op[x + i] = fmaxf(acc[h][i], 0.0f);
        }
      }
    }
  } else {
    const float w5 = tail.w5[n], b5 = tail.b5[0];
    float* op = out + b * cells + (long long)y * X;
#pragma unroll
    for (int h = 0; h < 2; h++) {
#pragma unroll
      for (int i = 0; i < 4; i++) {
        float v = w5 * fmaxf(acc[h][i], 0.0f);
        v += __shfl_xor(v, 1, 64); v += __shfl_xor(v, 2, 64); v += __shfl_xor(v, 4, 64); v += __shfl_xor(v, 8, 64);
        const int x = x0 + h * 16 + 4 * g + i;
        if (n == 0 && x < X && y < Y) op[x] = v + b5;
      }
    }
  }
}

template <int CIN4, bool FUSED, bool TAIL>
static void launch2(hipStream_t st, int B, int Y, int X, const float* in, const float* bfrag, const float* bias,
                    float* out, Conv2In ci, Conv2Tail tail) {
  const int grid = ((X + k2TX - 1) / k2TX) * ((Y + k2TY - 1) / k2TY) * B;
  TFL_TIMED(TAIL ? "k_conv2_mfma_tail" : (FUSED ? "k_conv2_mfma_in" : "k_conv2_mfma"), st);
  k_conv2_mfma<CIN4, FUSED, TAIL><<<grid, 256, 0, st>>>(B, Y, X, in, bfrag, bias, out, ci, tail);
}

// layer 1: {pDiv, div, flags, stats} -> 16 planar channels (+ReLU)
void conv2_mfma_first_fused(hipStream_t st, int B, int Y, int X, const float* pDiv, const float* div, const float* flags,
                            const double* stats, double count, const float* bfrag, const float* bias, float* out16) {
  Conv2In ci = {pDiv, div, flags, stats, count};
  Conv2Tail none = {nullptr, nullptr};
  launch2<4, true, false>(st, B, Y, X, nullptr, bfrag, bias, out16, ci, none);
}
// layers 2, 3: 16 -> 16 (+ReLU)
void conv2_mfma_mid(hipStream_t st, int B, int Y, int X, const float* in16, const float* bfrag, const float* bias,
                    float* out16) {
  Conv2In noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  Conv2Tail none = {nullptr, nullptr};
  launch2<16, false, false>(st, B, Y, X, in16, bfrag, bias, out16, noin, none);
}
// layer 4 (16 -> 16 + ReLU) with layer 5 (16 -> 1, k = 1) folded in; planar pressure out
void conv2_mfma_tail(hipStream_t st, int B, int Y, int X, const float* in16, const float* bfrag, const float* bias,
                     const float* w5, const float* b5, float* p_out) {
  Conv2In noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  Conv2Tail tail = {w5, b5};
  launch2<16, false, true>(st, B, Y, X, in16, bfrag, bias, p_out, noin, tail);
}

}  // namespace tfl
