// conv_mfma.hip -- 3x3x3, 8-output-channel convolution layers of the 3-D `default` projection net
// as an fp32 MFMA implicit GEMM (gfx950: v_mfma_f32_16x16x4_f32, exact fp32, 157 TFLOP/s peak).
//
// Replaces cudnn.VolumetricConvolution forward (torch/lib/model_utils.lua:104-116) for the layers
// 3->8, 8->8, 8->8 (k=3) of lib/model.lua:219-226, with the ReLU fused, and -- in the last of them --
// the two trailing 1x1x1 layers (8->8 + ReLU, 8->1) fused into the epilogue.
//
// GEMM shape problem: N = 8 output channels, but the narrowest MFMA tile is 16 wide. Instead of
// wasting half of it, one 16x16 tile covers 32 consecutive x-voxels x 8 channels ("x-phase packing"):
//   D[m][n]   m = 0..15, n = ph*8 + co : output channel co of voxel x0 + 2m + ph
//   A[m][k] = in[c][z+dz][y+dy][x0 + 2m + k - 1], k = 0..3   (a 4-wide x window per row)
//   B[k][n] = w[co][c][dz][dy][k - ph] if 0 <= k - ph <= 2 else 0
// so one MFMA (K = 4) consumes one (c, dz, dy) of the stencil for both phases: C_in*9 MFMAs per tile,
// 75% of the issued MACs useful (vs 50% with N padded to 16).
//
// Data layout: the block stages a channel-planar halo tile [C_in][TZ+2][TY+2][34(+2)] in LDS (zero
// padded at the domain boundary = the convolution's zero padding); an A operand is one ds_read_b32 at
// lane offset 2m + k, bank-conflict-free by construction (each 32-lane half touches 32 consecutive
// dwords). Inter-layer activations live in HBM channel-LAST ([Z][Y][X][8], 32 B per voxel) so staging
// loads and epilogue stores are 16/64-byte contiguous. B fragments (the weights, pre-arranged per lane
// on the host) sit in registers for the whole kernel: 9*C_in VGPRs.
//
// The MFMA is issued TRANSPOSED (weights as the A operand, the x window as B): D'[n][m] = D[m][n], so a lane
// ends up with FOUR CONSECUTIVE CHANNELS of one voxel (rows n = 4*(lane/16) + 0..3 of column m = lane%16):
// the epilogue is one 16-byte store per lane and row (a wave writes 1 KB contiguous), and D' is already in the
// B-operand layout of a further MFMA that contracts over n -- which is how the last layer's fused 1x1x1
// convolution (8 -> 8) is evaluated: 4 more MFMAs per row against a block-diagonal [16][16] copy of its weights,
// no cross-lane traffic.
//
// Block = 256 threads = 4 waves; block tile = 32(x) x 8(y) x 4(z); wave w owns z-plane w, 8 rows ->
// 8 independent accumulators (covers the 40-cycle dependent-MFMA latency at the 32-cycle issue rate).
// The input channels are staged 4 at a time: LDS 4*6*10*36*4 = 34.5 KB -> 4 blocks per CU, so while one
// block stages or stores, the other three keep the MFMA pipes busy (r01: 2 blocks/CU left them 67% busy).
#include "tfl_device.hpp"
#include "tfl_host.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

namespace tfl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTX = 32, kTY = 8, kTZ = 4, kLX = 36;  // LDS row pitch (34 used)

struct ConvTail {       // fused 1x1x1 layers (device pointers): h4 = relu(W4 h + b4); p = w5 . h4 + b5
  const float* w4;      // [8][8]  (out, in)
  const float* b4;      // [8]
  const float* w5;      // [8]
  const float* b5;      // [1]
};

struct ConvIn {         // fused network input (first layer): {pDiv/scale, div/scale, occupancy(flags)}
  const float* pDiv;    // [B][1][Z][Y][X]
  const float* div;
  const float* flags;
  const double* stats;  // [B][2] = sum u, sum u^2 (model.hip); nullptr -> read the planar `in` instead
  double count;
};

// CIN: input channels. IN_PLANAR: input is [CIN][Z][Y][X] (first layer) else channel-last [Z][Y][X][CIN].
// TAIL: fuse the two 1x1x1 layers and write planar pressure instead of channel-last activations.
template <int CIN, bool IN_PLANAR, bool TAIL>
__global__ __launch_bounds__(256, 4) void k_conv3_mfma(Dom d, int tiles_x, int tiles_y, int tiles_z, int n_tiles,
                                                       const float* __restrict__ in, const float* __restrict__ bfrag,
                                                       const float* __restrict__ bias, float* __restrict__ out,
                                                       ConvTail tail, ConvIn cin, int dbg, unsigned long long* __restrict__ trace) {
  // dbg (env TFL_CONV_DEBUG, timing experiments only -- results are garbage): bit 0 skips the MFMA loop,
  // bit 1 skips the staging loads. Block-uniform branches, free when 0.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  // phase timestamps of wave 0 of every block (TFL_CONV_TRACE=1, development aid): s_memtime = shader clock
#define TFL_STAMP(slot) do { if (trace && threadIdx.x == 0) trace[(long long)blockIdx.x * 8 + (slot)] = __builtin_amdgcn_s_memtime(); } while (0)
  // XCD-aware tile order: the dispatcher deals consecutive block ids round-robin over the 8 XCDs, so
  // give each XCD a contiguous run of tiles (neighbouring tiles share halo planes through its L2).
  const int per_xcd = (n_tiles + 7) / 8;
  const int tile = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
  if (tile >= n_tiles) return;
  TFL_STAMP(0);
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
  in += (long long)b * cells * CIN;

  const int tid = threadIdx.x;
  constexpr int kRows = (kTZ + 2) * (kTY + 2);      // 60 halo rows per channel
  constexpr int kPlane = kRows * kLX;               // floats per channel plane in LDS
  constexpr int CG = CIN < 4 ? CIN : 4;             // channels staged per pass (34.5 KB -> 4 blocks per CU:
                                                    // while one block stages, three keep the MFMA pipes busy)
  const int lane = tid & 63, wave = tid >> 6;
  float bf[CIN * 9];
#pragma unroll
  for (int q = 0; q < CIN * 9; q++) bf[q] = bfrag[q * 64 + lane];
  // D' layout: lane holds rows n = 4*g + i (i = 0..3) of column m = lane & 15; n = ph*8 + co
  const int g = lane >> 4, ph = g >> 1, co0 = 4 * (g & 1);
  const f32x4 bv = {bias[co0], bias[co0 + 1], bias[co0 + 2], bias[co0 + 3]};
  f32x4 acc[kTY];
#pragma unroll
  for (int r = 0; r < kTY; r++) acc[r] = bv;
  const int lane_off = 2 * (lane & 15) + (lane >> 4);
  float in_scale = 1.0f;
  if (IN_PLANAR && cin.stats) {  // lib/modules/variance.lua:44-76 (n-1) + Sqrt, as model.hip scale_from_stats
    const double s1 = cin.stats[b * 2], s2 = cin.stats[b * 2 + 1], n = cin.count;
    in_scale = (float)sqrt(fmax(n * s2 - s1 * s1, 0.0) / (n * (n - 1.0)));
  }

#pragma unroll
  for (int cg = 0; cg < CIN; cg += CG) {
    if (cg > 0) __syncthreads();   // everyone is done reading the previous channel group
    // ---- stage the halo tile of channels [cg, cg + CG) --------------------------------------------
    // element -> (row, xx) without integer division by 34: the 32 interior columns are dealt 32 per row
    // (shift/mask), the two halo columns (xx = 0, 33) of the 60 rows go to the first 120 threads: 9 elements per
    // thread. The loads are UNCONDITIONAL (address clamped into the grid, value zeroed afterwards) and issued
    // kBatch at a time before any of them is consumed: with a branch around each load the compiler serialises
    // them, and a stage then costs 9 dependent L2 round trips (stage-2 wait 25 k -> 13 k cycles, r01 trace).
    constexpr int kElems = (kRows * 32 + 255) / 256 + 1;     // 8 interior passes + 1 halo-column pass
    constexpr int kBatch = IN_PLANAR ? kElems : 4;           // registers: 4 float4 in flight beside 72 B fragments
    constexpr int NLD = IN_PLANAR ? 3 : 1;                   // loads per element
    auto elem = [&](int e, int& row, int& xx, bool& live) {
      if (e < kElems - 1) { row = (tid >> 5) + e * 8; xx = (tid & 31) + 1; live = row < kRows; }
      else { row = tid >> 1; xx = (tid & 1) * 33; live = tid < 2 * kRows; }
      if (!live) row = 0;
    };
    if (!(dbg & 2)) {
#pragma unroll
      for (int e0 = 0; e0 < kElems; e0 += kBatch) {
        float4 ld[kBatch][NLD];
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
          if (e0 + u >= kElems) continue;
          int row, xx; bool live;
          elem(e0 + u, row, xx, live);
          const int zz = (row * 205) >> 11;            // row / 10 for row < 1024
          const int yy = row - zz * (kTY + 2);
          const int gx = min(max(x0 - 1 + xx, 0), d.X - 1), gy = min(max(y0 - 1 + yy, 0), d.Y - 1);
          const int gz = min(max(z0 - 1 + zz, 0), d.Z - 1);
          const long long o = TFL_AT(d, gx, gy, gz);
          if (IN_PLANAR && cin.stats) {
            const long long bo = (long long)b * cells + o;
            ld[u][0].x = cin.pDiv[bo]; ld[u][1 % NLD].x = cin.div[bo]; ld[u][2 % NLD].x = cin.flags[bo];
          } else if (IN_PLANAR) {
#pragma unroll
            for (int c = 0; c < CG; c++) ld[u][c % NLD].x = in[o + (cg + c) * cells];
          } else {
            ld[u][0] = *reinterpret_cast<const float4*>(in + o * CIN + cg);
          }
        }
#pragma unroll
        for (int u = 0; u < kBatch; u++) {
          if (e0 + u >= kElems) continue;
          int row, xx; bool live;
          elem(e0 + u, row, xx, live);
          const int zz = (row * 205) >> 11;
          const int yy = row - zz * (kTY + 2);
          const int gx = x0 - 1 + xx, gy = y0 - 1 + yy, gz = z0 - 1 + zz;
          const bool ok = gx >= 0 && gx < d.X && gy >= 0 && gy < d.Y && gz >= 0 && gz < d.Z;
          float v[CG];
          if (IN_PLANAR && cin.stats) {
            // the net input is built here instead of by k_net_input: ApplyScale(true) = CDivTable
            // (apply_scale.lua:24-30), FlagsToOccupancy (generic/tfluids.cu:355-371)
            v[0] = ld[u][0].x / in_scale;
            v[1] = ld[u][1 % NLD].x / in_scale;
            const int f = (int)ld[u][2 % NLD].x;
            v[CG - 1] = (f == kFluid) ? 0.0f : ((f == kObstacle) ? 1.0f : -1.0f);
          } else if (IN_PLANAR) {
#pragma unroll
            for (int c = 0; c < CG; c++) v[c] = ld[u][c % NLD].x;
          } else {
            v[0] = ld[u][0].x; v[1] = ld[u][0].y; v[CG > 2 ? 2 : 0] = ld[u][0].z; v[CG > 3 ? 3 : 0] = ld[u][0].w;
          }
          if (live) {
#pragma unroll
            for (int c = 0; c < CG; c++) lds[c * kPlane + row * kLX + xx] = ok ? v[c] : 0.0f;
          }
        }
      }
    }
    __syncthreads();
    TFL_STAMP(1 + 2 * (cg / CG));
    // ---- implicit GEMM over these channels ---------------------------------------------------------
    // One step = one (c, dz): the 10 halo rows of LDS plane (wave + dz) feed 3 (dy) x 8 (rows) = 24 MFMAs;
    // consecutive uses of one accumulator are 8 MFMAs apart (> the 40-cycle dependent latency).
    if (!(dbg & 1))
#pragma unroll
    for (int cl = 0; cl < CG; cl++) {
#pragma unroll
      for (int dz = 0; dz < 3; dz++) {
        const float* base = lds + cl * kPlane + ((wave + dz) * (kTY + 2)) * kLX + lane_off;
        float a[kTY + 2];
#pragma unroll
        for (int q = 0; q < kTY + 2; q++) a[q] = base[q * kLX];
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
          const float bval = bf[((cg + cl) * 3 + dz) * 3 + dy];
#pragma unroll
          for (int r = 0; r < kTY; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(bval, a[r + dy], acc[r], 0, 0, 0);
        }
      }
    }
    TFL_STAMP(2 + 2 * (cg / CG));
  }
  TFL_STAMP(1 + 2 * ((CIN + CG - 1) / CG));
  // ---- epilogue ----------------------------------------------------------------------------------
  const int z = z0 + wave;
  const int x = x0 + 2 * (lane & 15) + ph;      // the lane's voxel; its channels co0 .. co0+3
  if (!TAIL) {
    out += (long long)b * cells * 8;
#pragma unroll
    for (int r = 0; r < kTY; r++) {
      const int y = y0 + r;
      if (x < d.X && y < d.Y && z < z_end) {
        const float4 v = make_float4(fmaxf(acc[r][0], 0.0f), fmaxf(acc[r][1], 0.0f), fmaxf(acc[r][2], 0.0f),
                                     fmaxf(acc[r][3], 0.0f));
        *reinterpret_cast<float4*>(out + (long long)TFL_AT(d, x, y, z) * 8 + co0) = v;
      }
    }
  } else {
    // 8 -> 8 (k = 1) + ReLU as a second MFMA: E'[n'][m] = b4 + sum_n W4big[n'][n] * relu(D')[n][m], where
    // W4big = diag(w4, w4) over the two x phases. relu(D') register i of lane (k = g, m) is element
    // [4k + i][m], i.e. exactly the B operand of the partial product over n = 4k + i, k = 0..3.
    float a4[4];
    {
      const int np = lane & 15, php = np >> 3, cop = np & 7;
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const int n = 4 * g + i;
        a4[i] = ((n >> 3) == php) ? tail.w4[cop * 8 + (n & 7)] : 0.0f;
      }
    }
    const f32x4 b4v = {tail.b4[co0], tail.b4[co0 + 1], tail.b4[co0 + 2], tail.b4[co0 + 3]};
    const float w5v[4] = {tail.w5[co0], tail.w5[co0 + 1], tail.w5[co0 + 2], tail.w5[co0 + 3]};
    const float b5 = tail.b5[0];
    out += (long long)b * cells;
#pragma unroll
    for (int r = 0; r < kTY; r++) {
      f32x4 e = b4v;
#pragma unroll
      for (int i = 0; i < 4; i++) e = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[i], fmaxf(acc[r][i], 0.0f), e, 0, 0, 0);
      // 8 -> 1: this lane's four channels, then the other four from lane ^ 16 (same voxel, co0 ^ 4)
      float v = w5v[0] * fmaxf(e[0], 0.0f);
#pragma unroll
      for (int i = 1; i < 4; i++) v = fmaf(w5v[i], fmaxf(e[i], 0.0f), v);
      const float other = __shfl_xor(v, 16, 64);
      const int y = y0 + r;
      if (co0 == 0 && x < d.X && y < d.Y && z < z_end) out[TFL_AT(d, x, y, z)] = (v + other) + b5;
    }
  }
  TFL_STAMP(7);
#undef TFL_STAMP
}

template <int CIN, bool IN_PLANAR, bool TAIL>
static void launch_mfma(hipStream_t st, const Dom& d, int B, const float* in, const float* bfrag, const float* bias,
                        float* out, ConvTail tail, ConvIn cin) {
  const int tx = (d.X + kTX - 1) / kTX, ty = (d.Y + kTY - 1) / kTY;
  const int tz = (d.n0 + kTZ - 1) / kTZ + (d.nw - d.n0 + kTZ - 1) / kTZ;   // z-tiles of the compute window's two plane runs
  const int n_tiles = tx * ty * tz * B;
  const int grid = ((n_tiles + 7) / 8) * 8;
  const size_t lds_bytes = sizeof(float) * (CIN < 4 ? CIN : 4) * (kTZ + 2) * (kTY + 2) * kLX;
  static int attr_dev = -1;                    // the attribute is a per-device setting
  int cur_dev = 0; (void)hipGetDevice(&cur_dev);
  if (attr_dev != cur_dev) {
    (void)hipFuncSetAttribute((const void*)k_conv3_mfma<CIN, IN_PLANAR, TAIL>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_dev = cur_dev;
    if (getenv("TFL_DEBUG")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_conv3_mfma<CIN, IN_PLANAR, TAIL>, 256,
                                                         lds_bytes);
      fprintf(stderr, "[tfl] k_conv3_mfma<%d,%d,%d>: dynamic LDS %zu B, occupancy %d blocks/CU, grid %d\n", CIN,
              (int)IN_PLANAR, (int)TAIL, lds_bytes, nb, grid);
    }
  }
  TFL_TIMED_EXT(TAIL ? "k_conv3_mfma_tail" : (IN_PLANAR ? "k_conv3_mfma_in" : "k_conv3_mfma"), st);
  static const int dbg = exp_env("TFL_CONV_DEBUG") ? atoi(exp_env("TFL_CONV_DEBUG")) : 0;
  static const bool want_trace = exp_env("TFL_CONV_TRACE") != nullptr;
  if (want_trace) {
    // development aid: per-block phase timestamps of this launch, summarised on stderr (synchronises!)
    unsigned long long* dev = nullptr;
    const size_t n = (size_t)grid * 8;
    if (hipMalloc((void**)&dev, n * 8) != hipSuccess) return;
    (void)hipMemsetAsync(dev, 0, n * 8, st);
    k_conv3_mfma<CIN, IN_PLANAR, TAIL><<<grid, 256, lds_bytes, st>>>(d, tx, ty, tz, n_tiles, in, bfrag, bias, out, tail, cin, dbg, dev);
    std::vector<unsigned long long> h(n);
    (void)hipStreamSynchronize(st);
    (void)hipMemcpy(h.data(), dev, n * 8, hipMemcpyDeviceToHost);
    (void)hipFree(dev);
    // s_memtime bases differ between XCDs: compare only within one XCD (block b runs on XCD b % 8). Blocks whose
    // start lies within 2000 cycles of their XCD's first start are the first dispatch round.
    const int nst = 2 * ((CIN + 3) / 4) + 1;   // slot of the epilogue start
    double seg[2][8] = {{0}}; int cnt[2] = {0, 0}; double life[2] = {0, 0}; double xspan = 0;
    for (int x = 0; x < 8; x++) {
      unsigned long long t0 = ~0ull, t1 = 0;
      for (int b = x; b < grid; b += 8) if (h[b * 8]) { t0 = std::min(t0, h[b * 8]); t1 = std::max(t1, h[b * 8 + 7]); }
      xspan += (double)(t1 - t0) / 8;
      for (int b = x; b < grid; b += 8) {
        if (!h[b * 8]) continue;
        const int late = (h[b * 8] - t0) > 2000 ? 1 : 0;
        cnt[late]++;
        for (int q = 1; q <= nst; q++) seg[late][q] += (double)(h[b * 8 + q] - h[b * 8 + q - 1]);
        seg[late][7] += (double)(h[b * 8 + 7] - h[b * 8 + nst]);
        life[late] += (double)(h[b * 8 + 7] - h[b * 8]);
      }
    }
    fprintf(stderr, "[tfl] conv trace <%d,%d,%d>: kernel span %.0f cyc (per XCD);", CIN, (int)IN_PLANAR, (int)TAIL, xspan);
    for (int l = 0; l < 2; l++) {
      if (!cnt[l]) continue;
      fprintf(stderr, " %s %d blocks: life %.0f =", l ? "| later" : "first-round", cnt[l], life[l] / cnt[l]);
      for (int q = 1; q < nst; q++) fprintf(stderr, " %s%d %.0f", (q & 1) ? "stage" : "mfma", (q + 1) / 2, seg[l][q] / cnt[l]);
      fprintf(stderr, " epilogue %.0f", seg[l][7] / cnt[l]);
    }
    fprintf(stderr, "\n");
    return;
  }
  TFL_LAUNCH_EXT((k_conv3_mfma<CIN, IN_PLANAR, TAIL>), grid, 256, lds_bytes, st, d, tx, ty, tz, n_tiles, in, bfrag, bias, out, tail,
                 cin, dbg, (unsigned long long*)nullptr);
}

// 3 -> 8 (planar in) / 8 -> 8 (channel-last in), k = 3, ReLU; channel-last [Z][Y][X][8] out.
void conv3_mfma_first(hipStream_t st, int B, int Z, int Y, int X, const float* in_planar3, const float* bfrag,
                      const float* bias, float* out_cl8) {
  ConvTail none = {nullptr, nullptr, nullptr, nullptr};
  ConvIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  launch_mfma<3, true, false>(st, make_dom(Z, Y, X), B, in_planar3, bfrag, bias, out_cl8, none, noin);
}
// the same with the network input {pDiv/scale, div/scale, occupancy} built on the fly while staging
void conv3_mfma_first_fused(hipStream_t st, int B, int Z, int Y, int X, const float* pDiv, const float* div,
                            const float* flags, const double* stats, double count, const float* bfrag,
                            const float* bias, float* out_cl8) {
  ConvTail none = {nullptr, nullptr, nullptr, nullptr};
  ConvIn ci = {pDiv, div, flags, stats, count};
  launch_mfma<3, true, false>(st, make_dom(Z, Y, X), B, pDiv, bfrag, bias, out_cl8, none, ci);
}
void conv3_mfma_mid(hipStream_t st, int B, int Z, int Y, int X, const float* in_cl8, const float* bfrag,
                    const float* bias, float* out_cl8) {
  ConvTail none = {nullptr, nullptr, nullptr, nullptr};
  ConvIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  launch_mfma<8, false, false>(st, make_dom(Z, Y, X), B, in_cl8, bfrag, bias, out_cl8, none, noin);
}
// 8 -> 8 k3 + ReLU, then 8 -> 8 k1 + ReLU, then 8 -> 1 k1; planar pressure out.
void conv3_mfma_tail(hipStream_t st, int B, int Z, int Y, int X, const float* in_cl8, const float* bfrag,
                     const float* bias, const float* w4, const float* b4, const float* w5, const float* b5,
                     float* p_out) {
  ConvTail tail = {w4, b4, w5, b5};
  ConvIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  launch_mfma<8, false, true>(st, make_dom(Z, Y, X), B, in_cl8, bfrag, bias, p_out, tail, noin);
}

}  // namespace tfl
