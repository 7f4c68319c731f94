// conv_valu.hip -- the 3x3x3, 8-output-channel convolution layers of the 3-D `default` projection net on the vector ALUs
// (gfx950), fp32, with the x-taps evaluated as Winograd F(2,3).
//
// Replaces cudnn.VolumetricConvolution forward (torch/lib/model_utils.lua:104-116) for the layers 3->8, 8->8, 8->8 (k=3)
// of lib/model.lua:219-226 with the ReLU fused and, in the last of them, the two trailing 1x1x1 layers (8->8 + ReLU,
// 8->1) evaluated in registers. Same entry points as conv_mfma.hip, which it supersedes for these layers
// (TFL_CONV_PATH=mfma brings the MFMA kernels back for comparison).
//
// Why not MFMA here: on gfx950 the f32-input MFMA has NO rate advantage -- it executes at the fp32 vector rate on the
// same pipe (MI355X_MICROARCH.md; profiles/r02_ubench_mfma_valu.txt) -- while its narrowest tile is 16 outputs wide and
// this net has 8 output channels: the x-phase packing of conv_mfma.hip wastes 25% of the issued MACs and every staging /
// epilogue VALU instruction comes out of the same budget. On the vector ALUs the weights are SCALAR operands (uniform
// s_load, one v_pk_fma_f32 per two FMAs), the data comes from an LDS halo tile, and exactly the useful FMAs are issued.
//
// Winograd along x: a lane owns an x-PAIR; from the four halo values d0..d3 of a row it forms
// V = (d0 - d2, d1 + d2, d2 - d1, d1 - d3) and accumulates M_p += V_p * U_p with the host-transformed weights
// U = (g0, (g0 + g1 + g2)/2, (g0 - g1 + g2)/2, g2); at the end y0 = M0 + M1 + M2, y1 = M1 - M2 - M3. 4 multiplies per 2
// outputs instead of 6: 18 FMAs per voxel and (c_in, c_out) instead of 27. Rounding differs from a direct fmaf chain in
// the last bits (rel-L2 6e-7 on the net's output against the generic direct kernels, tests/test_hip_simulate.py).
//
// Block = 256 threads = 4 waves, tile 64(x) x 2(y) x 4(z): wave w owns z-plane w; lane = (pair 0..31, row 0..1); a lane
// keeps 4 Winograd points x 8 output channels = 32 accumulators. Input channels are staged 4 at a time as channel-planar
// halo planes [4][6][4][66(+2)] = 26 KB of LDS; per (channel, dz) a lane issues 3 ds_read2_b64 for 96 FMAs. 96 VGPRs ->
// 5 blocks per CU.
// Staging loads of a stage are ALL issued before the first LDS write (written load->store per element, hipcc waits for
// every load in turn -- the mistake that sank two LDS advection kernels, profiles/r02_advect_experiments.txt).
// Activations between layers are channel-PLANAR [8][Z][Y][X]: a wave's staging load of a (channel, row) is one
// contiguous 256 B segment and its epilogue store of a channel is 8 B per lane, contiguous across the half-wave. The
// channel-last form this replaced ([Z][Y][X][8]: 16 B per lane at a 32 B or 64 B stride) touched 4x the cache lines per
// instruction in the texture addresser, and the store burst at the end of every block cost 12 us per layer at 128^3.
// The first layer builds {pDiv/scale, div/scale, occupancy} on the fly while staging.
#include "tfl_device.hpp"
#include "tfl_fastmath.hpp"
#include "tfl_host.hpp"

#include <cstdio>
#include <cstdlib>

namespace tfl {

namespace {

#ifndef TFL_WINO_VY
#define TFL_WINO_VY 1
#endif
#ifndef TFL_VALU_LB
#define TFL_VALU_LB 4
#endif
constexpr int kVX = 64, kVY = TFL_WINO_VY, kVZ = 4;       // rows per lane; block tile = kVX x 2 kVY x kVZ voxels
constexpr int kTY = 2 * kVY;
constexpr int kPX = kVX + 4;                              // LDS row pitch (66 used)
constexpr int kRowsP = kTY + 2;                           // halo rows per plane
constexpr int kRowsT = (kVZ + 2) * kRowsP;                // halo rows per channel
constexpr int kPlaneF = kRowsT * kPX;                     // floats per staged channel
constexpr int kPerWave = (kRowsT + 3) / 4;                // staged rows per wave

// fused 1x1x1 layers of the TAIL variant: h4 = relu(W4 h + b4); p = w5 . h4 + b5, read through the `bias` pointer, which
// then holds {bias[8], w4[8][8] (out, in), b4[8], w5[8], b5[1]} (one pointer instead of five: the four extra ones cost
// the kernel its 97th VGPR through SGPR spills, and with it the fifth block per CU)
constexpr int kTailW4 = 8, kTailB4 = 72, kTailW5 = 80, kTailB5 = 88;
struct VIn {            // fused network input (first layer): {pDiv/scale, div/scale, occupancy(flags)}
  const float* pDiv;    // [B][1][Z][Y][X]
  const float* div;
  const float* flags;
  const double* stats;  // [B][2] = sum u, sum u^2 (model.hip)
  double count;
};

}  // namespace

// CIN: 3 (first layer: inputs built from pDiv / div / flags) or 8 (channel-planar activations).
// TAIL: fuse the two 1x1x1 layers and write the pressure instead of 8 activation planes.
// wq: [dz][dy][CIN][p = 0..3][8] (tfl_model::wino): the 32 weights of a (dz, dy, c) are contiguous, two s_load_dwordx16.
template <int CIN, bool TAIL>
__global__ __launch_bounds__(256, TFL_VALU_LB) void k_conv3_wino(Dom d, int tiles_x, int tiles_y, int tiles_z, int n_tiles,
                                                                 const float* __restrict__ in, const float* __restrict__ wq,
                                                                 const float* __restrict__ bias, float* __restrict__ out, VIn cin) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr bool FIRST = CIN == 3;
  constexpr int CG = FIRST ? 3 : 4;                        // channels staged per pass
  // XCD-aware tile order (as conv_mfma.hip): consecutive block ids go round-robin over the 8 XCDs
  const int per_xcd = (n_tiles + 7) / 8;
  const int tile = (int)(blockIdx.x % 8) * per_xcd + (int)(blockIdx.x / 8);
  if (tile >= n_tiles) return;
  int t = tile;
  const int tx = t % tiles_x; t /= tiles_x;
  const int ty = t % tiles_y; t /= tiles_y;
  const int tz = t % tiles_z;
  const int b = t / tiles_z;
  // z-window (tfl_device.hpp Dom): the z-tiles cover the plane run [w0, w0 + n0) and then [w1, w1 + nw - n0)
  const int tz_a = (d.n0 + kVZ - 1) / kVZ;
  const int z0 = tz < tz_a ? d.w0 + tz * kVZ : d.w1 + (tz - tz_a) * kVZ;
  const int z_end = tz < tz_a ? d.w0 + d.n0 : d.w1 + (d.nw - d.n0);
  const int x0 = tx * kVX, y0 = ty * kTY;
  const long long cells = d.sc;
  // the wave index is uniform across a wave, but only readfirstlane tells the compiler: with it the row geometry of the
  // staging (row -> z, y, clamps, in-grid tests, LDS row offsets) is scalar-ALU work instead of ~300 VALU instructions
  // per tile taken out of the FMA budget
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int px = lane & 31, ly = lane >> 5;

  float in_scale = 1.0f;
  if (FIRST) {  // lib/modules/variance.lua:44-76 (n-1) + Sqrt, as model.hip scale_from_stats
    const double s1 = cin.stats[b * 2], s2 = cin.stats[b * 2 + 1], n = cin.count;
    in_scale = (float)sqrt(fmax(n * s2 - s1 * s1, 0.0) / (n * (n - 1.0)));
  } else {
    in += (long long)b * cells * CIN;
  }

  const bool scale_in_range = FIRST && in_scale >= 0x1p-12f && in_scale <= 0x1p21f;     // wave-uniform
  const float inv_scale = scale_in_range ? rcp_refined(in_scale) : 0.0f;
  float acc[kVY][4][8];
#pragma unroll
  for (int v = 0; v < kVY; v++)
#pragma unroll
    for (int p = 0; p < 4; p++)
#pragma unroll
      for (int co = 0; co < 8; co++) acc[v][p][co] = 0.0f;

  // ---- staging geometry: wave `wave` stages halo rows r = wave + 4 t (t < kPerWave); a lane loads column x0-1+lane of
  // each, and lanes 2q+e also the two remaining columns (x0+63, x0+64) of row wave + 4 q -----------------------------------
  const int gx = x0 - 1 + lane, gxc = min(max(gx, 0), d.X - 1);
  const bool gx_ok = gx >= 0 && gx < d.X;
  const int eq = lane >> 1, ee = lane & 1;
  const int ex = x0 + 63 + ee, exc = min(max(ex, 0), d.X - 1);
  const bool e_live = eq < kPerWave, ex_ok = ex < d.X;
  auto row_zy = [&](int r, int& gz, int& gy) { const int zz = r / kRowsP; gz = z0 - 1 + zz; gy = y0 - 1 + (r - zz * kRowsP); };

#pragma unroll 1
  for (int cg = 0; cg < CIN; cg += CG) {
    if (cg > 0) __syncthreads();   // everyone is done reading the previous channel group
    {
      // all loads of the stage first ...
      float ld[kPerWave + 1][CG];
#pragma unroll
      for (int tt = 0; tt <= kPerWave; tt++) {
        const bool edge = tt == kPerWave;
        const int r = min(edge ? wave + 4 * (e_live ? eq : 0) : wave + 4 * tt, kRowsT - 1);
        int gz, gy; row_zy(r, gz, gy);
        const int o = TFL_AT(d, edge ? exc : gxc, min(max(gy, 0), d.Y - 1), min(max(gz, 0), d.Z - 1));
        if (FIRST) {
          const long long bo = (long long)b * cells + o;
          ld[tt][0] = cin.pDiv[bo]; ld[tt][1] = cin.div[bo]; ld[tt][2] = cin.flags[bo];
        } else {
#pragma unroll
          for (int c = 0; c < CG; c++) ld[tt][c] = in[(cg + c) * cells + o];
        }
      }
      // ... then the LDS writes (zero outside the grid = the convolution's zero padding)
#pragma unroll
      for (int tt = 0; tt <= kPerWave; tt++) {
        const bool edge = tt == kPerWave;
        const int r = edge ? wave + 4 * (e_live ? eq : 0) : wave + 4 * tt;      // < kRowsT except in a ragged last pass
        int gz, gy; row_zy(r, gz, gy);
        const bool ok = (edge ? ex_ok : gx_ok) && gy >= 0 && gy < d.Y && gz >= 0 && gz < d.Z;
        float v[CG];
        if (FIRST) {
          // the net input is built here: ApplyScale(true) = CDivTable (apply_scale.lua:24-30), FlagsToOccupancy
          // (generic/tfluids.cu:355-371)
          // x / scale with ONE refined reciprocal per thread and an exact-remainder step per quotient: bit-equal to `/`
          // for a scale in [2^-12, 2^21] (tools/ubench/exact_math.hip, free-ratio run) at 3 instead of 11 instructions;
          // 37 staged values per thread and layer made the two divisions a third of this kernel's vector instructions
          if (scale_in_range) { v[0] = div_by<1>(ld[tt][0], in_scale, inv_scale); v[1] = div_by<1>(ld[tt][1], in_scale, inv_scale); }
          else { v[0] = ld[tt][0] / in_scale; v[1] = ld[tt][1] / in_scale; }
          const int f = (int)ld[tt][2];
          v[2] = (f == kFluid) ? 0.0f : ((f == kObstacle) ? 1.0f : -1.0f);
        } else {
#pragma unroll
          for (int c = 0; c < CG; c++) v[c] = ld[tt][c];
        }
        if ((!edge || e_live) && r < kRowsT) {
          const int col = edge ? 64 + ee : lane;
#pragma unroll
          for (int c = 0; c < CG; c++) lds[c * kPlaneF + r * kPX + col] = ok ? v[c] : 0.0f;
        }
      }
    }
    __syncthreads();
    // ---- F(2,3) over these channels: per (c, dz) the kVY + 2 halo rows of plane (wave + dz), 4 values each -----------
#pragma unroll 1
    for (int cl = 0; cl < CG; cl++) {
#pragma unroll
      for (int dz = 0; dz < 3; dz++) {
        const float* base = lds + cl * kPlaneF + ((wave + dz) * kRowsP + ly * kVY) * kPX + 2 * px;
        float V[kVY + 2][4];
#pragma unroll
        for (int r = 0; r < kVY + 2; r++) {
          const float2 a = *reinterpret_cast<const float2*>(base + r * kPX), c2 = *reinterpret_cast<const float2*>(base + r * kPX + 2);
          V[r][0] = a.x - c2.x; V[r][1] = a.y + c2.x; V[r][2] = c2.x - a.y; V[r][3] = a.y - c2.y;
        }
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
          for (int p = 0; p < 4; p++) {
            const float* wt = wq + ((((dz * 3 + dy) * CIN) + (cg + cl)) * 4 + p) * 8;     // wave-uniform -> s_load
#pragma unroll
            for (int co = 0; co < 8; co++) {
              const float wv = wt[co];
#pragma unroll
              for (int v = 0; v < kVY; v++) acc[v][p][co] = __builtin_fmaf(V[v + dy][p], wv, acc[v][p][co]);
            }
          }
      }
    }
  }

  // ---- epilogue: inverse transform, bias, ReLU (+ the 1x1x1 tail) --------------------------------------------------------
  const int x = x0 + 2 * px, z = z0 + wave;
  if (x >= d.X || z >= z_end) return;
  const bool pair_ok = x + 1 < d.X;
  const bool vec2 = (d.X & 1) == 0;          // x is even: an (x, x+1) pair is 8-byte aligned iff the row pitch is even
  const float b5 = TAIL ? bias[kTailB5] : 0.0f;
  out += (long long)b * cells * (TAIL ? 1 : 8);
#pragma unroll
  for (int v = 0; v < kVY; v++) {
    const int y = y0 + ly * kVY + v;
    if (y >= d.Y) continue;
    const int o = TFL_AT(d, x, y, z);
    float h[2][8];
#pragma unroll
    for (int co = 0; co < 8; co++) {
      h[0][co] = fmaxf(((acc[v][0][co] + acc[v][1][co]) + acc[v][2][co]) + bias[co], 0.0f);
      h[1][co] = fmaxf(((acc[v][1][co] - acc[v][2][co]) - acc[v][3][co]) + bias[co], 0.0f);
    }
    auto put = [&](float* dst, float a0, float a1) {
      if (vec2) { *reinterpret_cast<float2*>(dst) = make_float2(a0, a1); }      // pair_ok holds: X even, x even
      else { dst[0] = a0; if (pair_ok) dst[1] = a1; }
    };
    if (!TAIL) {
#pragma unroll
      for (int co = 0; co < 8; co++) put(out + co * cells + o, h[0][co], h[1][co]);
    } else {
      float p[2];
#pragma unroll
      for (int e = 0; e < 2; e++) {
        p[e] = b5;
#pragma unroll
        for (int j = 0; j < 8; j++) {        // 8 -> 8 (k = 1) + ReLU, then 8 -> 1
          float q = bias[kTailB4 + j];
#pragma unroll
          for (int i = 0; i < 8; i++) q = __builtin_fmaf(bias[kTailW4 + j * 8 + i], h[e][i], q);
          p[e] = __builtin_fmaf(bias[kTailW5 + j], fmaxf(q, 0.0f), p[e]);
        }
      }
      put(out + o, p[0], p[1]);
    }
  }
}

template <int CIN, bool TAIL>
static void launch_wino(hipStream_t st, const Dom& d, int B, const float* in, const float* wq, const float* bias, float* out,
                        VIn cin) {
  const int tx = (d.X + kVX - 1) / kVX, ty = (d.Y + kTY - 1) / kTY;
  const int tz = (d.n0 + kVZ - 1) / kVZ + (d.nw - d.n0 + kVZ - 1) / kVZ;   // z-tiles of the compute window's two plane runs
  const int n_tiles = tx * ty * tz * B;
  const int grid = ((n_tiles + 7) / 8) * 8;
  const size_t lds_bytes = sizeof(float) * (CIN == 3 ? 3 : 4) * kPlaneF;
  static int attr_dev = -1;                    // the attribute is a per-device setting
  int cur_dev = 0; (void)hipGetDevice(&cur_dev);
  if (attr_dev != cur_dev) {
    (void)hipFuncSetAttribute((const void*)k_conv3_wino<CIN, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_dev = cur_dev;
    if (getenv("TFL_DEBUG")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_conv3_wino<CIN, TAIL>, 256, lds_bytes);
      fprintf(stderr, "[tfl] k_conv3_wino<%d,%d>: dynamic LDS %zu B, occupancy %d blocks/CU, grid %d\n", CIN, (int)TAIL, lds_bytes, nb, grid);
    }
  }
  // profiler names kept from the MFMA kernels they replace (bench.py's per-layer flop table is keyed by them)
  TFL_TIMED_EXT(TAIL ? "k_conv3_tail" : (CIN == 3 ? "k_conv3_in" : "k_conv3_mid"), st);
  TFL_LAUNCH_EXT((k_conv3_wino<CIN, TAIL>), grid, 256, lds_bytes, st, d, tx, ty, tz, n_tiles, in, wq, bias, out, cin);
}

// first layer: {pDiv/scale, div/scale, occupancy} built while staging; activations out: channel-planar [B][8][Z][Y][X]
void conv3_valu_first_fused(hipStream_t st, int B, int Z, int Y, int X, const float* pDiv, const float* div, const float* flags,
                            const double* stats, double count, const float* wq, const float* bias, float* out_p8) {
  VIn ci = {pDiv, div, flags, stats, count};
  launch_wino<3, false>(st, make_dom(Z, Y, X), B, pDiv, wq, bias, out_p8, ci);
}
void conv3_valu_mid(hipStream_t st, int B, int Z, int Y, int X, const float* in_p8, const float* wq, const float* bias,
                    float* out_p8) {
  VIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  launch_wino<8, false>(st, make_dom(Z, Y, X), B, in_p8, wq, bias, out_p8, noin);
}
// 8 -> 8 k3 + ReLU, then 8 -> 8 k1 + ReLU, then 8 -> 1 k1; planar pressure out.
void conv3_valu_tail(hipStream_t st, int B, int Z, int Y, int X, const float* in_p8, const float* wq, const float* tail_pack,
                     float* p_out) {
  VIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  launch_wino<8, true>(st, make_dom(Z, Y, X), B, in_p8, wq, tail_pack, p_out, noin);
}

}  // namespace tfl
