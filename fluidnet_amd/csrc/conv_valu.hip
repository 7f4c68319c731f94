// conv_valu.hip -- the 3x3x3, 8-output-channel convolution layers of the 3-D `default` projection net as a DIRECT
// convolution on the vector ALUs (gfx950), exact fp32 (one fmaf chain per output, order c -> dz -> dy -> dx).
//
// Replaces cudnn.VolumetricConvolution forward (torch/lib/model_utils.lua:104-116) for the layers 3->8, 8->8, 8->8 (k=3)
// of lib/model.lua:219-226 with the ReLU fused and, in the last of them, the two trailing 1x1x1 layers (8->8 + ReLU,
// 8->1) evaluated in registers. Same entry points and data layouts as conv_mfma.hip, which it supersedes for these
// layers (TFL_CONV_PATH=mfma brings the MFMA kernels back for comparison).
//
// Why not MFMA here: on gfx950 the f32-input MFMA has NO rate advantage -- it executes at the fp32 vector rate on the
// same pipe (MI355X_MICROARCH.md; profiles/r02_ubench_mfma_valu.txt: MFMA TF + VALU TF stays constant when v_fma are
// mixed into an MFMA stream) -- while its narrowest tile is 16 outputs wide and this net has 8 output channels: the
// x-phase packing of conv_mfma.hip wastes 25% of the issued MACs, every staging / epilogue VALU instruction comes out of
// the same budget, and the layer sits at 0.50 of peak with the pipe 85% busy (r01/r02 PMC). A direct convolution issues
// exactly the useful FMAs: thread = voxel column, weights as SCALAR operands (uniform s_load, one v_pk_fma_f32 per two
// FMAs), data from an LDS halo tile. tools/ubench/valu_conv.hip measured the inner loop at 112-119 useful TFLOP/s
// against 77-81 for the MFMA kernel.
//
// Block = 256 threads = 4 waves, tile 64(x) x 2(y) x 4(z): wave w owns z-plane w; lane = x; each lane keeps 2 rows x 8
// output channels = 16 accumulators (register blocking along y: the 4 halo rows of a (channel, dz) feed 3 dy x 2 rows).
// Input channels are staged 4 at a time as channel-planar halo planes [4][6][4][66(+2)] = 26 KB of LDS -> 6 blocks per
// CU; per (channel, dz) a lane issues 12 ds_read_b32 (conflict-free: lanes read consecutive words) for 144 FMAs.
// Tile height: 4 rows per lane is 1% faster at 128^3 but 10-25% slower on small and slab-shaped grids (64^3, 24..40 x
// 128^2: half as many tiles to fill 256 CUs x 4+ slots) -- the shapes every rank of a z-slab run works on; 6 and 8 rows
// are slower everywhere (A/B in one session, -DTFL_VALU_VY / -DTFL_VALU_LB).
// Staging loads of a stage are ALL issued before the first LDS write (written load->store per element, hipcc waits for
// every load in turn -- the mistake that sank two LDS advection kernels, profiles/r02_advect_experiments.txt).
// Activations between layers: channel-last [Z][Y][X][8] (a lane's epilogue is two 16-byte stores, a wave writes 2 KB
// contiguous); the first layer builds {pDiv/scale, div/scale, occupancy} on the fly while staging.
#include "tfl_device.hpp"
#include "tfl_host.hpp"

#include <cstdio>
#include <cstdlib>

namespace tfl {

namespace {

#ifndef TFL_VALU_VY
#define TFL_VALU_VY 2
#endif
#ifndef TFL_VALU_LB
#define TFL_VALU_LB 4
#endif
constexpr int kVX = 64, kVY = TFL_VALU_VY, kVZ = 4;       // block tile (voxels); A/B knobs: -DTFL_VALU_VY=.. -DTFL_VALU_LB=..
constexpr int kPX = kVX + 4;                              // LDS row pitch (66 used)
constexpr int kRowsP = kVY + 2;                           // halo rows per plane
constexpr int kRowsT = (kVZ + 2) * kRowsP;                // halo rows per channel (36)
constexpr int kPlaneF = kRowsT * kPX;                     // floats per staged channel
constexpr int kPerWave = (kRowsT + 3) / 4;                // staged rows per wave (9)

struct VTail {          // fused 1x1x1 layers (device pointers): h4 = relu(W4 h + b4); p = w5 . h4 + b5
  const float* w4;      // [8][8]  (out, in)
  const float* b4;      // [8]
  const float* w5;      // [8]
  const float* b5;      // [1]
};
struct VIn {            // fused network input (first layer): {pDiv/scale, div/scale, occupancy(flags)}
  const float* pDiv;    // [B][1][Z][Y][X]
  const float* div;
  const float* flags;
  const double* stats;  // [B][2] = sum u, sum u^2 (model.hip)
  double count;
};

}  // namespace

// CIN: 3 (first layer: inputs built from pDiv / div / flags) or 8 (channel-last activations).
// TAIL: fuse the two 1x1x1 layers and write planar pressure instead of channel-last activations.
// w: [tap = (dz*3+dy)*3+dx][CIN][8] (tfl_layer::w), so the 8 output-channel weights of a (tap, c) are one s_load_dwordx8.
template <int CIN, bool TAIL>
__global__ __launch_bounds__(256, TFL_VALU_LB) void k_conv3_valu(Dom d, int tiles_x, int tiles_y, int tiles_z, int n_tiles,
                                                       const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ bias, float* __restrict__ out, VTail tail,
                                                       VIn cin) {
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
  const int x0 = tx * kVX, y0 = ty * kVY;
  const long long cells = d.sc;
  // the wave index is uniform across a wave, but only readfirstlane tells the compiler: with it the row geometry of the
  // staging (row -> z, y, clamps, in-grid tests, LDS row offsets) is scalar-ALU work instead of ~300 VALU instructions
  // per tile taken out of the FMA budget
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

  float in_scale = 1.0f;
  if (FIRST) {  // lib/modules/variance.lua:44-76 (n-1) + Sqrt, as model.hip scale_from_stats
    const double s1 = cin.stats[b * 2], s2 = cin.stats[b * 2 + 1], n = cin.count;
    in_scale = (float)sqrt(fmax(n * s2 - s1 * s1, 0.0) / (n * (n - 1.0)));
  } else {
    in += (long long)b * cells * CIN;
  }

  float acc[kVY][8];
#pragma unroll
  for (int v = 0; v < kVY; v++)
#pragma unroll
    for (int co = 0; co < 8; co++) acc[v][co] = bias[co];

  // ---- staging geometry: wave `wave` stages halo rows r = wave + 4 t (t < 9); a lane loads column x0-1+lane of each,
  // and lanes 0..17 also the two remaining columns (x0+63, x0+64) of those rows: lane = 2 q + e -> row wave + 4 q ------
  const int gx = x0 - 1 + lane, gxc = min(max(gx, 0), d.X - 1);
  const bool gx_ok = gx >= 0 && gx < d.X;
  const int eq = lane >> 1, ee = lane & 1;
  const int ex = x0 + 63 + ee, exc = min(max(ex, 0), d.X - 1);
  const bool e_live = eq < kPerWave, ex_ok = ex < d.X;
  auto row_zy = [&](int r, int& gz, int& gy) { const int zz = r / kRowsP; gz = z0 - 1 + zz; gy = y0 - 1 + (r - zz * kRowsP); };

#pragma unroll
  for (int cg = 0; cg < CIN; cg += CG) {
    if (cg > 0) __syncthreads();   // everyone is done reading the previous channel group
    {
      // all loads of the stage first ...
      float4 ld[kPerWave + 1];
      float f2[FIRST ? kPerWave + 1 : 1], f3[FIRST ? kPerWave + 1 : 1];
#pragma unroll
      for (int tt = 0; tt <= kPerWave; tt++) {
        const bool edge = tt == kPerWave;
        const int r = min(edge ? wave + 4 * (e_live ? eq : 0) : wave + 4 * tt, kRowsT - 1);
        int gz, gy; row_zy(r, gz, gy);
        const long long o = TFL_AT(d, edge ? exc : gxc, min(max(gy, 0), d.Y - 1), min(max(gz, 0), d.Z - 1));
        if (FIRST) {
          const long long bo = (long long)b * cells + o;
          ld[tt].x = cin.pDiv[bo]; f2[tt] = cin.div[bo]; f3[tt] = cin.flags[bo];
        } else {
          ld[tt] = *reinterpret_cast<const float4*>(in + o * CIN + cg);
        }
      }
      // ... then the LDS writes (zero outside the grid = the convolution's zero padding)
#pragma unroll
      for (int tt = 0; tt <= kPerWave; tt++) {
        const bool edge = tt == kPerWave;
        const int r = edge ? wave + 4 * (e_live ? eq : 0) : wave + 4 * tt;      // < kRowsT except in a ragged last pass
        int gz, gy; row_zy(r, gz, gy);
        const bool ok = (edge ? ex_ok : gx_ok) && gy >= 0 && gy < d.Y && gz >= 0 && gz < d.Z;
        float v[4];
        if (FIRST) {
          // the net input is built here: ApplyScale(true) = CDivTable (apply_scale.lua:24-30), FlagsToOccupancy
          // (generic/tfluids.cu:355-371)
          v[0] = ld[tt].x / in_scale; v[1] = f2[tt] / in_scale;
          const int f = (int)f3[tt];
          v[2] = (f == kFluid) ? 0.0f : ((f == kObstacle) ? 1.0f : -1.0f);
          v[3] = 0.0f;
        } else {
          v[0] = ld[tt].x; v[1] = ld[tt].y; v[2] = ld[tt].z; v[3] = ld[tt].w;
        }
        if ((!edge || e_live) && r < kRowsT) {
          const int col = edge ? 64 + ee : lane;
#pragma unroll
          for (int c = 0; c < CG; c++) lds[c * kPlaneF + r * kPX + col] = ok ? v[c] : 0.0f;
        }
      }
    }
    __syncthreads();
    // ---- direct convolution over these channels: per (c, dz) the 6 halo rows x 3 dx of plane (wave + dz) --------------
#pragma unroll 1
    for (int cl = 0; cl < CG; cl++) {
#pragma unroll
      for (int dz = 0; dz < 3; dz++) {
        const float* base = lds + cl * kPlaneF + ((wave + dz) * kRowsP) * kPX + lane;
        float row[kRowsP][3];
#pragma unroll
        for (int r = 0; r < kRowsP; r++)
#pragma unroll
          for (int dx = 0; dx < 3; dx++) row[r][dx] = base[r * kPX + dx];
#pragma unroll
        for (int dy = 0; dy < 3; dy++)
#pragma unroll
          for (int dx = 0; dx < 3; dx++) {
            const float* wt = w + (((dz * 3 + dy) * 3 + dx) * CIN + (cg + cl)) * 8;     // wave-uniform -> s_load
#pragma unroll
            for (int co = 0; co < 8; co++) {
              const float wv = wt[co];
#pragma unroll
              for (int v = 0; v < kVY; v++) acc[v][co] = __builtin_fmaf(row[v + dy][dx], wv, acc[v][co]);
            }
          }
      }
    }
  }

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  const int x = x0 + lane, z = z0 + wave;
  if (x >= d.X || z >= z_end) return;
  if (!TAIL) {
    out += (long long)b * cells * 8;
#pragma unroll
    for (int v = 0; v < kVY; v++) {
      const int y = y0 + v;
      if (y < d.Y) {
        float* o = out + (long long)TFL_AT(d, x, y, z) * 8;
        *reinterpret_cast<float4*>(o) = make_float4(fmaxf(acc[v][0], 0.0f), fmaxf(acc[v][1], 0.0f), fmaxf(acc[v][2], 0.0f), fmaxf(acc[v][3], 0.0f));
        *reinterpret_cast<float4*>(o + 4) = make_float4(fmaxf(acc[v][4], 0.0f), fmaxf(acc[v][5], 0.0f), fmaxf(acc[v][6], 0.0f), fmaxf(acc[v][7], 0.0f));
      }
    }
  } else {
    out += (long long)b * cells;
    const float b5 = tail.b5[0];
#pragma unroll
    for (int v = 0; v < kVY; v++) {
      float h[8];
#pragma unroll
      for (int i = 0; i < 8; i++) h[i] = fmaxf(acc[v][i], 0.0f);
      float p = b5;
#pragma unroll
      for (int j = 0; j < 8; j++) {        // 8 -> 8 (k = 1) + ReLU, then 8 -> 1
        float e = tail.b4[j];
#pragma unroll
        for (int i = 0; i < 8; i++) e = __builtin_fmaf(tail.w4[j * 8 + i], h[i], e);
        p = __builtin_fmaf(tail.w5[j], fmaxf(e, 0.0f), p);
      }
      const int y = y0 + v;
      if (y < d.Y) out[TFL_AT(d, x, y, z)] = p;
    }
  }
}

template <int CIN, bool TAIL>
static void launch_valu(hipStream_t st, const Dom& d, int B, const float* in, const float* w, const float* bias, float* out,
                        VTail tail, VIn cin) {
  const int tx = (d.X + kVX - 1) / kVX, ty = (d.Y + kVY - 1) / kVY;
  const int tz = (d.n0 + kVZ - 1) / kVZ + (d.nw - d.n0 + kVZ - 1) / kVZ;   // z-tiles of the compute window's two plane runs
  const int n_tiles = tx * ty * tz * B;
  const int grid = ((n_tiles + 7) / 8) * 8;
  const size_t lds_bytes = sizeof(float) * (CIN == 3 ? 3 : 4) * kPlaneF;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_conv3_valu<CIN, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    attr_set = true;
    if (getenv("TFL_DEBUG")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_conv3_valu<CIN, TAIL>, 256, lds_bytes);
      fprintf(stderr, "[tfl] k_conv3_valu<%d,%d>: dynamic LDS %zu B, occupancy %d blocks/CU, grid %d\n", CIN, (int)TAIL, lds_bytes, nb, grid);
    }
  }
  // profiler names kept from the MFMA kernels they replace (bench.py's per-layer flop table is keyed by them)
  TFL_TIMED_EXT(TAIL ? "k_conv3_tail" : (CIN == 3 ? "k_conv3_in" : "k_conv3_mid"), st);
  TFL_LAUNCH_EXT((k_conv3_valu<CIN, TAIL>), grid, 256, lds_bytes, st, d, tx, ty, tz, n_tiles, in, w, bias, out, tail, cin);
}

// first layer: {pDiv/scale, div/scale, occupancy} built while staging; w = tfl_layer::w ([tap][3][8])
void conv3_valu_first_fused(hipStream_t st, int B, int Z, int Y, int X, const float* pDiv, const float* div, const float* flags,
                            const double* stats, double count, const float* w, const float* bias, float* out_cl8) {
  VTail none = {nullptr, nullptr, nullptr, nullptr};
  VIn ci = {pDiv, div, flags, stats, count};
  launch_valu<3, false>(st, make_dom(Z, Y, X), B, pDiv, w, bias, out_cl8, none, ci);
}
void conv3_valu_mid(hipStream_t st, int B, int Z, int Y, int X, const float* in_cl8, const float* w, const float* bias,
                    float* out_cl8) {
  VTail none = {nullptr, nullptr, nullptr, nullptr};
  VIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  launch_valu<8, false>(st, make_dom(Z, Y, X), B, in_cl8, w, bias, out_cl8, none, noin);
}
// 8 -> 8 k3 + ReLU, then 8 -> 8 k1 + ReLU, then 8 -> 1 k1; planar pressure out.
void conv3_valu_tail(hipStream_t st, int B, int Z, int Y, int X, const float* in_cl8, const float* w, const float* bias,
                     const float* w4, const float* b4, const float* w5, const float* b5, float* p_out) {
  VTail tail = {w4, b4, w5, b5};
  VIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  launch_valu<8, true>(st, make_dom(Z, Y, X), B, in_cl8, w, bias, p_out, tail, noin);
}

}  // namespace tfl
