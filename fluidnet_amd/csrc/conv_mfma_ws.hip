// conv_mfma_ws.hip -- the 3x3x3 / 8-channel layers of the 3-D `default` projection net, WAVE-SPECIALISED.
//
// Same implicit GEMM as conv_mfma.hip (x-phase packing: one 16x16x4 fp32 MFMA = one (c, dz, dy) of the stencil
// for 32 x-voxels x 8 channels; see that file), different kernel structure. Measured on conv_mfma.hip (r01 phase
// ablation, mid layer): 98.6 us = 64 us MFMA (already the ideal issue rate) + 20 us staging + 17 us epilogue --
// the phases ADD, because a wave executes in order and the co-resident blocks of a CU run in lockstep. Here the
// non-matrix work gets its own waves:
//
//   block = 512 threads = 8 waves, persistent (one block per CU, walks its tiles)
//     waves 0-3  CONSUMERS : ds_read A rows + MFMA only (wave w owns z-plane w of the 32x8x4 tile, 8 accumulators)
//     waves 4-7  PRODUCERS : global loads of the NEXT stage -> registers -> LDS, and the PREVIOUS tile's output
//                            LDS -> global (one coalesced 1 KB row per wave instruction); in the first layer
//                            they also build the network input {pDiv/scale, div/scale, occupancy}, in the
//                            last layer they evaluate the two fused 1x1x1 layers (per-voxel FMAs, no shuffles)
//   LDS: input planes double-buffered (2 x 4 ch x 6 x 10 x 36 fp32) + transposed output tile double-buffered
//        (2 x 32 KB) = 131.5 KB; one s_barrier per stage (stage = one tile x 4 input channels = 288 MFMAs/wave).
// Every SIMD hosts one consumer and one producer wave, so matrix and memory/VALU/LDS pipes run concurrently.
//
// STATUS (r01, measured at 128^3 on MI355X): correct (same results as conv_mfma.hip to rounding) but SLOWER --
// mid layer 109-112 us vs 96 us. Ablation (TFL_CONV_DEBUG): with all global traffic removed the kernel still takes
// 85 us (ideal MFMA time 62-65 us): ONE MFMA-issuing wave per SIMD sustains only ~80 % of the issue rate even with
// the A rows prefetched a step ahead, and sharing the LDS pipe with the producers' ds_writes costs more; the staging
// loads add another 19 us because a one-stage prefetch distance (4.9 us) is shorter than load latency + commit under
// a 3 TB/s chip-wide load. Opt-in via TFL_CONV_PATH=mfma_ws, kept as the base for a deeper-pipelined version.
#include "tfl_device.hpp"
#include "tfl_host.hpp"

#include <cstdio>
#include <cstdlib>

namespace tfl {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace ws {
constexpr int kTX = 32, kTY = 8, kTZ = 4, kLX = 36;
constexpr int kRows = (kTZ + 2) * (kTY + 2);   // 60 halo rows per channel plane
constexpr int kPlane = kRows * kLX;            // floats per channel plane in LDS
constexpr int kOut = kTZ * kTY * 256;          // floats of one transposed output tile (32 KB)
constexpr int kNV = (kRows * 34 + 255) / 256;  // halo voxels per producer thread per stage (8)
}  // namespace ws

struct WsTail { const float* w4; const float* b4; const float* w5; const float* b5; };
struct WsIn { const float* pDiv; const float* div; const float* flags; const double* stats; double count; };

template <int CIN, bool FUSED_IN, bool TAIL>
__global__ __launch_bounds__(512, 2) void k_conv3_ws(Dom d, int tiles_x, int tiles_y, int tiles_z, int n_tiles,
                                                     const float* __restrict__ in, const float* __restrict__ bfrag,
                                                     const float* __restrict__ bias, float* __restrict__ out,
                                                     WsTail tail, WsIn cin, int dbg) {
  // dbg (env TFL_CONV_DEBUG; timing experiments only, results are garbage): 1 = no MFMAs, 2 = no staging loads,
  // 4 = no output stores
  using namespace ws;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int CG = CIN < 4 ? CIN : 4;   // input channels per stage
  constexpr int NCG = CIN / CG;           // stages per tile
  constexpr int INBUF = CG * kPlane;
  float* const inbuf = lds;               // [2][INBUF]
  float* const outbuf = lds + 2 * INBUF;  // [2][kOut]
  const int tid = threadIdx.x;
  const bool consumer = tid < 256;
  const long long cells = d.sc;

  // ---- this block's tiles: t = bid + it * G, mapped so that a block stays inside one XCD's contiguous run ----
  const int per_xcd = (n_tiles + 7) / 8;
  const int G = gridDim.x;
  struct Tile { int b, x0, y0, z0; };
  auto tile_valid = [&](int t) { return t / 8 < per_xcd && (t % 8) * per_xcd + t / 8 < n_tiles; };
  auto tile_of = [&](int t) {
    int r = (t % 8) * per_xcd + t / 8;
    Tile T;
    const int tx = r % tiles_x; r /= tiles_x;
    const int ty = r % tiles_y; r /= tiles_y;
    const int tz = r % tiles_z;
    T.b = r / tiles_z; T.x0 = tx * kTX; T.y0 = ty * kTY; T.z0 = tz * kTZ;
    return T;
  };
  int my_tiles = 0;
  while (tile_valid((int)blockIdx.x + my_tiles * G)) my_tiles++;
  if (my_tiles == 0) return;
  const int S = my_tiles * NCG;

  if (consumer) {
    // =========================================== CONSUMERS ===========================================
    const int lane = tid & 63, wave = tid >> 6;
    float bf[CIN * 9];
#pragma unroll
    for (int q = 0; q < CIN * 9; q++) bf[q] = bfrag[q * 64 + lane];
    const int co = lane & 7, g = lane >> 4, ph = (lane >> 3) & 1;
    const float bv = bias[co];
    const int lane_off = 2 * (lane & 15) + (lane >> 4);
    f32x4 acc[kTY];
    __syncthreads();   // stage 0 is in LDS
    for (int s = 0; s < S; s++) {
      const int cgi = s % NCG;
      if (cgi == 0) {
#pragma unroll
        for (int r = 0; r < kTY; r++) acc[r] = (f32x4){bv, bv, bv, bv};
      }
      const float* ib = inbuf + (s & 1) * INBUF;
      // one step = one (c, dz): 10 halo rows of LDS plane (wave + dz) feed 3 (dy) x 8 (rows) = 24 MFMAs;
      // consecutive uses of one accumulator are 8 MFMAs apart (> the 40-cycle dependent latency). This wave is
      // the only MFMA issuer of its SIMD, so LDS latency must be hidden inside the wave: the rows of step n+1
      // are fetched BEFORE the MFMAs of step n (sched_barrier keeps hipcc from sinking the reads to their uses).
      float a_cur[kTY + 2], a_nxt[kTY + 2];
      {
        const float* base = ib + (wave * (kTY + 2)) * kLX + lane_off;
#pragma unroll
        for (int q = 0; q < kTY + 2; q++) a_cur[q] = base[q * kLX];
      }
      if (!(dbg & 1))
#pragma unroll
      for (int step = 0; step < CG * 3; step++) {
        if (step + 1 < CG * 3) {
          const int cl = (step + 1) / 3, dz = (step + 1) % 3;
          const float* base = ib + cl * kPlane + ((wave + dz) * (kTY + 2)) * kLX + lane_off;
#pragma unroll
          for (int q = 0; q < kTY + 2; q++) a_nxt[q] = base[q * kLX];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dy = 0; dy < 3; dy++) {
          float bval;
          if (NCG == 1) bval = bf[step * 3 + dy];
          else bval = cgi == 0 ? bf[step * 3 + dy] : bf[(CG * 3 + step) * 3 + dy];
#pragma unroll
          for (int r = 0; r < kTY; r++) acc[r] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_cur[r + dy], bval, acc[r], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < kTY + 2; q++) a_cur[q] = a_nxt[q];
      }
      if (cgi == NCG - 1) {
        // ReLU + transpose into the output tile: [z = wave][row r][voxel 8g + 2i + ph][channel co]; the D layout
        // (lane = one channel of 4 voxels) becomes 1 KB-contiguous channel-last rows for the producers
        float* ob = outbuf + ((s / NCG) & 1) * kOut + wave * (kTY * 256);
#pragma unroll
        for (int r = 0; r < kTY; r++)
#pragma unroll
          for (int i = 0; i < 4; i++) ob[r * 256 + (8 * g + 2 * i + ph) * 8 + co] = fmaxf(acc[r][i], 0.0f);
      }
      __syncthreads();
    }
    return;
  }

  // ============================================= PRODUCERS =============================================
  const int ptid = tid - 256;
  // geometry of this thread's halo-voxel slots: tile-relative, computed once
  int slot_xyz[kNV], slot_goff[kNV], slot_loff[kNV];
#pragma unroll
  for (int q = 0; q < kNV; q++) {
    const int idx = ptid + q * 256;
    const int xx = idx % 34, row = idx / 34;
    const int yy = row % (kTY + 2), zz = row / (kTY + 2);
    slot_xyz[q] = idx < kRows * 34 ? (xx | (yy << 8) | (zz << 16)) : -1;
    slot_goff[q] = (xx - 1) + (yy - 1) * d.sy + (zz - 1) * d.sz;
    slot_loff[q] = row * kLX + xx;
  }
  float pre[kNV][CG];
  auto load_stage = [&](int s) {   // global -> registers
    const Tile T = tile_of((int)blockIdx.x + (s / NCG) * G);
    const int cg = (s % NCG) * CG;
    const long long tile_o = TFL_AT(d, T.x0, T.y0, T.z0);
    const bool interior = T.x0 >= 1 && T.x0 + kTX + 1 <= d.X && T.y0 >= 1 && T.y0 + kTY + 1 <= d.Y && T.z0 >= 1 &&
                          T.z0 + kTZ + 1 <= d.Z;
#pragma unroll
    for (int q = 0; q < kNV; q++) {
      const int xyz = slot_xyz[q];
      bool ok = xyz >= 0 && !(dbg & 2);
      if (!interior) {
        const int gx = T.x0 - 1 + (xyz & 255), gy = T.y0 - 1 + ((xyz >> 8) & 255), gz = T.z0 - 1 + (xyz >> 16);
        ok = ok && gx >= 0 && gx < d.X && gy >= 0 && gy < d.Y && gz >= 0 && gz < d.Z;
      }
#pragma unroll
      for (int c = 0; c < CG; c++) pre[q][c] = 0.0f;
      if (ok) {
        const long long o = tile_o + slot_goff[q];
        if (FUSED_IN) {
          const long long bo = (long long)T.b * cells + o;
          pre[q][0] = cin.pDiv[bo]; pre[q][1] = cin.div[bo]; pre[q][2] = cin.flags[bo];
        } else {
          const float4 f = *reinterpret_cast<const float4*>(in + ((long long)T.b * cells + o) * CIN + cg);
          pre[q][0] = f.x; pre[q][1] = f.y; pre[q][2] = f.z; pre[q][CG - 1] = f.w;
        }
      } else if (FUSED_IN) {
        pre[q][2] = -12345.0f;   // marker: outside the domain -> zero padding (not an occupancy value)
      }
    }
  };
  auto commit_stage = [&](int s) {   // registers -> LDS (+ the fused network-input transform)
    float* ib = inbuf + (s & 1) * INBUF;
    float in_scale = 1.0f;
    if (FUSED_IN) {   // lib/modules/variance.lua:44-76 (n-1) + Sqrt, as model.hip scale_from_stats
      const int b = tile_of((int)blockIdx.x + (s / NCG) * G).b;
      const double s1 = cin.stats[b * 2], s2 = cin.stats[b * 2 + 1], n = cin.count;
      in_scale = (float)sqrt((n * s2 - s1 * s1) / (n * (n - 1.0)));
    }
#pragma unroll
    for (int q = 0; q < kNV; q++) {
      if (slot_xyz[q] < 0) continue;
      float v[CG];
#pragma unroll
      for (int c = 0; c < CG; c++) v[c] = pre[q][c];
      if (FUSED_IN) {
        if (pre[q][2] == -12345.0f) {
          v[0] = v[1] = v[2] = 0.0f;
        } else {
          // ApplyScale(true) = CDivTable (apply_scale.lua:24-30); FlagsToOccupancy (generic/tfluids.cu:355-371)
          v[0] = pre[q][0] / in_scale;
          v[1] = pre[q][1] / in_scale;
          const int f = (int)pre[q][2];
          v[2] = (f == kFluid) ? 0.0f : ((f == kObstacle) ? 1.0f : -1.0f);
        }
      }
#pragma unroll
      for (int c = 0; c < CG; c++) ib[c * kPlane + slot_loff[q]] = v[c];
    }
  };
  auto store_tile = [&](int it) {   // transposed output tile (LDS) -> global
    if (dbg & 4) return;
    const Tile T = tile_of((int)blockIdx.x + it * G);
    const float* ob = outbuf + (it & 1) * kOut;
    if (!TAIL) {
      float* op = out + (long long)T.b * cells * 8;
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const int f = ptid + q * 256;          // float4 index within the tile: one wave instruction = one row
        const int row = f >> 6, within = f & 63;
        const int z = T.z0 + (row >> 3), y = T.y0 + (row & 7), x = T.x0 + (within >> 1);
        const float4 v4 = *reinterpret_cast<const float4*>(ob + row * 256 + within * 4);
        if (x < d.X && y < d.Y && z < d.Z)
          *reinterpret_cast<float4*>(op + ((long long)TFL_AT(d, T.x0, y, z)) * 8 + within * 4) = v4;
      }
    } else {
      // the two fused 1x1x1 layers, per voxel in registers: h4 = relu(W4 h3 + b4); p = w5 . h4 + b5
      float* op = out + (long long)T.b * cells;
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int vi = ptid + q * 256;         // voxel index within the tile
        const int row = vi >> 5, v = vi & 31;
        const float4 h0 = *reinterpret_cast<const float4*>(ob + row * 256 + v * 8);
        const float4 h1 = *reinterpret_cast<const float4*>(ob + row * 256 + v * 8 + 4);
        const float h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        float p = tail.b5[0];
#pragma unroll
        for (int c4 = 0; c4 < 8; c4++) {
          float h4 = tail.b4[c4];
#pragma unroll
          for (int c = 0; c < 8; c++) h4 = fmaf(tail.w4[c4 * 8 + c], h[c], h4);
          p = fmaf(tail.w5[c4], fmaxf(h4, 0.0f), p);
        }
        const int z = T.z0 + (row >> 3), y = T.y0 + (row & 7), x = T.x0 + v;
        if (x < d.X && y < d.Y && z < d.Z) op[TFL_AT(d, x, y, z)] = p;
      }
    }
  };

  load_stage(0);
  commit_stage(0);
  __syncthreads();   // stage 0 is in LDS
  for (int s = 0; s < S; s++) {
    if (s + 1 < S) load_stage(s + 1);                                        // loads in flight ...
    if (s >= 1 && (s - 1) % NCG == NCG - 1) store_tile((s - 1) / NCG);       // ... while the previous tile leaves
    if (s + 1 < S) commit_stage(s + 1);
    __syncthreads();
  }
  store_tile(my_tiles - 1);
}

template <int CIN, bool FUSED_IN, bool TAIL>
static void launch_ws(hipStream_t st, const Dom& d, int B, const float* in, const float* bfrag, const float* bias,
                      float* out, WsTail tail, WsIn cin) {
  using namespace ws;
  const int tx = (d.X + kTX - 1) / kTX, ty = (d.Y + kTY - 1) / kTY, tz = (d.Z + kTZ - 1) / kTZ;
  const int n_tiles = tx * ty * tz * B;
  static int num_cu = 0;
  if (!num_cu) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) num_cu = prop.multiProcessorCount;
    if (num_cu <= 0) num_cu = 256;
  }
  int grid = ((n_tiles + 7) / 8) * 8;           // a multiple of 8: a block stays on one XCD's tile run
  const int cap = ((num_cu + 7) / 8) * 8;       // persistent: one block per CU
  if (grid > cap) grid = cap;
  const size_t lds_bytes = sizeof(float) * (2 * (CIN < 4 ? CIN : 4) * kPlane + 2 * kOut);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)k_conv3_ws<CIN, FUSED_IN, TAIL>, hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds_bytes);
    attr_set = true;
    if (getenv("TFL_DEBUG")) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, (const void*)k_conv3_ws<CIN, FUSED_IN, TAIL>, 512, lds_bytes);
      fprintf(stderr, "[tfl] k_conv3_ws<%d,%d,%d>: dynamic LDS %zu B, occupancy %d blocks/CU, grid %d, tiles %d\n", CIN,
              (int)FUSED_IN, (int)TAIL, lds_bytes, nb, grid, n_tiles);
    }
  }
  TFL_TIMED(TAIL ? "k_conv3_mfma_tail" : (FUSED_IN ? "k_conv3_mfma_in" : "k_conv3_mfma"), st);
  static const int dbg = getenv("TFL_CONV_DEBUG") ? atoi(getenv("TFL_CONV_DEBUG")) : 0;
  k_conv3_ws<CIN, FUSED_IN, TAIL><<<grid, 512, lds_bytes, st>>>(d, tx, ty, tz, n_tiles, in, bfrag, bias, out, tail, cin, dbg);
}

void conv3_ws_first_fused(hipStream_t st, int B, int Z, int Y, int X, const float* pDiv, const float* div,
                          const float* flags, const double* stats, double count, const float* bfrag, const float* bias,
                          float* out_cl8) {
  WsTail none = {nullptr, nullptr, nullptr, nullptr};
  WsIn ci = {pDiv, div, flags, stats, count};
  launch_ws<3, true, false>(st, make_dom(Z, Y, X), B, nullptr, bfrag, bias, out_cl8, none, ci);
}
void conv3_ws_mid(hipStream_t st, int B, int Z, int Y, int X, const float* in_cl8, const float* bfrag, const float* bias,
                  float* out_cl8) {
  WsTail none = {nullptr, nullptr, nullptr, nullptr};
  WsIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  launch_ws<8, false, false>(st, make_dom(Z, Y, X), B, in_cl8, bfrag, bias, out_cl8, none, noin);
}
void conv3_ws_tail(hipStream_t st, int B, int Z, int Y, int X, const float* in_cl8, const float* bfrag, const float* bias,
                   const float* w4, const float* b4, const float* w5, const float* b5, float* p_out) {
  WsTail tail = {w4, b4, w5, b5};
  WsIn noin = {nullptr, nullptr, nullptr, nullptr, 0.0};
  launch_ws<8, false, true>(st, make_dom(Z, Y, X), B, in_cl8, bfrag, bias, p_out, tail, noin);
}

}  // namespace tfl
