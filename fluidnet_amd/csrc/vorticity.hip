// vorticity.hip -- vorticity confinement (gfx950). Replaces third_party/tfluids.cc:1312-1458 |
// tfluids.cu:1355-1497.
//
// The reference makes four sweeps through four temporaries (centered -> curl,|curl| -> force ->
// AddForceField; ~150 B/cell). Here it is two launches and two temporaries:
//   pass A : U -> curl[3], |curl|      (cell-centred velocities are recomputed from the MAC faces in
//            registers, with the reference's "zero on the border shell" rule applied per tap)
//   pass B : curl, |curl|, flags, U -> U   (the force of the cell and of its three -c neighbours is
//            evaluated in registers and face-averaged; no force grid is stored)
// Algorithmic bytes per cell (3-D): A 7 words + B 11 words = 72 B. HBM-bound.
#include "tfl_device.hpp"
#include "tfl_fastmath.hpp"
#include "tfl_host.hpp"
#include "tfl_vec4.hpp"
#include <type_traits>

#include <atomic>
#include <cstdlib>

// minimum waves per SIMD the register allocator has to leave room for (tools/ab_build.sh -DTFL_LB_...=n for A/B runs)
#ifndef TFL_LB_CURL
#define TFL_LB_CURL 1
#endif
#ifndef TFL_LB_CONFINE
#define TFL_LB_CONFINE 4      // 128 VGPRs without spills (140 unconstrained: a wave less per SIMD)
#endif

namespace tfl {

// centred velocity component AXIS of cell n, 0 on the border shell (tfluids.cc:1372-1386 + grid.cc:346-356)
template <bool IS3D, int AXIS>
__device__ __forceinline__ float centred_c(const Dom& d, const float* __restrict__ U, int i, int j, int k) {
  if (on_border<IS3D>(d, i, j, k)) return 0.0f;
  const int a = TFL_AT(d, i, j, k) + AXIS * d.sc;
  const int step = AXIS == 0 ? 1 : (AXIS == 1 ? d.sy : d.sz);
  return 0.5f * (U[a] + U[a + step]);
}

template <bool IS3D>
__global__ __launch_bounds__(256) void k_curl(Dom d, const float* __restrict__ U, float* __restrict__ curl,
                                              float* __restrict__ cnorm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  if (i >= d.X || j >= d.Y) return;
  const long long cells = d.sc;
  U += b * cells * (IS3D ? 3 : 2); curl += b * cells * 3; cnorm += b * cells;
  const int o = TFL_AT(d, i, j, k);
  v3 w = mk3(0.0f, 0.0f, 0.0f);
  float nrm = 0.0f;
  if (!on_border<IS3D>(d, i, j, k)) {  // VecGrid::curl, grid.cc:497-515
    w.z = 0.5f * ((centred_c<IS3D, 1>(d, U, i + 1, j, k) - centred_c<IS3D, 1>(d, U, i - 1, j, k)) -
                  (centred_c<IS3D, 0>(d, U, i, j + 1, k) - centred_c<IS3D, 0>(d, U, i, j - 1, k)));
    if (IS3D) {
      w.x = 0.5f * ((centred_c<IS3D, 2>(d, U, i, j + 1, k) - centred_c<IS3D, 2>(d, U, i, j - 1, k)) -
                    (centred_c<IS3D, 1>(d, U, i, j, k + 1) - centred_c<IS3D, 1>(d, U, i, j, k - 1)));
      w.y = 0.5f * ((centred_c<IS3D, 0>(d, U, i, j, k + 1) - centred_c<IS3D, 0>(d, U, i, j, k - 1)) -
                    (centred_c<IS3D, 2>(d, U, i + 1, j, k) - centred_c<IS3D, 2>(d, U, i - 1, j, k)));
    }
    nrm = norm3(w);
  }
  curl[o] = w.x; curl[o + d.sc] = w.y; curl[o + 2 * d.sc] = w.z;
  cnorm[o] = nrm;
}

// confinement force of cell n (0 on the border shell), tfluids.cc:1410-1436
template <bool IS3D>
__device__ __forceinline__ v3 force_at(const Dom& d, const float* __restrict__ curl, const float* __restrict__ cn,
                                       float strength, int i, int j, int k) {
  if (on_border<IS3D>(d, i, j, k)) return mk3(0.0f, 0.0f, 0.0f);
  const int o = TFL_AT(d, i, j, k);
  v3 g = mk3(0.5f * (cn[o + 1] - cn[o - 1]), 0.5f * (cn[o + d.sy] - cn[o - d.sy]), 0.0f);
  if (IS3D) g.z = 0.5f * (cn[o + d.sz] - cn[o - d.sz]);
  g = normalize3(g);
  const v3 w = mk3(curl[o], curl[o + d.sc], curl[o + 2 * d.sc]);
  return mk3(((g.y * w.z) - (g.z * w.y)) * strength, ((g.z * w.x) - (g.x * w.z)) * strength,
             ((g.x * w.y) - (g.y * w.x)) * strength);
}

template <bool IS3D>
__global__ __launch_bounds__(256) void k_confine(Dom d, float* __restrict__ U, const float* __restrict__ flags,
                                                 const float* __restrict__ curl, const float* __restrict__ cn,
                                                 float strength) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  if (i >= d.X || j >= d.Y) return;
  if (on_border<IS3D>(d, i, j, k)) return;
  const long long cells = d.sc;
  U += b * cells * (IS3D ? 3 : 2); flags += b * cells; curl += b * cells * 3; cn += b * cells;
  const int o = TFL_AT(d, i, j, k);
  const int fc = (int)flags[o];
  const bool cf = fc & kFluid, ce = fc & kEmpty;
  if (!cf && !ce) return;  // AddForceField, tfluids.cc:1312-1339
  const int nx = (int)flags[o - 1], ny = (int)flags[o - d.sy], nz = IS3D ? (int)flags[o - d.sz] : 0;
  const bool ax = (nx & kFluid) || (cf && (nx & kEmpty));
  const bool ay = (ny & kFluid) || (cf && (ny & kEmpty));
  const bool az = IS3D && ((nz & kFluid) || (cf && (nz & kEmpty)));
  if (!ax && !ay && !az) return;
  const v3 f0 = force_at<IS3D>(d, curl, cn, strength, i, j, k);
  if (ax) U[o] += (0.5f * (force_at<IS3D>(d, curl, cn, strength, i - 1, j, k).x + f0.x));
  if (ay) U[o + d.sc] += (0.5f * (force_at<IS3D>(d, curl, cn, strength, i, j - 1, k).y + f0.y));
  if (az) U[o + 2 * d.sc] += (0.5f * (force_at<IS3D>(d, curl, cn, strength, i, j, k - 1).z + f0.z));
}

// ---- four x-cells per thread (tfl_vec4.hpp): same per-cell arithmetic, 16-byte accesses -----------------
// k_curl_v4: 16 row loads per 4 cells instead of 24 dword loads per cell. Tap (i+-1, j+-1, k+-1) of a
// non-border cell is on the border shell iff its moved coordinate is 0 or N-1 -> contributes 0.
template <bool IS3D>
__global__ __launch_bounds__(256, TFL_LB_CURL) void k_curl_v4(Dom d, const float* __restrict__ U, float* __restrict__ curl,
                                                 float* __restrict__ cnorm, BlockOrder ord) {
  int bx_, by_, bz_; block_tile(ord, bx_, by_, bz_);
  const V4Ctx c = v4_ctx(d, bx_);
  const int j = by_ * blockDim.y + threadIdx.y;
  int b, k; dom_bk_of(d, bz_, b, k);
  const bool live = c.i0 < d.X && j < d.Y;
  const long long cells = d.sc;
  U += b * cells * (IS3D ? 3 : 2); curl += b * cells * 3; cnorm += b * cells;
  const int o = TFL_AT(d, c.i0, j, k);
  const bool in = live && j >= 1 && j <= d.Y - 2 && (!IS3D || (k >= 1 && k <= d.Z - 2));   // not a border row
  const float* Ux = U;
  const float* Uy = U + d.sc;
  const float* Uz = U + 2 * d.sc;   // only touched when IS3D
  const bool ypb = j + 1 == d.Y - 1, ymb = j - 1 == 0, zpb = k + 1 == d.Z - 1, zmb = k - 1 == 0;
  float uy_a[6], uy_b[6], ux_p[6], ux_m[6];
  v4_load6<true, true>(c, Uy, o, in, 0.0f, uy_a);
  v4_load6<true, true>(c, Uy, o + d.sy, in, 0.0f, uy_b);
  v4_load6<false, true>(c, Ux, o + d.sy, in, 0.0f, ux_p);
  v4_load6<false, true>(c, Ux, o - d.sy, in, 0.0f, ux_m);
  float uz_p0[4], uz_p1[4], uz_m0[4], uz_m1[4], uy_zp0[4], uy_zp1[4], uy_zm0[4], uy_zm1[4], ux_zp[6], ux_zm[6], uz_a[6], uz_b[6];
  if (IS3D) {
    v4_load(Uz, o + d.sy, in, 0.0f, uz_p0);
    v4_load(Uz, o + d.sy + d.sz, in, 0.0f, uz_p1);
    v4_load(Uz, o - d.sy, in, 0.0f, uz_m0);
    v4_load(Uz, o - d.sy + d.sz, in, 0.0f, uz_m1);
    v4_load(Uy, o + d.sz, in, 0.0f, uy_zp0);
    v4_load(Uy, o + d.sz + d.sy, in, 0.0f, uy_zp1);
    v4_load(Uy, o - d.sz, in, 0.0f, uy_zm0);
    v4_load(Uy, o - d.sz + d.sy, in, 0.0f, uy_zm1);
    v4_load6<false, true>(c, Ux, o + d.sz, in, 0.0f, ux_zp);
    v4_load6<false, true>(c, Ux, o - d.sz, in, 0.0f, ux_zm);
    v4_load6<true, true>(c, Uz, o, in, 0.0f, uz_a);
    v4_load6<true, true>(c, Uz, o + d.sz, in, 0.0f, uz_b);
  }
  float wx[4], wy[4], wz[4], nr[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = c.i0 + q;
    v3 w = mk3(0.0f, 0.0f, 0.0f);
    float nrm = 0.0f;
    if (in && i >= 1 && i <= d.X - 2) {   // VecGrid::curl, grid.cc:497-515
      const bool xpb = i + 1 == d.X - 1, xmb = i - 1 == 0;
      const float cy_xp = xpb ? 0.0f : 0.5f * (uy_a[q + 2] + uy_b[q + 2]);
      const float cy_xm = xmb ? 0.0f : 0.5f * (uy_a[q] + uy_b[q]);
      const float cx_yp = ypb ? 0.0f : 0.5f * (ux_p[q + 1] + ux_p[q + 2]);
      const float cx_ym = ymb ? 0.0f : 0.5f * (ux_m[q + 1] + ux_m[q + 2]);
      w.z = 0.5f * ((cy_xp - cy_xm) - (cx_yp - cx_ym));
      if (IS3D) {
        const float cz_yp = ypb ? 0.0f : 0.5f * (uz_p0[q] + uz_p1[q]);
        const float cz_ym = ymb ? 0.0f : 0.5f * (uz_m0[q] + uz_m1[q]);
        const float cy_zp = zpb ? 0.0f : 0.5f * (uy_zp0[q] + uy_zp1[q]);
        const float cy_zm = zmb ? 0.0f : 0.5f * (uy_zm0[q] + uy_zm1[q]);
        const float cx_zp = zpb ? 0.0f : 0.5f * (ux_zp[q + 1] + ux_zp[q + 2]);
        const float cx_zm = zmb ? 0.0f : 0.5f * (ux_zm[q + 1] + ux_zm[q + 2]);
        const float cz_xp = xpb ? 0.0f : 0.5f * (uz_a[q + 2] + uz_b[q + 2]);
        const float cz_xm = xmb ? 0.0f : 0.5f * (uz_a[q] + uz_b[q]);
        w.x = 0.5f * ((cz_yp - cz_ym) - (cy_zp - cy_zm));
        w.y = 0.5f * ((cx_zp - cx_zm) - (cz_xp - cz_xm));
      }
      nrm = norm3(w);
    }
    wx[q] = w.x; wy[q] = w.y; wz[q] = w.z; nr[q] = nrm;
  }
  if (live) {
    v4_store(curl, o, wx);
    v4_store(curl, o + d.sc, wy);
    v4_store(curl, o + 2 * d.sc, wz);
    v4_store(cnorm, o, nr);
  }
}

// confinement force of four cells of row (jj,kk) given that row's |curl| (6 wide) and its four neighbour
// rows; cells on the border shell get 0. NEED: which components the caller uses (bit 0 x, 1 y, 2 z).
template <bool IS3D>
__device__ __forceinline__ void force_row(const Dom& d, float strength, int i0, bool row_inner, const float* cn6,
                                          const float* cn_ym, const float* cn_yp, const float* cn_zm, const float* cn_zp,
                                          const float* wx, const float* wy, const float* wz, v3* f) {
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = i0 + q;
    f[q] = mk3(0.0f, 0.0f, 0.0f);
    if (row_inner && i >= 1 && i <= d.X - 2) {
      v3 g = mk3(0.5f * (cn6[q + 2] - cn6[q]), 0.5f * (cn_yp[q] - cn_ym[q]), 0.0f);
      if (IS3D) g.z = 0.5f * (cn_zp[q] - cn_zm[q]);
      g = normalize3(g);
      const v3 w = mk3(wx[q], wy[q], wz[q]);
      f[q] = mk3(((g.y * w.z) - (g.z * w.y)) * strength, ((g.z * w.x) - (g.x * w.z)) * strength,
                 ((g.x * w.y) - (g.y * w.x)) * strength);
    }
  }
}

// k_confine_v4: the force of the thread's own four cells, of the four cells one row down (y-1) and one plane
// down (z-1) are evaluated in registers from 10 |curl| rows + 7 curl rows; force.x of cell i0-1 comes from the
// previous lane. U is rewritten in full rows (unchanged cells get their own value back).
template <bool IS3D>
__global__ __launch_bounds__(256, TFL_LB_CONFINE) void k_confine_v4(Dom d, const float* Usrc, float* U, const float* __restrict__ flags,
                                                    const float* __restrict__ curl, const float* __restrict__ cn,
                                                    float strength, BcFoldArg folda, BlockOrder ord) {
  // U = Usrc + confinement force. Usrc == U: the reference's in-place operator; Usrc != U (round 5, tfl_vorticityConfinementFrom
  // on grids below the fused kernel's size): every cell of the window is written, which moves the velocity out of the
  // advection's scratch array for free
  int bx_, by_, bz_; block_tile(ord, bx_, by_, bz_);
  const V4Ctx c = v4_ctx(d, bx_);
  const int j = by_ * blockDim.y + threadIdx.y;
  int b, k; dom_bk_of(d, bz_, b, k);
  const bool fold_blk = fold_block(folda, (int)(by_ * blockDim.y), (int)(by_ * blockDim.y + blockDim.y - 1), k, k);
  const bool live = c.i0 < d.X && j < d.Y;
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  U += b * cells * C; Usrc += b * cells * C; flags += b * cells; curl += b * cells * 3; cn += b * cells;
  const int o = TFL_AT(d, c.i0, j, k);
  const bool in = live && j >= 1 && j <= d.Y - 2 && (!IS3D || (k >= 1 && k <= d.Z - 2));   // the cells' own row
  const bool in_ym = in && j - 1 >= 1;                 // row (j-1,k) is not a border row
  const bool in_zm = in && IS3D && k - 1 >= 1;         // row (j,k-1) is not a border row
  // |curl| rows. Own row and the two "minus" rows 6 wide (x gradient), the others 4 wide.
  float n_c[6], n_ym[6], n_zm[6], n_yp[4], n_zp[4], n_ym2[4], n_ymzp[4], n_ymzm[4], n_ypzm[4], n_zm2[4];
  v4_load6<true, true>(c, cn, o, in, 0.0f, n_c);
  v4_load6<true, true>(c, cn, o - d.sy, in, 0.0f, n_ym);
  v4_load(cn, o + d.sy, in, 0.0f, n_yp);
  v4_load(cn, o - 2 * d.sy, in_ym, 0.0f, n_ym2);
  if (IS3D) {
    v4_load6<true, true>(c, cn, o - d.sz, in, 0.0f, n_zm);
    v4_load(cn, o + d.sz, in, 0.0f, n_zp);
    v4_load(cn, o - d.sy + d.sz, in_ym, 0.0f, n_ymzp);
    v4_load(cn, o - d.sy - d.sz, in_ym || in_zm, 0.0f, n_ymzm);
    v4_load(cn, o + d.sy - d.sz, in_zm, 0.0f, n_ypzm);
    v4_load(cn, o - 2 * d.sz, in_zm, 0.0f, n_zm2);
  }
  float wx[4], wy[4], wz[4], wx_ym[4], wy_ym[4], wz_ym[4], wx_zm[4], wy_zm[4], wz_zm[4];
  v4_load(curl, o, in, 0.0f, wx);
  v4_load(curl, o + d.sc, in, 0.0f, wy);
  v4_load(curl, o + 2 * d.sc, in, 0.0f, wz);
  v4_load(curl, o - d.sy, in_ym, 0.0f, wx_ym);              // force.y needs w.x, w.z
  v4_load(curl, o - d.sy + 2 * d.sc, in_ym, 0.0f, wz_ym);
#pragma unroll
  for (int q = 0; q < 4; q++) { wy_ym[q] = 0.0f; wz_zm[q] = 0.0f; wx_zm[q] = 0.0f; wy_zm[q] = 0.0f; }
  if (IS3D) {
    v4_load(curl, o - d.sz, in_zm, 0.0f, wx_zm);            // force.z needs w.x, w.y
    v4_load(curl, o - d.sz + d.sc, in_zm, 0.0f, wy_zm);
  }
  float fl_c[6], fl_ym[4], fl_zm[4], u[3][4];
  v4_load6<true, false>(c, flags, o, in, 0.0f, fl_c);
  v4_load(flags, o - d.sy, in, 0.0f, fl_ym);
  v4_load(flags, o - d.sz, in && IS3D, 0.0f, fl_zm);
#pragma unroll
  for (int a = 0; a < 3; a++) v4_load(Usrc, o + a * d.sc, live && a < C, 0.0f, u[a]);

  v3 f0[4], fy[4], fz[4];
  force_row<IS3D>(d, strength, c.i0, in, n_c, n_ym + 1, n_yp, n_zm + 1, n_zp, wx, wy, wz, f0);
  force_row<IS3D>(d, strength, c.i0, in_ym, n_ym, n_ym2, n_c + 1, n_ymzm, n_ymzp, wx_ym, wy_ym, wz_ym, fy);
  if (IS3D) force_row<IS3D>(d, strength, c.i0, in_zm, n_zm, n_ymzm, n_ypzm, n_zm2, n_c + 1, wx_zm, wy_zm, wz_zm, fz);
  // force.x of cell i0-1: previous lane's last cell; at a segment start rebuilt from memory
  float fxl = from_lane_below(f0[3].x);
  {
    // (loads unconditional, tfl_vec4.hpp v4_load: every lane reads -- the lanes that need nothing, cell 0 of the field)
    const int i = c.i0 - 1;
    const bool need = c.first && in && i >= 1;   // i <= X-2 always
    const int oo = need ? o - 1 : 0;
    const float n_xm = cn[need ? oo - 1 : 0], n_ym1 = cn[need ? oo - d.sy : 0], n_yp1 = cn[need ? oo + d.sy : 0];
    const float n_zm1 = cn[need && IS3D ? oo - d.sz : 0], n_zp1 = cn[need && IS3D ? oo + d.sz : 0];
    const float w_y = curl[need ? oo + d.sc : 0], w_z = curl[need ? oo + 2 * d.sc : 0];
    if (c.first) {
      fxl = 0.0f;
      if (need) {
        v3 g = mk3(0.5f * (n_c[1] - n_xm), 0.5f * (n_yp1 - n_ym1), 0.0f);   // cn[oo + 1] is the thread's own first cell
        if (IS3D) g.z = 0.5f * (n_zp1 - n_zm1);
        g = normalize3(g);
        fxl = ((g.y * w_z) - (g.z * w_y)) * strength;
      }
    }
  }
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = c.i0 + q;
    if (!(in && i >= 1 && i <= d.X - 2)) continue;
    const int fc = (int)fl_c[q + 1];
    const bool cf = fc & kFluid, ce = fc & kEmpty;
    if (!cf && !ce) continue;  // AddForceField, tfluids.cc:1312-1339
    const int nx = (int)fl_c[q], ny = (int)fl_ym[q], nz = IS3D ? (int)fl_zm[q] : 0;
    const bool ax = (nx & kFluid) || (cf && (nx & kEmpty));
    const bool ay = (ny & kFluid) || (cf && (ny & kEmpty));
    const bool az = IS3D && ((nz & kFluid) || (cf && (nz & kEmpty)));
    const float fxm = q > 0 ? f0[q > 0 ? q - 1 : 0].x : fxl;
    if (ax) u[0][q] += (0.5f * (fxm + f0[q].x));
    if (ay) u[1][q] += (0.5f * (fy[q].y + f0[q].y));
    if (az) u[2][q] += (0.5f * (fz[q].z + f0[q].z));
  }
  if (live) {
    // the setConstVals that follows the forces in simulate() (tfl_host.hpp BcFold): only rows inside the pair's box load it
    BcFold fold = {nullptr, nullptr, 0, -1, 0, -1, 0, -1};
    if (fold_blk) fold = *folda.dev;
    if (fold_blk && fold_row(fold, j, k) && c.i0 <= fold.x1 && c.i0 + 3 >= fold.x0) {
#pragma unroll
      for (int a = 0; a < 3; a++) {
        if (a >= C) continue;
        float fb[4], fm[4];
        v4_load(fold.bc + b * cells * C, o + a * d.sc, true, 0.0f, fb);
        v4_load(fold.inv + b * cells * C, o + a * d.sc, true, 1.0f, fm);
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (fold_col(fold, c.i0 + q)) u[a][q] = u[a][q] * fm[q] + fb[q];
      }
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
      if (a < C) v4_store(U, o + a * d.sc, u[a]);
  }
}

// =====================================================================================================================
// Fused form (round 4): U_out = U_in + confinement(U_in) on a 3-D grid in ONE launch, no curl / |curl| arrays in HBM.
// A block owns a 64 x 8 column of the plane and walks a chunk of z. Per step one plane of U_in (with a halo of 3 cells in x
// and y) enters a ring of four planes in LDS; curl and |curl| of the plane two behind it are evaluated from the ring into
// their own rings (|curl|: three planes, curl: two); the force of the plane three behind it comes from those, its x / y
// face neighbours through a small exchange array, its z neighbour (the plane before) from the thread's own register; then
// the plane's velocities are written. Per-cell arithmetic is that of k_curl / k_confine, operation for operation: the
// results are bit-equal to the two-launch form (tests/test_hip_parity.py). HBM traffic: U (12 B/cell, ~3x with the halo
// re-reads through L2) + flags in, U out -- against 72 B/cell for the two launches.
// U_out must not alias U_in: neighbouring blocks read each other's input cells.
namespace {
constexpr int FBX = 64, FBY = 8;
constexpr int FUX = FBX + 6, FUY = FBY + 6, FUN = FUX * FUY;      // U tile 70 x 14, origin (x0 - 3, y0 - 3)
constexpr int FCX = FBX + 3, FCY = FBY + 3, FCN = FCX * FCY;      // curl tile 67 x 11, origin (x0 - 2, y0 - 2)
constexpr int FEX = FBX + 1, FEY = FBY + 1;                       // force exchange, origin (x0 - 1, y0 - 1)
constexpr int kFusedLds = (4 * 3 * FUN + 3 * FCN + 2 * 3 * FCN + 2 * FEY * FEX) * 4;   // 78 252 bytes: two blocks per CU
}  // namespace

// vec3::norm / normalize (generic/vec3.h:119-141) with the same thresholds and the same correctly rounded root and quotients
// as norm3 / normalize3 at a third of their instructions: sqrt_exact and ONE refined reciprocal for the three quotients
// (tfl_fastmath.hpp: the root checked exhaustively on [2^-40, 2^40), the quotient on a sample; the numerators are components
// of the vector whose norm divides them). Outside that range of the squared length, or for a NaN, the library forms.
// Branches: the library fall-back is taken by the WHOLE wave when any lane that counts (`use`) needs it -- a scalar branch that is
// never taken on real fields -- instead of a per-lane one (round 5: the kernel's 44 exec-mask branches and the register copies
// around them were a quarter of its instruction stream, and it runs at the length of that stream).
__device__ __forceinline__ float norm3_x(v3 a, bool use = true) {
  const float l2 = a.x * a.x + a.y * a.y + a.z * a.z;
  float n = (l2 > 1e-6f) ? sqrt_exact(l2) : 0.0f;
  if (__builtin_expect(__any(use && !(l2 < 0x1p40f)), 0)) {
    if (!(l2 < 0x1p40f)) n = norm3(a);
  }
  return n;
}
__device__ __forceinline__ v3 normalize3_x(v3 a, bool use = true) {
  const float l2 = a.x * a.x + a.y * a.y + a.z * a.z;
  const float n = (l2 > 1e-6f) ? sqrt_exact(l2) : 0.0f;
  const bool nz = n > 1e-6f;
  const float nn = nz ? n : 1.0f;                       // keeps the unused lanes' quotients finite
  const float r = rcp_refined(nn);
  v3 q = mk3(div_by<1>(a.x, nn, r), div_by<1>(a.y, nn, r), div_by<1>(a.z, nn, r));
  q = nz ? q : mk3(0.0f, 0.0f, 0.0f);
  if (__builtin_expect(__any(use && !(l2 < 0x1p40f)), 0)) {
    if (!(l2 < 0x1p40f)) q = normalize3(a);
  }
  return q;
}

__global__ __launch_bounds__(512) void k_vort_fused(Dom d, int cols_x, int cols_y, int cz, int chunks_a, int chunks, int n_blocks,
                                                    const float* __restrict__ Uin, float* __restrict__ Uout,
                                                    const float* __restrict__ flags, float strength, int xcd_order) {
  extern __shared__ float lds[];
  float* Ut = lds;                    // [4][3][FUN]  planes t & 3
  float* Cn = Ut + 4 * 3 * FUN;       // [3][FCN]     |curl|, planes z % 3
  float* Cv = Cn + 3 * FCN;           // [2][3][FCN]  curl, planes z & 1
  float* Fe = Cv + 2 * 3 * FCN;       // [2][FEY][FEX] force.x / force.y of the plane being finished
  // XCD-aware order (round 5): consecutive block ids go round-robin over the 8 XCDs, each with its own L2 -- so x / y neighbours
  // (blk +- 1, blk +- cols_x) never shared their 3-cell halo rings through an L2 and every block fetched its whole 70 x 14 tile,
  // 128-byte lines and all, from the fabric (45 B/cell read at 256^3 for 16 algorithmic). XCD k now owns a contiguous run of
  // tiles (x fastest, then y, then z chunk): its resident blocks are neighbours marching in step.
  const int blk = xcd_order ? (int)xcd_contiguous(blockIdx.x, (unsigned)n_blocks) : (int)blockIdx.x;
  if (blk >= n_blocks) return;
  int tq = blk;
  const int bx = tq % cols_x; tq /= cols_x;
  const int by = tq % cols_y; tq /= cols_y;
  const int ch = tq % chunks;
  const int b = tq / chunks;
  const int za = ch < chunks_a ? d.w0 + ch * cz : d.w1 + (ch - chunks_a) * cz;
  const int z_end = ch < chunks_a ? d.w0 + d.n0 : d.w1 + (d.nw - d.n0);
  const int zb = min(za + cz, z_end);
  const int x0 = bx * FBX, y0 = by * FBY;
  const long long cells = d.sc;
  Uin += b * cells * 3; Uout += b * cells * 3; flags += b * cells;
  const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
  const int i = x0 + tx, j = y0 + ty;

  // ---- per-thread geometry, fixed for the whole march --------------------------------------------------------------------
  // staging: the thread's one or two cells of a U plane (clamped: cells outside the array are never used, the border shell is 0)
  int st_o[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int it = min(tid + 512 * r, FUN - 1);
    const int uy = it / FUX, ux = it - uy * FUX;
    st_o[r] = TFL_AT(d, min(max(x0 - 3 + ux, 0), d.X - 1), min(max(y0 - 3 + uy, 0), d.Y - 1), 0);
  }
  // curl: the thread's one or two cells of the 67 x 11 curl tile; bits: 1 = in the grid and not on the x / y border shell,
  // 2 / 4 = its +x / -x neighbour is on the shell, 8 / 16 = +y / -y
  int c_it[2], c_base[2], c_bits[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int it = min(tid + 512 * r, FCN - 1);
    const int cy = it / FCX, cx = it - cy * FCX;
    const int gx = x0 - 2 + cx, gy = y0 - 2 + cy;
    c_it[r] = it;
    c_base[r] = (cy + 1) * FUX + (cx + 1);
    const bool in = tid + 512 * r < FCN && gx >= 1 && gx <= d.X - 2 && gy >= 1 && gy <= d.Y - 2;
    c_bits[r] = (in ? 1 : 0) | (gx + 1 == d.X - 1 ? 2 : 0) | (gx - 1 == 0 ? 4 : 0) | (gy + 1 == d.Y - 1 ? 8 : 0) | (gy - 1 == 0 ? 16 : 0);
  }
  // force: the thread's own cell, and (waves 0 and 1 only) one cell of the column / row before the block
  const int f_it = (ty + 2) * FCX + tx + 2;
  const bool f_in = i >= 1 && i <= d.X - 2 && j >= 1 && j <= d.Y - 2;
  const bool e_col = tid < FBY, e_row = tid >= 64 && tid < 64 + FBX;
  const int e_cx = e_col ? 1 : (tid - 64) + 2, e_cy = e_col ? tid + 2 : 1;
  const int e_it = (e_col || e_row) ? e_cy * FCX + e_cx : f_it;      // (the other lanes of waves 0 / 1 evaluate their own cell again, unused)
  const int e_gx = x0 - 2 + e_cx, e_gy = y0 - 2 + e_cy;
  const bool e_in = (e_col || e_row) && e_gx >= 1 && e_gx <= d.X - 2 && e_gy >= 1 && e_gy <= d.Y - 2;
  const int e_dst = e_col ? (tid + 1) * FEX : FEY * FEX + (tid - 64) + 1;
  const bool out_xy = i < d.X && j < d.Y;
  const int o_xy = TFL_AT(d, min(i, d.X - 1), min(j, d.Y - 1), 0);

  // confinement force of the cell `it` of the curl tile in plane zf (tfluids.cc:1410-1436); n0 / np / nm: ring slots of |curl|
  // (evaluated by every lane of the calling waves, `use` = the lane's value counts: the reads stay inside the rings)
  auto force = [&](int it, bool use, const float* cn0, const float* cnp, const float* cnm, const float* cv) -> v3 {
    const float* c0 = cn0 + it;
    v3 g = mk3(0.5f * (c0[1] - c0[-1]), 0.5f * (c0[FCX] - c0[-FCX]), 0.5f * (cnp[it] - cnm[it]));
    g = normalize3_x(g, use);
    const v3 w = mk3(cv[it], cv[FCN + it], cv[2 * FCN + it]);
    const v3 f = mk3(((g.y * w.z) - (g.z * w.y)) * strength, ((g.z * w.x) - (g.x * w.z)) * strength,
                     ((g.x * w.y) - (g.y * w.x)) * strength);
    return use ? f : mk3(0.0f, 0.0f, 0.0f);
  };

  float nu[2][3];                     // the U plane loaded one step ahead
  auto load_plane = [&](int t) {
    const int gz = min(max(t, 0), d.Z - 1) * d.sz;
#pragma unroll
    for (int r = 0; r < 2; r++) {
      if (r == 1 && tid >= FUN - 512) continue;
      const int o = st_o[r] + gz;
      nu[r][0] = Uin[o]; nu[r][1] = Uin[o + d.sc]; nu[r][2] = Uin[o + 2 * d.sc];
    }
  };
  load_plane(za - 3);
  float fz_prev = 0.0f;               // force.z of the thread's cell in the plane finished one step earlier
#pragma unroll 1
  for (int t = za - 3; t <= zb + 2; t++) {
    // ---- U plane t (in registers since the previous step) -> ring ----
    {
      float* dst = Ut + (t & 3) * 3 * FUN + tid;
      dst[0] = nu[0][0]; dst[FUN] = nu[0][1]; dst[2 * FUN] = nu[0][2];
      if (tid < FUN - 512) { dst[512] = nu[1][0]; dst[FUN + 512] = nu[1][1]; dst[2 * FUN + 512] = nu[1][2]; }
    }
    __syncthreads();
    load_plane(t + 1);
    // the inputs of the plane this step finishes (zo = t - 3), also asked for now
    const int zo = t - 3;
    const bool out_live = out_xy && zo >= za && zo < zb;
    const bool out_inner = out_live && f_in && zo >= 1 && zo <= d.Z - 2;
    float pu0 = 0.0f, pu1 = 0.0f, pu2 = 0.0f, pfc = 0.0f, pnx = 0.0f, pny = 0.0f, pnz = 0.0f;
    if (out_live) {
      // the plane's own velocities are still in the ring (plane t - 3 entered it three steps ago; its slot is overwritten at
      // the top of the NEXT step, behind this step's barriers) -- round 5: they were re-read from memory, 12 of the kernel's
      // ~61 fetched bytes per cell at 256^3 (profiles/r04_256_pmc_traffic.txt)
      const float* own = Ut + (zo & 3) * 3 * FUN + (ty + 3) * FUX + tx + 3;
      pu0 = own[0]; pu1 = own[FUN]; pu2 = own[2 * FUN];
      const int o = o_xy + zo * d.sz;
      if (out_inner) { pfc = flags[o]; pnx = flags[o - 1]; pny = flags[o - d.sy]; pnz = flags[o - d.sz]; }
    }
    // ---- curl, |curl| of plane zc = t - 2 (VecGrid::curl, grid.cc:497-515, centred velocities 0 on the border shell) ----
    const int zc = t - 2;
    if (zc >= za - 2 && zc <= zb) {       // block-uniform
      const bool z_in = zc >= 1 && zc <= d.Z - 2, zpb = zc + 1 == d.Z - 1, zmb = zc - 1 == 0;
      const float* P0 = Ut + (zc & 3) * 3 * FUN;          // planes zc, zc + 1, zc - 1 of the ring
      const float* Pp = Ut + ((zc + 1) & 3) * 3 * FUN;
      const float* Pm = Ut + ((zc - 1) & 3) * 3 * FUN;
      float* cvw = Cv + (zc & 1) * 3 * FCN;
      float* cnw = Cn + ((zc + 6) % 3) * FCN;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (r == 1 && (tid & ~63) >= FCN - 512) continue;      // wave-uniform: the waves with no cell in the second round
        // every lane evaluates (its reads stay inside the ring: c_base is a tile-interior index); `ok` selects
        const bool ok = z_in && (c_bits[r] & 1) && (r == 0 || tid < FCN - 512);
        const bool xpb = c_bits[r] & 2, xmb = c_bits[r] & 4, ypb = c_bits[r] & 8, ymb = c_bits[r] & 16;
        // (every tap is read first, unconditionally -- inside `cond ? 0 : f(load)` hipcc guards each load with its own
        // exec-mask branch: twelve of them per curl cell --, the border selects follow)
        const float* ux0 = P0 + c_base[r];
        const float* uy0 = ux0 + FUN;
        const float* uz0 = ux0 + 2 * FUN;
        const float* uxp = Pp + c_base[r];
        const float* uxm = Pm + c_base[r];
        const float y_xp0 = uy0[1], y_xp1 = uy0[1 + FUX], y_xm0 = uy0[-1], y_xm1 = uy0[-1 + FUX];
        const float x_yp0 = ux0[FUX], x_yp1 = ux0[FUX + 1], x_ym0 = ux0[-FUX], x_ym1 = ux0[-FUX + 1];
        const float z_yp0 = uz0[FUX], z_yp1 = uxp[2 * FUN + FUX], z_ym0 = uz0[-FUX], z_ym1 = uxp[2 * FUN - FUX];
        const float y_zp0 = uxp[FUN], y_zp1 = uxp[FUN + FUX], y_zm0 = uxm[FUN], y_zm1 = uxm[FUN + FUX];
        const float x_zp0 = uxp[0], x_zp1 = uxp[1], x_zm0 = uxm[0], x_zm1 = uxm[1];
        const float z_xp0 = uz0[1], z_xp1 = uxp[2 * FUN + 1], z_xm0 = uz0[-1], z_xm1 = uxp[2 * FUN - 1];
        const float cy_xp = xpb ? 0.0f : 0.5f * (y_xp0 + y_xp1);
        const float cy_xm = xmb ? 0.0f : 0.5f * (y_xm0 + y_xm1);
        const float cx_yp = ypb ? 0.0f : 0.5f * (x_yp0 + x_yp1);
        const float cx_ym = ymb ? 0.0f : 0.5f * (x_ym0 + x_ym1);
        v3 w;
        w.z = 0.5f * ((cy_xp - cy_xm) - (cx_yp - cx_ym));
        const float cz_yp = ypb ? 0.0f : 0.5f * (z_yp0 + z_yp1);
        const float cz_ym = ymb ? 0.0f : 0.5f * (z_ym0 + z_ym1);
        const float cy_zp = zpb ? 0.0f : 0.5f * (y_zp0 + y_zp1);
        const float cy_zm = zmb ? 0.0f : 0.5f * (y_zm0 + y_zm1);
        w.x = 0.5f * ((cz_yp - cz_ym) - (cy_zp - cy_zm));
        const float cx_zp = zpb ? 0.0f : 0.5f * (x_zp0 + x_zp1);
        const float cx_zm = zmb ? 0.0f : 0.5f * (x_zm0 + x_zm1);
        const float cz_xp = xpb ? 0.0f : 0.5f * (z_xp0 + z_xp1);
        const float cz_xm = xmb ? 0.0f : 0.5f * (z_xm0 + z_xm1);
        w.y = 0.5f * ((cx_zp - cx_zm) - (cz_xp - cz_xm));
        float nrm = norm3_x(w, ok);            // (unconditional: the wave-wide vote inside must not sit behind a lane branch)
        nrm = ok ? nrm : 0.0f;
        w = ok ? w : mk3(0.0f, 0.0f, 0.0f);
        if (r == 0 || tid < FCN - 512) {
          cvw[c_it[r]] = w.x; cvw[FCN + c_it[r]] = w.y; cvw[2 * FCN + c_it[r]] = w.z;
          cnw[c_it[r]] = nrm;
        }
      }
    }
    __syncthreads();
    // ---- force of plane zf = t - 3: the thread's own cell, and the cells one column / one row before the block ----------
    const int zf = t - 3;
    v3 f0 = mk3(0.0f, 0.0f, 0.0f);
    if (zf >= za - 1 && zf <= zb - 1) {   // block-uniform
      const bool z_in = zf >= 1 && zf <= d.Z - 2;
      const float* cn0 = Cn + ((zf + 6) % 3) * FCN;
      const float* cnp = Cn + ((zf + 7) % 3) * FCN;
      const float* cnm = Cn + ((zf + 5) % 3) * FCN;
      const float* cv = Cv + (zf & 1) * 3 * FCN;
      f0 = force(f_it, z_in && f_in, cn0, cnp, cnm, cv);
      Fe[(ty + 1) * FEX + tx + 1] = f0.x;
      Fe[(FEY + ty + 1) * FEX + tx + 1] = f0.y;
      if (tid < 128) {                      // waves 0 and 1 (wave-uniform): the column / the row before the block
        const v3 fe = force(e_it, z_in && e_in, cn0, cnp, cnm, cv);
        if (e_col || e_row) Fe[e_dst] = e_col ? fe.x : fe.y;
      }
    }
    __syncthreads();
    // ---- plane zf out: AddForceField (tfluids.cc:1312-1339); every cell of the plane is written (U_out is another array) ----
    if (out_live) {
      float u0 = pu0, u1 = pu1, u2 = pu2;
      {
        const int fc = (int)pfc;           // (all zero when !out_inner: nothing is added)
        const bool cf = fc & kFluid, ce = fc & kEmpty;
        const int nx = (int)pnx, ny = (int)pny, nz = (int)pnz;
        const bool any = out_inner && (cf || ce);
        const bool ax = any && ((nx & kFluid) || (cf && (nx & kEmpty)));
        const bool ay = any && ((ny & kFluid) || (cf && (ny & kEmpty)));
        const bool az = any && ((nz & kFluid) || (cf && (nz & kEmpty)));
        const float a0 = u0 + (0.5f * (Fe[(ty + 1) * FEX + tx] + f0.x));
        const float a1 = u1 + (0.5f * (Fe[(FEY + ty) * FEX + tx + 1] + f0.y));
        const float a2 = u2 + (0.5f * (fz_prev + f0.z));
        u0 = ax ? a0 : u0; u1 = ay ? a1 : u1; u2 = az ? a2 : u2;
      }
      const int o = o_xy + zf * d.sz;
      Uout[o] = u0; Uout[o + d.sc] = u1; Uout[o + 2 * d.sc] = u2;
    }
    fz_prev = f0.z;
  }
}

// =====================================================================================================================
// k_vort_pipe (round 5): the same fused operator, software-pipelined -- ONE barrier per plane step instead of three -- and
// with the CENTRED velocities in the ring instead of the MAC ones.
// * k_vort_fused's waves spend 61 % of their cycles waiting (SQ_WAIT_ANY, profiles/r05_vort_pipe.txt): a step is three
//   dependent phases (ring <- U plane | curl from the ring | force from the curl rings | store), a barrier behind each, 4.2
//   clocks per issued instruction and SIMD where the issue-bound kernels run at 2.5. Here every phase of step t works on data
//   an EARLIER step produced, so the phases of one step are independent instruction streams and one barrier closes the step.
// * VecGrid::curl (grid.cc:497-515) reads the centred velocity c(p) = 0.5 (u(p) + u(p + e)) of the six neighbours p of a
//   cell, 0 where p lies on the border shell: each c(p) was evaluated by up to two cells, from two ring reads, an add, a
//   multiply and a border select each time. The staging threads now evaluate it ONCE per staged cell (same expression: same
//   bits), border zero included, and the curl of a cell is 12 ring reads and 12 flops.
//     step t:  C ring[(t - 1) & 3] <- centred velocities of plane t - 1: c_x, c_y from the plane's own loads (registers since
//                                     step t - 1), c_z from u_z of planes t - 1 (carried) and t (loaded during step t - 1)
//              curl, |curl| of plane zc = t - 3  from C planes t - 4, t - 3, t - 2        -> Cv[zc % 3], Cn[zc & 3]
//              force of plane zf = t - 5          from Cn planes t - 6, t - 5, t - 4, Cv plane t - 5   -> Fe[zf & 1], registers
//              plane zo = t - 6 out               from Fe[zo & 1] (x / y neighbours), the thread's own force of planes zo and
//                                                 zo - 1 (registers), its own velocities and flags (loaded at the top of the step)
//   Nothing a step writes is read in the same step, and what step t + 1 overwrites (C plane t - 4, Cn plane t - 6, Cv plane
//   t - 5, Fe plane t - 6) was last read before step t's barrier.
// A block is 64 x 16 cells x a chunk of z with 1024 threads (one block per CU: 150 KB of LDS): staged cells per output 1.42
// (k_vort_fused: 1.91), curl cells 1.24 (1.44); the second rounds of the staging / the curl tile and the edge forces go to
// DIFFERENT waves (9-15 / 0-3 / 4-5). Pipeline fill: 9 steps per chunk (6). Per-cell arithmetic is k_curl / k_confine's,
// operation for operation: bit-equal to the two-launch form (tests/test_hip_parity.py).
namespace {
// k_vort_pipe's global accesses are BUFFER loads / stores: resource (the batch item's array) + the lane's 32-bit byte offset, fixed
// for the whole march, + the plane's byte offset in a scalar register. Written as pointers, the loop optimiser turns "moving
// base + fixed lane offset" into one 64-bit pointer PER LANE and advances it every step -- a v_lshl_add_u64 per access, 25 per
// step (round 6). An item's array must stay below 4 GiB (launch_vort_pipe checks).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t gbuf(const void* p, long long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)(unsigned)bytes, 0x00020000);
}
__device__ __forceinline__ float ldb(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned plane_off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)lane_off, (int)plane_off, 0));
}
__device__ __forceinline__ void stb(__amdgpu_buffer_rsrc_t r, unsigned lane_off, unsigned plane_off, float v) {
  __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, (int)lane_off, (int)plane_off, 0);
}
}  // namespace

namespace {
// a block = BX x BY cells (one thread each) x a chunk of z. 64 x 16: 1024 threads, 153 428 bytes of LDS, one block per CU;
// 32 x 16 (round 6): 512 threads, 80 852 bytes -- TWO blocks per CU, whose barrier and load waits overlap
template <int BX, int BY>
struct PipeGeo {
  static constexpr int PBX = BX, PBY = BY, NT = BX * BY;
  static constexpr int PCX = PBX + 3, PCY = PBY + 3, PCN = PCX * PCY;      // curl tile (67 x 19), origin (x0 - 2, y0 - 2)
  static constexpr int QX = PCX + 2, QY = PCY + 2, QN = QX * QY;           // centred-velocity tile (69 x 21), origin (x0 - 3, y0 - 3)
  static constexpr int PEX = PBX + 1, PEY = PBY + 1;                       // force exchange, origin (x0 - 1, y0 - 1)
  static constexpr int kLds = (4 * 3 * QN + 4 * PCN + 3 * 3 * PCN + 2 * 2 * PEY * PEX) * 4;
  // the threads that evaluate the column / the row of forces before the block: [kE, kE + PBY) and [kER, kER + PBX), in waves
  // that have no second curl round and as little of the second staging round as the block allows
  static constexpr int kE = BX == 64 ? 256 : 192, kER = BX == 64 ? 320 : kE + PBY;
  static_assert(QN <= 2 * NT && PCN <= 2 * NT && NT % 64 == 0, "two rounds cover the tiles");
};
constexpr int kPipeFill = 9;
}  // namespace

template <int BX, int BY>
__global__ __launch_bounds__(BX * BY) void k_vort_pipe(Dom d, int cols_x, int cols_y, int cz, int chunks_a, int chunks, int n_blocks,
                                                    const float* __restrict__ Uin, float* __restrict__ Uout,
                                                    const float* __restrict__ flags, float strength, int xcd_order, BcFoldArg folda) {
  using G = PipeGeo<BX, BY>;
  constexpr int PBX = G::PBX, PBY = G::PBY, NT = G::NT, PCX = G::PCX, PCN = G::PCN, QX = G::QX, QN = G::QN, PEX = G::PEX, PEY = G::PEY;
  extern __shared__ float lds[];
  float* Cr = lds;                    // [4][3][QN]   centred velocities, planes z & 3
  float* Cn = Cr + 4 * 3 * QN;        // [4][PCN]     |curl|, planes z & 3
  float* Cv = Cn + 4 * PCN;           // [3][3][PCN]  curl, planes z % 3
  float* Fe = Cv + 3 * 3 * PCN;       // [2][2][PEY][PEX] force.x / force.y, planes z & 1
  const int blk = xcd_order ? (int)xcd_contiguous(blockIdx.x, (unsigned)n_blocks) : (int)blockIdx.x;
  if (blk >= n_blocks) return;
  int tq = blk;
  const int bx = tq % cols_x; tq /= cols_x;
  const int by = tq % cols_y; tq /= cols_y;
  const int ch = tq % chunks;
  const int b = tq / chunks;
  const int za = ch < chunks_a ? d.w0 + ch * cz : d.w1 + (ch - chunks_a) * cz;
  const int z_end = ch < chunks_a ? d.w0 + d.n0 : d.w1 + (d.nw - d.n0);
  const int zb = min(za + cz, z_end);
  const int x0 = bx * PBX, y0 = by * PBY;
  const long long cells = d.sc;
  Uin += b * cells * 3; Uout += b * cells * 3; flags += b * cells;
  const int tid = threadIdx.x, tx = tid % PBX, ty = tid / PBX;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // (scalar: the per-wave roles below are scalar branches)
  const int i = x0 + tx, j = y0 + ty;

  // ---- per-thread geometry, fixed for the whole march --------------------------------------------------------------------
  // staging: round 0 = cell tid of the C tile, round 1 (waves 9-15) = cell 1024 + (tid - kS1). Per cell: the offsets of the
  // cell, its +x and its +y neighbour (clamped into the grid: cells outside it and shell cells stage 0) and the shell flag
  constexpr int kS1 = NT - (QN - NT);        // first thread of the second staging round (599)
  const bool two_st = wave * 64 + 63 >= kS1;     // this wave stages a second cell
  int st_it[2], st_o[2], st_ox[2], st_oy[2];
  bool st_sh[2];
  st_it[0] = tid; st_it[1] = tid >= kS1 ? NT + (tid - kS1) : tid;
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int qy = st_it[r] / QX, qx = st_it[r] - qy * QX;
    const int gx = x0 - 3 + qx, gy = y0 - 3 + qy;
    const int xa = min(max(gx, 0), d.X - 1), ya = min(max(gy, 0), d.Y - 1);
    st_o[r] = TFL_AT(d, xa, ya, 0);
    st_ox[r] = TFL_AT(d, min(xa + 1, d.X - 1), ya, 0);
    st_oy[r] = TFL_AT(d, xa, min(ya + 1, d.Y - 1), 0);
    st_sh[r] = gx <= 0 || gx >= d.X - 1 || gy <= 0 || gy >= d.Y - 1;
  }
  // curl: round 0 = cell tid of the 67 x 19 curl tile, round 1 (waves 0-3) = cell 1024 + tid
  constexpr int kC1 = PCN - NT;                // cells of the second curl round (249)
  int c_it[2], c_base[2];
  bool c_in[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const int it = min(tid + NT * r, PCN - 1);
    const int cy = it / PCX, cx = it - cy * PCX;
    const int gx = x0 - 2 + cx, gy = y0 - 2 + cy;
    c_it[r] = it;
    c_base[r] = (cy + 1) * QX + (cx + 1);
    c_in[r] = tid + NT * r < PCN && gx >= 1 && gx <= d.X - 2 && gy >= 1 && gy <= d.Y - 2;
  }
  // force: the thread's own cell; waves 4 / 5: one cell of the column / the row before the block
  const int f_it = (ty + 2) * PCX + tx + 2;
  const bool f_in = i >= 1 && i <= d.X - 2 && j >= 1 && j <= d.Y - 2;
  constexpr int kE = G::kE, kER = G::kER;
  const bool e_col = tid >= kE && tid < kE + PBY, e_row = tid >= kER && tid < kER + PBX;
  const bool e_wave = wave >= kE / 64 && wave <= (kER + PBX - 1) / 64;       // scalar
  const int e_cx = e_col ? 1 : (tid - kER) + 2, e_cy = e_col ? (tid - kE) + 2 : 1;
  const int e_it = (e_col || e_row) ? e_cy * PCX + e_cx : f_it;
  const int e_gx = x0 - 2 + e_cx, e_gy = y0 - 2 + e_cy;
  const bool e_in = (e_col || e_row) && e_gx >= 1 && e_gx <= d.X - 2 && e_gy >= 1 && e_gy <= d.Y - 2;
  const int e_dst = e_col ? ((tid - kE) + 1) * PEX : PEY * PEX + (tid - kER) + 1;
  const bool out_xy = i < d.X && j < d.Y;
  const int o_xy = TFL_AT(d, min(i, d.X - 1), min(j, d.Y - 1), 0);
  const int o_safe = o_xy + (i >= 1 ? 0 : 1) + (j >= 1 ? 0 : d.sy);      // a cell whose -x / -y neighbours exist
  // the same as byte offsets of the lane inside a plane, and the arrays as buffer resources (see ldb / stb above)
  const unsigned st_o4[2] = {(unsigned)st_o[0] * 4u, (unsigned)st_o[1] * 4u}, st_ox4[2] = {(unsigned)st_ox[0] * 4u, (unsigned)st_ox[1] * 4u};
  const unsigned st_oy4[2] = {(unsigned)st_oy[0] * 4u, (unsigned)st_oy[1] * 4u};
  const unsigned o_xy4 = (unsigned)o_xy * 4u, o_safe4 = (unsigned)o_safe * 4u, o_safe_x4 = (unsigned)(o_safe - 1) * 4u, o_safe_y4 = (unsigned)(o_safe - d.sy) * 4u;
  const unsigned sc4 = (unsigned)d.sc * 4u, sz4 = (unsigned)d.sz * 4u;
  const __amdgpu_buffer_rsrc_t rU = gbuf(Uin, cells * 12), rF = gbuf(flags, cells * 4), wU = gbuf(Uout, cells * 12);
  // the setConstVals that follows the forces in simulate() (tfl_host.hpp BcFold): only the blocks whose rows and planes can
  // touch the pair's box read the descriptor (for the plume's pair 1 block in 16-32)
  const bool fold_blk = fold_block(folda, y0, y0 + PBY - 1, za, zb - 1);
  BcFold fold = {nullptr, nullptr, 0, -1, 0, -1, 0, -1};
  if (fold_blk) { fold = *folda.dev; fold.bc += b * cells * 3; fold.inv += b * cells * 3; }
  const bool fold_xy = fold_blk && out_xy && j >= fold.y0 && j <= fold.y1 && fold_col(fold, i);

  auto force = [&](int it, bool use, const float* cn0, const float* cnp, const float* cnm, const float* cv) -> v3 {
    const float* c0 = cn0 + it;
    v3 g = mk3(0.5f * (c0[1] - c0[-1]), 0.5f * (c0[PCX] - c0[-PCX]), 0.5f * (cnp[it] - cnm[it]));
    g = normalize3_x(g, use);
    const v3 w = mk3(cv[it], cv[PCN + it], cv[2 * PCN + it]);
    const v3 f = mk3(((g.y * w.z) - (g.z * w.y)) * strength, ((g.z * w.x) - (g.x * w.z)) * strength,
                     ((g.x * w.y) - (g.y * w.x)) * strength);
    return use ? f : mk3(0.0f, 0.0f, 0.0f);
  };

  float nl[2][5];                     // the loads of the plane one step ahead: u_x(p), u_x(p + x), u_y(p), u_y(p + y), u_z(p)
  auto load_plane = [&](int t) {
    const unsigned px = (unsigned)min(max(t, 0), d.Z - 1) * sz4, py = px + sc4, pzc = py + sc4;      // (scalar)
#pragma unroll
    for (int r = 0; r < 2; r++) {
      if (r == 1 && !two_st) continue;
      nl[r][0] = ldb(rU, st_o4[r], px); nl[r][1] = ldb(rU, st_ox4[r], px); nl[r][2] = ldb(rU, st_o4[r], py); nl[r][3] = ldb(rU, st_oy4[r], py);
      nl[r][4] = ldb(rU, st_o4[r], pzc);
    }
  };
  const int t0 = za - 3, t1 = zb + 5;
  float cxp[2] = {0.0f, 0.0f}, cyp[2] = {0.0f, 0.0f}, uzp[2] = {0.0f, 0.0f};   // c_x, c_y, u_z of plane t - 1 at the top of step t
  // the store stage's inputs of its NEXT plane, asked for a step ahead: the flags of the cell, its -x and its -y neighbour, and the
  // thread's own velocities (un[]; round 5 carried those in a 7-deep register queue from the staging step -- 21 registers and
  // 21 moves per step); fzc = the cell's own flag of the plane before (the -z neighbour's)
  float fn[3] = {0.0f, 0.0f, 0.0f}, fzc = 0.0f, un[3] = {0.0f, 0.0f, 0.0f};
  v3 fcar = mk3(0.0f, 0.0f, 0.0f);    // the thread's force of plane t - 6 (computed in step t - 1)
  float fzcar = 0.0f;                 // force.z of plane t - 7
  int cv3 = ((t0 - 3) % 3 + 3) % 3;   // (zc % 3) of this step's curl plane, kept as a counter
  // one step; STEADY = every stage is active (no block-uniform guards: ONE basic block, so that the scheduler can run the
  // stages' ring reads and arithmetic against each other)
  auto step = [&](int t, auto steady_tag) {
    constexpr bool STEADY = decltype(steady_tag)::value;
    // ---- out stage, part 1: ask for the flags and the velocities of plane zo = t - 6 now ----
    const int zo = t - 6;
    const bool out_act = STEADY || (zo >= za && zo < zb);         // block-uniform
    const bool out_live = out_xy && out_act;
    const bool out_inner = out_live && f_in && zo >= 1 && zo <= d.Z - 2;
    float pfc = 0.0f, pnx = 0.0f, pny = 0.0f, pnz = 0.0f, pu0 = 0.0f, pu1 = 0.0f, pu2 = 0.0f;
    // (round 6) the flags of plane zo were asked for a step ago (fn[]: the cell, its -x, its -y neighbour); the -z neighbour's is
    // the cell's own flag of the plane before (fzc: carried). This step asks for plane zo + 1's.
    if (out_act) { pfc = out_inner ? fn[0] : 0.0f; pnx = out_inner ? fn[1] : 0.0f; pny = out_inner ? fn[2] : 0.0f; pnz = out_inner ? fzc : 0.0f; }
    pu0 = un[0]; pu1 = un[1]; pu2 = un[2];
    if (STEADY || (zo + 1 >= za && zo + 1 < zb)) {      // block-uniform
      fzc = fn[0];
      const unsigned pz = (unsigned)min(max(zo + 1, 0), d.Z - 1) * sz4;      // (scalar)
      fn[0] = ldb(rF, o_safe4, pz); fn[1] = ldb(rF, o_safe_x4, pz); fn[2] = ldb(rF, o_safe_y4, pz);
      un[0] = ldb(rU, o_xy4, pz); un[1] = ldb(rU, o_xy4, pz + sc4); un[2] = ldb(rU, o_xy4, pz + 2u * sc4);
    }
    // ---- centred velocities of plane t - 1 -> ring; then ask for plane t + 1 ----
    if (STEADY || t <= zb + 2) {        // block-uniform
      const bool zsh = t - 1 <= 0 || t - 1 >= d.Z - 1;
      float* dst = Cr + ((t - 1) & 3) * 3 * QN;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (r == 1 && !two_st) continue;
        const float cxn = 0.5f * (nl[r][0] + nl[r][1]), cyn = 0.5f * (nl[r][2] + nl[r][3]);     // plane t: next step's
        const float czp = 0.5f * (uzp[r] + nl[r][4]);
        const bool z0 = zsh || st_sh[r];
        if (STEADY || t > t0) { dst[st_it[r]] = z0 ? 0.0f : cxp[r]; dst[QN + st_it[r]] = z0 ? 0.0f : cyp[r]; dst[2 * QN + st_it[r]] = z0 ? 0.0f : czp; }
        cxp[r] = cxn; cyp[r] = cyn; uzp[r] = nl[r][4];
      }
      if (STEADY || t <= zb + 1) load_plane(t + 1);
    }
    // ---- curl, |curl| of plane zc = t - 3 from C planes t - 4, t - 3, t - 2 ----
    const int zc = t - 3;
    if (STEADY || (zc >= za - 2 && zc <= zb)) {       // block-uniform
      const bool z_in = zc >= 1 && zc <= d.Z - 2;
      const float* P0 = Cr + (zc & 3) * 3 * QN;
      const float* Pp = Cr + ((zc + 1) & 3) * 3 * QN;
      const float* Pm = Cr + ((zc - 1) & 3) * 3 * QN;
      float* cvw = Cv + cv3 * 3 * PCN;
      float* cnw = Cn + (zc & 3) * PCN;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (r == 1 && wave * 64 >= kC1) continue;              // waves 4-15 have no cell in the second round
        const bool mine = r == 0 || tid < kC1;
        const bool ok = z_in && c_in[r] && mine;
        const float* q0 = P0 + c_base[r];
        const float* qp = Pp + c_base[r];
        const float* qm = Pm + c_base[r];
        const float cy_xp = q0[QN + 1], cy_xm = q0[QN - 1], cx_yp = q0[QX], cx_ym = q0[-QX];
        const float cz_yp = q0[2 * QN + QX], cz_ym = q0[2 * QN - QX], cy_zp = qp[QN], cy_zm = qm[QN];
        const float cx_zp = qp[0], cx_zm = qm[0], cz_xp = q0[2 * QN + 1], cz_xm = q0[2 * QN - 1];
        v3 w;
        w.z = 0.5f * ((cy_xp - cy_xm) - (cx_yp - cx_ym));
        w.x = 0.5f * ((cz_yp - cz_ym) - (cy_zp - cy_zm));
        w.y = 0.5f * ((cx_zp - cx_zm) - (cz_xp - cz_xm));
        float nrm = norm3_x(w, ok);            // (unconditional: the wave-wide vote inside must not sit behind a lane branch)
        nrm = ok ? nrm : 0.0f;
        w = ok ? w : mk3(0.0f, 0.0f, 0.0f);
        if (mine) {
          cvw[c_it[r]] = w.x; cvw[PCN + c_it[r]] = w.y; cvw[2 * PCN + c_it[r]] = w.z;
          cnw[c_it[r]] = nrm;
        }
      }
    }
    // ---- force of plane zf = t - 5 from Cn planes t - 6, t - 5, t - 4 and Cv plane t - 5 ----
    const int zf = t - 5;
    v3 f0 = mk3(0.0f, 0.0f, 0.0f);
    if (STEADY || (zf >= za - 1 && zf <= zb - 1)) {   // block-uniform
      const bool z_in = zf >= 1 && zf <= d.Z - 2;
      const float* cn0 = Cn + (zf & 3) * PCN;
      const float* cnp = Cn + ((zf + 1) & 3) * PCN;
      const float* cnm = Cn + ((zf - 1) & 3) * PCN;
      const int cvf = cv3 >= 2 ? cv3 - 2 : cv3 + 1;          // (zc - 2) % 3
      const float* cv = Cv + cvf * 3 * PCN;
      float* fe = Fe + (zf & 1) * 2 * PEY * PEX;
      f0 = force(f_it, z_in && f_in, cn0, cnp, cnm, cv);
      fe[(ty + 1) * PEX + tx + 1] = f0.x;
      fe[(PEY + ty + 1) * PEX + tx + 1] = f0.y;
      if (e_wave) {         // the column / the row before the block
        const v3 fq = force(e_it, z_in && e_in, cn0, cnp, cnm, cv);
        if (e_col || e_row) fe[e_dst] = e_col ? fq.x : fq.y;
      }
    }
    // ---- plane zo = t - 6 out: AddForceField (tfluids.cc:1312-1339); every cell of the plane is written ----
    if (out_act) {
      const float* fe = Fe + (zo & 1) * 2 * PEY * PEX;
      float u0 = pu0, u1 = pu1, u2 = pu2;
      const int fc = (int)pfc;            // (all zero when !out_inner: nothing is added)
      const bool cf = fc & kFluid, ce = fc & kEmpty;
      const int nx = (int)pnx, ny = (int)pny, nz = (int)pnz;
      const bool any = out_inner && (cf || ce);
      const bool ax = any && ((nx & kFluid) || (cf && (nx & kEmpty)));
      const bool ay = any && ((ny & kFluid) || (cf && (ny & kEmpty)));
      const bool az = any && ((nz & kFluid) || (cf && (nz & kEmpty)));
      const float a0 = u0 + (0.5f * (fe[(ty + 1) * PEX + tx] + fcar.x));
      const float a1 = u1 + (0.5f * (fe[(PEY + ty) * PEX + tx + 1] + fcar.y));
      const float a2 = u2 + (0.5f * (fzcar + fcar.z));
      u0 = ax ? a0 : u0; u1 = ay ? a1 : u1; u2 = az ? a2 : u2;
      if (fold_blk) {                     // block-uniform
        const bool fq = fold_xy && out_live && zo >= fold.z0 && zo <= fold.z1;
        const long long o = fq ? (long long)zo * d.sz + o_xy : 0;
        const float m0 = fold.inv[o], m1 = fold.inv[o + d.sc], m2 = fold.inv[o + 2 * d.sc];
        const float b0 = fold.bc[o], b1 = fold.bc[o + d.sc], b2 = fold.bc[o + 2 * d.sc];
        const float g0 = u0 * m0 + b0, g1 = u1 * m1 + b1, g2 = u2 * m2 + b2;
        u0 = fq ? g0 : u0; u1 = fq ? g1 : u1; u2 = fq ? g2 : u2;
      }
      if (out_live) {
        const unsigned po = (unsigned)zo * sz4;      // (scalar)
        stb(wU, o_xy4, po, u0); stb(wU, o_xy4, po + sc4, u1); stb(wU, o_xy4, po + 2u * sc4, u2);
      }
    }
    // ---- rotate the registers that travel with the planes ----
    fzcar = fcar.z; fcar = f0;
    cv3 = cv3 == 2 ? 0 : cv3 + 1;
    __syncthreads();
  };
  // every stage is active for t in [za + 6, zb + 1]
  const int ts = min(max(za + 6, t0), t1 + 1), te = min(zb + 1, t1);
#ifdef TFL_VORT_TIMING
  long long tk[40]; int nk = 0;
  tk[nk++] = wall_clock64();
#define TFL_VT() if (nk < 40) tk[nk++] = wall_clock64()
#else
#define TFL_VT()
#endif
  // ---- prologue (round 6): what the steps t0 .. za would do -- nothing but staging: C planes za - 3, za - 2, za - 1 into the ring,
  // plane za's c_x, c_y, u_z into the carry registers -- with the loads of all four planes in flight TOGETHER and one barrier
  // instead of four (those four steps were 4.8 of a 128^3 chunk's 30.7 us: each waited out a memory round trip on its own)
  {
    float pl[4][2][5];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const unsigned px = (unsigned)min(max(t0 + q, 0), d.Z - 1) * sz4, py = px + sc4, pzc = py + sc4;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (r == 1 && !two_st) continue;
        pl[q][r][0] = ldb(rU, st_o4[r], px); pl[q][r][1] = ldb(rU, st_ox4[r], px); pl[q][r][2] = ldb(rU, st_o4[r], py); pl[q][r][3] = ldb(rU, st_oy4[r], py);
        pl[q][r][4] = ldb(rU, st_o4[r], pzc);
      }
    }
    load_plane(za + 1);
    fn[0] = ldb(rF, o_safe4, (unsigned)min(max(za - 1, 0), d.Z - 1) * sz4);      // becomes fzc when the step before the first store asks for plane za's flags
#pragma unroll
    for (int q = 0; q < 3; q++) {
      const int pz = t0 + q;
      const bool zsh = pz <= 0 || pz >= d.Z - 1;
      float* dst = Cr + (pz & 3) * 3 * QN;
#pragma unroll
      for (int r = 0; r < 2; r++) {
        if (r == 1 && !two_st) continue;
        const float cx = 0.5f * (pl[q][r][0] + pl[q][r][1]), cy = 0.5f * (pl[q][r][2] + pl[q][r][3]);
        const float cz = 0.5f * (pl[q][r][4] + pl[q + 1][r][4]);
        const bool z0 = zsh || st_sh[r];
        dst[st_it[r]] = z0 ? 0.0f : cx; dst[QN + st_it[r]] = z0 ? 0.0f : cy; dst[2 * QN + st_it[r]] = z0 ? 0.0f : cz;
      }
    }
#pragma unroll
    for (int r = 0; r < 2; r++) {
      if (r == 1 && !two_st) continue;
      cxp[r] = 0.5f * (pl[3][r][0] + pl[3][r][1]); cyp[r] = 0.5f * (pl[3][r][2] + pl[3][r][3]); uzp[r] = pl[3][r][4];
    }
    cv3 = (cv3 + 4) % 3;
    __syncthreads();
  }
  TFL_VT();
  int t = za + 1;
#pragma unroll 1
  for (; t < ts; t++) { step(t, std::false_type{}); TFL_VT(); }
#pragma unroll 1
  for (; t <= te; t++) { step(t, std::true_type{}); TFL_VT(); }
#pragma unroll 1
  for (; t <= t1; t++) { step(t, std::false_type{}); TFL_VT(); }
#ifdef TFL_VORT_TIMING
  if (tid == 0 && (blk == 0 || blk == n_blocks / 2 + 3)) {
    for (int q = 1; q < nk; q++) printf("vt blk %d step %d (t-za %d) %lld\n", blk, q - 1, q - 1, tk[q] - tk[q - 1]);     // (step 0 = the prologue)
  }
#endif
}

// block slots of k_vort_fused on the current device, asked once per device: 0 = the device cannot run it (its 78 KB of dynamic
// LDS refused, or no resident block: ADVICE r04 -- a part with 64 KB of LDS per workgroup), and both entry points below then
// say so instead of launching into an error
static int vort_fused_slots() {
  static std::atomic<int> slots_of[64];       // 0 = not asked yet, -1 = unsupported
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  int slots = slots_of[dev].load();
  if (!slots) {
    int cus = 256, per = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const bool attr_ok = hipFuncSetAttribute((const void*)k_vort_fused, hipFuncAttributeMaxDynamicSharedMemorySize, kFusedLds) == hipSuccess;
    if (!attr_ok || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)k_vort_fused, 512, kFusedLds) != hipSuccess || per <= 0) {
      (void)hipGetLastError();                // the refusal must not surface as the next launch's error
      slots = -1;
    } else slots = cus * per;
    slots_of[dev].store(slots);
  }
  return slots > 0 ? slots : 0;
}

// block slots of k_vort_pipe<BX, BY> on the current device (64 x 16: one 1024-thread block with 154 KB of LDS per CU; 32 x 16: two
// 512-thread blocks with 81 KB each); 0 = this device cannot run it
template <int BX, int BY>
static int vort_pipe_slots_of() {
  static std::atomic<int> slots_of[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  int slots = slots_of[dev].load();
  if (!slots) {
    int cus = 256, per = 0;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    using G = PipeGeo<BX, BY>;
    const bool attr_ok = hipFuncSetAttribute((const void*)k_vort_pipe<BX, BY>, hipFuncAttributeMaxDynamicSharedMemorySize, G::kLds) == hipSuccess;
    if (!attr_ok || hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, (const void*)k_vort_pipe<BX, BY>, G::NT, G::kLds) != hipSuccess || per <= 0) {
      (void)hipGetLastError();
      slots = -1;
    } else slots = cus * per;
    slots_of[dev].store(slots);
  }
  return slots > 0 ? slots : 0;
}

static int vort_pipe_slots() { return vort_pipe_slots_of<64, 16>(); }

// chunk length of a z-marched launch: rounds of resident blocks x (planes written + pipeline fill)
static int march_chunk(long long tiles, int na, int nb, int slots, int fill, int cmin) {
  int cz = cmin;
  long long best = -1;
  for (int c = cmin; c <= 64; c++) {
    const long long blocks = tiles * ((na + c - 1) / c + (nb + c - 1) / c);
    const long long cost = ((blocks + slots - 1) / slots) * (c + fill);
    if (best < 0 || cost < best) { best = cost; cz = c; }
  }
  return cz;
}


template <int BX, int BY>
static bool launch_vort_pipe(hipStream_t st, const Dom& d, int B, int X, int Y, int na, int nb, const float* Uin, float* Uout,
                             const float* flags, float strength, int xcd_order) {
  using G = PipeGeo<BX, BY>;
  const int pslots = vort_pipe_slots_of<BX, BY>();
  const int pcx = (X + BX - 1) / BX, pcy = (Y + BY - 1) / BY;
  const long long tiles = (long long)pcx * pcy * B;
  if (pslots <= 0 || tiles * (na + nb) <= 0) return false;
  if ((long long)d.Z * Y * X * 12 >= (1ll << 32)) return false;      // the kernel addresses an item's array with 32-bit byte offsets (buffer accesses)
  int cz = march_chunk(tiles, na, nb, pslots, kPipeFill, 4);
  if (const char* e = exp_env("TFL_VORT_CZ")) cz = atoi(e) > 0 ? atoi(e) : cz;
  const int chunks_a = (na + cz - 1) / cz, chunks = chunks_a + (nb + cz - 1) / cz;
  const int n_blocks = (int)(pcx * pcy * chunks * B);
  TFL_TIMED_EXT("k_vort_fused", st);
  const BcFoldArg fold = take_fold();    // the kernel writes the operator's result: it applies the pair that follows
  TFL_LAUNCH_EXT((k_vort_pipe<BX, BY>), n_blocks, G::NT, G::kLds, st, d, pcx, pcy, cz, chunks_a, chunks, n_blocks, Uin, Uout, flags, strength, xcd_order, fold);
  return true;
}

// does the native step (and tfl_vorticityConfinementFrom) route the confinement through a fused kernel? TFL_VORT_FUSED = 1 | 0
// forces. By measurement (profiles/r05_vort_pipe.txt, us: two launches / k_vort_pipe): 64^3 13 / 18, 96^3 24 / 23, 112^3 32 / 27,
// 128^3 36 / 29, 160^3 80 / 50, 192^3 147 / 77, 256^3 310 / 165. Inside the 128^3 step (with the setConstVals pair folded into
// either form: 37.5 / 35 us) the step gains 1.5 % on most boxes of the pool and nothing on one whose memory-bound kernels all
// ran slow that day -- never a loss. So: k_vort_pipe from 2 M cells per batch item on, on arrays at least 32 planes deep (round 6: a
// z-slab rank's 40-plane array of 256^3 on 8 ranks was measured too -- 35.8 us against 40.7, the rank-step 0.293 -> 0.279 ms);
// where the device cannot hold the pipelined kernel's block, k_vort_fused from 3 M cells (160^3: 69 / 80).
bool vorticity_confinement_fused_ok(bool is3d, int Z, int Y, int X) {
  static const int mode = getenv("TFL_VORT_FUSED") ? atoi(getenv("TFL_VORT_FUSED")) : -1;
  static const int pipe_mode = getenv("TFL_VORT_PIPE") ? atoi(getenv("TFL_VORT_PIPE")) : -1;
  if (!is3d || Z < 3 || mode == 0) return false;
  const long long cells = (long long)Z * Y * X;
  const bool pipe = pipe_mode != 0 && vort_pipe_slots() > 0;
  if (mode == 1) return pipe || vort_fused_slots() > 0;
  if (pipe && cells >= 2000000ll && Z >= 32) return true;      // (round 6: a 40-plane slab of 256^3 on 8 ranks: 35.8 us against 40.7 for the two launches)
  // round 6, after the kernel's prologue: planes that fill the 64 x 16 tiles exactly, from 0.6 M cells on -- the z-slab ranks of
  // the metric's 128^3 series: 36-40 planes on 4 ranks (rank-step 0.115 -> 0.112 / 0.121 -> 0.118 ms), 68 planes on 2 (0.169 ->
  // 0.159: one launch less, and k_bcs_div_stats runs 12.7 -> 9.1 us behind this kernel); 24 planes on 8 ranks stay with the two
  // launches (0.094 / 0.095). Whole grids with ragged tiles keep the 2 M rule (80^3 0.1240 / 0.1253 ms, 96^3 0.1537 / 0.1542,
  // 112^3 0.2114 / 0.2084 two launches / fused).
  if (pipe && cells >= 600000ll && Z >= 32 && X % PipeGeo<64, 16>::PBX == 0 && Y % PipeGeo<64, 16>::PBY == 0) return true;
  return cells >= 3000000ll && (pipe || vort_fused_slots() > 0);
}

// false = shape not supported by the fused kernel (the caller copies and runs the two-launch form)
bool vorticity_confinement_fused(hipStream_t st, int B, int Z, int Y, int X, const float* Uin, float* Uout, const float* flags,
                                 float strength) {
  if (Z < 3 || Uin == Uout) return false;
  const Dom d = make_dom(Z, Y, X);
  const int na = d.n0, nb = d.nw - d.n0;
  const int xcd_order = xcd_order_enabled() ? 1 : 0;
  // the software-pipelined form (one barrier per step; 64 x 16 tiles, 9 steps of fill) wherever the device can hold its block:
  // faster than k_vort_fused at every size measured, 9-step fill and all (profiles/r05_vort_pipe.txt). TFL_VORT_PIPE=0: the
  // three-barrier kernel
  static const int pipe_mode = getenv("TFL_VORT_PIPE") ? atoi(getenv("TFL_VORT_PIPE")) : -1;
  if (pipe_mode != 0) {
#ifdef TFL_EXPERIMENTS
    // 32 x 16 tiles, two 512-thread blocks per CU (round 6): level with 64 x 16 at 128^3 (34.9 against 34.3 us) and at 256^3
    // (182.6 / 180.5) -- a CU issues the two blocks' steps no faster than the one big block's; kept for A/B only
    static const int tile = exp_env("TFL_VORT_TILE") ? atoi(exp_env("TFL_VORT_TILE")) : 64;
    if (tile == 32 && launch_vort_pipe<32, 16>(st, d, B, X, Y, na, nb, Uin, Uout, flags, strength, xcd_order)) return true;
#endif
    if (launch_vort_pipe<64, 16>(st, d, B, X, Y, na, nb, Uin, Uout, flags, strength, xcd_order)) return true;
  }
  const int cxn = (X + FBX - 1) / FBX, cyn = (Y + FBY - 1) / FBY;
  if ((long long)cxn * cyn * (na + nb) * B <= 0) return true;
  const int slots = vort_fused_slots();     // (the dynamic-LDS attribute is a per-device setting: asked once per device)
  if (slots <= 0) return false;             // the caller copies and runs the two-launch form
  int cz = march_chunk((long long)cxn * cyn * B, na, nb, slots, 6, 4);
  if (const char* e = exp_env("TFL_VORT_CZ")) cz = atoi(e) > 0 ? atoi(e) : cz;
  const int chunks_a = (na + cz - 1) / cz, chunks = chunks_a + (nb + cz - 1) / cz;
  const int n_blocks = cxn * cyn * chunks * B;
  TFL_TIMED_EXT("k_vort_fused", st);
  TFL_LAUNCH_EXT(k_vort_fused, n_blocks, 512, kFusedLds, st, d, cxn, cyn, cz, chunks_a, chunks, n_blocks, Uin, Uout, flags, strength, xcd_order);
  return true;
}

bool vorticity_confinement(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* U, const float* flags,
                           float strength, float* curl, float* curl_norm, int stages, const float* Usrc) {
  const Dom d = make_dom(Z, Y, X);
  const dim3 blk(64, 4, 1), grd((X + 63) / 64, (Y + 3) / 4, (unsigned)(d.nw * B));
  const float* Uin = Usrc ? Usrc : U;
  const Vec4Launch v = vec4_launch(B, Z, Y, X, {U, flags, curl, curl_norm, Uin});
  const bool pa = stages & 1, pb = stages & 2;
  if (v.ok) {
    const BcFoldArg fold = pb ? take_fold() : no_fold();   // pass B writes the operator's result
    const BlockOrder ord = make_block_order(v.grd.x, v.grd.y, v.grd.z, is3d && xcd_order_enabled(), xcd_run(v.grd.x, v.grd.y));
    if (is3d) {
      if (pa) { TFL_TIMED_EXT("k_curl", st); TFL_LAUNCH_EXT((k_curl_v4<true>), v.grd, v.blk, 0, st, d, Uin, curl, curl_norm, ord); }
      if (pb) { TFL_TIMED_EXT("k_confine", st); TFL_LAUNCH_EXT((k_confine_v4<true>), v.grd, v.blk, 0, st, d, Uin, U, flags, (const float*)curl, (const float*)curl_norm, strength, fold, ord); }
    } else {
      if (pa) { TFL_TIMED("k_curl", st); k_curl_v4<false><<<v.grd, v.blk, 0, st>>>(d, Uin, curl, curl_norm, ord); }
      if (pb) { TFL_TIMED("k_confine", st); k_confine_v4<false><<<v.grd, v.blk, 0, st>>>(d, Uin, U, flags, curl, curl_norm, strength, fold, ord); }
    }
    return true;
  }
  if (Usrc && Usrc != U) return false;      // the one-cell kernels update in place: the caller copies first
  if (is3d) {
    if (pa) { TFL_TIMED("k_curl", st); k_curl<true><<<grd, blk, 0, st>>>(d, U, curl, curl_norm); }
    if (pb) { TFL_TIMED("k_confine", st); k_confine<true><<<grd, blk, 0, st>>>(d, U, flags, curl, curl_norm, strength); }
  } else {
    if (pa) { TFL_TIMED("k_curl", st); k_curl<false><<<grd, blk, 0, st>>>(d, U, curl, curl_norm); }
    if (pb) { TFL_TIMED("k_confine", st); k_confine<false><<<grd, blk, 0, st>>>(d, U, flags, curl, curl_norm, strength); }
  }
  return true;
}

}  // namespace tfl
