// advect.hip -- semi-Lagrangian / MacCormack advection on the MAC grid (gfx950).
//
// Replaces third_party/tfluids.cc:23-920 | tfluids.cu:21-963. The reference runs MacCormack as four
// grid sweeps plus a copy (fwd, bwd, correct, clamp, copy-back); here it is TWO launches, the minimum
// the method allows because the backward pass samples the complete forward field:
//   pass A : forward trace + sample -> fwd   (scalar "Ours": also the clamp bounds of the forward
//            position, so pass B never needs the traced position again)
//   pass B : backward trace on fwd + MacCormack correction + clamp -> dst
// One thread per cell, x fastest: a 64-lane wave covers 64 consecutive x of one row, so the
// cell-aligned loads/stores are 256-B coalesced rows; the data-dependent taps of the back-trace
// gather through L1/L2 (neighbouring lanes land in neighbouring cells).
//
// Algorithmic HBM bytes per cell (fp32, 3-D): advectScalar 52 B (A: s,U3,flags -> fwd; B: fwd,s,U3,
// flags -> dst), advectVel 68 B (A: U3,flags -> fwd3; B: fwd3,U3,flags -> dst3); SURVEY.md 8d.
#include "tfl_advect.hpp"
#include "tfl_vec4.hpp"

#include <cstdlib>
#include <cstring>

namespace tfl {


// Manta SemiLagrange, tfluids.cc:209-218
template <bool IS3D>
__device__ __forceinline__ float sl_manta(const Dom& d, const float* U, const float* src, float dt, int i, int j, int k) {
  const v3 c = cell_centre(d, i, j, k), u = get_centered<IS3D>(d, U, i, j, k);
  return interpol<IS3D>(d, src, mk3(c.x - u.x * dt, c.y - u.y * dt, c.z - u.z * dt));
}

// SemiLagrangeRK2Ours, tfluids.cc:23-77
template <bool IS3D>
__device__ float sl_rk2_ours(const AdvArgs& a, const float* flags, const float* U, const float* src, int i, int j, int k) {
  const v3 c = cell_centre(a.d, i, j, k);
  v3 half, back;
  int hit = line_trace(a.d, flags, c, scale3(get_centered<IS3D>(a.d, U, i, j, k), -a.dt * 0.5f), half);
  count_trace_error(hit, a.err);
  if (hit > 0) return sample_s<IS3D>(a.d, src, flags, half, a.outside);
  count_trace_error(line_trace(a.d, flags, c, scale3(sample_vel<IS3D>(a.d, U, half), -a.dt), back), a.err);
  return sample_s<IS3D>(a.d, src, flags, back, a.outside);
}

// SemiLagrangeRK3Ours, tfluids.cc:79-147 (CPU variant: a k3 hit samples at k3_pos)
template <bool IS3D>
__device__ float sl_rk3_ours(const AdvArgs& a, const float* flags, const float* U, const float* src, int i, int j, int k) {
  const v3 c = cell_centre(a.d, i, j, k);
  const float dt = a.dt;
  v3 p2, p3, back;
  const v3 k1 = get_centered<IS3D>(a.d, U, i, j, k);
  int hit = line_trace(a.d, flags, c, scale3(k1, -dt * 0.5f), p2);
  count_trace_error(hit, a.err);
  if (hit > 0) return sample_s<IS3D>(a.d, src, flags, p2, a.outside);
  const v3 k2 = sample_vel<IS3D>(a.d, U, p2);
  hit = line_trace(a.d, flags, c, scale3(k2, -dt * 0.75f), p3);
  count_trace_error(hit, a.err);
  if (hit > 0) return sample_s<IS3D>(a.d, src, flags, p3, a.outside);
  const v3 k3 = sample_vel<IS3D>(a.d, U, p3);
  // (real)(2.0/9.0) etc. are rounded to float before the multiply, tfluids.cc:135-137
  const v3 e1 = scale3(k1, -dt * (float)(2.0 / 9.0));
  const v3 e2 = scale3(k2, -dt * (float)(3.0 / 9.0));
  const v3 e3 = scale3(k3, -dt * (float)(4.0 / 9.0));
  const v3 disp = mk3((e1.x + e2.x) + e3.x, (e1.y + e2.y) + e3.y, (e1.z + e2.z) + e3.z);
  count_trace_error(line_trace(a.d, flags, c, disp, back), a.err);
  return sample_s<IS3D>(a.d, src, flags, back, a.outside);
}

// Manta MacCormackClamp (scalar), tfluids.cc:297-327
template <bool IS3D>
__device__ float manta_clamp_scalar(const Dom& d, const float* flags, const float* U, float dval, const float* orig,
                                    float fwd, float dt, int i, int j, int k) {
  const v3 ijk = mk3((float)i, (float)j, (float)(k + d.zg));
  const v3 ud = scale3(get_centered<IS3D>(d, U, i, j, k), dt);
  dval = manta_clamp_component<IS3D>(d, dval, orig, fwd, ijk, ud);
  const int fx = (int)((ijk.x + 0.5f) - ud.x), fy = (int)((ijk.y + 0.5f) - ud.y), fz = (int)((ijk.z + 0.5f) - ud.z);
  const int bx = (int)((ijk.x + 0.5f) + ud.x), by = (int)((ijk.y + 0.5f) + ud.y), bz = (int)((ijk.z + 0.5f) + ud.z);
  const int ux = d.X - 1, uy = d.Y - 1, uz = d.Zg - 1;
  if (fx < 0 || fy < 0 || fz < 0 || bx < 0 || by < 0 || bz < 0 || fx > ux || fy > uy || (fz > uz && IS3D) ||
      bx > ux || by > uy || (bz > uz && IS3D))
    return fwd;
  if ((flag_at(d, flags, fx, fy, slab_plane(d, fz, d.Z - 1)) & kObstacle) || (flag_at(d, flags, bx, by, slab_plane(d, bz, d.Z - 1)) & kObstacle)) return fwd;
  return dval;
}

// Precomputed clamp-bound grid: lo3/hi3[cell] = min/max of src over the (fluid) 3^dim neighbourhood of
// the cell (getClampBounds, tfluids.cc:331-378, evaluated once per CELL instead of once per traced
// POSITION: pass A then needs two gathers at int(fwd_pos) instead of 27 x 2). min/max are exact, so the
// evaluation order is free. Block = 64x4 threads of one z-plane; the masked values of the 66x6x3 halo
// tile are staged in LDS as two planes (lo: non-fluid -> +inf, hi: non-fluid -> -inf).
template <bool IS3D>
__global__ __launch_bounds__(256) void k_minmax3(Dom d, int outside, const float* __restrict__ s,
                                                 const float* __restrict__ flags, float* __restrict__ lo3,
                                                 float* __restrict__ hi3) {
  constexpr int NZ = IS3D ? 3 : 1;
  // {lo, hi} packed per tile cell: one ds_read_b64 fetches both (same LDS cycles as a b32 read)
  __shared__ float2 tile[NZ][6][66];
  int b, k; dom_bk(d, b, k);
  const long long cells = d.sc;
  s += b * cells; flags += b * cells; lo3 += b * cells; hi3 += b * cells;
  const int x0 = blockIdx.x * 64 - 1, y0 = blockIdx.y * 4 - 1, z0 = IS3D ? k - 1 : 0;
  const int tid = threadIdx.y * 64 + threadIdx.x;
  for (int idx = tid; idx < NZ * 6 * 66; idx += 256) {
    const int xx = idx % 66, yy = (idx / 66) % 6, zz = idx / (66 * 6);
    const int gx = x0 + xx, gy = y0 + yy, gz = z0 + zz;
    float2 v = make_float2(__builtin_inff(), -__builtin_inff());
    if (gx >= 0 && gx < d.X && gy >= 0 && gy < d.Y && gz >= 0 && gz < d.Z) {
      const int o = TFL_AT(d, gx, gy, gz);
      if (outside || (((int)flags[o]) & kFluid)) { v.x = s[o]; v.y = v.x; }
    }
    tile[zz][yy][xx] = v;
  }
  __syncthreads();
  const int i = blockIdx.x * 64 + threadIdx.x, j = blockIdx.y * 4 + threadIdx.y;
  if (i >= d.X || j >= d.Y) return;
  float lo = __builtin_inff(), hi = -__builtin_inff();
#pragma unroll
  for (int zz = 0; zz < NZ; zz++)
#pragma unroll
    for (int yy = 0; yy < 3; yy++)
#pragma unroll
      for (int xx = 0; xx < 3; xx++) {
        const float2 a = tile[zz][threadIdx.y + yy][threadIdx.x + xx];
        if (a.x < lo) lo = a.x;
        if (a.y > hi) hi = a.y;
      }
  const int o = TFL_AT(d, i, j, k);
  lo3[o] = lo; hi3[o] = hi;
}

// The same grid with four x-cells per thread and no LDS (tfl_vec4.hpp): per cell row of the 3^dim
// neighbourhood one 16-byte load of src and of flags; a masked-out cell is carried as NaN (min/max ignore it --
// exactly the reference's "skip", and a NaN in src itself is skipped by the reference's comparisons too), so
// ONE value per cell crosses lanes.
template <bool IS3D>
__global__ __launch_bounds__(256) void k_minmax3_v4(Dom d, int outside, const float* __restrict__ s,
                                                    const float* __restrict__ flags, float* __restrict__ lo3,
                                                    float* __restrict__ hi3) {
  const V4Ctx c = v4_ctx(d);
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  const bool live = c.i0 < d.X && j < d.Y;
  const long long cells = d.sc;
  s += b * cells; flags += b * cells; lo3 += b * cells; hi3 += b * cells;
  const float qnan = __builtin_nanf("");
  auto masked = [&](float v, float f) { return (outside || (((int)f) & kFluid)) ? v : qnan; };
  // Separable: first the extrema of each of the thread's four COLUMNS over the 3 (2-D) / 9 (3-D) rows of the
  // neighbourhood, then a 3-wide min/max along x; the columns i0-1 and i0+4 are the neighbouring lanes' last /
  // first column (4 DPP moves per thread instead of 2 per row). v_min3 / v_max3 skip NaN operands like the
  // reference's comparisons do; against its compare-and-keep chain only the sign of a zero bound can differ
  // (see manta_clamp_bounds).
  // v_min3_f32 / v_max3_f32 issued directly: through fminf / fmaxf hipcc first canonicalises every loaded operand
  // (v_max_f32 x, x, x: 152 of this kernel's 677 vector instructions) because it cannot rule out signalling NaNs; the
  // hardware instructions skip a NaN operand either way, which is all the masking needs. Rows are folded in pairs.
  auto min3r = [](float a, float b, float c2) { float r; asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c2)); return r; };
  auto max3r = [](float a, float b, float c2) { float r; asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c2)); return r; };
  float clo[6], chi[6], pend[6];
#pragma unroll
  for (int q = 0; q < 6; q++) { clo[q] = __builtin_inff(); chi[q] = -__builtin_inff(); pend[q] = qnan; }
  // all rows of the neighbourhood are asked for first (one memory round trip; round 4: a round trip per row before),
  // then folded. A segment's end lanes walk the outside columns themselves (only when the row continues: X > 128); those
  // loads are unconditional too (tfl_vec4.hpp v4_load: the lanes that need nothing read cell 0 of the field).
  constexpr int NR = IS3D ? 9 : 3;
  float svr[NR][4], fvr[NR][4], slr[NR], glr[NR], srr[NR], grr[NR];
  bool okr[NR], nlr[NR], nrr[NR];
#pragma unroll
  for (int r = 0; r < NR; r++) {
    const int dz = IS3D ? r / 3 - 1 : 0, dy = r % 3 - 1;
    const int jj = j + dy, kk = k + dz;
    const bool ok = live && jj >= 0 && jj < d.Y && kk >= 0 && kk < d.Z;
    const int o = TFL_AT(d, c.i0, jj, kk);
    v4_load(s, o, ok, qnan, svr[r]);
    v4_load(flags, o, ok, 0.0f, fvr[r]);
    const bool nl = c.first && ok && c.has_l, nr = c.last && ok && c.has_r;
    slr[r] = s[nl ? o - 1 : 0]; glr[r] = flags[nl ? o - 1 : 0]; srr[r] = s[nr ? o + 4 : 0]; grr[r] = flags[nr ? o + 4 : 0];
    okr[r] = ok; nlr[r] = nl; nrr[r] = nr;
  }
  int nrow = 0;
#pragma unroll
  for (int r = 0; r < NR; r++) {
    {
      const bool ok = okr[r];
      float m[6];
#pragma unroll
      for (int q = 0; q < 4; q++) m[q + 1] = ok ? masked(svr[r][q], fvr[r][q]) : qnan;
      m[0] = nlr[r] ? masked(slr[r], glr[r]) : qnan; m[5] = nrr[r] ? masked(srr[r], grr[r]) : qnan;
      if (nrow & 1) {
#pragma unroll
        for (int q = 0; q < 6; q++) { clo[q] = min3r(clo[q], pend[q], m[q]); chi[q] = max3r(chi[q], pend[q], m[q]); }
      } else {
#pragma unroll
        for (int q = 0; q < 6; q++) pend[q] = m[q];
      }
      nrow++;
    }
  }
  if (nrow & 1) {
#pragma unroll
    for (int q = 0; q < 6; q++) { clo[q] = min3r(clo[q], pend[q], pend[q]); chi[q] = max3r(chi[q], pend[q], pend[q]); }
  }
  {
    const float l0 = from_lane_below(clo[4]), h0 = from_lane_below(chi[4]);
    const float l5 = from_lane_above(clo[1]), h5 = from_lane_above(chi[1]);
    if (!c.first) { clo[0] = l0; chi[0] = h0; }
    if (!c.last) { clo[5] = l5; chi[5] = h5; }
  }
  float lo[4], hi[4];
#pragma unroll
  for (int q = 0; q < 4; q++) {
    lo[q] = min3r(clo[q], clo[q + 1], clo[q + 2]);
    hi[q] = max3r(chi[q], chi[q + 1], chi[q + 2]);
  }
  if (live) {
    const int o = TFL_AT(d, c.i0, j, k);
    v4_store(lo3, o, lo);
    v4_store(hi3, o, hi);
  }
}

// 8 waves per SIMD for the scalar passes (their traces are latency-bound; measured -2 us each at 128^3, r01)
#define TFL_SCALAR_OCC __attribute__((amdgpu_waves_per_eu(8, 8)))

// ---- advectScalar ------------------------------------------------------------------------------
// Pass A / single-pass methods. For kMacCormackOurs the clamp bounds go to bounds[0], bounds[1]
// (two channel planes of the caller's fwdPos temp; empty neighbourhood is stored as lo=+inf > hi).
template <bool IS3D, int METHOD>
__global__ __launch_bounds__(256) TFL_SCALAR_OCC void k_scalar_fwd(AdvArgs a, const float* __restrict__ s, const float* __restrict__ U,
                                                    const float* __restrict__ flags, float* __restrict__ out,
                                                    float* __restrict__ bounds, const float* __restrict__ lo3,
                                                    const float* __restrict__ hi3) {
  TFL_CELL_INDEX();
  const int C = IS3D ? 3 : 2;
  s += b * cells; flags += b * cells; out += b * cells; U += b * cells * C;
  const int o = TFL_AT(d, i, j, k);
  if (on_border<IS3D>(d, i, j, k)) { out[o] = 0.0f; return; }
  float v;
  if (METHOD == kEuler || METHOD == kMacCormack) {
    v = sl_manta<IS3D>(d, U, s, a.dt, i, j, k);
  } else {
    const bool fl = fluid_at(d, flags, i, j, k);
    v3 back = cell_centre(d, i, j, k);
    if (!fl) v = s[o];
    else if (METHOD == kRK2Ours) v = sl_rk2_ours<IS3D>(a, flags, U, s, i, j, k);
    else if (METHOD == kRK3Ours) v = sl_rk3_ours<IS3D>(a, flags, U, s, i, j, k);
    else v = sl_euler_ours<IS3D>(a, flags, U, s, a.dt, i, j, k, back);
    if (METHOD == kMacCormackOurs) {
      // clamp bounds of the forward position = the precomputed 3^dim min/max of the cell it falls in
      const int i0 = iclampi((int)back.x, 0, d.X - 1), j0 = iclampi((int)back.y, 0, d.Y - 1);
      const int k0 = IS3D ? slab_plane(d, iclampi((int)back.z, 0, d.Zg - 1), d.Z - 1) : 0;
      const long long g = b * cells + TFL_AT(d, i0, j0, k0);
      bounds += b * cells * C;
      bounds[o] = lo3[g];
      bounds[o + d.sc] = hi3[g];
    }
  }
  out[o] = v;
}

// Pass B of the two MacCormack flavours: bwd + correct (tfluids.cc:220-234) + clamp (:297-413).
template <bool IS3D, int METHOD>
__global__ __launch_bounds__(256) TFL_SCALAR_OCC void k_scalar_bwd(AdvArgs a, const float* __restrict__ s, const float* __restrict__ U,
                                                    const float* __restrict__ flags, const float* __restrict__ fwd,
                                                    const float* __restrict__ bounds, float* __restrict__ dst) {
  TFL_CELL_INDEX();
  const int C = IS3D ? 3 : 2;
  s += b * cells; flags += b * cells; fwd += b * cells; dst += b * cells; U += b * cells * C;
  const int o = TFL_AT(d, i, j, k);
  const bool border = on_border<IS3D>(d, i, j, k);
  const bool fl = fluid_at(d, flags, i, j, k);
  const float f = fwd[o];
  float bwd;
  if (border) bwd = 0.0f;
  else if (METHOD == kMacCormack) bwd = sl_manta<IS3D>(d, U, fwd, -a.dt, i, j, k);
  else if (!fl) bwd = f;
  else { v3 back; bwd = sl_euler_ours<IS3D>(a, flags, U, fwd, -a.dt, i, j, k, back); }
  // MacCormackCorrect has no border test; the unsuffixed 0.5 makes the reference evaluate the
  // correction in double and round once (tfluids.cc:231)
  float v = f;
  if (fl) v = (float)((double)f + (double)a.strength * 0.5 * (double)(s[o] - bwd));
  if (!border) {
    if (METHOD == kMacCormack) {
      v = manta_clamp_scalar<IS3D>(d, flags, U, v, s, f, a.dt, i, j, k);
    } else {
      bounds += b * cells * C;
      const float lo = bounds[o], hi = bounds[o + d.sc];
      v = (lo > hi) ? f : fclampf(v, lo, hi);
    }
  }
  dst[o] = v;
}

// Kernel structure (both passes): FIRST every load whose address does not depend on a back-trace -- the
// three MAC-averaged face velocities (18 distinct taps), the cell's own words and, in pass B, the 2 x 2^dim
// clamp corners of all three components -- issued as one batch, THEN the three data-dependent chains
// trace -> 8-tap sample. hipcc does not hoist loads across the trace loops on its own; in source order
// "component x completely, then y, then z" the kernel spent 68% of its wave cycles in s_waitcnt (r01 PMC).
template <bool IS3D, bool OURS>
__global__ __launch_bounds__(256) void k_vel_fwd(AdvArgs a, const float* __restrict__ U, const float* __restrict__ flags,
                                                 float* __restrict__ out) {
  TFL_CELL_INDEX();
  const int C = IS3D ? 3 : 2;
  flags += b * cells; U += b * cells * C; out += b * cells * C;
  const int o = TFL_AT(d, i, j, k);
  float vx = 0.0f, vy = 0.0f, vz = 0.0f;
  if (!on_border<IS3D>(d, i, j, k)) {
    if (OURS && !fluid_at(d, flags, i, j, k)) {  // tfluids.cc:598-601
      vx = U[o]; vy = U[o + d.sc]; if (IS3D) vz = U[o + 2 * d.sc];
    } else {
      const v3 u0 = get_at_mac<IS3D, 0>(d, U, i, j, k);
      const v3 u1 = get_at_mac<IS3D, 1>(d, U, i, j, k);
      const v3 u2 = IS3D ? get_at_mac<IS3D, 2>(d, U, i, j, k) : mk3(0.0f, 0.0f, 0.0f);
      vx = sl_mac_from_u<IS3D, OURS, 0>(a, flags, U, u0, a.dt, i, j, k);
      vy = sl_mac_from_u<IS3D, OURS, 1>(a, flags, U, u1, a.dt, i, j, k);
      if (IS3D) vz = sl_mac_from_u<IS3D, OURS, 2>(a, flags, U, u2, a.dt, i, j, k);
    }
  }
  out[o] = vx; out[o + d.sc] = vy; if (IS3D) out[o + 2 * d.sc] = vz;
}

template <bool IS3D, bool OURS>
__global__ __launch_bounds__(256) void k_vel_bwd(AdvArgs a, const float* __restrict__ U, const float* __restrict__ flags,
                                                 const float* __restrict__ fwd, float* __restrict__ dst) {
  TFL_CELL_INDEX();
  const int C = IS3D ? 3 : 2;
  flags += b * cells; U += b * cells * C; fwd += b * cells * C; dst += b * cells * C;
  const int o = TFL_AT(d, i, j, k);
  const bool border = on_border<IS3D>(d, i, j, k);
  const bool fl = fluid_at(d, flags, i, j, k);
  bool skip[3];
  skip[0] = !fl || (i > 0 && !fluid_at(d, flags, i - 1, j, k));
  skip[1] = !fl || (j > 0 && !fluid_at(d, flags, i, j - 1, k));
  skip[2] = IS3D ? (!fl || (k > 0 && !fluid_at(d, flags, i, j, k - 1))) : true;
  float f[3], uo[3], lo[3], hi[3], bwd[3] = {0.0f, 0.0f, 0.0f};
  bool ok[3] = {false, false, false};
  v3 u[3];
#pragma unroll
  for (int c = 0; c < C; c++) { f[c] = fwd[o + c * d.sc]; uo[c] = U[o + c * d.sc]; }
  if (!border) {
    u[0] = get_at_mac<IS3D, 0>(d, U, i, j, k);
    u[1] = get_at_mac<IS3D, 1>(d, U, i, j, k);
    u[2] = IS3D ? get_at_mac<IS3D, 2>(d, U, i, j, k) : mk3(0.0f, 0.0f, 0.0f);
    const v3 ijk = mk3((float)i, (float)j, (float)(k + d.zg));
    // MacCormackClampMAC bounds, tfluids.cc:748-774
#pragma unroll
    for (int c = 0; c < C; c++) ok[c] = manta_clamp_bounds<IS3D>(d, U + c * d.sc, ijk, scale3(u[c], a.dt), lo[c], hi[c]);
    if (OURS && !fl) {
#pragma unroll
      for (int c = 0; c < C; c++) bwd[c] = f[c];
    } else {
      bwd[0] = sl_mac_from_u<IS3D, OURS, 0>(a, flags, fwd, u[0], -a.dt, i, j, k);
      bwd[1] = sl_mac_from_u<IS3D, OURS, 1>(a, flags, fwd, u[1], -a.dt, i, j, k);
      if (IS3D) bwd[2] = sl_mac_from_u<IS3D, OURS, 2>(a, flags, fwd, u[2], -a.dt, i, j, k);
    }
  }
#pragma unroll
  for (int c = 0; c < C; c++) {
    float v = f[c];
    // MacCormackCorrectMAC, tfluids.cc:660-699 (double arithmetic through the unsuffixed 0.5, :693)
    if (!skip[c]) v = (float)((double)f[c] + (double)a.strength * 0.5 * (double)(uo[c] - bwd[c]));
    if (!border) v = ok[c] ? fclampf(v, lo[c], hi[c]) : f[c];
    dst[o + c * d.sc] = v;
  }
}

// ---- host launchers ----------------------------------------------------------------------------
template <bool IS3D>
static void launch_scalar(hipStream_t st, int method, const AdvArgs& a, int B, const float* s, const float* U,
                          const float* flags, float* fwd, float* bounds, float* mm, float* dst, int stages) {
  const dim3 blk(64, 4, 1), grd = cell_grid(a.d, B, blk);
  // stages (tfl_set_stages): 1 = the 3^dim min/max grid, 2 = pass A, 4 = pass B; single-pass methods are "pass A"
  const bool pm = stages & 1, pa = stages & 2, pb = stages & 4;
  if (method != kMacCormack && method != kMacCormackOurs && !pa) return;
  // the trace-based methods on a 3-D grid: LDS-tiled fast-path kernels without a min/max grid (advect_scalar3.hip)
  if (IS3D && (method == kEulerOurs || method == kMacCormackOurs) &&
      advect_scalar3(st, method == kMacCormackOurs, a, B, s, U, flags, fwd, bounds, dst, stages))
    return;
  switch (method) {
    case kEuler: { TFL_TIMED("k_scalar_fwd", st); k_scalar_fwd<IS3D, kEuler><<<grd, blk, 0, st>>>(a, s, U, flags, dst, nullptr, nullptr, nullptr); break; }
    case kEulerOurs: { TFL_TIMED("k_scalar_fwd", st); k_scalar_fwd<IS3D, kEulerOurs><<<grd, blk, 0, st>>>(a, s, U, flags, dst, nullptr, nullptr, nullptr); break; }
    case kRK2Ours: { TFL_TIMED("k_scalar_fwd", st); k_scalar_fwd<IS3D, kRK2Ours><<<grd, blk, 0, st>>>(a, s, U, flags, dst, nullptr, nullptr, nullptr); break; }
    case kRK3Ours: { TFL_TIMED("k_scalar_fwd", st); k_scalar_fwd<IS3D, kRK3Ours><<<grd, blk, 0, st>>>(a, s, U, flags, dst, nullptr, nullptr, nullptr); break; }
    case kMacCormack:
      if (pa) { TFL_TIMED("k_scalar_fwd", st); k_scalar_fwd<IS3D, kMacCormack><<<grd, blk, 0, st>>>(a, s, U, flags, fwd, nullptr, nullptr, nullptr); }
      if (pb) { TFL_TIMED("k_scalar_bwd", st); k_scalar_bwd<IS3D, kMacCormack><<<grd, blk, 0, st>>>(a, s, U, flags, fwd, nullptr, dst); }
      break;
    default:
      if (pm) minmax3(st, IS3D, B, a.d.Z, a.d.Y, a.d.X, a.outside, s, flags, mm, mm + (long long)B * a.d.sc);
      if (pa) { TFL_TIMED_EXT("k_scalar_fwd", st); TFL_LAUNCH_EXT((k_scalar_fwd<IS3D, kMacCormackOurs>), grd, blk, 0, st, a, s, U, flags, fwd, bounds, (const float*)mm, (const float*)(mm + (long long)B * a.d.sc)); }
      if (pb) { TFL_TIMED_EXT("k_scalar_bwd", st); TFL_LAUNCH_EXT((k_scalar_bwd<IS3D, kMacCormackOurs>), grd, blk, 0, st, a, s, U, flags, (const float*)fwd, (const float*)bounds, dst); }
      break;
  }
}

void minmax3(hipStream_t st, bool is3d, int B, int Z, int Y, int X, int outside, const float* s, const float* flags,
             float* lo3, float* hi3) {
  const Dom d = make_dom(Z, Y, X);
  const dim3 blk(64, 4, 1), grd = cell_grid(d, B, blk);
  const Vec4Launch v = vec4_launch(B, Z, Y, X, {s, flags, lo3, hi3});
  if (v.ok) {
    TFL_TIMED_EXT("k_minmax3", st);
    if (is3d) TFL_LAUNCH_EXT((k_minmax3_v4<true>), v.grd, v.blk, 0, st, d, outside, s, flags, lo3, hi3);
    else TFL_LAUNCH_EXT((k_minmax3_v4<false>), v.grd, v.blk, 0, st, d, outside, s, flags, lo3, hi3);
    return;
  }
  TFL_TIMED("k_minmax3", st);
  if (is3d) k_minmax3<true><<<grd, blk, 0, st>>>(d, outside, s, flags, lo3, hi3);
  else k_minmax3<false><<<grd, blk, 0, st>>>(d, outside, s, flags, lo3, hi3);
}

void advect_scalar(hipStream_t st, bool is3d, int method, int B, int Z, int Y, int X, float dt, float strength,
                   int outside, unsigned long long* err, const float* s, const float* U, const float* flags,
                   float* fwd, float* bounds, float* mm, float* dst, int stages) {
  AdvArgs a; a.d = make_dom(Z, Y, X); a.dt = dt; a.strength = strength; a.outside = outside; a.err = err; a.fast = g_advect_fast;
  if (is3d) launch_scalar<true>(st, method, a, B, s, U, flags, fwd, bounds, mm, dst, stages);
  else launch_scalar<false>(st, method, a, B, s, U, flags, fwd, bounds, mm, dst, stages);
}

template <bool IS3D>
static void launch_vel(hipStream_t st, int method, const AdvArgs& a, int B, const float* U, const float* flags,
                       float* fwd, float* dst, int stages) {
  const dim3 blk(64, 4, 1), grd = cell_grid(a.d, B, blk);
  const bool pa = stages & 2, pb = stages & 4;   // as in launch_scalar
  if (method != kMacCormack && method != kMacCormackOurs && !pa) return;
  // the trace-based methods on a 3-D grid: LDS-tiled fast-path kernels (advect_vel3.hip)
  if (IS3D && (method == kEulerOurs || method == kMacCormackOurs) &&
      advect_vel3(st, method == kMacCormackOurs, a, B, U, flags, fwd, dst, stages))
    return;
  switch (method) {
    case kEuler: { TFL_TIMED("k_vel_fwd", st); k_vel_fwd<IS3D, false><<<grd, blk, 0, st>>>(a, U, flags, dst); break; }
    case kEulerOurs: { TFL_TIMED("k_vel_fwd", st); k_vel_fwd<IS3D, true><<<grd, blk, 0, st>>>(a, U, flags, dst); break; }
    case kMacCormack:
      if (pa) { TFL_TIMED("k_vel_fwd", st); k_vel_fwd<IS3D, false><<<grd, blk, 0, st>>>(a, U, flags, fwd); }
      if (pb) { TFL_TIMED("k_vel_bwd", st); k_vel_bwd<IS3D, false><<<grd, blk, 0, st>>>(a, U, flags, fwd, dst); }
      break;
    default:
      if (pa) { TFL_TIMED_EXT("k_vel_fwd", st); TFL_LAUNCH_EXT((k_vel_fwd<IS3D, true>), grd, blk, 0, st, a, U, flags, fwd); }
      if (pb) { TFL_TIMED_EXT("k_vel_bwd", st); TFL_LAUNCH_EXT((k_vel_bwd<IS3D, true>), grd, blk, 0, st, a, U, flags, fwd, dst); }
      break;
  }
}

void advect_vel(hipStream_t st, bool is3d, int method, int B, int Z, int Y, int X, float dt, float strength,
                unsigned long long* err, const float* U, const float* flags, float* fwd, float* dst, int stages) {
  if (method == kRK2Ours || method == kRK3Ours) method = kMacCormackOurs;  // tfluids.cc:799-802
  AdvArgs a; a.d = make_dom(Z, Y, X); a.dt = dt; a.strength = strength; a.outside = 0; a.err = err; a.fast = g_advect_fast;
  if (is3d) launch_vel<true>(st, method, a, B, U, flags, fwd, dst, stages);
  else launch_vel<false>(st, method, a, B, U, flags, fwd, dst, stages);
}

}  // namespace tfl
