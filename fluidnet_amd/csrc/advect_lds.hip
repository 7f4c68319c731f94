// advect_lds.hip -- LDS-tiled MacCormack ("maccormackOurs") advection for gfx950.
//
// Same arithmetic, same results (bit-exact) as advect.hip; what changes is where the operands come
// from. advect.hip gathers everything through the vector memory path: ~60 (pass A) / ~110 (pass B of
// advectVel) dword loads per cell, each costing the TA/L1 ~10 cycles per wave because a wave's taps
// straddle cache lines -- measured 13% of HBM peak, load-issue bound (profiles/r01). Here a block
// stages the fields it samples -- tile + 1-cell halo -- into LDS once (coalesced 136-byte rows), and
// every stencil tap / trilinear corner / flag test of the back-trace becomes a ds_read_b32 (2 cycles per
// wave, consecutive lanes -> consecutive banks). A tap that falls outside the tile (|u|*dt > 1 cell, or
// a trace that ran along a wall) transparently falls back to the global load, so the result does not
// depend on the tile shape.
//
// Block = 256 threads, one cell per thread; 3-D tile 32 x 8 x 1 cells (+ halo = 34 x 10 x 3), 2-D tile
// 64 x 4. LDS: 4 KB per field in 3-D; pass B of advectVel stages 7 fields (U3, fwd3, flags) = 28 KB.
//
// STATUS (r01, measured on MI355X at 128^3): bit-identical to advect.hip but SLOWER -- k_vel_bwd 128 us
// here (177 us with 32x8x4 tiles and a 4-plane march per thread) vs 79 us for the plain-gather kernel.
// The gathers of advect.hip already hit in the hardware L1; staging every field (3-4x read amplification
// from the halo) plus the dual LDS/global code path and its register cost outweigh the cheaper taps.
// Opt-in via TFL_ADVECT_PATH=lds; the default path is advect.hip.
#include "tfl_device.hpp"
#include "tfl_host.hpp"

namespace tfl {

struct AdvArgsL {
  Dom d;
  float dt;
  float strength;
  int outside;
  unsigned long long* err;
};

template <bool IS3D>
struct TileDims {
  static constexpr int BX = IS3D ? 32 : 64, BY = IS3D ? 8 : 4, BZ = 1;
  static constexpr int NX = BX + 2, NY = BY + 2, NZ = IS3D ? BZ + 2 : 1;
  static constexpr int N = NX * NY * NZ;
};

// A field seen through the tile: LDS inside, global memory outside.
template <bool IS3D>
struct TField {
  const float* t;  // LDS copy of [z0, z0+NZ) x [y0, y0+NY) x [x0, x0+NX)
  const float* g;  // the same batch item / channel in global memory
  int x0, y0, z0;
  using TD = TileDims<IS3D>;
  __device__ __forceinline__ bool inside(int i, int j, int k) const {
    return (unsigned)(i - x0) < (unsigned)TD::NX && (unsigned)(j - y0) < (unsigned)TD::NY &&
           (!IS3D || (unsigned)(k - z0) < (unsigned)TD::NZ);
  }
  __device__ __forceinline__ int lidx(int i, int j, int k) const {
    return (i - x0) + TD::NX * ((j - y0) + (IS3D ? TD::NY * (k - z0) : 0));
  }
  __device__ __forceinline__ float at(const Dom& d, int i, int j, int k) const {
    return inside(i, j, k) ? t[lidx(i, j, k)] : g[TFL_AT(d, i, j, k)];
  }
};

template <bool IS3D>
__device__ __forceinline__ void stage(float* lds, const float* __restrict__ g, const Dom& d, int x0, int y0, int z0) {
  using TD = TileDims<IS3D>;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < TD::N; idx += 256) {
    const int xx = idx % TD::NX, yy = (idx / TD::NX) % TD::NY, zz = idx / (TD::NX * TD::NY);
    const int gx = x0 + xx, gy = y0 + yy, gz = z0 + zz;
    float v = 0.0f;
    if (gx >= 0 && gx < d.X && gy >= 0 && gy < d.Y && gz >= 0 && gz < d.Z) v = g[TFL_AT(d, gx, gy, gz)];
    lds[idx] = v;
  }
}

// ---- samplers on tile fields (same expressions as tfl_device.hpp) --------------------------------
template <bool IS3D>
__device__ __forceinline__ float t_interpol(const Dom& d, const TField<IS3D>& f, v3 pos) {
  using TD = TileDims<IS3D>;
  const Lerp L = build_index<IS3D>(d, pos);
  float c000, c010, c100, c110, c001 = 0, c011 = 0, c101 = 0, c111 = 0;
  if (f.inside(L.xi, L.yi, L.zi) && f.inside(L.xi + 1, L.yi + 1, IS3D ? L.zi + 1 : L.zi)) {
    const float* p = f.t + f.lidx(L.xi, L.yi, L.zi);
    c000 = p[0]; c010 = p[TD::NX]; c100 = p[1]; c110 = p[1 + TD::NX];
    if (IS3D) {
      const float* q = p + TD::NX * TD::NY;
      c001 = q[0]; c011 = q[TD::NX]; c101 = q[1]; c111 = q[1 + TD::NX];
    }
  } else {
    const float* p = f.g + TFL_AT(d, L.xi, L.yi, L.zi);
    c000 = p[0]; c010 = p[d.sy]; c100 = p[1]; c110 = p[1 + d.sy];
    if (IS3D) {
      const float* q = p + d.sz;
      c001 = q[0]; c011 = q[d.sy]; c101 = q[1]; c111 = q[1 + d.sy];
    }
  }
  const float lo = (c000 * L.t0 + c010 * L.t1) * L.s0 + (c100 * L.t0 + c110 * L.t1) * L.s1;
  if (!IS3D) return lo;
  const float hi = (c001 * L.t0 + c011 * L.t1) * L.s0 + (c101 * L.t0 + c111 * L.t1) * L.s1;
  return lo * L.f0 + hi * L.f1;
}

template <bool IS3D>
__device__ __forceinline__ float t_interpol_with_fluid(const Dom& d, const TField<IS3D>& f, const TField<IS3D>& fl, v3 pos) {
  using TD = TileDims<IS3D>;
  const Lerp L = build_index<IS3D>(d, pos);
  float c[8];
  int m[8];
  if (f.inside(L.xi, L.yi, L.zi) && f.inside(L.xi + 1, L.yi + 1, IS3D ? L.zi + 1 : L.zi)) {
    const int o = f.lidx(L.xi, L.yi, L.zi);
    const int off[8] = {0, TD::NX, 1, 1 + TD::NX, TD::NX * TD::NY, TD::NX * TD::NY + TD::NX, TD::NX * TD::NY + 1,
                        TD::NX * TD::NY + 1 + TD::NX};
#pragma unroll
    for (int q = 0; q < (IS3D ? 8 : 4); q++) { c[q] = f.t[o + off[q]]; m[q] = (int)fl.t[o + off[q]] & kFluid; }
  } else {
    const int o = TFL_AT(d, L.xi, L.yi, L.zi);
    const int off[8] = {0, d.sy, 1, 1 + d.sy, d.sz, d.sz + d.sy, d.sz + 1, d.sz + 1 + d.sy};
#pragma unroll
    for (int q = 0; q < (IS3D ? 8 : 4); q++) { c[q] = f.g[o + off[q]]; m[q] = (int)fl.g[o + off[q]] & kFluid; }
  }
  bool f_ab, f_cd, f_abcd, fo;
  float v_ab, v_cd, v_abcd, val;
  lerp_fluid(c[0], m[0] != 0, c[1], m[1] != 0, L.t0, L.t1, f_ab, v_ab);
  lerp_fluid(c[2], m[2] != 0, c[3], m[3] != 0, L.t0, L.t1, f_cd, v_cd);
  lerp_fluid(v_ab, f_ab, v_cd, f_cd, L.s0, L.s1, f_abcd, v_abcd);
  if (IS3D) {
    bool f_ef, f_gh, f_efgh;
    float v_ef, v_gh, v_efgh;
    lerp_fluid(c[4], m[4] != 0, c[5], m[5] != 0, L.t0, L.t1, f_ef, v_ef);
    lerp_fluid(c[6], m[6] != 0, c[7], m[7] != 0, L.t0, L.t1, f_gh, v_gh);
    lerp_fluid(v_ef, f_ef, v_gh, f_gh, L.s0, L.s1, f_efgh, v_efgh);
    lerp_fluid(v_abcd, f_abcd, v_efgh, f_efgh, L.f0, L.f1, fo, val);
  } else {
    fo = f_abcd; val = v_abcd;
  }
  if (fo) return val;
  // all taps non-fluid: plain interpolation of the same 8 values (grid.cc:224-332 falls back to interpol)
  const float lo = (c[0] * L.t0 + c[1] * L.t1) * L.s0 + (c[2] * L.t0 + c[3] * L.t1) * L.s1;
  if (!IS3D) return lo;
  const float hi = (c[4] * L.t0 + c[5] * L.t1) * L.s0 + (c[6] * L.t0 + c[7] * L.t1) * L.s1;
  return lo * L.f0 + hi * L.f1;
}

template <bool IS3D>
__device__ __forceinline__ int t_blocked_at(const Dom& d, const TField<IS3D>& fl, v3 p) {
  const int i = (int)p.x, j = (int)p.y, k = (int)p.z;
  if (i < 0 || i >= d.X || j < 0 || j >= d.Y || k < 0 || k >= d.Z) return -1;
  return (((int)fl.at(d, i, j, k)) & kFluid) ? 0 : 1;
}

// calcLineTrace (tfl_device.hpp line_trace) with the flag tests served from the tile
template <bool IS3D>
__device__ inline int t_line_trace(const Dom& d, const TField<IS3D>& fl, v3 pos, v3 delta, v3& out) {
  out = pos;
  const float length = norm3(delta);
  if (length <= TFL_TRACE_EPS) return 0;
  const v3 dt = mk3(delta.x / length, delta.y / length, delta.z / length);
  float cur = 0.0f;
  while (cur < (length - TFL_HIT_MARGIN)) {
    const float step = stdmin(length - cur, 1.0f);
    v3 next = mk3(out.x + dt.x * step, out.y + dt.y * step, out.z + dt.z * step);
    if (out_of_domain(d, next)) {
      v3 ip;
      if (!ray_border(d, out, next, ip)) {
        ip.x = stdmin(stdmax(next.x, TFL_HIT_MARGIN), (float)d.X - TFL_HIT_MARGIN);
        ip.y = stdmin(stdmax(next.y, TFL_HIT_MARGIN), (float)d.Y - TFL_HIT_MARGIN);
        ip.z = stdmin(stdmax(next.z, TFL_HIT_MARGIN), (float)d.Z - TFL_HIT_MARGIN);
      }
      if (out_of_domain(d, ip)) return -3;
      const int blk = t_blocked_at<IS3D>(d, fl, ip);
      if (blk < 0) return -4;
      if (!blk) { out = ip; return 1; }
      next = ip;
    }
    int blk = t_blocked_at<IS3D>(d, fl, next);
    if (blk < 0) return -4;
    if (blk) {
      for (int count = 0; count <= 4; count++) {
        blk = t_blocked_at<IS3D>(d, fl, next);
        if (blk < 0) return -4;
        if (!blk) break;
        if (count == 4) return -5;
        const float cx = (float)((int)next.x) + 0.5f, cy = (float)((int)next.y) + 0.5f, cz = (float)((int)next.z) + 0.5f;
        const float lo[3] = {cx - 0.5f - TFL_HIT_MARGIN, cy - 0.5f - TFL_HIT_MARGIN, cz - 0.5f - TFL_HIT_MARGIN};
        const float hi[3] = {cx + 0.5f + TFL_HIT_MARGIN, cy + 0.5f + TFL_HIT_MARGIN, cz + 0.5f + TFL_HIT_MARGIN};
        const float org[3] = {out.x, out.y, out.z};
        const float dir[3] = {dt.x, dt.y, dt.z};
        float hitp[3];
        if (!ray_box(lo, hi, org, dir, hitp)) return 1;
        next = mk3(hitp[0], hitp[1], hitp[2]);
      }
      out = next;
      if (out_of_domain(d, out)) return -6;
      if (t_blocked_at<IS3D>(d, fl, out) != 0) return -7;
      return 1;
    }
    out = next;
    cur += step;
  }
  return 0;
}

// MAC averages of an interior cell: all taps are +-1 around the cell, i.e. always inside the tile.
template <bool IS3D>
__device__ __forceinline__ v3 t_get_centered(const float* tx, const float* ty, const float* tz, int l) {
  using TD = TileDims<IS3D>;
  v3 r;
  r.x = 0.5f * (tx[l] + tx[l + 1]);
  r.y = 0.5f * (ty[l] + ty[l + TD::NX]);
  r.z = IS3D ? 0.5f * (tz[l] + tz[l + TD::NX * TD::NY]) : 0.0f;
  return r;
}
template <bool IS3D, int AXIS>
__device__ __forceinline__ v3 t_get_at_mac(const float* Ux, const float* Uy, const float* Uz, int a) {
  using TD = TileDims<IS3D>;
  constexpr int sy = TD::NX, sz = TD::NX * TD::NY;
  v3 r;
  if (AXIS == 0) {
    r.x = Ux[a];
    r.y = 0.25f * (Uy[a] + Uy[a - 1] + Uy[a + sy] + Uy[a - 1 + sy]);
    r.z = IS3D ? 0.25f * (Uz[a] + Uz[a - 1] + Uz[a + sz] + Uz[a - 1 + sz]) : 0.0f;
  } else if (AXIS == 1) {
    r.x = 0.25f * (Ux[a] + Ux[a - sy] + Ux[a + 1] + Ux[a + 1 - sy]);
    r.y = Uy[a];
    r.z = IS3D ? 0.25f * (Uz[a] + Uz[a - sy] + Uz[a + sz] + Uz[a - sy + sz]) : 0.0f;
  } else {
    r.x = 0.25f * (Ux[a] + Ux[a - sz] + Ux[a + 1] + Ux[a + 1 - sz]);
    r.y = 0.25f * (Uy[a] + Uy[a - sz] + Uy[a + sy] + Uy[a + sy - sz]);
    r.z = IS3D ? Uz[a] : 0.0f;
  }
  return r;
}

__device__ __forceinline__ void mm2(float& lo, float& hi, float v) {
  if (v < lo) lo = v;
  if (v > hi) hi = v;
}

// doClampComponentMAC, tfluids.cc:701-746, with the 2 x 2^dim corners served from the tile
template <bool IS3D>
__device__ float t_manta_clamp_component(const Dom& d, float dst, const TField<IS3D>& f, float fwd, v3 pos, v3 vel) {
  using TD = TileDims<IS3D>;
  float lo = 3.402823466e+38f, hi = -3.402823466e+38f;
#pragma unroll
  for (int l = 0; l < 2; l++) {
    int px, py, pz;
    if (l == 0) { px = (int)(pos.x - vel.x); py = (int)(pos.y - vel.y); pz = (int)(pos.z - vel.z); }
    else { px = (int)(pos.x + vel.x); py = (int)(pos.y + vel.y); pz = (int)(pos.z + vel.z); }
    const int i0 = iclampi(px, 0, d.X - 2), j0 = iclampi(py, 0, d.Y - 2), k0 = iclampi(pz, 0, IS3D ? (d.Z - 2) : 1);
    const int i1 = i0 + 1, j1 = j0 + 1, k1 = IS3D ? k0 + 1 : k0;
    if (IS3D) { if (k0 < 0 || k1 >= d.Z) return fwd; }
    else if (k0 != 0 || k1 != 0) return fwd;
    if (i0 < 0 || j0 < 0 || i1 >= d.X || j1 >= d.Y) return fwd;
    if (f.inside(i0, j0, k0) && f.inside(i1, j1, k1)) {
      const float* p = f.t + f.lidx(i0, j0, k0);
      mm2(lo, hi, p[0]); mm2(lo, hi, p[1]); mm2(lo, hi, p[TD::NX]); mm2(lo, hi, p[1 + TD::NX]);
      if (IS3D) {
        const float* q = p + TD::NX * TD::NY;
        mm2(lo, hi, q[0]); mm2(lo, hi, q[1]); mm2(lo, hi, q[TD::NX]); mm2(lo, hi, q[1 + TD::NX]);
      }
    } else {
      const float* p = f.g + TFL_AT(d, i0, j0, k0);
      mm2(lo, hi, p[0]); mm2(lo, hi, p[1]); mm2(lo, hi, p[d.sy]); mm2(lo, hi, p[1 + d.sy]);
      if (IS3D) {
        const float* q = p + d.sz;
        mm2(lo, hi, q[0]); mm2(lo, hi, q[1]); mm2(lo, hi, q[d.sy]); mm2(lo, hi, q[1 + d.sy]);
      }
    }
  }
  return fclampf(dst, lo, hi);
}

// ---- block/tile bookkeeping ----------------------------------------------------------------------
template <bool IS3D>
struct BlockPos {
  int b, x0, y0, z0;  // batch item, tile origin (global cell coords of LDS element 0 = first halo cell)
  int i, j, k0;       // this thread's first cell
  __device__ BlockPos(const Dom& d) {
    using TD = TileDims<IS3D>;
    const int ntz = IS3D ? (d.Z + TD::BZ - 1) / TD::BZ : 1;
    b = blockIdx.z / ntz;
    const int bz = blockIdx.z - b * ntz;
    x0 = blockIdx.x * TD::BX - 1; y0 = blockIdx.y * TD::BY - 1; z0 = IS3D ? bz * TD::BZ - 1 : 0;
    i = blockIdx.x * TD::BX + (threadIdx.x % TD::BX);
    j = blockIdx.y * TD::BY + (threadIdx.x / TD::BX);
    k0 = IS3D ? bz * TD::BZ : 0;
  }
};

// ---- advectVel, maccormackOurs ---------------------------------------------------------------------
template <bool IS3D, int AXIS>
__device__ __forceinline__ float t_sl_mac_comp(const AdvArgsL& a, const TField<IS3D>& fl, const float* Ux, const float* Uy,
                                               const float* Uz, const TField<IS3D>& src, float dt, int i, int j, int k,
                                               int l) {
  const v3 ctr = cell_centre(i, j, k);
  const v3 u = t_get_at_mac<IS3D, AXIS>(Ux, Uy, Uz, l);
  v3 p;
  count_trace_error(t_line_trace<IS3D>(a.d, fl, ctr, scale3(u, -dt), p), a.err);
  return t_interpol<IS3D>(a.d, src, p);
}

template <bool IS3D>
__global__ __launch_bounds__(256) void k_vel_fwd_lds(AdvArgsL a, const float* __restrict__ U, const float* __restrict__ flags,
                                                     float* __restrict__ out) {
  using TD = TileDims<IS3D>;
  __shared__ float sU[3][TD::N];
  __shared__ float sF[TD::N];
  const Dom& d = a.d;
  const BlockPos<IS3D> bp(d);
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  U += bp.b * cells * C; flags += bp.b * cells; out += bp.b * cells * C;
#pragma unroll
  for (int c = 0; c < C; c++) stage<IS3D>(sU[c], U + c * cells, d, bp.x0, bp.y0, bp.z0);
  stage<IS3D>(sF, flags, d, bp.x0, bp.y0, bp.z0);
  __syncthreads();
  if (bp.i >= d.X || bp.j >= d.Y) return;
  TField<IS3D> fl = {sF, flags, bp.x0, bp.y0, bp.z0};
  TField<IS3D> fu[3];
#pragma unroll
  for (int c = 0; c < 3; c++) fu[c] = {sU[c < C ? c : 0], U + (c < C ? c : 0) * cells, bp.x0, bp.y0, bp.z0};
#pragma unroll 1
  for (int kk = 0; kk < TD::BZ; kk++) {
    const int i = bp.i, j = bp.j, k = bp.k0 + kk;
    if (k >= d.Z) break;
    const int o = TFL_AT(d, i, j, k);
    const int l = fl.lidx(i, j, k);
    float vx = 0.0f, vy = 0.0f, vz = 0.0f;
    if (!on_border<IS3D>(d, i, j, k)) {
      if (!(((int)sF[l]) & kFluid)) {
        vx = sU[0][l]; vy = sU[1][l]; if (IS3D) vz = sU[2][l];
      } else {
        vx = t_sl_mac_comp<IS3D, 0>(a, fl, sU[0], sU[1], sU[2], fu[0], a.dt, i, j, k, l);
        vy = t_sl_mac_comp<IS3D, 1>(a, fl, sU[0], sU[1], sU[2], fu[1], a.dt, i, j, k, l);
        if (IS3D) vz = t_sl_mac_comp<IS3D, 2>(a, fl, sU[0], sU[1], sU[2], fu[2], a.dt, i, j, k, l);
      }
    }
    out[o] = vx; out[o + d.sc] = vy; if (IS3D) out[o + 2 * d.sc] = vz;
  }
}

template <bool IS3D, int AXIS>
__device__ __forceinline__ float t_vel_bwd_comp(const AdvArgsL& a, const TField<IS3D>& fl, const float* Ux, const float* Uy,
                                                const float* Uz, const TField<IS3D>& fU, const TField<IS3D>& fF,
                                                bool border, bool fluid, bool skip, int i, int j, int k, int l) {
  const float f = fF.t[l];
  float bwd = 0.0f;
  if (!border) {
    if (!fluid) bwd = f;
    else bwd = t_sl_mac_comp<IS3D, AXIS>(a, fl, Ux, Uy, Uz, fF, -a.dt, i, j, k, l);
  }
  float v = f;
  if (!skip) v = (float)((double)f + (double)a.strength * 0.5 * (double)(fU.t[l] - bwd));  // tfluids.cc:693
  if (!border) {
    const v3 ud = scale3(t_get_at_mac<IS3D, AXIS>(Ux, Uy, Uz, l), a.dt);
    v = t_manta_clamp_component<IS3D>(a.d, v, fU, f, mk3((float)i, (float)j, (float)k), ud);
  }
  return v;
}

template <bool IS3D>
__global__ __launch_bounds__(256) void k_vel_bwd_lds(AdvArgsL a, const float* __restrict__ U, const float* __restrict__ flags,
                                                     const float* __restrict__ fwd, float* __restrict__ dst) {
  using TD = TileDims<IS3D>;
  __shared__ float sU[3][TD::N];
  __shared__ float sW[3][TD::N];
  __shared__ float sF[TD::N];
  const Dom& d = a.d;
  const BlockPos<IS3D> bp(d);
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  U += bp.b * cells * C; flags += bp.b * cells; fwd += bp.b * cells * C; dst += bp.b * cells * C;
#pragma unroll
  for (int c = 0; c < C; c++) {
    stage<IS3D>(sU[c], U + c * cells, d, bp.x0, bp.y0, bp.z0);
    stage<IS3D>(sW[c], fwd + c * cells, d, bp.x0, bp.y0, bp.z0);
  }
  stage<IS3D>(sF, flags, d, bp.x0, bp.y0, bp.z0);
  __syncthreads();
  if (bp.i >= d.X || bp.j >= d.Y) return;
  TField<IS3D> fl = {sF, flags, bp.x0, bp.y0, bp.z0};
  TField<IS3D> fu[3], fw[3];
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const int cc = c < C ? c : 0;
    fu[c] = {sU[cc], U + cc * cells, bp.x0, bp.y0, bp.z0};
    fw[c] = {sW[cc], fwd + cc * cells, bp.x0, bp.y0, bp.z0};
  }
#pragma unroll 1
  for (int kk = 0; kk < TD::BZ; kk++) {
    const int i = bp.i, j = bp.j, k = bp.k0 + kk;
    if (k >= d.Z) break;
    const int o = TFL_AT(d, i, j, k);
    const int l = fl.lidx(i, j, k);
    const bool border = on_border<IS3D>(d, i, j, k);
    const bool fluid = ((int)sF[l]) & kFluid;
    // -c neighbours are inside the tile (halo 1) whenever they exist
    const bool sx = !fluid || (i > 0 && !(((int)sF[l - 1]) & kFluid));
    const bool sy = !fluid || (j > 0 && !(((int)sF[l - TD::NX]) & kFluid));
    dst[o] = t_vel_bwd_comp<IS3D, 0>(a, fl, sU[0], sU[1], sU[2], fu[0], fw[0], border, fluid, sx, i, j, k, l);
    dst[o + d.sc] = t_vel_bwd_comp<IS3D, 1>(a, fl, sU[0], sU[1], sU[2], fu[1], fw[1], border, fluid, sy, i, j, k, l);
    if (IS3D) {
      const bool sz = !fluid || (k > 0 && !(((int)sF[l - TD::NX * TD::NY]) & kFluid));
      dst[o + 2 * d.sc] = t_vel_bwd_comp<IS3D, 2>(a, fl, sU[0], sU[1], sU[2], fu[2], fw[2], border, fluid, sz, i, j, k, l);
    }
  }
}

// ---- advectScalar, maccormackOurs -------------------------------------------------------------------
template <bool IS3D>
__global__ __launch_bounds__(256) void k_scalar_fwd_lds(AdvArgsL a, const float* __restrict__ s, const float* __restrict__ U,
                                                        const float* __restrict__ flags, float* __restrict__ out,
                                                        float* __restrict__ bounds, const float* __restrict__ lo3,
                                                        const float* __restrict__ hi3) {
  using TD = TileDims<IS3D>;
  __shared__ float sU[3][TD::N];
  __shared__ float sS[TD::N];
  __shared__ float sF[TD::N];
  const Dom& d = a.d;
  const BlockPos<IS3D> bp(d);
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  s += bp.b * cells; flags += bp.b * cells; out += bp.b * cells; U += bp.b * cells * C; bounds += bp.b * cells * C;
#pragma unroll
  for (int c = 0; c < C; c++) stage<IS3D>(sU[c], U + c * cells, d, bp.x0, bp.y0, bp.z0);
  stage<IS3D>(sS, s, d, bp.x0, bp.y0, bp.z0);
  stage<IS3D>(sF, flags, d, bp.x0, bp.y0, bp.z0);
  __syncthreads();
  if (bp.i >= d.X || bp.j >= d.Y) return;
  TField<IS3D> fl = {sF, flags, bp.x0, bp.y0, bp.z0};
  TField<IS3D> fs = {sS, s, bp.x0, bp.y0, bp.z0};
#pragma unroll 1
  for (int kk = 0; kk < TD::BZ; kk++) {
    const int i = bp.i, j = bp.j, k = bp.k0 + kk;
    if (k >= d.Z) break;
    const int o = TFL_AT(d, i, j, k);
    if (on_border<IS3D>(d, i, j, k)) { out[o] = 0.0f; continue; }
    const int l = fl.lidx(i, j, k);
    v3 back = cell_centre(i, j, k);
    float v;
    if (!(((int)sF[l]) & kFluid)) {
      v = sS[l];
    } else {
      const v3 disp = scale3(t_get_centered<IS3D>(sU[0], sU[1], sU[2], l), -a.dt);
      count_trace_error(t_line_trace<IS3D>(d, fl, back, disp, back), a.err);
      v = a.outside ? t_interpol<IS3D>(d, fs, back) : t_interpol_with_fluid<IS3D>(d, fs, fl, back);
    }
    const int i0 = iclampi((int)back.x, 0, d.X - 1), j0 = iclampi((int)back.y, 0, d.Y - 1);
    const int k0 = IS3D ? iclampi((int)back.z, 0, d.Z - 1) : 0;
    const long long g = bp.b * cells + TFL_AT(d, i0, j0, k0);
    bounds[o] = lo3[g];
    bounds[o + d.sc] = hi3[g];
    out[o] = v;
  }
}

template <bool IS3D>
__global__ __launch_bounds__(256) void k_scalar_bwd_lds(AdvArgsL a, const float* __restrict__ s, const float* __restrict__ U,
                                                        const float* __restrict__ flags, const float* __restrict__ fwd,
                                                        const float* __restrict__ bounds, float* __restrict__ dst) {
  using TD = TileDims<IS3D>;
  __shared__ float sU[3][TD::N];
  __shared__ float sW[TD::N];
  __shared__ float sF[TD::N];
  const Dom& d = a.d;
  const BlockPos<IS3D> bp(d);
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  s += bp.b * cells; flags += bp.b * cells; fwd += bp.b * cells; dst += bp.b * cells; U += bp.b * cells * C;
  bounds += bp.b * cells * C;
#pragma unroll
  for (int c = 0; c < C; c++) stage<IS3D>(sU[c], U + c * cells, d, bp.x0, bp.y0, bp.z0);
  stage<IS3D>(sW, fwd, d, bp.x0, bp.y0, bp.z0);
  stage<IS3D>(sF, flags, d, bp.x0, bp.y0, bp.z0);
  __syncthreads();
  if (bp.i >= d.X || bp.j >= d.Y) return;
  TField<IS3D> fl = {sF, flags, bp.x0, bp.y0, bp.z0};
  TField<IS3D> fw = {sW, fwd, bp.x0, bp.y0, bp.z0};
#pragma unroll 1
  for (int kk = 0; kk < TD::BZ; kk++) {
    const int i = bp.i, j = bp.j, k = bp.k0 + kk;
    if (k >= d.Z) break;
    const int o = TFL_AT(d, i, j, k);
    const int l = fl.lidx(i, j, k);
    const bool border = on_border<IS3D>(d, i, j, k);
    const bool fluid = ((int)sF[l]) & kFluid;
    const float f = sW[l];
    float bwd;
    if (border) bwd = 0.0f;
    else if (!fluid) bwd = f;
    else {
      v3 back;
      const v3 disp = scale3(t_get_centered<IS3D>(sU[0], sU[1], sU[2], l), a.dt);   // -(-dt)
      count_trace_error(t_line_trace<IS3D>(d, fl, cell_centre(i, j, k), disp, back), a.err);
      bwd = a.outside ? t_interpol<IS3D>(d, fw, back) : t_interpol_with_fluid<IS3D>(d, fw, fl, back);
    }
    float v = f;
    if (fluid) v = (float)((double)f + (double)a.strength * 0.5 * (double)(s[o] - bwd));  // tfluids.cc:231
    if (!border) {
      const float lo = bounds[o], hi = bounds[o + d.sc];
      v = (lo > hi) ? f : fclampf(v, lo, hi);
    }
    dst[o] = v;
  }
}

// ---- host launchers ----------------------------------------------------------------------------------
template <bool IS3D>
static dim3 tile_grid(const Dom& d, int B) {
  using TD = TileDims<IS3D>;
  const int ntz = IS3D ? (d.Z + TD::BZ - 1) / TD::BZ : 1;
  return dim3((d.X + TD::BX - 1) / TD::BX, (d.Y + TD::BY - 1) / TD::BY, (unsigned)(ntz * B));
}

void advect_vel_ours_lds(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float dt, float strength,
                         unsigned long long* err, const float* U, const float* flags, float* fwd, float* dst) {
  AdvArgsL a; a.d = make_dom(Z, Y, X); a.dt = dt; a.strength = strength; a.outside = 0; a.err = err;
  if (is3d) {
    const dim3 g = tile_grid<true>(a.d, B);
    { TFL_TIMED("k_vel_fwd", st); k_vel_fwd_lds<true><<<g, 256, 0, st>>>(a, U, flags, fwd); }
    { TFL_TIMED("k_vel_bwd", st); k_vel_bwd_lds<true><<<g, 256, 0, st>>>(a, U, flags, fwd, dst); }
  } else {
    const dim3 g = tile_grid<false>(a.d, B);
    { TFL_TIMED("k_vel_fwd", st); k_vel_fwd_lds<false><<<g, 256, 0, st>>>(a, U, flags, fwd); }
    { TFL_TIMED("k_vel_bwd", st); k_vel_bwd_lds<false><<<g, 256, 0, st>>>(a, U, flags, fwd, dst); }
  }
}

void advect_scalar_ours_lds(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float dt, float strength,
                            int outside, unsigned long long* err, const float* s, const float* U, const float* flags,
                            float* fwd, float* bounds, const float* lo3, const float* hi3, float* dst) {
  AdvArgsL a; a.d = make_dom(Z, Y, X); a.dt = dt; a.strength = strength; a.outside = outside; a.err = err;
  if (is3d) {
    const dim3 g = tile_grid<true>(a.d, B);
    { TFL_TIMED("k_scalar_fwd", st); k_scalar_fwd_lds<true><<<g, 256, 0, st>>>(a, s, U, flags, fwd, bounds, lo3, hi3); }
    { TFL_TIMED("k_scalar_bwd", st); k_scalar_bwd_lds<true><<<g, 256, 0, st>>>(a, s, U, flags, fwd, bounds, dst); }
  } else {
    const dim3 g = tile_grid<false>(a.d, B);
    { TFL_TIMED("k_scalar_fwd", st); k_scalar_fwd_lds<false><<<g, 256, 0, st>>>(a, s, U, flags, fwd, bounds, lo3, hi3); }
    { TFL_TIMED("k_scalar_bwd", st); k_scalar_bwd_lds<false><<<g, 256, 0, st>>>(a, s, U, flags, fwd, bounds, dst); }
  }
}

}  // namespace tfl
