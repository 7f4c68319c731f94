// stencil.hip -- the one-sweep MAC-grid operators (gfx950): setWallBcs, velocityDivergence,
// velocityUpdate, addBuoyancy, addGravity, emptyDomain, flagsToOccupancy.
//
// All are HBM-bound: each cell reads its own words plus the -c / +c face neighbours, which the
// neighbouring lanes (x) or the L2 (y, z rows) already hold. One thread per cell, x fastest, 64x4
// thread blocks => 256-B coalesced rows. Algorithmic bytes per cell (3-D): setWallBcs 28,
// divergence 20, velocityUpdate 32, addBuoyancy 32.
#include "tfl_device.hpp"
#include "tfl_host.hpp"

#include <cstdlib>

namespace tfl {

#define TFL_STENCIL_INDEX()                                          \
  const int i = blockIdx.x * blockDim.x + threadIdx.x;               \
  const int j = blockIdx.y * blockDim.y + threadIdx.y;               \
  int b, k; dom_bk(d, b, k);                                         \
  if (i >= d.X || j >= d.Y) return;                                  \
  const long long cells = (long long)d.sc;                           \
  const int C = IS3D ? 3 : 2;                                        \
  (void)C; (void)cells;                                              \
  const int o = TFL_AT(d, i, j, k)

// third_party/tfluids.cc:926-1002
template <bool IS3D>
__global__ __launch_bounds__(256) void k_set_wall_bcs(Dom d, float* __restrict__ U, const float* __restrict__ flags) {
  TFL_STENCIL_INDEX();
  U += b * cells * C; flags += b * cells;
  // the seven flag words in ONE memory round trip (round 4: each was a guarded load of its own, seven round trips in a row --
  // DESIGN.md 3.6): a neighbour that does not exist reads the cell itself and counts as 0
  const int fc = (int)flags[o];
  const float gxm = flags[i > 0 ? o - 1 : o], gym = flags[j > 0 ? o - d.sy : o], gzm = flags[(IS3D && k > 0) ? o - d.sz : o];
  const float gxp = flags[i < d.X - 1 ? o + 1 : o], gyp = flags[j < d.Y - 1 ? o + d.sy : o];
  const float gzp = flags[(IS3D && k < d.Z - 1) ? o + d.sz : o];
  const int fxm = i > 0 ? (int)gxm : 0, fym = j > 0 ? (int)gym : 0, fzm = (IS3D && k > 0) ? (int)gzm : 0;
  const int fxp = i < d.X - 1 ? (int)gxp : 0, fyp = j < d.Y - 1 ? (int)gyp : 0, fzp = (IS3D && k < d.Z - 1) ? (int)gzp : 0;
  const bool cf = fc & kFluid, co = fc & kObstacle;
  bool zx = (fxm & kObstacle) || (co && (fxm & kFluid));
  bool zy = (fym & kObstacle) || (co && (fym & kFluid));
  bool zz = IS3D && ((fzm & kObstacle) || (co && (fzm & kFluid)));
  if (cf) {
    if ((fxm & kStick) || (fxp & kStick)) { zy = true; zz = IS3D; }
    if ((fym & kStick) || (fyp & kStick)) { zx = true; zz = zz || IS3D; }
    if (IS3D && ((fzm & kStick) || (fzp & kStick))) { zx = true; zy = true; }
  }
  const bool act = cf || co;
  if (act && zx) U[o] = 0.0f;
  if (act && zy) U[o + d.sc] = 0.0f;
  if (IS3D && act && zz) U[o + 2 * d.sc] = 0.0f;
}

// third_party/tfluids.cc:1008-1066 (negative divergence, Manta makeRhs convention)
template <bool IS3D>
__global__ __launch_bounds__(256) void k_divergence(Dom d, const float* __restrict__ U, const float* __restrict__ flags,
                                                    float* __restrict__ div) {
  TFL_STENCIL_INDEX();
  U += b * cells * C; flags += b * cells; div += b * cells;
  float v = 0.0f;
  if (!on_border<IS3D>(d, i, j, k) && (((int)flags[o]) & kFluid)) {
    v = U[o] - U[o + 1] + U[o + d.sc] - U[o + d.sc + d.sy];
    if (IS3D) v += (U[o + 2 * d.sc] - U[o + 2 * d.sc + d.sz]);
  }
  div[o] = v;
}

// third_party/tfluids.cc:1072-1156
template <bool IS3D>
__global__ __launch_bounds__(256) void k_velocity_update(Dom d, float* __restrict__ U, const float* __restrict__ flags,
                                                         const float* __restrict__ p) {
  TFL_STENCIL_INDEX();
  if (on_border<IS3D>(d, i, j, k)) return;
  U += b * cells * C; flags += b * cells; p += b * cells;
  // every operand in ONE memory round trip (round 4, DESIGN.md 3.6: the pressure and velocity loads sat behind the flag tests)
  const int fc = (int)flags[o];
  const int fx = (int)flags[o - 1], fy = (int)flags[o - d.sy], fz = IS3D ? (int)flags[o - d.sz] : 0;
  const float pc = p[o], pxm = p[o - 1], pym = p[o - d.sy], pzm = IS3D ? p[o - d.sz] : 0.0f;
  const float ux0 = U[o], uy0 = U[o + d.sc], uz0 = IS3D ? U[o + 2 * d.sc] : 0.0f;
  if (fc & kFluid) {
    float ux = ux0, uy = uy0;
    if (fx & kFluid) ux -= (pc - pxm);
    if (fy & kFluid) uy -= (pc - pym);
    if (fx & kEmpty) ux -= pc;
    if (fy & kEmpty) uy -= pc;
    U[o] = ux; U[o + d.sc] = uy;
    if (IS3D) {
      float uz = uz0;
      if (fz & kFluid) uz -= (pc - pzm);
      if (fz & kEmpty) uz -= pc;
      U[o + 2 * d.sc] = uz;
    }
  } else if ((fc & kEmpty) && !(fc & kOutflow)) {
    U[o] = (fx & kFluid) ? ux0 + pxm : 0.0f;
    U[o + d.sc] = (fy & kFluid) ? uy0 + pym : 0.0f;
    if (IS3D) U[o + 2 * d.sc] = (fz & kFluid) ? uz0 + pzm : 0.0f;
  }
}

// third_party/tfluids.cc:1162-1233; (sx,sy,sz) = -gravity * dt / dx computed on the host
template <bool IS3D>
__global__ __launch_bounds__(256) void k_add_buoyancy(Dom d, float* __restrict__ U, const float* __restrict__ flags,
                                                      const float* __restrict__ rho, float sx, float sy, float sz) {
  TFL_STENCIL_INDEX();
  if (on_border<IS3D>(d, i, j, k)) return;
  U += b * cells * C; flags += b * cells; rho += b * cells;
  if (!(((int)flags[o]) & kFluid)) return;
  const float r = rho[o];
  if (((int)flags[o - 1]) & kFluid) U[o] += (0.5f * sx * (r + rho[o - 1]));
  if (((int)flags[o - d.sy]) & kFluid) U[o + d.sc] += (0.5f * sy * (r + rho[o - d.sy]));
  if (IS3D && (((int)flags[o - d.sz]) & kFluid)) U[o + 2 * d.sc] += (0.5f * sz * (r + rho[o - d.sz]));
}

// The same, four consecutive x cells per thread (X % 4 == 0, 16-byte aligned rows): every access is a
// 16-byte vector; the x-1 neighbour of the first cell is one extra scalar load. Identical arithmetic per cell.
template <bool IS3D>
__global__ __launch_bounds__(256) void k_add_buoyancy_v4(Dom d, const float* __restrict__ Usrc, float* __restrict__ U,
                                                         const float* __restrict__ flags, const float* __restrict__ rho,
                                                         float sx, float sy, float sz, BcFoldArg folda) {
  // U = Usrc + buoyancy. Usrc == U: the reference's in-place op. Usrc != U (tfl_addBuoyancyFrom): every cell is
  // written, which folds the `U:copy(advected)` that precedes it in simulate() into this pass.
  const int i0 = (blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  if (i0 >= d.X || j >= d.Y) return;
  const bool inner = !(j < 1 || j > d.Y - 2 || (IS3D && (k < 1 || k > d.Z - 2)));
  if (!inner && Usrc == U) return;
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  U += b * cells * C; Usrc += b * cells * C; flags += b * cells; rho += b * cells;
  const int o = TFL_AT(d, i0, j, k);
  const float4 z4 = make_float4(0, 0, 0, 0);
  float4 f4 = z4, r4 = z4, fy4 = z4, ry4 = z4, fz4 = z4, rz4 = z4;
  float fm = 0.0f, rm = 0.0f;
  if (inner) {
    f4 = *reinterpret_cast<const float4*>(flags + o);
    r4 = *reinterpret_cast<const float4*>(rho + o);
    fy4 = *reinterpret_cast<const float4*>(flags + o - d.sy);
    ry4 = *reinterpret_cast<const float4*>(rho + o - d.sy);
    if (IS3D) { fz4 = *reinterpret_cast<const float4*>(flags + o - d.sz); rz4 = *reinterpret_cast<const float4*>(rho + o - d.sz); }
    if (i0 > 0) { fm = flags[o - 1]; rm = rho[o - 1]; }
  }
  float4 ux = *reinterpret_cast<const float4*>(Usrc + o);
  float4 uy = *reinterpret_cast<const float4*>(Usrc + o + d.sc);
  float4 uz = z4;
  if (IS3D) uz = *reinterpret_cast<const float4*>(Usrc + o + 2 * d.sc);
  const float fc[4] = {f4.x, f4.y, f4.z, f4.w}, rc[4] = {r4.x, r4.y, r4.z, r4.w};
  const float fxm[4] = {fm, f4.x, f4.y, f4.z}, rxm[4] = {rm, r4.x, r4.y, r4.z};
  const float fym[4] = {fy4.x, fy4.y, fy4.z, fy4.w}, rym[4] = {ry4.x, ry4.y, ry4.z, ry4.w};
  const float fzm[4] = {fz4.x, fz4.y, fz4.z, fz4.w}, rzm[4] = {rz4.x, rz4.y, rz4.z, rz4.w};
  float vx[4] = {ux.x, ux.y, ux.z, ux.w}, vy[4] = {uy.x, uy.y, uy.z, uy.w}, vz[4] = {uz.x, uz.y, uz.z, uz.w};
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int i = i0 + q;
    if (!inner || i < 1 || i > d.X - 2) continue;            // border shell
    if (!(((int)fc[q]) & kFluid)) continue;
    if (((int)fxm[q]) & kFluid) vx[q] += (0.5f * sx * (rc[q] + rxm[q]));
    if (((int)fym[q]) & kFluid) vy[q] += (0.5f * sy * (rc[q] + rym[q]));
    if (IS3D && (((int)fzm[q]) & kFluid)) vz[q] += (0.5f * sz * (rc[q] + rzm[q]));
  }
  // the setConstVals that follows the forces in simulate() (tfl_host.hpp BcFold; only asked for when Usrc != U: every cell written)
  BcFold fold = {nullptr, nullptr, 0, -1, 0, -1, 0, -1};
  const bool fold_blk = fold_block(folda, (int)(blockIdx.y * blockDim.y), (int)(blockIdx.y * blockDim.y + blockDim.y - 1), k, k);
  if (fold_blk) fold = *folda.dev;
  if (fold_blk && fold_row(fold, j, k) && i0 <= fold.x1 && i0 + 3 >= fold.x0) {
    const float* fb = fold.bc + b * cells * C + o;
    const float* fm = fold.inv + b * cells * C + o;
#pragma unroll
    for (int a = 0; a < 3; a++) {
      if (a >= C) continue;
      const float4 b4 = *reinterpret_cast<const float4*>(fb + a * d.sc), m4 = *reinterpret_cast<const float4*>(fm + a * d.sc);
      const float bb[4] = {b4.x, b4.y, b4.z, b4.w}, mm[4] = {m4.x, m4.y, m4.z, m4.w};
      float* v = a == 0 ? vx : (a == 1 ? vy : vz);
#pragma unroll
      for (int q = 0; q < 4; q++)
        if (fold_col(fold, i0 + q)) v[q] = v[q] * mm[q] + bb[q];
    }
  }
  *reinterpret_cast<float4*>(U + o) = make_float4(vx[0], vx[1], vx[2], vx[3]);
  *reinterpret_cast<float4*>(U + o + d.sc) = make_float4(vy[0], vy[1], vy[2], vy[3]);
  if (IS3D) *reinterpret_cast<float4*>(U + o + 2 * d.sc) = make_float4(vz[0], vz[1], vz[2], vz[3]);
}

// third_party/tfluids.cc:1239-1306
template <bool IS3D>
__global__ __launch_bounds__(256) void k_add_gravity(Dom d, float* __restrict__ U, const float* __restrict__ flags,
                                                     float fx, float fy, float fz) {
  TFL_STENCIL_INDEX();
  if (on_border<IS3D>(d, i, j, k)) return;
  U += b * cells * C; flags += b * cells;
  const int fc = (int)flags[o];
  const bool cf = fc & kFluid, ce = fc & kEmpty;
  if (!cf && !ce) return;
  const int nx = (int)flags[o - 1], ny = (int)flags[o - d.sy], nz = IS3D ? (int)flags[o - d.sz] : 0;
  if ((nx & kFluid) || (cf && (nx & kEmpty))) U[o] += fx;
  if ((ny & kFluid) || (cf && (ny & kEmpty))) U[o + d.sc] += fy;
  if (IS3D && ((nz & kFluid) || (cf && (nz & kEmpty)))) U[o + 2 * d.sc] += fz;
}

// generic/tfluids.cc:136-167
template <bool IS3D>
__global__ __launch_bounds__(256) void k_empty_domain(Dom d, int bnd, float* __restrict__ flags) {
  TFL_STENCIL_INDEX();
  flags += b * cells;
  const bool border = i < bnd || i > d.X - 1 - bnd || j < bnd || j > d.Y - 1 - bnd ||
                      (IS3D && (k < bnd || k > d.Z - 1 - bnd));
  flags[o] = border ? (float)kObstacle : (float)kFluid;
}

// generic/tfluids.cc:173-210 | generic/tfluids.cu:355-371 (other cell types -> -1 like the CUDA path)
__global__ __launch_bounds__(256) void k_flags_to_occupancy(long long n, const float* __restrict__ flags,
                                                            float* __restrict__ occ) {
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int f = (int)flags[t];
    occ[t] = (f == kFluid) ? 0.0f : ((f == kObstacle) ? 1.0f : -1.0f);
  }
}

// *out = max(*out, max_i |x[i]|): the z-slab reach check (max |u_z| * dt bounds how many planes a back-trace crosses).
// Non-negative floats order like their bit patterns, so the maximum is one integer atomic per block.
// (Round 6: ONE atomic per block of at most 256 blocks, 16-byte loads -- the first form issued one same-address atomic per WAVE of
// 1024 blocks, 4096 of them serialised in L2: 55 us on a 24-plane slab, more than half of the rank-step it was guarding.)
__global__ __launch_bounds__(256) void k_absmax(long long n, const float* __restrict__ x, float* __restrict__ out) {
  float m = 0.0f;
  const long long n4 = ((reinterpret_cast<uintptr_t>(x) & 15) == 0) ? n / 4 : 0;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n4; t += (long long)gridDim.x * blockDim.x) {
    const float4 v = x4[t];
    m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  for (long long t = n4 * 4 + (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(x[t]));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_down(m, off, 64));
  __shared__ float wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(reinterpret_cast<unsigned int*>(out), __float_as_uint(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]))));
}

// flags[r - 1] = 1.0 where *maxu * dt >= r (r = 1 .. n): the one-hot form of "back-trace reach needed" that a SUM all-reduce can
// combine over ranks (tfl_simulate_step_slab, check_reach = 2); the product is formed in fp32 like the host-side check
// flags[n] (range_count given) = this rank's conv stack has clamped activations at the fp16 range since the count was last read
__global__ void k_reach_flags(const float* __restrict__ maxu, float dt, int n, double* __restrict__ flags, const unsigned long long* __restrict__ range_count) {
  const int r = threadIdx.x + 1;
  if (r <= n) flags[r - 1] = (*maxu * dt >= (float)r) ? 1.0 : 0.0;
  if (r == n + 1) flags[n] = (range_count && *range_count != 0ull) ? 1.0 : 0.0;
}
void reach_flags(hipStream_t st, const float* maxu, float dt, int n, double* flags, const unsigned long long* range_count) {
  k_reach_flags<<<1, 64, 0, st>>>(maxu, dt, n, flags, range_count);
}

void absmax(hipStream_t st, long long n, const float* x, float* out, bool reset) {
  if (reset) (void)hipMemsetAsync(out, 0, sizeof(float), st);
  const long long want = (n / 4 + 255) / 256;
  const int blocks = (int)(want < 256 ? want : 256);
  { TFL_TIMED("k_absmax", st); k_absmax<<<blocks > 0 ? blocks : 1, 256, 0, st>>>(n, x, out); }
}

// The yard-stick of every HBM fraction bench.py reports: a plain streaming copy, 16 bytes per lane and access, four
// independent accesses per thread in flight (the guide's "float4 copy": ~6.3 of the 8 TB/s pin rate on this part).
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream_copy(const f4v* __restrict__ src, f4v* __restrict__ dst, long long n4) {
  const long long i0 = (long long)blockIdx.x * 1024 + threadIdx.x;
  f4v v[4];
#pragma unroll
  for (int u = 0; u < 4; u++) { const long long i = i0 + u * 256; if (i < n4) v[u] = __builtin_nontemporal_load(src + i); }
#pragma unroll
  for (int u = 0; u < 4; u++) { const long long i = i0 + u * 256; if (i < n4) __builtin_nontemporal_store(v[u], dst + i); }
}
void stream_copy(hipStream_t st, long long n4, const float* src, float* dst) {
  if (n4 <= 0) return;
  TFL_TIMED_EXT("k_stream_copy", st);
  TFL_LAUNCH_EXT(k_stream_copy, (unsigned)((n4 + 1023) / 1024), 256, 0, st, reinterpret_cast<const f4v*>(src),
                 reinterpret_cast<f4v*>(dst), n4);
}

static inline dim3 cgrid(const Dom& d, int B, dim3 blk) {
  return dim3((d.X + blk.x - 1) / blk.x, (d.Y + blk.y - 1) / blk.y, (unsigned)(d.nw * B));
}
#define TFL_LAUNCH(kern, ...)                                        \
  do {                                                               \
    const Dom d = make_dom(Z, Y, X);                                 \
    const dim3 blk(64, 4, 1), grd = cgrid(d, B, blk);                \
    TFL_TIMED(#kern, st);                                            \
    if (is3d) kern<true><<<grd, blk, 0, st>>>(d, __VA_ARGS__);       \
    else kern<false><<<grd, blk, 0, st>>>(d, __VA_ARGS__);           \
  } while (0)

void set_wall_bcs(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* U, const float* flags) {
  TFL_LAUNCH(k_set_wall_bcs, U, flags);
}
void velocity_divergence(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* U, const float* flags,
                         float* div) {
  TFL_LAUNCH(k_divergence, U, flags, div);
}
void velocity_update(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* U, const float* flags,
                     const float* p) {
  TFL_LAUNCH(k_velocity_update, U, flags, p);
}
void add_buoyancy(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* Usrc, float* U, const float* flags,
                  const float* density, float sx, float sy, float sz) {
  const bool vec = (X % 4 == 0) && ((((uintptr_t)U | (uintptr_t)Usrc | (uintptr_t)flags | (uintptr_t)density) & 15) == 0) &&
                   !exp_env("TFL_NO_VEC4");
  if (vec) {
    const Dom d = make_dom(Z, Y, X);
    const dim3 blk(32, 8, 1), grd((X / 4 + 31) / 32, (Y + 7) / 8, (unsigned)(d.nw * B));
    TFL_TIMED_EXT("k_add_buoyancy", st);
    const BcFoldArg fold = Usrc != U ? take_fold() : no_fold();
    if (is3d) TFL_LAUNCH_EXT((k_add_buoyancy_v4<true>), grd, blk, 0, st, d, Usrc, U, flags, density, sx, sy, sz, fold);
    else TFL_LAUNCH_EXT((k_add_buoyancy_v4<false>), grd, blk, 0, st, d, Usrc, U, flags, density, sx, sy, sz, fold);
    return;
  }
  if (Usrc != U)   // the one-cell-per-thread kernel skips the cells it does not change
    (void)hipMemcpyAsync(U, Usrc, sizeof(float) * (size_t)B * (is3d ? 3 : 2) * Z * Y * X, hipMemcpyDeviceToDevice, st);
  TFL_LAUNCH(k_add_buoyancy, U, flags, density, sx, sy, sz);
}
void add_gravity(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* U, const float* flags, float fx,
                 float fy, float fz) {
  TFL_LAUNCH(k_add_gravity, U, flags, fx, fy, fz);
}
// ---- rectangularBlur / signedDistanceField (generic/tfluids.cc:642-821): criterion-side helpers, not on the step ----------
// One thread per line; the running sum is sequential along the line in the reference's order (bit-exact). Lines of a pass:
// base(line) = (line / n_inner) * outer + (line % n_inner); z pass n_inner = Y*X, y pass n_inner = X (threads walk
// neighbouring lines: coalesced), x pass n_inner = 1 (a thread owns a row).
__global__ __launch_bounds__(256) void k_blur_axis(const float* __restrict__ src, float* __restrict__ dst, long long lines,
                                                   long long n_inner, long long outer, int size, long long stride, int rad) {
  const long long line = (long long)blockIdx.x * 256 + threadIdx.x;
  if (line >= lines) return;
  const long long base = (line / n_inner) * outer + (line % n_inner);
  const float* s = src + base;
  float* o = dst + base;
  float val = s[0] * (float)(rad + 1);
  for (int i = 0; i < size && i < rad; i++) val += s[i * stride];
  const float mul_const = 1.0f / (float)(rad * 2 + 1);
  for (int i = 0; i < size; i++) {
    const int iminus = max(0, i - rad - 1), iplus = min(size - 1, i + rad);
    val -= s[iminus * stride];
    val += s[iplus * stride];
    o[i * stride] = val * mul_const;
  }
}

__global__ __launch_bounds__(256) void k_signed_distance(int rad, int Z, int Y, int X, const float* __restrict__ flags,
                                                         float* __restrict__ dst) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y * blockDim.y + threadIdx.y;
  const int b = blockIdx.z / Z, z = blockIdx.z - b * Z;
  if (x >= X || y >= Y) return;
  const long long N = (long long)Z * Y * X;
  const float* f = flags + b * N;
  const long long o = (long long)z * Y * X + (long long)y * X + x;
  if ((int)f[o] & kObstacle) { dst[b * N + o] = 0.0f; return; }
  float dist_sq = (float)(rad * rad);
  for (int zs = max(0, z - rad); zs <= min(Z - 1, z + rad); zs++)
    for (int ys = max(0, y - rad); ys <= min(Y - 1, y + rad); ys++)
      for (int xs = max(0, x - rad); xs <= min(X - 1, x + rad); xs++)
        if ((int)f[(long long)zs * Y * X + (long long)ys * X + xs] & kObstacle) {
          const float cur = (float)((z - zs) * (z - zs) + (y - ys) * (y - ys) + (x - xs) * (x - xs));
          if (dist_sq > cur) dist_sq = cur;
        }
  dst[b * N + o] = sqrtf(dist_sq);
}

void rectangular_blur(hipStream_t st, bool is3d, int B, int C, int Z, int Y, int X, int rad, const float* src, float* dst,
                      float* tmp) {
  const long long F = (long long)B * C, sy = X, sz = (long long)Y * X, sf = (long long)Z * Y * X;
  auto pass = [&](const float* in, float* out, long long lines, long long n_inner, long long outer, int size, long long stride) {
    TFL_TIMED("k_blur_axis", st);
    k_blur_axis<<<(unsigned)((lines + 255) / 256), 256, 0, st>>>(in, out, lines, n_inner, outer, size, stride, rad);
  };
  const float* cur_src = src;
  float* cur_dst = is3d ? dst : tmp;
  if (is3d) {
    pass(cur_src, cur_dst, F * Y * X, (long long)Y * X, sf, Z, sz);
    cur_src = dst; cur_dst = tmp;
  }
  pass(cur_src, cur_dst, F * Z * X, X, sz, Y, sy);
  pass(tmp, dst, F * Z * Y, 1, X, X, 1);
}

void signed_distance_field(hipStream_t st, int B, int Z, int Y, int X, int rad, const float* flags, float* dst) {
  const dim3 blk(64, 4, 1), grd((X + 63) / 64, (Y + 3) / 4, (unsigned)(Z * B));
  TFL_TIMED("k_signed_distance", st);
  k_signed_distance<<<grd, blk, 0, st>>>(rad, Z, Y, X, flags, dst);
}

void empty_domain(hipStream_t st, bool is3d, int bnd, int B, int Z, int Y, int X, float* flags) {
  TFL_LAUNCH(k_empty_domain, bnd, flags);
}
void flags_to_occupancy(hipStream_t st, long long numel, const float* flags, float* occ) {
  const int blocks = (int)((numel + 255) / 256 < 2048 ? (numel + 255) / 256 : 2048);
  { TFL_TIMED("k_flags_to_occupancy", st); k_flags_to_occupancy<<<blocks > 0 ? blocks : 1, 256, 0, st>>>(numel, flags, occ); }
}

}  // namespace tfl
