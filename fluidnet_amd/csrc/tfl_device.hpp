// tfl_device.hpp -- device-side grid primitives of the MI355X tfluids path (gfx950 only).
//
// Semantics follow the reference's CPU float path (the parity target):
//   grid views / samplers   torch/tfluids/third_party/grid.cc:26-515
//   vec3 thresholded norm   torch/tfluids/generic/vec3.h:119-141
//   line trace              torch/tfluids/generic/calc_line_trace.cc:313-503
// Everything is fp32 with the reference's association order; the library is compiled with
// -ffp-contract=off so no FMA contraction can flip a branch inside the trace or a clamp.
//
// Data layout in HBM: contiguous [B][C][Z][Y][X] fp32, x fastest. A MAC component c of cell
// (i,j,k) lives on the cell's NEGATIVE c-face; cell centres sit at (i+.5, j+.5, k+.5).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

namespace tfl {

enum : int { kFluid = 1, kObstacle = 2, kEmpty = 4, kInflow = 8, kOutflow = 16, kOpen = 32,
             kStick = 128 };
// generic/advect_type.h:20-27
enum : int { kEuler = 0, kMacCormack = 1, kEulerOurs = 2, kRK2Ours = 3, kRK3Ours = 4,
             kMacCormackOurs = 5 };

struct v3 { float x, y, z; };

__device__ __forceinline__ v3 mk3(float x, float y, float z) { v3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ v3 scale3(v3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }

// One batch item's geometry. Element strides fit int32 for every grid the path supports
// (C*Z*Y*X < 2^31); the batch offset is applied to the base pointers in 64-bit by the caller.
struct Dom {
  int X, Y, Z;
  int sy, sz, sc;  // element strides of y, z and channel
  int one;         // always 1, but opaque to the compiler (see TFL_P1)
  // Compute window: a launch covers the z-planes [w0, w0 + n0) and then [w1, w1 + (nw - n0)) of the array (nw planes
  // per batch item in all; default = all Z planes). Addressing always uses the full Z: a z-slab rank computes each
  // phase only on the planes whose inputs are valid (tfl_set_z_window, csrc/simulate.cpp), in at most two runs
  // (the two boundary strips of an interior/boundary split go out as ONE launch).
  int w0, n0, w1, nw;
  // z-slab ranks: the array holds planes [zg, zg + Z) of a Zg-deep grid (tfl_set_z_origin; default zg = 0, Zg = Z).
  // Back-trace positions are formed in GLOBAL z -- (k + zg) + 0.5 - u*dt rounds exactly as the unsplit grid's
  // k_global + 0.5 - u*dt does, a position relative to the slab would round differently whenever the two indices
  // fall into different binades -- and are turned into local plane indices only to address memory.
  int zg, Zg;
};

// ---- XCD-contiguous block order (round 5) --------------------------------------------------------------------------------
// The dispatcher deals consecutive workgroup ids (x fastest, then y, then z) round-robin over the chip's 8 XCDs, and each XCD
// has its own L2. With the natural (blockIdx.x, blockIdx.y, blockIdx.z) -> tile map the x / y neighbours of a tile therefore run
// on OTHER XCDs: the halo rows and the 128-byte lines two tiles share are fetched from the fabric once per tile (PMC: 1.9-2.6x
// the algorithmic reads of the advection kernels, 2.8x for k_vort_fused at 256^3). With this order XCD k takes the k-th
// contiguous eighth of the tiles (x fastest, then y, then z): the blocks resident on an XCD are neighbours, plane after plane.
// Exact unsigned division by a launch constant (Granlund-Montgomery: q = (t + ((n - t) >> s1)) >> s2, t = mulhi(m, n)):
// the decode is ~20 scalar instructions, no v_rcp sequences.
struct Div32 { unsigned m, s1, s2; };
inline Div32 make_div32(unsigned d) {
  unsigned l = 0;
  while ((1ull << l) < d) l++;
  Div32 v;
  v.m = (unsigned)((((1ull << l) - d) << 32) / d + 1);
  v.s1 = l < 1 ? l : 1; v.s2 = l > 0 ? l - 1 : 0;
  return v;
}
__host__ __device__ __forceinline__ unsigned div32(unsigned n, const Div32& v) {
#if defined(__HIP_DEVICE_COMPILE__)
  const unsigned t = __umulhi(v.m, n);
#else
  const unsigned t = (unsigned)(((unsigned long long)v.m * n) >> 32);
#endif
  return (t + ((n - t) >> v.s1)) >> v.s2;
}
// id-th block of an n-block 1-D launch -> its place in the XCD-contiguous order (a bijection of [0, n))
__host__ __device__ __forceinline__ unsigned xcd_contiguous(unsigned id, unsigned n) {
  const unsigned q = n >> 3, r = n & 7, k = id & 7;
  return k * q + (k < r ? k : r) + (id >> 3);
}
struct BlockOrder {       // all zero = the hardware's own order
  unsigned on, gx, gy, P, n;
  unsigned S, full;       // run length (tiles one XCD takes in a row before the next eight runs start); full = the blocks in whole rounds of 8 runs
  Div32 dgx, dP, dS;
};
// run = tiles per run: 0 = one run per XCD (n / 8: XCD k sweeps the k-th eighth of the launch, plane after plane);
// P / 8 = an eighth of every plane per XCD (x and most y neighbours share an L2, all XCDs advance through z together)
inline BlockOrder make_block_order(unsigned gx, unsigned gy, unsigned gz, bool enable, unsigned run = 0) {
  BlockOrder o{};
  const unsigned long long n = (unsigned long long)gx * gy * gz;
  if (!enable || n < 16 || n >= (1ull << 31)) return o;
  o.on = 1; o.gx = gx; o.gy = gy; o.P = gx * gy; o.n = (unsigned)n;
  o.S = run ? run : o.n >> 3;
  if (o.S > (o.n >> 3)) o.S = o.n >> 3;
  o.full = o.n / (8 * o.S) * (8 * o.S);
  o.dgx = make_div32(gx); o.dP = make_div32(o.P); o.dS = make_div32(o.S);
  return o;
}
// the tile (x, y, z) of this block, as blockIdx would give it under the natural order
__device__ __forceinline__ void block_tile(const BlockOrder& o, int& bx, int& by, int& bz) {
  if (!o.on) { bx = (int)blockIdx.x; by = (int)blockIdx.y; bz = (int)blockIdx.z; return; }
  const unsigned L = blockIdx.x + o.gx * (blockIdx.y + o.gy * blockIdx.z);
  unsigned T;
  if (L < o.full) {
    const unsigned idx = L >> 3, c = div32(idx, o.dS);
    T = (c * 8 + (L & 7)) * o.S + (idx - c * o.S);
  } else {
    T = o.full + xcd_contiguous(L - o.full, o.n - o.full);
  }
  const unsigned z = div32(T, o.dP), p = T - z * o.P;
  const unsigned y = div32(p, o.dgx);
  bx = (int)(p - y * o.gx); by = (int)y; bz = (int)z;
}
// the same for a block that is the L-th of a gx x gy x gz launch in the natural order (x fastest) WITHOUT being one: the pair
// kernels of advect_pair3.hip run two launches' blocks as two ranges of one 1-D grid (round 6)
__device__ __forceinline__ void block_tile_linear(const BlockOrder& o, unsigned L, unsigned gx, unsigned gy, int& bx, int& by, int& bz) {
  if (!o.on) { const unsigned r = L / gx; bx = (int)(L - r * gx); bz = (int)(r / gy); by = (int)(r - (unsigned)bz * gy); return; }
  unsigned T;
  if (L < o.full) {
    const unsigned idx = L >> 3, c = div32(idx, o.dS);
    T = (c * 8 + (L & 7)) * o.S + (idx - c * o.S);
  } else {
    T = o.full + xcd_contiguous(L - o.full, o.n - o.full);
  }
  const unsigned z = div32(T, o.dP), p = T - z * o.P;
  const unsigned y = div32(p, o.dgx);
  bx = (int)(p - y * o.gx); by = (int)y; bz = (int)z;
}
// The decode above assumes EIGHT XCDs fed round-robin: an MI355X / MI350X (and gfx942's MI300X) in SPX mode. A partitioned part
// (DPX / QPX / CPX: 4 / 2 / 1 XCDs behind one device) or a one-die part would pay for the decode and gain nothing, so the order
// is taken only where the device shows the CU count of a whole eight-XCD package (>= 200 CUs; asked once per device -- ADVICE
// r05). EXPERIMENTS flavour: TFL_XCD_ORDER=0 restores the hardware order everywhere (A/B switch; read once)
inline bool device_has_8_xcds() {
  static int known[64];       // 0 = not asked yet, 1 = yes, -1 = no (benign race: every thread writes the same value)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  if (!known[dev]) {
    int cus = 0;
    const bool ok = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess;
    known[dev] = (ok && cus >= 200) ? 1 : -1;
  }
  return known[dev] > 0;
}
inline bool xcd_order_enabled() {
#ifdef TFL_EXPERIMENTS
  static const bool on = !(getenv("TFL_XCD_ORDER") && atoi(getenv("TFL_XCD_ORDER")) == 0);
  return on && device_has_8_xcds();
#else
  return device_has_8_xcds();
#endif
}
// run length of a gx x gy (x gz) launch: an eighth of a plane per XCD (measured best, or level with one run per XCD, for the
// scalar advection and the curl / confinement kernels at 128^3 and 256^3: profiles/r05_xcd_order.txt). TFL_XCD_RUN = tiles
// per run and TFL_XCD_ORDER = 1 (one run per XCD) are the switches of the EXPERIMENTS flavour.
inline unsigned xcd_run(unsigned gx, unsigned gy) {
#ifdef TFL_EXPERIMENTS
  static const int run = getenv("TFL_XCD_RUN") ? atoi(getenv("TFL_XCD_RUN")) : 0;
  static const int mode = getenv("TFL_XCD_ORDER") ? atoi(getenv("TFL_XCD_ORDER")) : 2;
  if (run > 0) return (unsigned)run;
  if (mode == 1) return 0;
#endif
  return gx * gy >= 8 ? gx * gy / 8 : 1;
}

// (batch item, z-plane) of a block: blockIdx.z enumerates the window's planes, batch item by batch item
__device__ __forceinline__ void dom_bk(const Dom& d, int& b, int& k) {
  b = (int)blockIdx.z / d.nw;
  const int r = (int)blockIdx.z - b * d.nw;
  k = r < d.n0 ? d.w0 + r : d.w1 + (r - d.n0);
}

#define TFL_AT(d, i, j, k) ((i) + (j) * (d).sy + (k) * (d).sz)
// The +1 x-neighbour of a GATHERED tap (interpolation corners, clamp boxes: positions that depend on a back-trace).
// hipcc merges g[a], g[a + 1] into one global_load_dwordx2 at a 4-byte-aligned address, and on gfx950 that costs
// more texture-addresser time than two dword loads (r01 A/B: k_vel_bwd 57.4 -> 53.5 us, k_vel_fwd 26.4 -> 24.3,
// scalar passes -1 us each). Dom::one is a kernel argument equal to 1: adding it keeps the two loads apart. The
// cell-aligned MAC taps (get_at_mac) stay merged: there the wide load wins.
#define TFL_P1(d) ((d).one)

// global plane -> plane of the local array, held inside it. On the whole grid (zg = 0, Zg = Z) the clamp never acts. On
// a z-slab it acts only when a back-trace left the halo (|u_z| dt above the layout's reach): that step's owned planes
// are wrong either way and the reach check fails the NEXT call (csrc/simulate.cpp) -- but the read must stay inside the
// array, a GPU memory fault would take the process down before the error can be reported.
__device__ __forceinline__ int slab_plane(const Dom& d, int k_global, int hi) { return min(max(k_global - d.zg, 0), hi); }

template <bool IS3D>
__device__ __forceinline__ bool on_border(const Dom& d, int i, int j, int k) {
  // bnd = 1 is hard-coded in every op (third_party/tfluids.cc:467)
  return i < 1 || i > d.X - 2 || j < 1 || j > d.Y - 2 || (IS3D && (k < 1 || k > d.Z - 2));
}

__device__ __forceinline__ int flag_at(const Dom& d, const float* __restrict__ f, int i, int j, int k) {
  return (int)f[TFL_AT(d, i, j, k)];
}
__device__ __forceinline__ bool fluid_at(const Dom& d, const float* __restrict__ f, int i, int j, int k) {
  return (flag_at(d, f, i, j, k) & kFluid) != 0;
}

__device__ __forceinline__ int iclampi(int v, int lo, int hi) { return max(min(v, hi), lo); }
// std::min<real>(hi, std::max<real>(lo, v)), third_party/tfluids.cc:246-248
__device__ __forceinline__ float fclampf(float v, float lo, float hi) {
  float m = (lo < v) ? v : lo;
  return (m < hi) ? m : hi;
}
__device__ __forceinline__ float stdmin(float a, float b) { return (b < a) ? b : a; }
__device__ __forceinline__ float stdmax(float a, float b) { return (a < b) ? b : a; }

// vec3::norm / normalize with their thresholds (float kEpsilon = 1e-6f), generic/vec3.h:119-141
__device__ __forceinline__ float norm3(v3 a) {
  float l2 = a.x * a.x + a.y * a.y + a.z * a.z;
  return (l2 > 1e-6f) ? sqrtf(l2) : 0.0f;
}
__device__ __forceinline__ v3 normalize3(v3 a) {
  float n = norm3(a);
  if (n > 1e-6f) return mk3(a.x / n, a.y / n, a.z / n);
  return mk3(0.0f, 0.0f, 0.0f);
}

// ---- MAC samplers, third_party/grid.cc:346-417 ------------------------------------------------
template <bool IS3D>
__device__ __forceinline__ v3 get_centered(const Dom& d, const float* __restrict__ U, int i, int j, int k) {
  const int a = TFL_AT(d, i, j, k);
  v3 r;
  r.x = 0.5f * (U[a] + U[a + TFL_P1(d)]);
  r.y = 0.5f * (U[a + d.sc] + U[a + d.sc + d.sy]);
  r.z = IS3D ? 0.5f * (U[a + 2 * d.sc] + U[a + 2 * d.sc + d.sz]) : 0.0f;
  return r;
}

template <bool IS3D, int AXIS>
__device__ __forceinline__ v3 get_at_mac(const Dom& d, const float* __restrict__ U, int i, int j, int k) {
  const int a = TFL_AT(d, i, j, k);
  const float* Ux = U;
  const float* Uy = U + d.sc;
  const float* Uz = U + 2 * d.sc;
  v3 r;
  if (AXIS == 0) {
    r.x = Ux[a];
    r.y = 0.25f * (Uy[a] + Uy[a - 1] + Uy[a + d.sy] + Uy[a - 1 + d.sy]);
    r.z = IS3D ? 0.25f * (Uz[a] + Uz[a - 1] + Uz[a + d.sz] + Uz[a - 1 + d.sz]) : 0.0f;
  } else if (AXIS == 1) {
    r.x = 0.25f * (Ux[a] + Ux[a - d.sy] + Ux[a + 1] + Ux[a + 1 - d.sy]);
    r.y = Uy[a];
    r.z = IS3D ? 0.25f * (Uz[a] + Uz[a - d.sy] + Uz[a + d.sz] + Uz[a - d.sy + d.sz]) : 0.0f;
  } else {
    r.x = 0.25f * (Ux[a] + Ux[a - d.sz] + Ux[a + 1] + Ux[a + 1 - d.sz]);
    r.y = 0.25f * (Uy[a] + Uy[a - d.sz] + Uy[a + d.sy] + Uy[a + d.sy - d.sz]);
    r.z = IS3D ? Uz[a] : 0.0f;
  }
  return r;
}

// ---- interpolation, third_party/grid.cc:82-130 (buildIndex), :182-202, :204-332 ----------------
struct Lerp { int xi, yi, zi; float s0, s1, t0, t1, f0, f1; };

// buildIndex clamps in two branches per axis (pos < 0.5 -> cell 0, weights 1/0; cell >= N-1 -> cell N-2, weights
// 0/1). Clamping the shifted coordinate to [0, N-1] FIRST gives the same cell and the same weights without
// branches (6 VALU per axis instead of 12): inside the range nothing changes (s1 = px - float(int(px)), the same
// subtraction), below it px = 0 -> cell 0, s1 = 0, s0 = 1; above it px = N-1 -> cell N-2, s1 = 1, s0 = 0.
__device__ __forceinline__ void lerp_axis(float p, int n, int& idx, float& w0, float& w1) {
  const float pc = __builtin_fminf(__builtin_fmaxf(p, 0.0f), (float)(n - 1));
  idx = min((int)pc, n - 2);
  w1 = pc - (float)idx;
  w0 = 1.0f - w1;
}

template <bool IS3D>
__device__ __forceinline__ Lerp build_index(const Dom& d, v3 pos) {
  Lerp L;
  lerp_axis(pos.x - 0.5f, d.X, L.xi, L.s0, L.s1);
  lerp_axis(pos.y - 0.5f, d.Y, L.yi, L.t0, L.t1);
  if (IS3D) {
    lerp_axis(pos.z - 0.5f, d.Zg, L.zi, L.f0, L.f1);
    L.zi = slab_plane(d, L.zi, d.Z - 2);
  } else {
    // Z == 1: the 2-D samplers only ever touch plane 0 (weights as the reference computes them)
    const float pz = pos.z - 0.5f;
    L.zi = (int)pz;
    L.f1 = pz - (float)L.zi; L.f0 = 1.0f - L.f1;
    if (pz < 0.0f) { L.f0 = 1.0f; L.f1 = 0.0f; }
    L.zi = 0;
  }
  return L;
}

// plain bi/tri-linear sample of one channel plane
template <bool IS3D>
__device__ __forceinline__ float interpol(const Dom& d, const float* __restrict__ g, v3 pos) {
  const Lerp L = build_index<IS3D>(d, pos);
  const int a = TFL_AT(d, L.xi, L.yi, L.zi);
  const float lo = (g[a] * L.t0 + g[a + d.sy] * L.t1) * L.s0 +
                   (g[a + TFL_P1(d)] * L.t0 + g[a + TFL_P1(d) + d.sy] * L.t1) * L.s1;
  if (!IS3D) return lo;
  const int b = a + d.sz;
  const float hi = (g[b] * L.t0 + g[b + d.sy] * L.t1) * L.s0 +
                   (g[b + TFL_P1(d)] * L.t0 + g[b + TFL_P1(d) + d.sy] * L.t1) * L.s1;
  return lo * L.f0 + hi * L.f1;
}

// 1-D lerp that drops non-fluid taps, grid.cc:204-222
__device__ __forceinline__ void lerp_fluid(float va, bool fa, float vb, bool fb, float ta, float tb,
                                           bool& fo, float& vo) {
  if (!fa && !fb) { vo = 0.0f; fo = false; }
  else if (!fa) { vo = vb; fo = true; }
  else if (!fb) { vo = va; fo = true; }
  else { vo = va * ta + vb * tb; fo = true; }
}

template <bool IS3D>
__device__ __forceinline__ float interpol_with_fluid(const Dom& d, const float* __restrict__ g,
                                                     const float* __restrict__ flags, v3 pos) {
  const Lerp L = build_index<IS3D>(d, pos);
  const int a = TFL_AT(d, L.xi, L.yi, L.zi);
  auto fl = [&](int off) { return ((int)flags[off] & kFluid) != 0; };
  bool f_ab, f_cd, f_abcd, fo;
  float v_ab, v_cd, v_abcd, val;
  lerp_fluid(g[a], fl(a), g[a + d.sy], fl(a + d.sy), L.t0, L.t1, f_ab, v_ab);
  lerp_fluid(g[a + TFL_P1(d)], fl(a + TFL_P1(d)), g[a + TFL_P1(d) + d.sy], fl(a + TFL_P1(d) + d.sy), L.t0, L.t1, f_cd, v_cd);
  lerp_fluid(v_ab, f_ab, v_cd, f_cd, L.s0, L.s1, f_abcd, v_abcd);
  if (IS3D) {
    const int b = a + d.sz;
    bool f_ef, f_gh, f_efgh;
    float v_ef, v_gh, v_efgh;
    lerp_fluid(g[b], fl(b), g[b + d.sy], fl(b + d.sy), L.t0, L.t1, f_ef, v_ef);
    lerp_fluid(g[b + TFL_P1(d)], fl(b + TFL_P1(d)), g[b + TFL_P1(d) + d.sy], fl(b + TFL_P1(d) + d.sy), L.t0, L.t1, f_gh, v_gh);
    lerp_fluid(v_ef, f_ef, v_gh, f_gh, L.s0, L.s1, f_efgh, v_efgh);
    lerp_fluid(v_abcd, f_abcd, v_efgh, f_efgh, L.f0, L.f1, fo, val);
  } else {
    fo = f_abcd; val = v_abcd;
  }
  return fo ? val : interpol<IS3D>(d, g, pos);
}

template <bool IS3D>
__device__ __forceinline__ v3 sample_vel(const Dom& d, const float* __restrict__ U, v3 p) {
  v3 r;
  r.x = interpol<IS3D>(d, U, p);
  r.y = interpol<IS3D>(d, U + d.sc, p);
  r.z = IS3D ? interpol<IS3D>(d, U + 2 * d.sc, p) : 0.0f;
  return r;
}

// ---- line trace, generic/calc_line_trace.cc ----------------------------------------------------
#define TFL_HIT_MARGIN 1e-5f  // calc_line_trace.cc:22
#define TFL_TRACE_EPS 1e-12f  // calc_line_trace.cc:23

__device__ __forceinline__ bool out_of_domain(const Dom& d, v3 p) {  // :43-51 (walls count as outside)
  return p.x <= 0.0f || p.x >= (float)d.X || p.y <= 0.0f || p.y >= (float)d.Y || p.z <= 0.0f ||
         p.z >= (float)d.Zg;
}
// :86-91 + :53-62; -1 when the cell index falls outside the grid (the CPU reference raises)
__device__ __forceinline__ int blocked_at(const Dom& d, const float* __restrict__ flags, v3 p) {
  const int i = (int)p.x, j = (int)p.y, k = (int)p.z - d.zg;
  if (i < 0 || i >= d.X || j < 0 || j >= d.Y || k < 0 || k >= d.Z) return -1;
  return fluid_at(d, flags, i, j, k) ? 0 : 1;
}

// blocked_at for a position already known to be inside the domain (0 < p < N on every axis, so the truncated
// cell index is in range and the reference's out-of-grid error cannot fire): skips the six range tests
__device__ __forceinline__ int blocked_inside(const Dom& d, const float* __restrict__ flags, v3 p) {
  return fluid_at(d, flags, (int)p.x, (int)p.y, slab_plane(d, (int)p.z, d.Z - 1)) ? 0 : 1;
}

// Ray/box test, calc_line_trace.cc:101-171
__device__ inline bool ray_box(const float* lo, const float* hi, const float* org, const float* dir,
                               float* out) {
  bool inside = true;
  int quad[3];
  float plane[3], tmax[3];
  const float err_tol = 1e-6f;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (org[a] < lo[a]) { quad[a] = 1; plane[a] = lo[a]; inside = false; }
    else if (org[a] > hi[a]) { quad[a] = 0; plane[a] = hi[a]; inside = false; }
    else { quad[a] = 2; plane[a] = 0.0f; }
  }
  if (inside) { out[0] = org[0]; out[1] = org[1]; out[2] = org[2]; return true; }
#pragma unroll
  for (int a = 0; a < 3; a++)
    tmax[a] = (quad[a] != 2 && dir[a] != 0.0f) ? (plane[a] - org[a]) / dir[a] : -1.0f;
  int which = 0;
  if (tmax[which] < tmax[1]) which = 1;
  if (tmax[which] < tmax[2]) which = 2;
  if (tmax[which] < 0.0f) return false;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (which != a) {
      out[a] = org[a] + tmax[which] * dir[a];
      if (out[a] < (lo[a] - err_tol) || out[a] > (hi[a] + err_tol)) return false;
    } else {
      out[a] = plane[a];
    }
  }
  return true;
}

// calc_line_trace.cc:205-286: pull `next` back onto the domain wall inset by the hit margin
__device__ inline bool ray_border(const Dom& d, v3 pos, v3 next, v3& ipos) {
  float min_step = 3.402823466e+38f;
  const float p[3] = {pos.x, pos.y, pos.z}, n[3] = {next.x, next.y, next.z};
  const float sz[3] = {(float)d.X, (float)d.Y, (float)d.Zg};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (n[a] <= TFL_HIT_MARGIN) {
      const float dd = n[a] - p[a];
      if (fabsf(dd) >= TFL_TRACE_EPS) { const float st = (TFL_HIT_MARGIN - p[a]) / dd; if (st < min_step) min_step = st; }
    }
  }
#pragma unroll
  for (int a = 0; a < 3; a++) {
    if (n[a] >= (sz[a] - TFL_HIT_MARGIN)) {
      const float dd = n[a] - p[a];
      if (fabsf(dd) >= TFL_TRACE_EPS) { const float st = (sz[a] - TFL_HIT_MARGIN - p[a]) / dd; if (st < min_step) min_step = st; }
    }
  }
  if (min_step < 0.0f || min_step >= 3.402823466e+38f) return false;
  ipos.x = min_step * (next.x - pos.x) + pos.x;
  ipos.y = min_step * (next.y - pos.y) + pos.y;
  ipos.z = min_step * (next.z - pos.z) + pos.z;
  return true;
}

// calcLineTrace, calc_line_trace.cc:313-503. Returns 1 = hit, 0 = no hit, <0 = an invariant the
// CPU reference raises on (callers count these into tfl_ctx's error word and carry on with the last
// valid position, the way the reference's CUDA build does).
__device__ inline int line_trace(const Dom& d, const float* __restrict__ flags, v3 pos, v3 delta,
                                 v3& out) {
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
        ip.z = stdmin(stdmax(next.z, TFL_HIT_MARGIN), (float)d.Zg - TFL_HIT_MARGIN);
      }
      if (out_of_domain(d, ip)) return -3;
      if (!blocked_inside(d, flags, ip)) { out = ip; return 1; }
      next = ip;
    }
    // here `next` is inside the domain (either it never left, or it is the checked `ip`)
    int blk = blocked_inside(d, flags, next);
    if (blk) {
      for (int count = 0; count <= 4; count++) {
        blk = blocked_at(d, flags, next);
        if (blk < 0) return -4;
        if (!blk) break;
        if (count == 4) return -5;
        const float cx = (float)((int)next.x) + 0.5f;
        const float cy = (float)((int)next.y) + 0.5f;
        const float cz = (float)((int)next.z) + 0.5f;
        const float lo[3] = {cx - 0.5f - TFL_HIT_MARGIN, cy - 0.5f - TFL_HIT_MARGIN, cz - 0.5f - TFL_HIT_MARGIN};
        const float hi[3] = {cx + 0.5f + TFL_HIT_MARGIN, cy + 0.5f + TFL_HIT_MARGIN, cz + 0.5f + TFL_HIT_MARGIN};
        const float org[3] = {out.x, out.y, out.z};
        const float dir[3] = {dt.x, dt.y, dt.z};
        float hitp[3];
        if (!ray_box(lo, hi, org, dir, hitp)) return 1;  // keep `out` (still a valid fluid position)
        next = mk3(hitp[0], hitp[1], hitp[2]);
      }
      out = next;
      if (out_of_domain(d, out)) return -6;
      if (blocked_at(d, flags, out) != 0) return -7;
      return 1;
    }
    out = next;
    cur += step;
  }
  return 0;
}

__device__ __forceinline__ v3 cell_centre(const Dom& d, int i, int j, int k) {   // k: local plane; position: global z
  return mk3((float)i + 0.5f, (float)j + 0.5f, (float)(k + d.zg) + 0.5f);
}

__device__ __forceinline__ void count_trace_error(int rc, unsigned long long* err) {
  if (rc < 0) atomicAdd(err, 1ull);
}

// Sum of `count` (sum u, sum u^2) pairs at p by the 256 threads of a block in a FIXED order -- thread t accumulates pairs
// t, t + 256, ... in that order, then the tree sh[t] += sh[t + w], w = 128 ... 1 over the 256 partial sums: bit-reproducible
// run to run and the same bits whoever evaluates it (k_reduce_stats, or every block of the first conv layer for itself:
// round 6). The association is that of the round-2 shared-memory tree; what changed in round 6 is how it is evaluated: the
// pairs of a thread are loaded eight at a time (16-byte loads, all in flight together: the runtime-length loop made them
// eight dependent L2 round trips) and levels 32 ... 1 of the tree run inside wave 0 by lane shifts instead of six more
// block barriers. sh = 512 doubles of LDS; on return EVERY thread holds the two sums and sh may be reused.
// (lib/modules/variance.lua:44-76 sums in fp32 THC reductions; here fp64, order fixed.)
__device__ __forceinline__ void block_sum_pairs(const double* __restrict__ p, long long count, double* sh, int tid, double& o1, double& o2) {
  double s1 = 0.0, s2 = 0.0;
  const double2* p2 = reinterpret_cast<const double2*>(p);
  for (long long base = 0; base < count; base += 2048) {
    double2 v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {      // unconditional loads (a lane past the end reads the last pair and drops it)
      const long long t = base + tid + 256 * u;
      v[u] = p2[t < count ? t : count - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; u++) {
      const bool ok = base + tid + 256 * u < count;
      s1 += ok ? v[u].x : 0.0; s2 += ok ? v[u].y : 0.0;      // (+ 0.0 leaves a sum unchanged)
    }
  }
  sh[tid] = s1; sh[256 + tid] = s2;
  __syncthreads();
  if (tid < 64) {                        // levels 128 and 64 of the tree, then 32 ... 1 inside the wave
    double a1 = (sh[tid] + sh[tid + 128]) + (sh[tid + 64] + sh[tid + 192]);
    double a2 = (sh[256 + tid] + sh[256 + tid + 128]) + (sh[256 + tid + 64] + sh[256 + tid + 192]);
#pragma unroll
    for (int w = 32; w > 0; w >>= 1) { a1 += __shfl_down(a1, w, 64); a2 += __shfl_down(a2, w, 64); }
    if (tid == 0) { sh[0] = a1; sh[256] = a2; }
  }
  __syncthreads();
  o1 = sh[0]; o2 = sh[256];
  __syncthreads();
}

}  // namespace tfl
