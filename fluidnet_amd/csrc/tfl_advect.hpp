// tfl_advect.hpp -- pieces of the advection operators shared by advect.hip (all methods, 2-D and 3-D, the gather
// kernels) and advect_vel3.hip (the LDS-tiled 3-D velocity passes, whose rare lanes fall back to these).
#pragma once
#include "tfl_device.hpp"
#include "tfl_host.hpp"

namespace tfl {

struct AdvArgs {
  Dom d;
  float dt;
  float strength;
  int outside;  // sampleOutsideFluid
  int fast;     // tfl_set_advect_mode: 1 = the tolerance mode of the LDS-tiled 3-D kernels (advect_vel3.hip, advect_scalar3.hip)
  unsigned long long* err;
  BlockOrder ord;   // block -> tile order of the LDS-tiled 3-D kernels (tfl_device.hpp; zero = blockIdx as it comes)
};

#define TFL_CELL_INDEX()                                             \
  const int i = blockIdx.x * blockDim.x + threadIdx.x;               \
  const int j = blockIdx.y * blockDim.y + threadIdx.y;               \
  const Dom& d = a.d;                                                \
  int b, k; dom_bk(d, b, k);                                         \
  if (i >= d.X || j >= d.Y) return;                                  \
  const long long cells = (long long)d.sc;                           \
  (void)cells

__device__ __forceinline__ void minmax(float& lo, float& hi, float v) {
  if (v < lo) lo = v;
  if (v > hi) hi = v;
}

// doClampComponent[MAC], tfluids.cc:250-295 and :701-746, split in two so that the 2 x 2^dim corner loads
// (whose addresses depend only on the cell and its MAC velocity) can be issued BEFORE the back-trace and
// overlap its latency: manta_clamp_bounds gathers min/max (false <=> the reference returns `fwd`),
// manta_clamp_component applies them. g = channel plane of `orig`.
template <bool IS3D>
__device__ __forceinline__ bool manta_clamp_bounds(const Dom& d, const float* __restrict__ g, v3 pos, v3 vel, float& lo,
                                                   float& hi) {
  lo = 3.402823466e+38f; hi = -3.402823466e+38f;
#pragma unroll
  for (int l = 0; l < 2; l++) {
    int px, py, pz;
    if (l == 0) { px = (int)(pos.x - vel.x); py = (int)(pos.y - vel.y); pz = (int)(pos.z - vel.z); }
    else { px = (int)(pos.x + vel.x); py = (int)(pos.y + vel.y); pz = (int)(pos.z + vel.z); }
    const int i0 = iclampi(px, 0, d.X - 2);
    const int j0 = iclampi(py, 0, d.Y - 2);
    const int k0 = iclampi(pz, 0, IS3D ? (d.Zg - 2) : 1);      // global plane (pos.z carries the slab's z origin)
    const int i1 = i0 + 1, j1 = j0 + 1, k1 = IS3D ? k0 + 1 : k0;
    // isInBounds(p, 0), grid.cc:42-52: in 2-D z must be exactly 0
    if (IS3D) { if (k0 < 0 || k1 >= d.Zg) return false; }
    else if (k0 != 0 || k1 != 0) return false;
    if (i0 < 0 || j0 < 0 || i1 >= d.X || j1 >= d.Y) return false;
    const int a = TFL_AT(d, i0, j0, IS3D ? slab_plane(d, k0, d.Z - 2) : k0);
#ifdef TFL_EXACT_MINMAX
    minmax(lo, hi, g[a]);
    minmax(lo, hi, g[a + 1]);
    minmax(lo, hi, g[a + d.sy]);
    minmax(lo, hi, g[a + 1 + d.sy]);
    if (IS3D) {
      const int c = a + d.sz;
      minmax(lo, hi, g[c]);
      minmax(lo, hi, g[c + 1]);
      minmax(lo, hi, g[c + d.sy]);
      minmax(lo, hi, g[c + 1 + d.sy]);
    }
#else
    // v_min3_f32 / v_max3_f32: one instruction per corner pair instead of four. Equal to the reference's
    // compare-and-keep chain in value; only the SIGN of a zero bound can differ (min prefers -0, the chain the
    // first zero it met), which no later operation of the step can turn into a different number.
    lo = __builtin_fminf(__builtin_fminf(lo, g[a]), g[a + TFL_P1(d)]);
    hi = __builtin_fmaxf(__builtin_fmaxf(hi, g[a]), g[a + TFL_P1(d)]);
    lo = __builtin_fminf(__builtin_fminf(lo, g[a + d.sy]), g[a + TFL_P1(d) + d.sy]);
    hi = __builtin_fmaxf(__builtin_fmaxf(hi, g[a + d.sy]), g[a + TFL_P1(d) + d.sy]);
    if (IS3D) {
      const int c = a + d.sz;
      lo = __builtin_fminf(__builtin_fminf(lo, g[c]), g[c + TFL_P1(d)]);
      hi = __builtin_fmaxf(__builtin_fmaxf(hi, g[c]), g[c + TFL_P1(d)]);
      lo = __builtin_fminf(__builtin_fminf(lo, g[c + d.sy]), g[c + TFL_P1(d) + d.sy]);
      hi = __builtin_fmaxf(__builtin_fmaxf(hi, g[c + d.sy]), g[c + TFL_P1(d) + d.sy]);
    }
#endif
  }
  return true;
}
template <bool IS3D>
__device__ __forceinline__ float manta_clamp_component(const Dom& d, float dst, const float* __restrict__ g, float fwd,
                                                       v3 pos, v3 vel) {
  float lo, hi;
  return manta_clamp_bounds<IS3D>(d, g, pos, vel, lo, hi) ? fclampf(dst, lo, hi) : fwd;
}

// SemiLagrange[EulerOurs]MAC of one face component, tfluids.cc:594-658, given the MAC-averaged velocity
// u of that face. disp = u * (-dt) for the trace, p = centre - u*dt for Manta's variant.
template <bool IS3D, bool OURS, int AXIS>
__device__ __forceinline__ float sl_mac_from_u(const AdvArgs& a, const float* flags, const float* src, v3 u, float dt,
                                               int i, int j, int k) {
  const v3 ctr = cell_centre(a.d, i, j, k);
  v3 p;
  if (OURS) count_trace_error(line_trace(a.d, flags, ctr, scale3(u, -dt), p), a.err);
  else p = mk3(ctr.x - u.x * dt, ctr.y - u.y * dt, ctr.z - u.z * dt);
  return interpol<IS3D>(a.d, src + AXIS * a.d.sc, p);
}

template <bool IS3D>
__device__ __forceinline__ float sample_s(const Dom& d, const float* g, const float* flags, v3 p, int outside) {
  return outside ? interpol<IS3D>(d, g, p) : interpol_with_fluid<IS3D>(d, g, flags, p);
}

// SemiLagrangeEulerOurs[SavePos], tfluids.cc:152-207 (fluid cells only; caller handles the rest)
template <bool IS3D>
__device__ __forceinline__ float sl_euler_ours(const AdvArgs& a, const float* flags, const float* U, const float* src,
                                               float dt, int i, int j, int k, v3& back) {
  const v3 c = cell_centre(a.d, i, j, k);
  const v3 disp = scale3(get_centered<IS3D>(a.d, U, i, j, k), -dt);
  count_trace_error(line_trace(a.d, flags, c, disp, back), a.err);
  return sample_s<IS3D>(a.d, src, flags, back, a.outside);
}

static inline dim3 cell_grid(const Dom& d, int B, dim3 blk) {
  return dim3((d.X + blk.x - 1) / blk.x, (d.Y + blk.y - 1) / blk.y, (unsigned)(d.nw * B));
}


// advect_vel3.hip: advectVel of the trace-based methods on a 3-D grid; false = shape not supported (caller falls back)
bool advect_vel3(hipStream_t st, bool two_pass, const AdvArgs& a, int B, const float* U, const float* flags, float* fwd,
                 float* dst, int stages);
// advect_scalar3.hip: advectScalar of the trace-based methods on a 3-D grid (no min/max grid: `bounds` = two planes)
bool advect_scalar3(hipStream_t st, bool two_pass, const AdvArgs& a, int B, const float* s, const float* U, const float* flags,
                    float* fwd, float* bounds, float* dst, int stages);

// advect_pair3.hip (round 6): the passes A (stages & 2) / the passes B (stages & 4) of advectScalar AND advectVel in one launch each
bool advect_pair3(hipStream_t st, const AdvArgs& a, int B, const float* s, const float* U, const float* flags, float* sfwd, float* sbounds,
                  float* sdst, float* vfwd, float* vdst, int stages, const BcFoldArg& fold_s, const BcFoldArg& fold_v);

}  // namespace tfl
