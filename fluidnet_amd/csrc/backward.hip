// backward.hip -- the training-side operators of the path's second caller (lib/run_epoch.lua's
// long-term divergence rollout back-propagates through these) and the nearest-neighbour volume resampler of the
// multi-resolution model variant (8f-1), gfx950.
//
//   velocityDivergenceBackward   generic/tfluids.cc:49-130  | generic/tfluids.cu:224-291 (atomicAdd scatter)
//   velocityUpdateBackward       generic/tfluids.cc:216-344 | generic/tfluids.cu:407-513 (atomicAdd scatter)
//   volumetricUpSamplingNearest{Forward,Backward}  generic/tfluids.cc:509-633 | generic/tfluids.cu:516-686
//
// The reference scatters with atomics (each forward cell adds into up to 6 gradient words). Here every output
// word GATHERS its (at most 2, resp. 7) contributions in the order the reference's serial loop (k, j, i
// ascending) delivers them, so the result is deterministic and bit-equal to the reference's single-threaded CPU
// result -- the reference's own multi-threaded / CUDA results vary in the last bit with the atomic order.
#include "tfl_device.hpp"
#include "tfl_host.hpp"

namespace tfl {

template <bool IS3D>
__device__ __forceinline__ bool contributes(const Dom& d, const float* __restrict__ flags, int i, int j, int k) {
  // interior (bnd = 1) fluid cell: the only cells whose forward divergence is non-zero
  return !on_border<IS3D>(d, i, j, k) && (((int)flags[TFL_AT(d, i, j, k)]) & kFluid);
}

// forward: div(n) = sum_c U_c(n) - U_c(n + 1_c)  =>  gradU_c(n) = [n contributes] go(n) - [n - 1_c contributes] go(n - 1_c)
template <bool IS3D>
__global__ __launch_bounds__(256) void k_divergence_bwd(Dom d, const float* __restrict__ flags,
                                                        const float* __restrict__ go, float* __restrict__ gU) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  if (i >= d.X || j >= d.Y) return;
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  flags += b * cells; go += b * cells; gU += b * cells * C;
  const int o = TFL_AT(d, i, j, k);
  const bool self = contributes<IS3D>(d, flags, i, j, k);
  const float g0 = self ? go[o] : 0.0f;
  float gx = 0.0f, gy = 0.0f, gz = 0.0f;
  // serial order of the reference: the -= from cell n - 1_c arrives before the += of cell n itself for x (same
  // row, smaller i), y (smaller j) and z (smaller k); two-term sums are order-independent anyway.
  if (i > 0 && contributes<IS3D>(d, flags, i - 1, j, k)) gx -= go[o - 1];
  if (self) gx += g0;
  if (j > 0 && contributes<IS3D>(d, flags, i, j - 1, k)) gy -= go[o - d.sy];
  if (self) gy += g0;
  gU[o] = gx; gU[o + d.sc] = gy;
  if (IS3D) {
    if (k > 0 && contributes<IS3D>(d, flags, i, j, k - 1)) gz -= go[o - d.sz];
    if (self) gz += g0;
    gU[o + 2 * d.sc] = gz;
  }
}

// gradP(n): contributions in the reference's serial delivery order --
//   from cell n itself (when interior fluid): -go.x, -go.y, -go.z for fluid -c neighbours, then -go.x, -go.y,
//   -go.z for empty -c neighbours; then +go_x(n + 1_x), +go_y(n + 1_y), +go_z(n + 1_z) from the +c neighbours
//   (interior; fluid with fluid -c neighbour n, or empty-non-outflow with fluid -c neighbour n).
template <bool IS3D>
__global__ __launch_bounds__(256) void k_velocity_update_bwd(Dom d, const float* __restrict__ flags,
                                                             const float* __restrict__ go, float* __restrict__ gP) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j = blockIdx.y * blockDim.y + threadIdx.y;
  int b, k; dom_bk(d, b, k);
  if (i >= d.X || j >= d.Y) return;
  const long long cells = d.sc;
  const int C = IS3D ? 3 : 2;
  flags += b * cells; go += b * cells * C; gP += b * cells;
  const int o = TFL_AT(d, i, j, k);
  const int fc = (int)flags[o];
  float g = 0.0f;
  if (!on_border<IS3D>(d, i, j, k) && (fc & kFluid)) {
    const int fx = (int)flags[o - 1], fy = (int)flags[o - d.sy], fz = IS3D ? (int)flags[o - d.sz] : 0;
    if (fx & kFluid) g -= go[o];
    if (fy & kFluid) g -= go[o + d.sc];
    if (IS3D && (fz & kFluid)) g -= go[o + 2 * d.sc];
    if (fx & kEmpty) g -= go[o];
    if (fy & kEmpty) g -= go[o + d.sc];
    if (IS3D && (fz & kEmpty)) g -= go[o + 2 * d.sc];
  }
  if (fc & kFluid) {   // the +c neighbour adds go_c only when its -c neighbour (this cell) is fluid
    auto takes = [&](int ii, int jj, int kk) {
      if (ii >= d.X || jj >= d.Y || kk >= d.Z || on_border<IS3D>(d, ii, jj, kk)) return false;
      const int f = (int)flags[TFL_AT(d, ii, jj, kk)];
      return (f & kFluid) || ((f & kEmpty) && !(f & kOutflow));
    };
    if (takes(i + 1, j, k)) g += go[o + 1];
    if (takes(i, j + 1, k)) g += go[o + d.sy + d.sc];
    if (IS3D && takes(i, j, k + 1)) g += go[o + d.sz + 2 * d.sc];
  }
  gP[o] = g;
}

// out[b][f][z][y][x] = in[b][f][z/r][y/r][x/r]
__global__ __launch_bounds__(256) void k_upsample_fwd(int ratio, long long rows, int Zo, int Yo, int Xo,
                                                      const float* __restrict__ in, float* __restrict__ out) {
  const long long n = rows * Zo * Yo * Xo;
  const int Zi = Zo / ratio, Yi = Yo / ratio, Xi = Xo / ratio;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(t % Xo);
    long long r = t / Xo;
    const int y = (int)(r % Yo); r /= Yo;
    const int z = (int)(r % Zo);
    const long long row = r / Zo;
    out[t] = in[((row * Zi + z / ratio) * Yi + y / ratio) * Xi + x / ratio];
  }
}

// gradIn = sum over the ratio^3 window of gradOut, accumulated in float in (z, y, x) order like the reference
__global__ __launch_bounds__(256) void k_upsample_bwd(int ratio, long long rows, int Zi, int Yi, int Xi,
                                                      const float* __restrict__ go, float* __restrict__ gi) {
  const long long n = rows * Zi * Yi * Xi;
  const int Yo = Yi * ratio, Xo = Xi * ratio, Zo = Zi * ratio;
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (long long)gridDim.x * blockDim.x) {
    const int x = (int)(t % Xi);
    long long r = t / Xi;
    const int y = (int)(r % Yi); r /= Yi;
    const int z = (int)(r % Zi);
    const long long row = r / Zi;
    float sum = 0.0f;
    for (int zu = 0; zu < ratio; zu++)
      for (int yu = 0; yu < ratio; yu++)
        for (int xu = 0; xu < ratio; xu++)
          sum += go[((row * Zo + z * ratio + zu) * Yo + y * ratio + yu) * (long long)Xo + x * ratio + xu];
    gi[t] = sum;
  }
}

#define TFL_BWD_LAUNCH(kern, name, ...)                                             \
  do {                                                                              \
    const Dom d = make_dom(Z, Y, X);                                                \
    const dim3 blk(64, 4, 1), grd((X + 63) / 64, (Y + 3) / 4, (unsigned)(zwin_planes(Z) * B)); \
    TFL_TIMED(name, st);                                                            \
    if (is3d) kern<true><<<grd, blk, 0, st>>>(d, __VA_ARGS__);                      \
    else kern<false><<<grd, blk, 0, st>>>(d, __VA_ARGS__);                          \
  } while (0)

void velocity_divergence_bwd(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* flags,
                             const float* grad_out, float* grad_U) {
  TFL_BWD_LAUNCH(k_divergence_bwd, "k_divergence_bwd", flags, grad_out, grad_U);
}
void velocity_update_bwd(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* flags,
                         const float* grad_out, float* grad_p) {
  TFL_BWD_LAUNCH(k_velocity_update_bwd, "k_velocity_update_bwd", flags, grad_out, grad_p);
}
static int stream_blocks(long long n) {
  long long b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
void upsample_nearest_fwd(hipStream_t st, int ratio, long long rows, int Zo, int Yo, int Xo, const float* in, float* out) {
  TFL_TIMED("k_upsample_fwd", st);
  k_upsample_fwd<<<stream_blocks(rows * Zo * Yo * Xo), 256, 0, st>>>(ratio, rows, Zo, Yo, Xo, in, out);
}
void upsample_nearest_bwd(hipStream_t st, int ratio, long long rows, int Zi, int Yi, int Xi, const float* go, float* gi) {
  TFL_TIMED("k_upsample_bwd", st);
  k_upsample_bwd<<<stream_blocks(rows * Zi * Yi * Xi), 256, 0, st>>>(ratio, rows, Zi, Yi, Xi, go, gi);
}

}  // namespace tfl
