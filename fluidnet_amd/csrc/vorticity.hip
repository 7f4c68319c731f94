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
#include "tfl_host.hpp"

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
  const int b = blockIdx.z / d.Z, k = blockIdx.z - b * d.Z;
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
  const int b = blockIdx.z / d.Z, k = blockIdx.z - b * d.Z;
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

void vorticity_confinement(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* U, const float* flags,
                           float strength, float* curl, float* curl_norm) {
  const Dom d = make_dom(Z, Y, X);
  const dim3 blk(64, 4, 1), grd((X + 63) / 64, (Y + 3) / 4, (unsigned)(Z * B));
  if (is3d) {
    { TFL_TIMED("k_curl", st); k_curl<true><<<grd, blk, 0, st>>>(d, U, curl, curl_norm); }
    { TFL_TIMED("k_confine", st); k_confine<true><<<grd, blk, 0, st>>>(d, U, flags, curl, curl_norm, strength); }
  } else {
    { TFL_TIMED("k_curl", st); k_curl<false><<<grd, blk, 0, st>>>(d, U, curl, curl_norm); }
    { TFL_TIMED("k_confine", st); k_confine<false><<<grd, blk, 0, st>>>(d, U, flags, curl, curl_norm, strength); }
  }
}

}  // namespace tfl
