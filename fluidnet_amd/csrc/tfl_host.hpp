// tfl_host.hpp -- host-side launcher prototypes shared by the .hip translation units and abi.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdlib>

#include "tfl_device.hpp"

namespace tfl {

// A switch that only the EXPERIMENTS flavour of the library reads (-DTFL_EXPERIMENTS, `make exp`: the earlier and the
// measured-slower kernel forms, chunk-length / block-order overrides, test hooks); in the product library it is always unset.
inline const char* exp_env(const char* name) {
#ifdef TFL_EXPERIMENTS
  return getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// The z-window of the calling thread's current operator (set by abi.cpp from tfl_set_z_window, cleared after the
// launch): planes [a0, a1) and [b0, b1) in array indices; all zero = the whole array.
struct ZWin { int a0, a1, b0, b1; };
extern thread_local ZWin g_zwin;
// z origin of the array inside the whole grid (tfl_set_z_origin): {first global plane, planes of the whole grid}; {0, 0} = none
struct ZOrigin { int first, total; };
extern thread_local ZOrigin g_zorigin;
// tfl_set_advect_mode of the calling thread's current operator: 1 = tolerance mode of the LDS-tiled 3-D advection kernels
extern thread_local int g_advect_fast;
// A setConstVals pair (lib/simulate.lua:130-160: x = x * invMask + bc) that the calling thread's current operator MAY apply
// to the field it writes (round 4): the pair is the identity outside the box [x0, x1] x [y0, y1] x [z0, z1] (inclusive, array
// indices; the plume's pair covers four y rows), so a producing kernel compares its rows against the box and only the
// threads inside it load bc / invMask ([B][C][Z][Y][X] like the field) -- nothing per cell elsewhere. The descriptor lives in
// DEVICE memory (the plan owns it) and a kernel gets its address: one pointer argument, read with scalar loads at the store
// -- ten scalars by value stayed live through the whole kernel and cost the advection kernels 30 SGPR spills each. A launcher
// that hands the pointer to its kernel sets g_fold_done; tfl_simulate_step launches the sparse kernel (k_apply_bcs_indexed)
// when nobody did.
struct BcFold { const float* bc; const float* inv; int x0, x1, y0, y1, z0, z1; };
// What a kernel is handed: the descriptor's device address and, by value, the box's y / z range packed into two words
// (lo = y0 | z0 << 16, hi = y1 | z1 << 16): the block-uniform gate at the top of the kernel needs no memory access.
struct BcFoldArg { const BcFold* dev; unsigned lo, hi; };     // dev == nullptr: no request
extern thread_local BcFoldArg g_fold;
extern thread_local bool g_fold_done;
// the request for a kernel launch (and the acknowledgement), or an empty one
inline BcFoldArg take_fold() { if (g_fold.dev) g_fold_done = true; return g_fold; }
inline BcFoldArg no_fold() { return BcFoldArg{nullptr, 0u, 0u}; }

// addBuoyancy (third_party/tfluids.cc:1162-1233) folded into the kernel that writes the advected velocity (round 5). In
// simulate() the buoyancy force follows advectVel and the first setConstVals directly (lib/simulate.lua:196-226), it is
// pointwise in U and needs the ADVECTED density -- which advectScalar has delivered (pair applied) before advectVel runs. So
// tfl_simulate_step hands pass B of advectVel a request {rho, (sx, sy, sz) = -gravity dt / dx}; a launcher that takes it
// (g_buoy_done) adds 0.5 s_c (rho(i) + rho(i - e_c)) on the faces between two fluid cells behind the folded pair, from two
// to four more loads per cell, and the 32 B/cell launch of k_add_buoyancy disappears. Only taken together with the U pair
// (or when there is none): the force must see the boundary values.
struct BuoyFold { const float* rho; float sx, sy, sz; };      // rho == nullptr: no request
extern thread_local BuoyFold g_buoy;
extern thread_local bool g_buoy_done;
inline BuoyFold no_buoy() { return BuoyFold{nullptr, 0.0f, 0.0f, 0.0f}; }
inline BuoyFold take_buoy() { if (g_buoy.rho) g_buoy_done = true; return g_buoy; }

// does row (j, k) / cell i of the field lie inside the pair's box
__device__ __forceinline__ bool fold_row(const BcFold& f, int j, int k) {
  return j >= f.y0 && j <= f.y1 && k >= f.z0 && k <= f.z1;
}
__device__ __forceinline__ bool fold_col(const BcFold& f, int i) { return i >= f.x0 && i <= f.x1; }
// Block-uniform gate, evaluated once at the top of a kernel (two argument words, dead right after): can any cell of rows
// [ya, yb] x planes [za, zb] lie in the pair's box?
// Only the blocks that answer yes read the descriptor again at their stores -- for the plume's pair 1 block in 16-32.
__device__ __forceinline__ bool fold_block(const BcFoldArg& a, int ya, int yb, int za, int zb) {
  return (a.dev != nullptr) & (ya <= (int)(a.hi & 0xffffu)) & (yb >= (int)(a.lo & 0xffffu)) & (za <= (int)(a.hi >> 16)) &
         (zb >= (int)(a.lo >> 16));
}

inline Dom make_dom(int Z, int Y, int X) {
  Dom d; d.X = X; d.Y = Y; d.Z = Z; d.sy = X; d.sz = X * Y; d.sc = X * Y * Z; d.one = 1;
  d.w0 = 0; d.n0 = Z; d.w1 = 0; d.nw = Z;
  d.zg = 0; d.Zg = Z;
  if (g_zorigin.total > 0) { d.zg = g_zorigin.first; d.Zg = g_zorigin.total; }
  const ZWin w = g_zwin;
  if (w.a1 > w.a0 || w.b1 > w.b0) {
    auto clip = [Z](int v) { return v < 0 ? 0 : (v > Z ? Z : v); };
    const int a0 = clip(w.a0), a1 = clip(w.a1) > a0 ? clip(w.a1) : a0;
    const int b0 = clip(w.b0), b1 = clip(w.b1) > b0 ? clip(w.b1) : b0;
    d.w0 = a0; d.n0 = a1 - a0; d.w1 = b0; d.nw = d.n0 + (b1 - b0);
  }
  return d;
}
// number of planes a launch over a Z-deep array covers under the current window (grid.z = this * B)
inline int zwin_planes(int Z) { return make_dom(Z, 1, 1).nw; }

// Built-in per-kernel timing (the reference only has host timers around the projection,
// lib/simulate.lua:254-260,306-318). When a profile is active on this thread every kernel launch is
// bracketed by hipEvents on the launch stream; tfl_profile_end() sums them per kernel name.
struct KernelTimer {
  // ext = false: the events are recorded on the stream around whatever the scope launches (adds the command
  // processor's event handling to the reading: +2..7 us per kernel). ext = true: nothing is recorded here; the
  // scope passes start()/stop() to hipExtLaunchKernelGGL, which stamps them with the dispatch's own begin / end
  // (the numbers rocprofv3 reports). Both are null when no profile is active.
  KernelTimer(const char* name, hipStream_t st, bool ext = false);
  ~KernelTimer();
  hipEvent_t start() const;
  hipEvent_t stop() const;
  int slot_;
  hipStream_t st_;
  bool ext_;
};
#define TFL_TIMED(name, st) ::tfl::KernelTimer tfl_timer_(name, st)
#define TFL_TIMED_EXT(name, st) ::tfl::KernelTimer tfl_timer_(name, st, true)
// one launch inside a TFL_TIMED_EXT scope; `kernel` in parentheses when it is a template-id with commas
// (the plain launch when no profile is active: hipExtLaunchKernelGGL costs a few us more on the host)
#define TFL_LAUNCH_EXT(kernel, grid, block, shmem, st, ...)                                                        \
  do {                                                                                                             \
    if (tfl_timer_.start())                                                                                        \
      hipExtLaunchKernelGGL(kernel, dim3(grid), dim3(block), (uint32_t)(shmem), st, tfl_timer_.start(),            \
                            tfl_timer_.stop(), 0, __VA_ARGS__);                                                    \
    else                                                                                                           \
      hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (uint32_t)(shmem), st, __VA_ARGS__);                      \
  } while (0)

// advect.hip
void advect_scalar(hipStream_t st, bool is3d, int method, int B, int Z, int Y, int X, float dt, float strength,
                   int outside, unsigned long long* err, const float* s, const float* U, const float* flags,
                   float* fwd, float* bounds, float* mm, float* dst, int stages = 7);
void advect_vel(hipStream_t st, bool is3d, int method, int B, int Z, int Y, int X, float dt, float strength,
                unsigned long long* err, const float* U, const float* flags, float* fwd, float* dst, int stages = 7);

void minmax3(hipStream_t st, bool is3d, int B, int Z, int Y, int X, int outside, const float* s, const float* flags,
             float* lo3, float* hi3);

// stencil.hip
void set_wall_bcs(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* U, const float* flags);
void velocity_divergence(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* U, const float* flags,
                         float* div);
void velocity_update(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* U, const float* flags,
                     const float* p);
void add_buoyancy(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* Usrc, float* U, const float* flags,
                  const float* density, float sx, float sy, float sz);
void add_gravity(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* U, const float* flags, float fx,
                 float fy, float fz);
void empty_domain(hipStream_t st, bool is3d, int bnd, int B, int Z, int Y, int X, float* flags);
void flags_to_occupancy(hipStream_t st, long long numel, const float* flags, float* occ);
void rectangular_blur(hipStream_t st, bool is3d, int B, int C, int Z, int Y, int X, int rad, const float* src, float* dst,
                      float* tmp);
void signed_distance_field(hipStream_t st, int B, int Z, int Y, int X, int rad, const float* flags, float* dst);
void stream_copy(hipStream_t st, long long n4, const float* src, float* dst);   // n4 float4s, 16-byte aligned
void absmax(hipStream_t st, long long n, const float* x, float* out, bool reset);   // *out = max(*out, max |x|)
void reach_flags(hipStream_t st, const float* maxu, float dt, int n, double* flags, const unsigned long long* range_count = nullptr);   // flags[r-1] = (*maxu * dt >= r), r = 1..n (n <= 62); flags[n] = (*range_count != 0) when given

// vorticity.hip
// stages: bit 0 = pass A (U -> curl, |curl|), bit 1 = pass B (curl, |curl|, flags, U -> U); a z-slab rank runs the two
// passes under different z-windows
// Usrc (round 5): U = Usrc + force with every cell of the window written (the four-cells-per-thread kernels only: false =
// not possible here, nothing launched, the caller copies Usrc into U and calls again without it)
bool vorticity_confinement(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* U, const float* flags,
                           float strength, float* curl, float* curl_norm, int stages = 3, const float* Usrc = nullptr);

// U_out = U_in + confinement(U_in), 3-D, one fused launch without curl arrays (U_out != U_in); false = not supported here
bool vorticity_confinement_fused_ok(bool is3d, int Z, int Y, int X);   // does the native step use the fused kernel for this grid
bool vorticity_confinement_fused(hipStream_t st, int B, int Z, int Y, int X, const float* Uin, float* Uout, const float* flags,
                                 float strength);

// jacobi.hip
void jacobi_iteration(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* p_prev, const float* flags,
                      const float* div, float* p, double* resid_sq /* [B] or nullptr */);

// a 2-D grid of up to 16 K cells: the whole Jacobi solve in one launch, p ping-pongs in LDS (false = not taken)
bool jacobi_solve_lds(hipStream_t st, int B, int Y, int X, const float* flags, const float* div, float* p, float* p_prev,
                      int iters, double* resid_sq);

// pcg.hip
long long pcg_workspace_floats(int Z, int Y, int X);
int pcg_solve(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* p, const float* flags, const float* div,
              int precond, float tol, int max_iter, int verbose, float* workspace, float* residual, char* msg, size_t msg_len, bool allow_wavefronts = true);

long long npm_workspace_floats(int Z, int Y, int X);
int normalize_pressure_mean(hipStream_t st, bool is3d, int B, int Z, int Y, int X, float* p, const float* flags,
                            float* workspace, char* msg, size_t msg_len);

// model.hip
long long model_stat_blocks(int B, int Z, int Y, int X);
bool model_stats_fold_requested();
long long model_stat_pairs_per_plane(int B, int Z, int Y, int X, const float* U, const float* flags, const float* Ubc, const float* div);   // as model_pre lays them out
void model_pre(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* U, const float* flags, float* Ubc,
               float* div, double* partials, double* stats, int zlo, int zhi, int stages = 3, unsigned* ticket = nullptr,
               const unsigned short* wall_code = nullptr);      // wall_code: the flags' tfl_wall_plan (round 6), or null
void wall_code(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* flags, unsigned short* code);
void model_net_input(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* pDiv, const float* div,
                     const float* flags, const double* stats, double count, float* x3);
// the general net input of lib/model.lua:130-148: channels {pDiv/scale?, SetWallBcs(U)/scale (C)?, div/scale?, occupancy} in
// this order into x [B][in_c][Z][Y][X]; Ubc = the wall-BC'd velocity tfl_model_begin left in UOut
void model_net_input_gen(hipStream_t st, bool is3d, int B, int Z, int Y, int X, int in_pDiv, int in_UDiv, int in_div,
                         const float* pDiv, const float* Ubc, const float* div, const float* flags, const double* stats,
                         double count, float* x);
// stats[b] = {mode 0: sum x, sum x^2 | mode 1: 0, sum x^2 | mode 2: 0, 1} over the n floats of sample b of `field`
// (the three ways tfl_model_opts sets the input scale through scale_from_stats: std, l2 norm with count = 2, none)
void model_field_stats(hipStream_t st, int B, long long n, const float* field, int mode, double* stats);
// dst[b][ch][cell] = pDiv[b][cell] / scale_b: the joined pressure-skip channel (model.lua:356-360); dst has `och` planes per item
void model_skip_channel(hipStream_t st, int B, long long cells, const float* pDiv, const double* stats, double count,
                        float* dst, int och, int ch);
// returns true when the launch also folded max |u_z| of what it wrote into *reach_acc (round 6: k_project_v4 on full blocks)
bool model_project(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* pPred, const float* flags,
                   const double* stats, double count, float* Uio, float* pOut, const float* UBC, const float* UInvMask,
                   int do_clamp, float lo, float hi, const unsigned long long* range_src = nullptr,
                   unsigned long long* range_dst = nullptr, const float* reach_src = nullptr, float* reach_dst = nullptr,
                   float* reach_acc = nullptr, const unsigned short* wall_code = nullptr, unsigned* reach_tick = nullptr);
void apply_bcs(hipStream_t st, long long n, float* x, const float* bcv, const float* inv, int do_clamp, float lo,
               float hi);

void bc_scan(hipStream_t st, long long n, const float* bcv, const float* inv, int* counters, int* idx);
void apply_bcs_indexed_multi(hipStream_t st, int count, const long long* n, const int* const* idx, float* const* x,
                             const float* const* bcv, const float* const* inv);
void apply_bcs_indexed(hipStream_t st, long long n, const int* idx, float* x, const float* bcv, const float* inv);

// planes [zlo[i], zlo[i] + nplanes[i]) of field i (rows[i] = B*C rows of zstride floats each) <-> buf; returns the
// number of floats moved
long long pack_planes(hipStream_t st, int n, float* const* ptrs, const int* rows, const int* zlo, const int* nplanes,
                      long long zstride, long long yx, float* buf, int unpack, float* const* bufs = nullptr);

// conv.hip
// upf > 1: the result goes to sub-position `sub` (= (c*upf + b)*upf + a) of an upf-times finer output grid (pixel shuffle)
// act: 0 none | 1 ReLU | 2 ReLU6 | 3 sigmoid; out_ch: channel planes per batch item of `out` (0 = cout)
bool conv_direct(hipStream_t st, bool is3d, int B, int Z, int Y, int X, int cin, int cout, int ksz, int act,
                 const float* in, const float* w, const float* bias, float* out, int upf = 1, int sub = 0, int out_ch = 0);
// 2x average pooling of `rows` = B*C planes-stacks [Z][Y][X] -> [Z/2 (3-D)][Y/2][X/2]
void avg_pool2(hipStream_t st, bool is3d, int rows, int Z, int Y, int X, const float* in, float* out);

// conv_mfma.hip (3-D default topology: k=3, 8 output channels; x-phase-packed fp32 MFMA)
void conv3_mfma_first(hipStream_t st, int B, int Z, int Y, int X, const float* in_planar3, const float* bfrag,
                      const float* bias, float* out_cl8);
void conv3_mfma_first_fused(hipStream_t st, int B, int Z, int Y, int X, const float* pDiv, const float* div,
                            const float* flags, const double* stats, double count, const float* bfrag,
                            const float* bias, float* out_cl8);
void conv3_mfma_mid(hipStream_t st, int B, int Z, int Y, int X, const float* in_cl8, const float* bfrag,
                    const float* bias, float* out_cl8);
void conv3_mfma_tail(hipStream_t st, int B, int Z, int Y, int X, const float* in_cl8, const float* bfrag,
                     const float* bias, const float* w4, const float* b4, const float* w5, const float* b5,
                     float* p_out);

// conv_valu.hip: the same three layers on the vector ALUs, x-taps as Winograd F(2,3) (wq = tfl_model::wino,
// [dz][dy][cin][4][8]); activations between the layers are channel-planar [B][8][Z][Y][X]
void conv3_valu_first_fused(hipStream_t st, int B, int Z, int Y, int X, const float* pDiv, const float* div, const float* flags,
                            const double* stats, double count, const float* wq, const float* bias, float* out_p8);
void conv3_valu_mid(hipStream_t st, int B, int Z, int Y, int X, const float* in_p8, const float* wq, const float* bias,
                    float* out_p8);
// tail_pack: {bias of the k3 layer [8], w4 [8][8] (out, in), b4 [8], w5 [8], b5 [1]} (tfl_model::tail_pack)
void conv3_valu_tail(hipStream_t st, int B, int Z, int Y, int X, const float* in_p8, const float* wq, const float* tail_pack,
                     float* p_out);

// conv_mfma16.hip: the same three layers as a split-operand fp16 MFMA implicit GEMM; activations between the layers are
// "h2": per (b, z, y) two rows [x][8] of fp16 (hi, lo): 32 B per voxel. wfrag / post from conv3_m16_pack_weights.
void conv3_m16_first_fused(hipStream_t st, int B, int Z, int Y, int X, const float* pDiv, const float* div, const float* flags,
                           const double* stats, double count, const void* wfrag, const float* bias, float post, void* out_h2,
                           unsigned long long* range_err, const double* partials = nullptr, long long per_sample = 0,
                           double* stats_out = nullptr);
// round 6: can the first layer sum k_bcs_div_stats' partial pairs itself (partials / per_sample / stats_out above)? Then
// tfl_model_forward skips the launch of k_reduce_stats between the two kernels.
bool conv3_m16_first_sums_partials();
void conv3_m16_mid(hipStream_t st, int B, int Z, int Y, int X, const void* in_h2, const void* wfrag, const float* bias, float post,
                   void* out_h2, unsigned long long* range_err);
void conv3_m16_tail(hipStream_t st, int B, int Z, int Y, int X, const void* in_h2, const void* wfrag, const float* tail_pack,
                    float post, float* p_out, unsigned long long* range_err);
// layers 1 + 2 in ONE launch (round 5: the 64 B/voxel between them stay in LDS); false = not taken (z-window set, or switched
// off): the caller runs conv3_m16_first_fused + conv3_m16_mid
bool conv3_m16_first2_fused(hipStream_t st, int B, int Z, int Y, int X, const float* pDiv, const float* div, const float* flags,
                            const double* stats, double count, const void* wfrag1, const float* bias1, float post1,
                            const void* wfrag2, const float* bias2, float post2, void* out_h2, unsigned long long* range_err);
bool conv3_m16_fuse12_requested();      // EXPERIMENTS flavour + TFL_M16_FUSE12=1 (the default library: always false)
size_t conv3_m16_frag_halves(int cin);
float conv3_m16_pack_tail(const float* w4 /* [8][8] (out, in) */, uint16_t* frag_buf_of_a_cin8_layer);
float conv3_m16_pack_weights(const float* w, int cin, uint16_t* out);

// backward.hip
void velocity_divergence_bwd(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* flags,
                             const float* grad_out, float* grad_U);
void velocity_update_bwd(hipStream_t st, bool is3d, int B, int Z, int Y, int X, const float* flags,
                         const float* grad_out, float* grad_p);
void upsample_nearest_fwd(hipStream_t st, int ratio, long long rows, int Zo, int Yo, int Xo, const float* in, float* out);
void upsample_nearest_bwd(hipStream_t st, int ratio, long long rows, int Zi, int Yi, int Xi, const float* go, float* gi);

// conv2d_mfma.hip (2-D default topology: 16 channels, k = 3; one MFMA = one tap x four input channels)
void conv2_mfma_first_fused(hipStream_t st, int B, int Y, int X, const float* pDiv, const float* div, const float* flags,
                            const double* stats, double count, const float* bfrag, const float* bias, float* out16);
void conv2_mfma_mid(hipStream_t st, int B, int Y, int X, const float* in16, const float* bfrag, const float* bias,
                    float* out16);
void conv2_mfma_tail(hipStream_t st, int B, int Y, int X, const float* in16, const float* bfrag, const float* bias,
                     const float* w5, const float* b5, float* p_out);

}  // namespace tfl
