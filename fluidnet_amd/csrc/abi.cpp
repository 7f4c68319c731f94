// abi.cpp -- the C ABI of libtfluids_hip.so (include/tfluids_hip.h): argument validation that
// mirrors torch/tfluids/init.lua's asserts, then kernel launches on the context's stream.
// No torch / Lua / C++ types cross the boundary; errors come back as codes + tfl_last_error().
#include "../../include/tfluids_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include <vector>

#include "tfl_device.hpp"
#include "tfl_host.hpp"
#include "tfl_advect.hpp"

struct tfl_layer {
  int cin = 0, cout = 0, k = 0;
  int pool = 1;        // 2: 2x average pooling after the non-linearity (model_utils.lua:184-208)
  int up = 1;          // 2: {Spatial,Volumetric}ConvolutionUpsample -- up^dim sub-position convolutions, pixel-shuffled
  float* w = nullptr;  // device, [sub][tap][cin][cout]
  float* b = nullptr;  // device, [sub][cout]
};
struct tfl_model {
  bool is3d = false;
  int max_c = 0;
  bool multires = false;   // some layer pools / upsamples (the `tog` topologies)
  int max_down = 1;        // coarsest activation resolution = grid / max_down
  std::vector<tfl_layer> layers;
  // 3-D `default` topology (3->8 k3, 8->8 k3, 8->8 k3, 8->8 k1, 8->1 k1): MFMA path (conv_mfma.hip)
  bool mfma3d = false;
  bool valu3d = false;                            // the same three layers on the vector ALUs (conv_valu.hip): the default
  float* bfrag[3] = {nullptr, nullptr, nullptr};  // per-lane B fragments of the three k=3 layers
  float* wino[3] = {nullptr, nullptr, nullptr};   // F(2,3)-in-x transformed weights of the same layers, [dz][dy][cin][4][8]
  float* tail_w4 = nullptr;                       // [8][8] (out, in) of the 8->8 k1 layer
  float* tail_w5 = nullptr;                       // [8] of the 8->1 k1 layer
  float* tail_pack = nullptr;                     // conv_valu.hip: {bias3[8], w4[8][8], b4[8], w5[8], b5[1]} in one buffer
  // conv_mfma16.hip: split-operand fp16 MFMA (TFL_CONV_PATH=mfma16)
  bool m16 = false;
  void* wfrag16[3] = {nullptr, nullptr, nullptr}; // A fragments of the three k=3 layers (conv3_m16_pack_weights)
  float post16[3] = {0.0f, 0.0f, 0.0f};           // 2^-(11 + e) of each layer
  float post16_tail = 0.0f;                       // the same for the tail's 8 -> 8 (k = 1) layer (conv3_m16_pack_tail)
  unsigned long long* d_range_err = nullptr;      // blocks that clamped an activation at the fp16 range
  unsigned long long* h_range = nullptr;          // pinned, device-mapped: the projection kernel copies a non-zero count here, so the
  unsigned long long* d_range_host = nullptr;     // NEXT call sees it without a stream sync (d_range_host = its device address)
  // 2-D `default` topology (3->16, 16->16 x3 k3, 16->1 k1): MFMA path (conv2d_mfma.hip)
  bool mfma2d = false;
  float* bfrag2[4] = {nullptr, nullptr, nullptr, nullptr};
  double* d_stats = nullptr;  // [2 * kMaxBatch]: sum(u), sum(u^2) per sample
  long long stat_pairs_per_plane = 0;   // partial pairs per z-plane the last tfl_model_begin wrote (model.hip model_pre's layout)
  unsigned* d_ticket = nullptr;   // model.hip publish_and_maybe_reduce: blocks of the current k_bcs_div_stats launch that have published
  // forward-graph switches (tfl_model_opts); `custom` = any of them differs from default_conf.lua -> generic kernels only
  tfl_model_opts opts = {1, 0, 1, 1, TFL_NORM_UDIV, TFL_NORMFUNC_STD, TFL_NONLIN_RELU, 0};
  bool custom = false;
  int in_c = 3;               // net input channels
};

#include "tfl_ctx.hpp"
static const int kMaxBatch = 1024;

// ---- per-kernel event profiler (TFL_TIMED in the launchers) ---------------------------------------
namespace tfl {
thread_local ZWin g_zwin = {0, 0, 0, 0};
thread_local ZOrigin g_zorigin = {0, 0};
thread_local int g_advect_fast = 0;
thread_local BcFoldArg g_fold = {nullptr, 0u, 0u};
thread_local bool g_fold_done = false;
thread_local BuoyFold g_buoy = {nullptr, 0.0f, 0.0f, 0.0f};
thread_local bool g_buoy_done = false;
struct ProfRec { const char* name; hipEvent_t e0, e1; };
struct Profiler { std::vector<ProfRec> recs; };
static thread_local Profiler* g_prof = nullptr;

KernelTimer::KernelTimer(const char* name, hipStream_t st, bool ext) : slot_(-1), st_(st), ext_(ext) {
  if (!g_prof) return;
  ProfRec r; r.name = name;
  if (hipEventCreate(&r.e0) != hipSuccess) return;
  if (hipEventCreate(&r.e1) != hipSuccess) { (void)hipEventDestroy(r.e0); return; }
  if (!ext_) (void)hipEventRecord(r.e0, st);
  g_prof->recs.push_back(r);
  slot_ = (int)g_prof->recs.size() - 1;
}
KernelTimer::~KernelTimer() {
  if (slot_ >= 0 && g_prof && !ext_) (void)hipEventRecord(g_prof->recs[slot_].e1, st_);
}
hipEvent_t KernelTimer::start() const { return (slot_ >= 0 && g_prof) ? g_prof->recs[slot_].e0 : nullptr; }
hipEvent_t KernelTimer::stop() const { return (slot_ >= 0 && g_prof) ? g_prof->recs[slot_].e1 : nullptr; }
}  // namespace tfl

namespace {

int fail(tfl_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  if (ctx) ctx->err = buf;
  return code;
}

#define HIP_TRY(ctx, call)                                                                     \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) return fail(ctx, TFL_EHIP, "%s: %s", #call, hipGetErrorString(e_)); \
  } while (0)

int check_launch(tfl_ctx* ctx, const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(ctx, TFL_EHIP, "%s: %s", what, hipGetErrorString(e));
  return TFL_OK;
}

bool same_dims(const tfl_tensor* a, const tfl_tensor* b) {
  return a->B == b->B && a->Z == b->Z && a->Y == b->Y && a->X == b->X;
}

// init.lua's shape asserts (e.g. :99-120): flags scalar, U has 2 (2-D, Z==1) or 3 channels, all same B/Z/Y/X.
int check_flags(tfl_ctx* ctx, const char* op, const tfl_tensor* flags) {
  if (!ctx) return TFL_EINVAL;
  if (!flags || !flags->data) return fail(ctx, TFL_EINVAL, "%s: flags is null", op);
  if (flags->C != 1) return fail(ctx, TFL_EINVAL, "%s: flags is not scalar", op);
  if (flags->B < 1 || flags->Z < 1 || flags->Y < 1 || flags->X < 1) return fail(ctx, TFL_EINVAL, "%s: empty grid", op);
  if ((long long)flags->Z * flags->Y * flags->X * 3 >= (1ll << 31))
    return fail(ctx, TFL_EINVAL, "%s: grid too large for 32-bit cell offsets", op);
  if ((long long)flags->Z * flags->B > 65535) return fail(ctx, TFL_EINVAL, "%s: B*Z exceeds the launch grid limit", op);
  return TFL_OK;
}
int check_vel(tfl_ctx* ctx, const char* op, const char* name, const tfl_tensor* U, const tfl_tensor* flags, int is3D) {
  if (!U || !U->data) return fail(ctx, TFL_EINVAL, "%s: %s is null", op, name);
  if (!same_dims(U, flags)) return fail(ctx, TFL_EINVAL, "%s: %s size mismatch", op, name);
  if (is3D) {
    if (U->C != 3) return fail(ctx, TFL_EINVAL, "%s: 3D velocity field must have 3 channels", op);
  } else {
    if (flags->Z != 1) return fail(ctx, TFL_EINVAL, "%s: 2D velocity field but zdepth > 1", op);
    if (U->C != 2) return fail(ctx, TFL_EINVAL, "%s: 2D velocity field must have only 2 channels", op);
  }
  return TFL_OK;
}
int check_scalar(tfl_ctx* ctx, const char* op, const char* name, const tfl_tensor* s, const tfl_tensor* flags) {
  if (!s || !s->data) return fail(ctx, TFL_EINVAL, "%s: %s is null", op, name);
  if (s->C != 1 || !same_dims(s, flags)) return fail(ctx, TFL_EINVAL, "%s: %s size mismatch", op, name);
  return TFL_OK;
}
#define TRY(x) do { int rc_ = (x); if (rc_ != TFL_OK) return rc_; } while (0)

// Operators that honour tfl_set_z_window open one of these before launching: the launchers read the window from the
// calling thread (tfl_host.hpp make_dom); it is cleared again on the way out so that every other operator -- and every
// other context used from this thread -- sees the whole array.
// The same scope carries tfl_simulate_step's setConstVals request (tfl_host.hpp BcFold) to the launchers and the
// acknowledgement back into the context.
struct WindowScope {
  explicit WindowScope(tfl_ctx* c) : c_(c) {
    tfl::g_zwin = c->zwin; tfl::g_zorigin = c->zorigin; tfl::g_advect_fast = c->advect_fast;
    tfl::g_fold = c->fold; tfl::g_fold_done = false;
    tfl::g_buoy = c->buoy; tfl::g_buoy_done = false;
  }
  ~WindowScope() {
    tfl::g_zwin = tfl::ZWin{0, 0, 0, 0}; tfl::g_zorigin = tfl::ZOrigin{0, 0}; tfl::g_advect_fast = 0;
    c_->fold_done = c_->fold_done || tfl::g_fold_done;
    tfl::g_fold = tfl::no_fold(); tfl::g_fold_done = false;
    c_->buoy_done = c_->buoy_done || tfl::g_buoy_done;
    tfl::g_buoy = tfl::no_buoy(); tfl::g_buoy_done = false;
  }
  tfl_ctx* c_;
};
int stages_of(const tfl_ctx* c) { return c->stages ? c->stages : 0xff; }

// generic/advect_type.cc:18-37
int parse_method(const char* m) {
  if (!m) return -1;
  static const char* names[] = {"euler", "maccormack", "eulerOurs", "rk2Ours", "rk3Ours", "maccormackOurs"};
  for (int i = 0; i < 6; i++) if (strcmp(m, names[i]) == 0) return i;
  return -1;
}

float get_dx(const tfl_ctx* c, const tfl_tensor* f) {  // grid.cc:37-40
  if (c && c->dx_dim > 0) return 1.0f / (float)c->dx_dim;
  if (c && c->dx_override > 0.0f) return c->dx_override;
  int m = f->X > f->Y ? f->X : f->Y;
  if (f->Z > m) m = f->Z;
  return 1.0f / (float)m;
}

}  // namespace

// (library-internal, C++ linkage: the device word of a model's fp16 range counter, for the z-slab step's collective gate -- simulate.cpp)
namespace tfl { const unsigned long long* model_range_counter(const tfl_model* m) { return m ? m->d_range_err : nullptr; } }

extern "C" {

int tfl_abi_version(void) { return TFL_ABI_VERSION; }

tfl_ctx* tfl_create(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device < 0 || device >= n) return nullptr;
  if (hipSetDevice(device) != hipSuccess) return nullptr;
  tfl_ctx* c = new tfl_ctx();
  c->device = device;
  if (hipMalloc((void**)&c->d_trace_err, sizeof(unsigned long long)) != hipSuccess ||
      hipMemset(c->d_trace_err, 0, sizeof(unsigned long long)) != hipSuccess ||
      hipMalloc((void**)&c->d_resid, sizeof(double) * kMaxBatch) != hipSuccess ||
      hipHostMalloc((void**)&c->h_resid, sizeof(double) * kMaxBatch, hipHostMallocDefault) != hipSuccess ||
      hipMalloc((void**)&c->d_reach, sizeof(float)) != hipSuccess ||
      hipMemset(c->d_reach, 0, sizeof(float)) != hipSuccess ||      // (a sticky maximum since round 6: nothing resets it per step)
      hipHostMalloc((void**)&c->h_reach, 2 * sizeof(float), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&c->d_reach_host, c->h_reach, 0) != hipSuccess ||
      hipMalloc((void**)&c->d_reach_tick, sizeof(unsigned)) != hipSuccess ||
      hipMemset(c->d_reach_tick, 0, sizeof(unsigned)) != hipSuccess) {
    tfl_destroy(c);
    return nullptr;
  }
  c->h_reach[0] = 0.0f; reinterpret_cast<unsigned*>(c->h_reach)[1] = 0u;
  if (const char* m = getenv("TFL_ADVECT_MODE")) c->advect_fast = (strcmp(m, "fast") == 0 || strcmp(m, "1") == 0) ? 1 : 0;
  return c;
}

int tfl_set_advect_mode(tfl_ctx* c, int mode) {
  if (!c) return TFL_EINVAL;
  if (mode != TFL_ADVECT_EXACT && mode != TFL_ADVECT_FAST) return fail(c, TFL_EINVAL, "set_advect_mode: unknown mode %d", mode);
  c->advect_fast = mode == TFL_ADVECT_FAST ? 1 : 0;
  return TFL_OK;
}
int tfl_get_advect_mode(const tfl_ctx* c) { return c ? (c->advect_fast ? TFL_ADVECT_FAST : TFL_ADVECT_EXACT) : TFL_EINVAL; }

void tfl_destroy(tfl_ctx* c) {
  if (!c) return;
  { std::lock_guard<std::mutex> lock(c->wall_mu); for (tfl_wall_plan* p : c->wall_plans) p->owner = nullptr; c->wall_plans.clear(); }      // the host still owns (and frees) them
  if (c->d_reach) (void)hipFree(c->d_reach);
  if (c->h_reach) (void)hipHostFree(c->h_reach);
  if (c->d_reach_tick) (void)hipFree(c->d_reach_tick);
  if (c->h_reach_flags) (void)hipHostFree(c->h_reach_flags);
  if (c->d_trace_err) (void)hipFree(c->d_trace_err);
  if (c->d_resid) (void)hipFree(c->d_resid);
  if (c->h_resid) (void)hipHostFree(c->h_resid);
  delete c;
}

int tfl_set_stream(tfl_ctx* c, void* s) {
  if (!c) return TFL_EINVAL;
  c->stream = (hipStream_t)s;
  return TFL_OK;
}

const char* tfl_last_error(const tfl_ctx* c) { return c ? c->err.c_str() : "null context"; }

int tfl_set_dx_override(tfl_ctx* c, float dx) {
  if (!c) return TFL_EINVAL;
  c->dx_override = dx > 0.0f ? dx : 0.0f;
  return TFL_OK;
}

int tfl_set_z_window(tfl_ctx* c, int a0, int a1, int b0, int b1) {
  if (!c) return TFL_EINVAL;
  if (a0 < 0 || a1 < a0 || b0 < 0 || b1 < b0 || (a1 > a0 && b1 > b0 && b0 < a1))
    return fail(c, TFL_EINVAL, "set_z_window: [%d,%d) [%d,%d) is not an ordered pair of plane runs", a0, a1, b0, b1);
  c->zwin = tfl::ZWin{a0, a1, b0, b1};
  return TFL_OK;
}

int tfl_set_z_origin(tfl_ctx* c, int z_first, int z_total) {
  if (!c || z_first < 0 || z_total < 0 || (z_total > 0 && z_first >= z_total)) return TFL_EINVAL;
  c->zorigin = tfl::ZOrigin{z_total > 0 ? z_first : 0, z_total};
  return TFL_OK;
}

int tfl_set_stages(tfl_ctx* c, int mask) {
  if (!c || mask < 0) return TFL_EINVAL;
  c->stages = mask;
  return TFL_OK;
}

int tfl_synchronize(tfl_ctx* c) {
  if (!c) return TFL_EINVAL;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return TFL_OK;
}

int tfl_profile_begin(tfl_ctx* c) {
  if (!c) return TFL_EINVAL;
  if (tfl::g_prof) return fail(c, TFL_EINVAL, "profile_begin: a profile is already active on this thread");
  tfl::g_prof = new tfl::Profiler();
  return TFL_OK;
}

int tfl_profile_end(tfl_ctx* c, char* buf, int64_t cap) {
  if (!c) return TFL_EINVAL;
  tfl::Profiler* p = tfl::g_prof;
  if (!p) return fail(c, TFL_EINVAL, "profile_end: no active profile");
  tfl::g_prof = nullptr;
  (void)hipStreamSynchronize(c->stream);
  (void)hipDeviceSynchronize();
  std::vector<std::string> names;
  std::vector<double> ms;
  std::vector<long long> calls;
  for (auto& r : p->recs) {
    float t = 0.0f;
    if (hipEventElapsedTime(&t, r.e0, r.e1) != hipSuccess) t = 0.0f;
    (void)hipEventDestroy(r.e0);
    (void)hipEventDestroy(r.e1);
    size_t i = 0;
    for (; i < names.size(); i++) if (names[i] == r.name) break;
    if (i == names.size()) { names.push_back(r.name); ms.push_back(0.0); calls.push_back(0); }
    ms[i] += t; calls[i] += 1;
  }
  delete p;
  std::string js = "{";
  for (size_t i = 0; i < names.size(); i++) {
    char tmp[256];
    snprintf(tmp, sizeof(tmp), "%s\"%s\": {\"calls\": %lld, \"ms\": %.6f}", i ? ", " : "", names[i].c_str(), calls[i], ms[i]);
    js += tmp;
  }
  js += "}";
  if (buf && cap > 0) {
    const size_t n = js.size() < (size_t)cap - 1 ? js.size() : (size_t)cap - 1;
    memcpy(buf, js.data(), n);
    buf[n] = 0;
  }
  return (int)names.size();
}

int64_t tfl_trace_errors(tfl_ctx* c) {
  if (!c) return -1;
  unsigned long long v = 0;
  if (hipMemcpyAsync(&v, c->d_trace_err, sizeof(v), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
  if (hipMemsetAsync(c->d_trace_err, 0, sizeof(v), c->stream) != hipSuccess) return -1;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
  return (int64_t)v;
}

int tfl_advectScalar(tfl_ctx* c, float dt, const tfl_tensor* s, const tfl_tensor* U, const tfl_tensor* flags,
                     const tfl_tensor* fwd, const tfl_tensor* bwd, int is3D, const char* method,
                     const tfl_tensor* fwdPos, const tfl_tensor* bwdPos, int boundaryWidth, int sampleOutsideFluid,
                     float maccormackStrength, const tfl_tensor* sDst) {
  (void)bwd; (void)boundaryWidth;  // bnd = 1 like the reference (tfluids.cc:436,467)
  TRY(check_flags(c, "advectScalar", flags));
  TRY(check_vel(c, "advectScalar", "U", U, flags, is3D));
  TRY(check_scalar(c, "advectScalar", "s", s, flags));
  TRY(check_scalar(c, "advectScalar", "sDst", sDst, flags));
  const int m = parse_method(method);
  if (m < 0) return fail(c, TFL_EINVAL, "advectScalar: unknown advection method '%s'", method ? method : "(null)");
  // maccormackOurs' pass B reads `s` only at the cell it writes, so sDst may alias s (in-place, saves the
  // wrapper's copy-back); every other method gathers from s while writing sDst.
  if (sDst->data == s->data && m != tfl::kMacCormackOurs)
    return fail(c, TFL_EINVAL, "advectScalar: sDst must not alias s for method '%s'", method);
  float* fwd_p = nullptr;
  float* bounds_p = nullptr;
  float* mm_p = nullptr;
  if (m == tfl::kMacCormack || m == tfl::kMacCormackOurs) {
    TRY(check_scalar(c, "advectScalar", "fwd", fwd, flags));
    fwd_p = fwd->data;
  }
  if (m == tfl::kMacCormackOurs) {
    TRY(check_vel(c, "advectScalar", "fwdPos", fwdPos, flags, is3D));
    TRY(check_vel(c, "advectScalar", "bwdPos", bwdPos, flags, is3D));
    bounds_p = fwdPos->data;   // two planes: clamp bounds of each cell's forward position
    mm_p = bwdPos->data;       // two planes: 3^dim fluid min / max grid of s
  }
  WindowScope win(c);
  tfl::advect_scalar(c->stream, is3D != 0, m, flags->B, flags->Z, flags->Y, flags->X, dt, maccormackStrength,
                     sampleOutsideFluid != 0, c->d_trace_err, s->data, U->data, flags->data, fwd_p, bounds_p, mm_p,
                     sDst->data, stages_of(c));
  return check_launch(c, "advectScalar");
}

int tfl_advectVel(tfl_ctx* c, float dt, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* fwd,
                  const tfl_tensor* bwd, int is3D, const char* method, int boundaryWidth, float maccormackStrength,
                  const tfl_tensor* UDst) {
  (void)bwd; (void)boundaryWidth;
  TRY(check_flags(c, "advectVel", flags));
  TRY(check_vel(c, "advectVel", "U", U, flags, is3D));
  TRY(check_vel(c, "advectVel", "UDst", UDst, flags, is3D));
  int m = parse_method(method);
  if (m < 0) return fail(c, TFL_EINVAL, "advectVel: unknown advection method '%s'", method ? method : "(null)");
  if (UDst->data == U->data) return fail(c, TFL_EINVAL, "advectVel: UDst must not alias U");
  float* fwd_p = nullptr;
  if (m != tfl::kEuler && m != tfl::kEulerOurs) {
    TRY(check_vel(c, "advectVel", "fwd", fwd, flags, is3D));
    fwd_p = fwd->data;
  }
  WindowScope win(c);
  tfl::advect_vel(c->stream, is3D != 0, m, flags->B, flags->Z, flags->Y, flags->X, dt, maccormackStrength,
                  c->d_trace_err, U->data, flags->data, fwd_p, UDst->data, stages_of(c));
  return check_launch(c, "advectVel");
}

}  // extern "C"
// Library-internal (simulate.cpp's z-slab step; C++ linkage): the MacCormack(Ours) advection of one density channel AND of the
// velocity on a 3-D grid, their passes A as one launch (stage bit 2) and their passes B as one launch (stage bit 4) --
// advect_pair3.hip. Honours the z-window / z-origin / advect mode of the context like the two operators. fold_s / fold_v: the
// setConstVals pairs the passes B apply (tfl_host.hpp BcFold; dev == nullptr: none). TFL_EUNSUPPORTED = not taken, nothing
// launched: the caller runs tfl_advectScalar and tfl_advectVel.
namespace tfl {
int advect_pair(tfl_ctx* c, float dt, float strength, const tfl_tensor* s, const tfl_tensor* U, const tfl_tensor* flags,
                const tfl_tensor* sfwd, const tfl_tensor* sbounds, const tfl_tensor* sDst, const tfl_tensor* vfwd, const tfl_tensor* UDst,
                const BcFoldArg& fold_s, const BcFoldArg& fold_v) {
  TRY(check_flags(c, "advectPair", flags));
  TRY(check_vel(c, "advectPair", "U", U, flags, 1));
  TRY(check_vel(c, "advectPair", "UDst", UDst, flags, 1));
  TRY(check_vel(c, "advectPair", "vfwd", vfwd, flags, 1));
  TRY(check_scalar(c, "advectPair", "s", s, flags));
  TRY(check_scalar(c, "advectPair", "sDst", sDst, flags));
  TRY(check_scalar(c, "advectPair", "sfwd", sfwd, flags));
  if (!sbounds || !sbounds->data || sbounds->C < 2 || !same_dims(sbounds, flags)) return fail(c, TFL_EINVAL, "advectPair: bounds needs two planes of the flags size");
  if (UDst->data == U->data) return fail(c, TFL_EINVAL, "advectPair: UDst must not alias U");
  WindowScope win(c);
  AdvArgs a; a.d = make_dom(flags->Z, flags->Y, flags->X); a.dt = dt; a.strength = strength; a.outside = 0; a.err = c->d_trace_err; a.fast = g_advect_fast;
  a.ord = BlockOrder{};
  if (!advect_pair3(c->stream, a, flags->B, s->data, U->data, flags->data, sfwd->data, sbounds->data, sDst->data, vfwd->data, UDst->data,
                    stages_of(c), fold_s, fold_v))
    return TFL_EUNSUPPORTED;
  return check_launch(c, "advectPair");
}
}  // namespace tfl
extern "C" {

int tfl_setWallBcsForward(tfl_ctx* c, const tfl_tensor* U, const tfl_tensor* flags, int is3D) {
  TRY(check_flags(c, "setWallBcsForward", flags));
  TRY(check_vel(c, "setWallBcsForward", "U", U, flags, is3D));
  tfl::set_wall_bcs(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, U->data, flags->data);
  return check_launch(c, "setWallBcsForward");
}

int tfl_velocityDivergenceForward(tfl_ctx* c, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* UDiv,
                                  int is3D) {
  TRY(check_flags(c, "velocityDivergenceForward", flags));
  TRY(check_vel(c, "velocityDivergenceForward", "U", U, flags, is3D));
  TRY(check_scalar(c, "velocityDivergenceForward", "UDiv", UDiv, flags));
  tfl::velocity_divergence(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, U->data, flags->data,
                           UDiv->data);
  return check_launch(c, "velocityDivergenceForward");
}

int tfl_velocityUpdateForward(tfl_ctx* c, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* p,
                              int is3D) {
  TRY(check_flags(c, "velocityUpdateForward", flags));
  TRY(check_vel(c, "velocityUpdateForward", "U", U, flags, is3D));
  TRY(check_scalar(c, "velocityUpdateForward", "p", p, flags));
  tfl::velocity_update(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, U->data, flags->data, p->data);
  return check_launch(c, "velocityUpdateForward");
}

int tfl_vorticityConfinement(tfl_ctx* c, const tfl_tensor* U, const tfl_tensor* flags, float strength,
                             const tfl_tensor* centered, const tfl_tensor* curl, const tfl_tensor* curlNorm,
                             const tfl_tensor* force, int is3D) {
  (void)centered; (void)force;  // fused away (vorticity.hip)
  TRY(check_flags(c, "vorticityConfinement", flags));
  TRY(check_vel(c, "vorticityConfinement", "U", U, flags, is3D));
  if (!curl || !curl->data || curl->C != 3 || !same_dims(curl, flags))
    return fail(c, TFL_EINVAL, "vorticityConfinement: curl must be a 3-channel grid of the flags size");
  TRY(check_scalar(c, "vorticityConfinement", "curlNorm", curlNorm, flags));
  WindowScope win(c);
  const int stg = stages_of(c);   // tfl_set_stages: 2 = pass A (curl), 4 = pass B (confinement force)
  tfl::vorticity_confinement(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, U->data, flags->data,
                             strength, curl->data, curlNorm->data, ((stg & 2) ? 1 : 0) | ((stg & 4) ? 2 : 0));
  return check_launch(c, "vorticityConfinement");
}

int tfl_vorticityConfinementFrom(tfl_ctx* c, const tfl_tensor* USrc, const tfl_tensor* U, const tfl_tensor* flags, float strength,
                                 const tfl_tensor* curl, const tfl_tensor* curlNorm, int is3D) {
  TRY(check_flags(c, "vorticityConfinementFrom", flags));
  TRY(check_vel(c, "vorticityConfinementFrom", "U", U, flags, is3D));
  TRY(check_vel(c, "vorticityConfinementFrom", "USrc", USrc, flags, is3D));
  if (USrc->data == U->data) return fail(c, TFL_EINVAL, "vorticityConfinementFrom: U must not alias USrc (use vorticityConfinement)");
  if (!curl || !curl->data || curl->C != 3 || !same_dims(curl, flags))
    return fail(c, TFL_EINVAL, "vorticityConfinementFrom: curl must be a 3-channel grid of the flags size");
  TRY(check_scalar(c, "vorticityConfinementFrom", "curlNorm", curlNorm, flags));
  WindowScope win(c);
  if (is3D && !c->vort_from_two_launch && tfl::vorticity_confinement_fused_ok(true, flags->Z, flags->Y, flags->X) &&
      tfl::vorticity_confinement_fused(c->stream, flags->B, flags->Z, flags->Y, flags->X, USrc->data, U->data, flags->data, strength))
    return check_launch(c, "vorticityConfinementFrom");
  // 2-D, a grid below the fused kernel's size, or TFL_VORT_FUSED=0: the two launches read USrc and write U (round 5) ...
  // Under a z-window pass A must cover what pass B taps: |curl| two planes below / one above, curl one below -- the
  // window widened by two planes either way, its two runs merged where they meet; pass B then runs on the window itself.
  const tfl::ZWin w0 = tfl::g_zwin;
  const bool windowed = w0.a1 > w0.a0 || w0.b1 > w0.b0;
  if (windowed) {
    tfl::ZWin wa = {std::max(w0.a0 - 2, 0), std::min(w0.a1 + 2, (int)flags->Z), 0, 0};
    if (w0.b1 > w0.b0) { wa.b0 = std::max(w0.b0 - 2, 0); wa.b1 = std::min(w0.b1 + 2, (int)flags->Z); }
    if (w0.a1 <= w0.a0) { wa.a0 = wa.b0; wa.a1 = wa.b1; wa.b0 = wa.b1 = 0; }
    else if (wa.b1 > wa.b0 && wa.b0 <= wa.a1) { wa.a1 = wa.b1; wa.b0 = wa.b1 = 0; }
    tfl::g_zwin = wa;
    const bool a_ok = tfl::vorticity_confinement(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, U->data, flags->data, strength,
                                                 curl->data, curlNorm->data, 1, USrc->data);
    tfl::g_zwin = w0;
    if (a_ok && tfl::vorticity_confinement(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, U->data, flags->data, strength,
                                           curl->data, curlNorm->data, 2, USrc->data))
      return check_launch(c, "vorticityConfinementFrom");
    if (a_ok) return fail(c, TFL_EUNSUPPORTED, "vorticityConfinementFrom: pass B could not read USrc on this grid under a z-window");
  } else if (tfl::vorticity_confinement(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, U->data, flags->data, strength,
                                        curl->data, curlNorm->data, 3, USrc->data))
    return check_launch(c, "vorticityConfinementFrom");
  if (windowed)      // (the one-cell kernels below run in place on a copy: the same two windows are needed there -- not built)
    return fail(c, TFL_EUNSUPPORTED, "vorticityConfinementFrom: this grid takes the one-cell kernels, which do not run from USrc under a z-window; "
                                     "copy USrc into U and call tfl_vorticityConfinement per pass (tfl_set_stages) with the passes' own windows");
  // ... or, where only the one-cell kernels apply (X % 4 != 0, misaligned views), the planes of the window are copied and the
  // two-launch form runs in place
  {
    const tfl::Dom d = tfl::make_dom(flags->Z, flags->Y, flags->X);
    const size_t plane = sizeof(float) * (size_t)flags->Y * flags->X;
    const int C = is3D ? 3 : 2;
    for (int b = 0; b < flags->B; b++)
      for (int ch = 0; ch < C; ch++) {
        const size_t row = ((size_t)b * C + ch) * flags->Z;
        if (d.n0 > 0) HIP_TRY(c, hipMemcpyAsync(U->data + (row + d.w0) * (plane / 4), USrc->data + (row + d.w0) * (plane / 4), plane * d.n0, hipMemcpyDeviceToDevice, c->stream));
        if (d.nw > d.n0) HIP_TRY(c, hipMemcpyAsync(U->data + (row + d.w1) * (plane / 4), USrc->data + (row + d.w1) * (plane / 4), plane * (d.nw - d.n0), hipMemcpyDeviceToDevice, c->stream));
      }
  }
  tfl::vorticity_confinement(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, U->data, flags->data, strength,
                             curl->data, curlNorm->data, 3);
  return check_launch(c, "vorticityConfinementFrom");
}

int tfl_addBuoyancyFrom(tfl_ctx* c, const tfl_tensor* USrc, const tfl_tensor* U, const tfl_tensor* flags,
                        const tfl_tensor* density, const float gravity[3], float dt, int is3D) {
  TRY(check_flags(c, "addBuoyancy", flags));
  TRY(check_vel(c, "addBuoyancy", "U", U, flags, is3D));
  TRY(check_vel(c, "addBuoyancy", "USrc", USrc, flags, is3D));
  TRY(check_scalar(c, "addBuoyancy", "density", density, flags));
  if (!gravity) return fail(c, TFL_EINVAL, "addBuoyancy: gravity is null");
  const float sc = dt / get_dx(c, flags);  // strength = -gravity * (dt / dx), tfluids.cc:1190-1192
  WindowScope win(c);
  tfl::add_buoyancy(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, USrc->data, U->data, flags->data,
                    density->data, -gravity[0] * sc, -gravity[1] * sc, -gravity[2] * sc);
  return check_launch(c, "addBuoyancy");
}

int tfl_addBuoyancy(tfl_ctx* c, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* density,
                    const float gravity[3], float* strengthTmp, float dt, int is3D) {
  (void)strengthTmp;
  return tfl_addBuoyancyFrom(c, U, U, flags, density, gravity, dt, is3D);
}

int tfl_addGravity(tfl_ctx* c, const tfl_tensor* U, const tfl_tensor* flags, const float gravity[3], float dt,
                   int is3D, float* forceTmp) {
  (void)forceTmp;
  TRY(check_flags(c, "addGravity", flags));
  TRY(check_vel(c, "addGravity", "U", U, flags, is3D));
  if (!gravity) return fail(c, TFL_EINVAL, "addGravity: gravity is null");
  const float sc = dt / get_dx(c, flags);  // force = gravity * (dt / dx), tfluids.cc:1265-1267
  WindowScope win(c);
  tfl::add_gravity(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, U->data, flags->data,
                   gravity[0] * sc, gravity[1] * sc, gravity[2] * sc);
  return check_launch(c, "addGravity");
}

int tfl_emptyDomain(tfl_ctx* c, const tfl_tensor* flags, int is3D, int bnd) {
  TRY(check_flags(c, "emptyDomain", flags));
  if (!is3D && flags->Z != 1) return fail(c, TFL_EINVAL, "emptyDomain: 2D domain but zdepth > 1");
  tfl::empty_domain(c->stream, is3D != 0, bnd, flags->B, flags->Z, flags->Y, flags->X, flags->data);
  return check_launch(c, "emptyDomain");
}

int tfl_flagsToOccupancy(tfl_ctx* c, const tfl_tensor* flags, const tfl_tensor* occupancy) {
  TRY(check_flags(c, "flagsToOccupancy", flags));
  TRY(check_scalar(c, "flagsToOccupancy", "occupancy", occupancy, flags));
  tfl::flags_to_occupancy(c->stream, (long long)flags->B * flags->Z * flags->Y * flags->X, flags->data,
                          occupancy->data);
  return check_launch(c, "flagsToOccupancy");
}

int tfl_rectangularBlur(tfl_ctx* c, const tfl_tensor* src, int blurRad, int is3D, const tfl_tensor* dst,
                        const tfl_tensor* tmp) {
  if (!c) return TFL_EINVAL;
  if (!src || !dst || !tmp || !src->data || !dst->data || !tmp->data) return fail(c, TFL_EINVAL, "rectangularBlur: null tensor");
  if (!same_dims(dst, src) || dst->C != src->C || !same_dims(tmp, src) || tmp->C != src->C)
    return fail(c, TFL_EINVAL, "rectangularBlur: src, dst and tmp must have the same size");
  if (blurRad <= 0) return fail(c, TFL_EINVAL, "rectangularBlur: blurRad must be a positive, non-zero integer");   // init.lua:586-587
  if (!is3D && src->Z != 1) return fail(c, TFL_EINVAL, "rectangularBlur: 2D field but zdepth > 1");
  if (dst->data == src->data || tmp->data == src->data || tmp->data == dst->data)
    return fail(c, TFL_EINVAL, "rectangularBlur: src, dst and tmp must not alias");
  tfl::rectangular_blur(c->stream, is3D != 0, src->B, src->C, src->Z, src->Y, src->X, blurRad, src->data, dst->data, tmp->data);
  return check_launch(c, "rectangularBlur");
}

int tfl_signedDistanceField(tfl_ctx* c, const tfl_tensor* flags, int searchRad, int is3D, const tfl_tensor* dst) {
  TRY(check_flags(c, "signedDistanceField", flags));
  TRY(check_scalar(c, "signedDistanceField", "dst", dst, flags));
  if (searchRad <= 0) return fail(c, TFL_EINVAL, "signedDistanceField: searchRad must be a positive, non-zero integer");
  if (!is3D && flags->Z != 1) return fail(c, TFL_EINVAL, "signedDistanceField: 2D domain but zdepth > 1");
  tfl::signed_distance_field(c->stream, flags->B, flags->Z, flags->Y, flags->X, searchRad, flags->data, dst->data);
  return check_launch(c, "signedDistanceField");
}

int64_t tfl_pcg_workspace_floats(int32_t Z, int32_t Y, int32_t X) { return tfl::pcg_workspace_floats(Z, Y, X); }

int tfl_solveLinearSystemPCG(tfl_ctx* c, const tfl_tensor* p, const tfl_tensor* flags, const tfl_tensor* div, int is3D,
                             const char* precondType, float tol, int maxIter, int verbose, float* workspace,
                             int64_t workspace_floats, float* residual) {
  TRY(check_flags(c, "solveLinearSystemPCG", flags));
  TRY(check_scalar(c, "solveLinearSystemPCG", "p", p, flags));
  TRY(check_scalar(c, "solveLinearSystemPCG", "div", div, flags));
  if (!is3D && flags->Z != 1) return fail(c, TFL_EINVAL, "solveLinearSystemPCG: d > 1 for a 2D domain");
  int pc = -1;
  const std::string pt = precondType ? precondType : "ic0";
  if (pt == "none") pc = 0; else if (pt == "ilu0") pc = 1; else if (pt == "ic0") pc = 2;
  if (pc < 0) return fail(c, TFL_EINVAL, "solveLinearSystemPCG: precondType is not supported.");
  if (!workspace || ((uintptr_t)workspace & 7) != 0) return fail(c, TFL_EINVAL, "solveLinearSystemPCG: workspace must be 8-byte aligned");
  if (workspace_floats < tfl::pcg_workspace_floats(flags->Z, flags->Y, flags->X))
    return fail(c, TFL_EINVAL, "solveLinearSystemPCG: workspace too small (tfl_pcg_workspace_floats)");
  if ((long long)flags->Z * flags->Y * flags->X >= (1ll << 31)) return fail(c, TFL_EINVAL, "solveLinearSystemPCG: grid too large for 32-bit cell indices");
  char msg[256] = {0};
  // the pipelined sweeps need every sub-box resident at once: when something else holds part of the GPU they time out (~1 s)
  // and the solve is repeated with one launch per hyperplane -- remembered per context, so that only the FIRST solve pays
  int rc = tfl::pcg_solve(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, p->data, flags->data, div->data,
                          pc, tol, maxIter, verbose, workspace, residual, msg, sizeof(msg), c->wf_skip == 0);
  if (c->wf_skip > 0) c->wf_skip--;
  if (rc == -5) {
    // back off for 16, 32, ... 1024 solves, then try the pipelined sweeps again; say so once per latch
    c->wf_timeouts++;
    c->wf_skip = 16 << (c->wf_timeouts < 7 ? c->wf_timeouts - 1 : 6);
    fprintf(stderr, "[tfl] solveLinearSystemPCG: a pipelined triangular sweep timed out (something else holds part of the GPU?); "
                    "this solve and the next %d on this context use hyperplane sweeps (slower), then the pipelined form is retried\n", c->wf_skip);
  }
  if (rc == -5)
    rc = tfl::pcg_solve(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, p->data, flags->data, div->data,
                        pc, tol, maxIter, verbose, workspace, residual, msg, sizeof(msg), false);
  if (rc == -1 || rc == -3) return fail(c, TFL_EINVAL, "%s", msg);
  if (rc == -2) return fail(c, TFL_EINVAL, "%s", msg);
  if (rc != 0) return fail(c, TFL_EHIP, "%s", msg);
  return check_launch(c, "solveLinearSystemPCG");
}

int64_t tfl_normalize_workspace_floats(int32_t Z, int32_t Y, int32_t X) { return tfl::npm_workspace_floats(Z, Y, X); }

int tfl_normalizePressureMean(tfl_ctx* c, const tfl_tensor* p, const tfl_tensor* flags, int is3D, float* workspace,
                              int64_t workspace_floats) {
  TRY(check_flags(c, "normalizePressureMean", flags));
  TRY(check_scalar(c, "normalizePressureMean", "p", p, flags));
  if (!is3D && flags->Z != 1) return fail(c, TFL_EINVAL, "normalizePressureMean: 2D domain but zdepth > 1");
  if (!workspace || ((uintptr_t)workspace & 7) != 0) return fail(c, TFL_EINVAL, "normalizePressureMean: workspace must be 8-byte aligned");
  if (workspace_floats < tfl::npm_workspace_floats(flags->Z, flags->Y, flags->X))
    return fail(c, TFL_EINVAL, "normalizePressureMean: workspace too small (tfl_normalize_workspace_floats)");
  if ((long long)flags->Z * flags->Y * flags->X >= (1ll << 31)) return fail(c, TFL_EINVAL, "normalizePressureMean: grid too large for 32-bit cell indices");
  char msg[256] = {0};
  const int rc = tfl::normalize_pressure_mean(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, p->data,
                                              flags->data, workspace, msg, sizeof(msg));
  if (rc != 0) return fail(c, TFL_EHIP, "%s", msg);
  return check_launch(c, "normalizePressureMean");
}

int tfl_solveLinearSystemJacobi(tfl_ctx* c, const tfl_tensor* p, const tfl_tensor* flags, const tfl_tensor* div,
                                const tfl_tensor* pPrev, const tfl_tensor* pDelta, const tfl_tensor* pDeltaNorm,
                                int is3D, float pTol, int maxIter, int verbose, float* residual) {
  (void)pDelta; (void)pDeltaNorm;
  TRY(check_flags(c, "solveLinearSystemJacobi", flags));
  TRY(check_scalar(c, "solveLinearSystemJacobi", "p", p, flags));
  TRY(check_scalar(c, "solveLinearSystemJacobi", "div", div, flags));
  TRY(check_scalar(c, "solveLinearSystemJacobi", "pPrev", pPrev, flags));
  if (!is3D && flags->Z != 1) return fail(c, TFL_EINVAL, "solveLinearSystemJacobi: 2D domain but zdepth > 1");
  if (maxIter < 1) return fail(c, TFL_EINVAL, "solveLinearSystemJacobi: At least 1 iteration is needed (maxIter < 1)");
  const int B = flags->B;
  if (B > kMaxBatch) return fail(c, TFL_EINVAL, "solveLinearSystemJacobi: batch size above %d", kMaxBatch);
  const size_t bytes = sizeof(float) * (size_t)B * flags->Z * flags->Y * flags->X;
  // a small 2-D grid with a fixed iteration count (BASELINE config 1): one launch for the whole solve (jacobi.hip)
  if (!is3D && flags->Z == 1 && !(pTol > 0.0f) && !verbose) {
    const bool want = residual != nullptr;
    if (tfl::jacobi_solve_lds(c->stream, B, flags->Y, flags->X, flags->data, div->data, p->data, pPrev->data, maxIter,
                              want ? c->d_resid : nullptr)) {
      if (want) {
        HIP_TRY(c, hipMemcpyAsync(c->h_resid, c->d_resid, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        float res = 0.0f;
        for (int b = 0; b < B; b++) res = std::max(res, (float)std::sqrt(c->h_resid[b]));
        *residual = res;
      }
      return check_launch(c, "solveLinearSystemJacobi");
    }
  }
  // generic/tfluids.cu:1869-1872: both buffers start at zero
  HIP_TRY(c, hipMemsetAsync(p->data, 0, bytes, c->stream));
  HIP_TRY(c, hipMemsetAsync(pPrev->data, 0, bytes, c->stream));
  float* cur = p->data;
  float* prev = pPrev->data;
  float res = 0.0f;
  const bool every = pTol > 0.0f;  // the reference reads the residual back every iteration (:1886);
                                   // with pTol <= 0 it can never stop early, so only the last one matters
  int iter = 0;
  for (;;) {
    const bool last = (iter + 1 >= maxIter);
    const bool want = every || (last && residual != nullptr);
    if (want) HIP_TRY(c, hipMemsetAsync(c->d_resid, 0, sizeof(double) * B, c->stream));
    tfl::jacobi_iteration(c->stream, is3D != 0, B, flags->Z, flags->Y, flags->X, prev, flags->data, div->data, cur,
                          want ? c->d_resid : nullptr);
    if (want) {
      HIP_TRY(c, hipMemcpyAsync(c->h_resid, c->d_resid, sizeof(double) * B, hipMemcpyDeviceToHost, c->stream));
      HIP_TRY(c, hipStreamSynchronize(c->stream));
      res = 0.0f;
      for (int b = 0; b < B; b++) {
        const float r = (float)std::sqrt(c->h_resid[b]);
        if (r > res) res = r;
      }
      if (verbose) printf("Jacobi iteration %d: residual %e\n", iter + 1, (double)res);
      if (every && res < pTol) break;
    }
    iter++;
    if (iter >= maxIter) break;
    float* t = cur; cur = prev; prev = t;
  }
  if (cur != p->data) HIP_TRY(c, hipMemcpyAsync(p->data, cur, bytes, hipMemcpyDeviceToDevice, c->stream));
  if (residual) *residual = res;
  return check_launch(c, "solveLinearSystemJacobi");
}

// TFL_CONV_PATH: unset / "mfma16" = split-operand fp16 MFMA (conv_mfma16.hip: the default), "winograd" = fp32 Winograd on
// the vector ALUs (conv_valu.hip), "mfma" = fp32-operand MFMA (conv_mfma.hip), "direct" = the shape-generic kernels
static bool conv3d_default_is_m16(const char* force) { return !force || !*force || strcmp(force, "mfma16") == 0; }

tfl_model* tfl_model_create(tfl_ctx* c, int is3D, int nlayers, const int32_t* cin, const int32_t* cout,
                            const int32_t* ksize, const float* const* weights, const float* const* biases) {
  return tfl_model_create_ex(c, is3D, nlayers, cin, cout, ksize, nullptr, nullptr, weights, biases);
}

tfl_model* tfl_model_create_ex(tfl_ctx* c, int is3D, int nlayers, const int32_t* cin, const int32_t* cout,
                               const int32_t* ksize, const int32_t* pool, const int32_t* up,
                               const float* const* weights, const float* const* biases) {
  return tfl_model_create_opts(c, is3D, nlayers, cin, cout, ksize, pool, up, weights, biases, nullptr);
}

tfl_model* tfl_model_create_opts(tfl_ctx* c, int is3D, int nlayers, const int32_t* cin, const int32_t* cout,
                                 const int32_t* ksize, const int32_t* pool, const int32_t* up,
                                 const float* const* weights, const float* const* biases, const tfl_model_opts* opts) {
  if (!c) return nullptr;
  auto bad = [&](const char* m) -> tfl_model* { fail(c, TFL_EINVAL, "model_create: %s", m); return nullptr; };
  if (nlayers < 1 || !cin || !cout || !ksize || !weights || !biases) return bad("null or empty layer description");
  tfl_model_opts o = {1, 0, 1, 1, TFL_NORM_UDIV, TFL_NORMFUNC_STD, TFL_NONLIN_RELU, 0};
  if (opts) {
    o = *opts;
    o.in_pDiv = o.in_pDiv != 0; o.in_UDiv = o.in_UDiv != 0; o.in_div = o.in_div != 0; o.normalize = o.normalize != 0;
    o.pressure_skip = o.pressure_skip != 0;
    if (!o.in_pDiv && !o.in_UDiv && !o.in_div) return bad("Are you sure you dont want any (U, div or p) fields?");   // model.lua:50-52
    if (o.norm_chan < TFL_NORM_UDIV || o.norm_chan > TFL_NORM_DIV) return bad("Incorrect normalize input channel.");
    if (o.norm_func != TFL_NORMFUNC_STD && o.norm_func != TFL_NORMFUNC_L2) return bad("Incorrect normalize input function");
    if (o.nonlin < TFL_NONLIN_RELU || o.nonlin > TFL_NONLIN_SIGMOID) return bad("Bad mconf.nonlinType");
    if (o.pressure_skip && nlayers < 2) return bad("addPressureSkip needs a hidden layer to join pDiv to");
    if (o.pressure_skip && (pool || up)) {
      // the skip joins pDiv at full resolution in front of the last layer: with the layer table in hand, refuse the
      // combinations tfl_model_finish cannot run (it used to report them on every forward instead: ADVICE r02)
      bool multires = false;
      for (int l = 0; l < nlayers; l++) multires = multires || (pool && pool[l] > 1) || (up && up[l] > 1);
      const bool tail = (pool && (pool[nlayers - 1] > 1 || pool[nlayers - 2] > 1)) || (up && (up[nlayers - 1] > 1 || up[nlayers - 2] > 1));
      if (tail || (nlayers > 2 && multires)) {
        fail(c, TFL_EUNSUPPORTED, "model_create: addPressureSkip with pooling / upsampling layers");
        return nullptr;
      }
    }
  }
  const int in_c = o.in_pDiv + o.in_UDiv * (is3D ? 3 : 2) + o.in_div + 1;
  if (cin[0] != in_c) {
    fail(c, TFL_EINVAL, "model_create: the first layer must take %d input channels {pDiv, UDiv, div, occupancy as selected}", in_c);
    return nullptr;
  }
  if (cout[nlayers - 1] != 1) return bad("the last layer must output 1 channel (pressure)");
  tfl_model* m = new tfl_model();
  m->is3d = is3D != 0;
  m->opts = o; m->in_c = in_c;
  m->custom = !(o.in_pDiv && !o.in_UDiv && o.in_div && o.normalize && o.norm_chan == TFL_NORM_UDIV &&
                o.norm_func == TFL_NORMFUNC_STD && o.nonlin == TFL_NONLIN_RELU && !o.pressure_skip);
  auto cleanup = [&](const char* msg) -> tfl_model* { tfl_model_destroy(c, m); return bad(msg); };
  if (hipMalloc((void**)&m->d_stats, sizeof(double) * 2 * kMaxBatch) != hipSuccess) return cleanup("hipMalloc failed");
#ifdef TFL_EXPERIMENTS      // the ticket words of the producer-side stats fold (model.hip kStatTickets): EXPERIMENTS flavour only
  if (hipMalloc((void**)&m->d_ticket, sizeof(unsigned) * (1 << 16)) != hipSuccess || hipMemset(m->d_ticket, 0, sizeof(unsigned) * (1 << 16)) != hipSuccess)
    return cleanup("hipMalloc failed");
#endif
  // The shape-generic kernels are instantiated for 1, 2, 4, 8, 16, 32, 64 output channels. Any other width (the
  // `yang` topology of model.lua:188-205 has 6) is zero-padded to the next one: the extra channels carry
  // relu(0 + 0) = 0 into zero weights of the next layer, i.e. every sum gains exact `+ 0 * 0` terms only.
  auto padded = [](int cch) { for (int w : {1, 2, 4, 8, 16, 32, 64}) if (cch <= w) return w; return -1; };
  int prev_cout_padded = in_c;
  int res_num = 1, res_den = 1;   // resolution of the current activations relative to the grid = res_num / res_den
  for (int l = 0; l < nlayers; l++) {
    tfl_layer L;
    if (cin[l] < 1 || cout[l] < 1 || ksize[l] < 1 || (ksize[l] % 2) != 1) return cleanup("convolution size must be odd and positive");
    const bool skip_in = m->opts.pressure_skip && l + 1 == nlayers;     // this layer also reads pDiv/scale (last channel)
    if (l > 0 && cin[l] != cout[l - 1] + (skip_in ? 1 : 0)) return cleanup("layer channel counts do not chain");
    L.pool = pool ? pool[l] : 1; L.up = up ? up[l] : 1;
    if ((L.pool != 1 && L.pool != 2) || (L.up != 1 && L.up != 2)) return cleanup("pooling / upsampling factors must be 1 or 2");
    if (L.pool > 1 && L.up > 1) return cleanup("Pooling and upsampling in the same layer!");               // model.lua:322-324
    if (L.pool > 1 && l + 1 == nlayers) return cleanup("Pooling is not allowed in the last layer");        // model.lua:247
    res_num *= L.up; res_den *= L.pool;
    if (res_num > res_den) return cleanup("a layer upsamples beyond the grid resolution");
    if (L.pool > 1 || L.up > 1 || res_num != res_den) m->multires = true;
    if (res_den / res_num > m->max_down) m->max_down = res_den / res_num;
    const int cout_p = (l + 1 == nlayers) ? cout[l] : padded(cout[l]);
    if (cout_p < 0) return cleanup("unsupported output channel count (at most 64)");
    const int cin_p = prev_cout_padded + (skip_in ? 1 : 0);     // the skip channel sits after the (padded) hidden channels
    L.cin = cin_p; L.cout = cout_p; L.k = ksize[l];
    prev_cout_padded = cout_p;
    const int taps = m->is3d ? L.k * L.k * L.k : L.k * L.k;
    const int S = m->is3d ? L.up * L.up * L.up : L.up * L.up;
    // cudnn weight [cout*S][cin][taps] (output channel index = o*S + sub) -> [sub][tap][cin_p][cout_p]
    std::vector<float> relaid((size_t)S * taps * L.cin * L.cout, 0.0f);
    for (int sub = 0; sub < S; sub++)
      for (int co = 0; co < cout[l]; co++)
        for (int ci = 0; ci < cin[l]; ci++)
          for (int t = 0; t < taps; t++) {
            const int cid = (skip_in && ci == cin[l] - 1) ? L.cin - 1 : ci;
            relaid[(((size_t)sub * taps + t) * L.cin + cid) * L.cout + co] =
                weights[l][(((size_t)co * S + sub) * cin[l] + ci) * taps + t];
          }
    std::vector<float> bias_p((size_t)S * L.cout, 0.0f);
    for (int sub = 0; sub < S; sub++)
      for (int co = 0; co < cout[l]; co++) bias_p[(size_t)sub * L.cout + co] = biases[l][(size_t)co * S + sub];
    if (hipMalloc((void**)&L.w, relaid.size() * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&L.b, bias_p.size() * sizeof(float)) != hipSuccess ||
        hipMemcpy(L.w, relaid.data(), relaid.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(L.b, bias_p.data(), bias_p.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
      m->layers.push_back(L);
      return cleanup("uploading weights failed");
    }
    m->layers.push_back(L);
    if (L.cout + (m->opts.pressure_skip ? 1 : 0) > m->max_c && l + 1 < nlayers) m->max_c = L.cout + (m->opts.pressure_skip ? 1 : 0);
  }
  if (res_num != res_den) return cleanup("the layers do not return to the grid resolution (pool / up factors)");
  if (m->max_c < 1) m->max_c = 1;
  // ---- MFMA path for the 3-D default topology (TFL_CONV_PATH=direct forces the generic kernels) ----
  const char* force = getenv("TFL_CONV_PATH");
  const bool want_mfma = !(force && strcmp(force, "direct") == 0) && !m->custom;
  const int dflt[5][3] = {{3, 8, 3}, {8, 8, 3}, {8, 8, 3}, {8, 8, 1}, {8, 1, 1}};
  bool match = m->is3d && nlayers == 5 && !m->multires;
  for (int l = 0; match && l < 5; l++) match = cin[l] == dflt[l][0] && cout[l] == dflt[l][1] && ksize[l] == dflt[l][2];
  if (match && want_mfma) {
    for (int l = 0; l < 3; l++) {
      const int ci_n = cin[l];
      std::vector<float> frag((size_t)ci_n * 9 * 64);
      for (int c = 0; c < ci_n; c++)
        for (int kz = 0; kz < 3; kz++)
          for (int ky = 0; ky < 3; ky++)
            for (int lane = 0; lane < 64; lane++) {
              const int k = lane >> 4, n = lane & 15, ph = n >> 3, co = n & 7, kx = k - ph;
              float v = 0.0f;
              if (kx >= 0 && kx <= 2) v = weights[l][((((size_t)co * ci_n + c) * 3 + kz) * 3 + ky) * 3 + kx];
              frag[(((size_t)c * 3 + kz) * 3 + ky) * 64 + lane] = v;
            }
      if (hipMalloc((void**)&m->bfrag[l], frag.size() * sizeof(float)) != hipSuccess ||
          hipMemcpy(m->bfrag[l], frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return cleanup("uploading MFMA weight fragments failed");
    }
    if (hipMalloc((void**)&m->tail_w4, 64 * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&m->tail_w5, 8 * sizeof(float)) != hipSuccess ||
        hipMemcpy(m->tail_w4, weights[3], 64 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(m->tail_w5, weights[4], 8 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
      return cleanup("uploading tail weights failed");
    m->mfma3d = true;
    // default: the vector-ALU kernels (conv_valu.hip; faster than the fp32-MFMA form for 8 output channels, see there).
    // TFL_CONV_PATH=mfma keeps the MFMA kernels.
    m->valu3d = !(force && strcmp(force, "mfma") == 0);
    m->m16 = conv3d_default_is_m16(force);
    // their weights, Winograd-transformed along x: U = (g0, (g0 + g1 + g2)/2, (g0 - g1 + g2)/2, g2) over the three x-taps
    // of every (dz, dy, c_in, c_out)
    if (m->m16) {
      for (int l = 0; l < 3; l++) {
        std::vector<uint16_t> fr(tfl::conv3_m16_frag_halves(cin[l]));
        m->post16[l] = tfl::conv3_m16_pack_weights(weights[l], cin[l], fr.data());
        if (l == 2) m->post16_tail = tfl::conv3_m16_pack_tail(weights[3], fr.data());     // the 8 -> 8 (k = 1) layer's A fragment
        if (hipMalloc(&m->wfrag16[l], fr.size() * sizeof(uint16_t)) != hipSuccess ||
            hipMemcpy(m->wfrag16[l], fr.data(), fr.size() * sizeof(uint16_t), hipMemcpyHostToDevice) != hipSuccess)
          return cleanup("uploading fp16 MFMA weight fragments failed");
      }
      if (hipMalloc((void**)&m->d_range_err, sizeof(unsigned long long)) != hipSuccess ||
          hipMemset(m->d_range_err, 0, sizeof(unsigned long long)) != hipSuccess)
        return cleanup("hipMalloc failed");
      // the word a later call reads without a sync; failing to get mapped memory only loses the early warning
      if (hipHostMalloc((void**)&m->h_range, sizeof(unsigned long long), hipHostMallocMapped) == hipSuccess) {
        *m->h_range = 0;
        if (hipHostGetDevicePointer((void**)&m->d_range_host, m->h_range, 0) != hipSuccess) { (void)hipHostFree(m->h_range); m->h_range = nullptr; m->d_range_host = nullptr; }
      } else m->h_range = nullptr;
    }
    if (m->valu3d) {
      std::vector<float> tp(8 + 64 + 8 + 8 + 1 + 1);
      for (int i = 0; i < 8; i++) tp[i] = biases[2][i];
      for (int i = 0; i < 64; i++) tp[8 + i] = weights[3][i];
      for (int i = 0; i < 8; i++) { tp[72 + i] = biases[3][i]; tp[80 + i] = weights[4][i]; }
      tp[88] = biases[4][0];
      tp[89] = m->post16_tail;        // conv_mfma16.hip kTailPost4 (0 on the fp32 paths, which do not read it)
      if (hipMalloc((void**)&m->tail_pack, tp.size() * sizeof(float)) != hipSuccess ||
          hipMemcpy(m->tail_pack, tp.data(), tp.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return cleanup("uploading tail weights failed");
      for (int l = 0; l < 3; l++) {
        const int ci_n = cin[l];
        std::vector<float> u((size_t)9 * 4 * ci_n * 8);
        for (int co = 0; co < 8; co++)
          for (int c = 0; c < ci_n; c++)
            for (int zy = 0; zy < 9; zy++) {
              const float* g = weights[l] + (((size_t)co * ci_n + c) * 9 + zy) * 3;
              const float uu[4] = {g[0], 0.5f * ((g[0] + g[2]) + g[1]), 0.5f * ((g[0] + g[2]) - g[1]), g[2]};
              for (int p = 0; p < 4; p++) u[(((size_t)zy * ci_n + c) * 4 + p) * 8 + co] = uu[p];
            }
        if (hipMalloc((void**)&m->wino[l], u.size() * sizeof(float)) != hipSuccess ||
            hipMemcpy(m->wino[l], u.data(), u.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
          return cleanup("uploading transformed weights failed");
      }
    }
  }
  const int dflt2[5][3] = {{3, 16, 3}, {16, 16, 3}, {16, 16, 3}, {16, 16, 3}, {16, 1, 1}};
  bool match2 = !m->is3d && nlayers == 5 && !m->multires;
  for (int l = 0; match2 && l < 5; l++) match2 = cin[l] == dflt2[l][0] && cout[l] == dflt2[l][1] && ksize[l] == dflt2[l][2];
  if (match2 && want_mfma) {
    for (int l = 0; l < 4; l++) {
      const int ci_n = cin[l], c4n = (ci_n + 3) / 4;
      // B[k][n] of MFMA (tap, c4): lane = k*16 + n holds w[n][4*c4 + k][dy][dx] (0 for padded channels)
      std::vector<float> frag((size_t)9 * c4n * 64);
      for (int tap = 0; tap < 9; tap++)
        for (int c4 = 0; c4 < c4n; c4++)
          for (int lane = 0; lane < 64; lane++) {
            const int k = lane >> 4, n = lane & 15, c = 4 * c4 + k;
            frag[((size_t)tap * c4n + c4) * 64 + lane] = c < ci_n ? weights[l][((size_t)n * ci_n + c) * 9 + tap] : 0.0f;
          }
      if (hipMalloc((void**)&m->bfrag2[l], frag.size() * sizeof(float)) != hipSuccess ||
          hipMemcpy(m->bfrag2[l], frag.data(), frag.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
        return cleanup("uploading 2-D MFMA weight fragments failed");
    }
    if (hipMalloc((void**)&m->tail_w5, 16 * sizeof(float)) != hipSuccess ||
        hipMemcpy(m->tail_w5, weights[4], 16 * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
      return cleanup("uploading tail weights failed");
    m->mfma2d = true;
  }
  return m;
}

void tfl_model_destroy(tfl_ctx* c, tfl_model* m) {
  (void)c;
  if (!m) return;
  for (auto& L : m->layers) { if (L.w) (void)hipFree(L.w); if (L.b) (void)hipFree(L.b); }
  for (int l = 0; l < 3; l++) if (m->bfrag[l]) (void)hipFree(m->bfrag[l]);
  for (int l = 0; l < 3; l++) if (m->wino[l]) (void)hipFree(m->wino[l]);
  for (int l = 0; l < 3; l++) if (m->wfrag16[l]) (void)hipFree(m->wfrag16[l]);
  if (m->d_range_err) (void)hipFree(m->d_range_err);
  if (m->h_range) (void)hipHostFree(m->h_range);
  for (int l = 0; l < 4; l++) if (m->bfrag2[l]) (void)hipFree(m->bfrag2[l]);
  if (m->tail_w4) (void)hipFree(m->tail_w4);
  if (m->tail_pack) (void)hipFree(m->tail_pack);
  if (m->tail_w5) (void)hipFree(m->tail_w5);
  if (m->d_stats) (void)hipFree(m->d_stats);
  if (m->d_ticket) (void)hipFree(m->d_ticket);
  delete m;
}

int64_t tfl_model_range_errors(tfl_ctx* c, tfl_model* m) {
  if (!c || !m) return -1;
  if (!m->d_range_err) return 0;
  unsigned long long v = 0;
  if (hipMemcpyAsync(&v, m->d_range_err, sizeof(v), hipMemcpyDeviceToHost, c->stream) != hipSuccess) return -1;
  if (hipMemsetAsync(m->d_range_err, 0, sizeof(v), c->stream) != hipSuccess) return -1;
  if (hipStreamSynchronize(c->stream) != hipSuccess) return -1;
  if (m->h_range) *(volatile unsigned long long*)m->h_range = 0;      // acknowledged: the next forward runs again
  return (int64_t)v;
}


int64_t tfl_model_range_flag(tfl_ctx* c, tfl_model* m) {
  if (!c || !m) return -1;
  return m->h_range ? (int64_t)*(volatile unsigned long long*)m->h_range : 0;
}

// A forward pass after one that clamped activations at the fp16 range would build on wrong values where the reference's
// fp32 cuDNN carries on: refuse it until the host has acknowledged the count (tfl_model_range_errors).
// The word is written by the device whenever the projection of an earlier pass gets there (no synchronisation), so WHICH later
// call sees it first depends on how far the host runs ahead: the gate is best-effort in time, but it is taken at ONE defined
// point per unit of work -- the entry of a forward pass, or the entry of tfl_simulate_step[_slab] (which then marks the
// context `in_step`, so that the model calls inside the step do not refuse half-way through it).
static int range_gate(tfl_ctx* c, tfl_model* m, const char* who) {
  if (c && c->in_step) return TFL_OK;
  if (m && m->h_range && *(volatile unsigned long long*)m->h_range != 0)
    return fail(c, TFL_ERANGE, "%s: an earlier forward pass of this model clamped activations at the fp16 range (|x| > 65504 after the "
                               "std normalisation: a blown-up simulation); its pressure is not the reference's. Read and reset the count with "
                               "tfl_model_range_errors, or create the model under TFL_CONV_PATH=winograd (strict fp32, no range limit)", who);
  return TFL_OK;
}

int64_t tfl_model_workspace_floats(const tfl_model* m, int B, int Z, int Y, int X) {
  if (!m) return -1;
  const int64_t n = (int64_t)B * Z * Y * X;
  // per-block fp64 stat partials (2 doubles = 4 floats per block, kept first for 8-byte alignment)
  // + div[1] + net input[in_c] + two ping-pong activation buffers[max_c] + pPred[1]
  // (+4: the fp16 MFMA path aligns its activation buffers to 16 bytes)
  return 4 * tfl::model_stat_blocks(B, Z, Y, X) + n * (1 + (int64_t)m->in_c + 2 * (int64_t)m->max_c + 1) + (m->m16 ? 4 : 0);
}

namespace {
struct ModelWs { double* partials; float* div; float* x3; float* act[2]; float* pPred; };
int model_ws(tfl_ctx* c, const tfl_model* m, const tfl_tensor* flags, float* workspace, int64_t workspace_floats,
             ModelWs* w) {
  const int B = flags->B, Z = flags->Z, Y = flags->Y, X = flags->X;
  if (B > kMaxBatch) return fail(c, TFL_EINVAL, "model: batch size above %d", kMaxBatch);
  const int64_t n = (int64_t)B * Z * Y * X;
  if (!workspace || workspace_floats < tfl_model_workspace_floats(m, B, Z, Y, X))
    return fail(c, TFL_EINVAL, "model: workspace too small (%lld floats needed)",
                (long long)tfl_model_workspace_floats(m, B, Z, Y, X));
  if (((uintptr_t)workspace & 7) != 0) return fail(c, TFL_EINVAL, "model: workspace must be 8-byte aligned");
  w->partials = (double*)workspace;
  w->div = workspace + 4 * tfl::model_stat_blocks(B, Z, Y, X);
  w->x3 = w->div + n;
  w->act[0] = w->x3 + (int64_t)m->in_c * n;
  if (m->m16) w->act[0] = (float*)(((uintptr_t)w->act[0] + 15) & ~(uintptr_t)15);
  w->act[1] = w->act[0] + (int64_t)m->max_c * n;
  w->pPred = w->act[1] + (int64_t)m->max_c * n;
  return TFL_OK;
}
}  // namespace

float* tfl_model_div(const tfl_model* m, int B, int Z, int Y, int X, float* workspace) {
  if (!m || !workspace) return nullptr;
  return workspace + 4 * tfl::model_stat_blocks(B, Z, Y, X);
}

// the wall codes of THESE flags, if the host registered them with this context (tfl_wall_plan_create), else null
static const unsigned short* wall_code_of(tfl_ctx* c, const tfl_model* m, const tfl_tensor* flags) {
  std::lock_guard<std::mutex> lock(c->wall_mu);
  for (const tfl_wall_plan* wp : c->wall_plans)
    if (wp->flags == flags->data && wp->is3d == m->is3d && wp->B == flags->B && wp->Z == flags->Z && wp->Y == flags->Y && wp->X == flags->X) return wp->code;
  return nullptr;
}

int tfl_model_begin(tfl_ctx* c, tfl_model* m, const tfl_tensor* UDiv, const tfl_tensor* flags, const tfl_tensor* UOut,
                    float* workspace, int64_t workspace_floats, int zlo, int zhi, double* stats) {
  TRY(check_flags(c, "model_begin", flags));
  if (!m) return fail(c, TFL_EINVAL, "model_begin: null model");
  TRY(range_gate(c, m, "model_begin"));
  const int is3D = m->is3d ? 1 : 0;
  TRY(check_vel(c, "model_begin", "UDiv", UDiv, flags, is3D));
  TRY(check_vel(c, "model_begin", "UOut", UOut, flags, is3D));
  if (zlo < 0 || zhi > flags->Z || zlo >= zhi) return fail(c, TFL_EINVAL, "model_begin: bad z range [%d, %d)", zlo, zhi);
  ModelWs w;
  TRY(model_ws(c, m, flags, workspace, workspace_floats, &w));
  // SetWallBcs(UDiv) lands in UOut. UOut may alias UDiv: a thread rewrites only the cell it read, and
  // neighbour values are re-derived from the flags, so a neighbour already holding the BC-applied
  // value gives the identical result (the BC is idempotent).
  WindowScope win(c);
  int stg = stages_of(c);   // tfl_set_stages: 2 = wall BCs + divergence + partial sums, 4 = reduce [zlo, zhi)
  if (c->defer_stats) stg &= ~4;  // tfl_model_forward: the first conv layer reduces the partials itself (round 6)
  m->stat_pairs_per_plane = tfl::model_stat_pairs_per_plane(flags->B, flags->Z, flags->Y, flags->X, UDiv->data, flags->data, UOut->data, w.div);
  const unsigned short* code = wall_code_of(c, m, flags);
  tfl::model_pre(c->stream, m->is3d, flags->B, flags->Z, flags->Y, flags->X, UDiv->data, flags->data, UOut->data,
                 w.div, w.partials, stats ? stats : m->d_stats, zlo, zhi, ((stg & 2) ? 1 : 0) | ((stg & 4) ? 2 : 0), m->d_ticket, code);
  return check_launch(c, "model_begin");
}

tfl_wall_plan* tfl_wall_plan_create(tfl_ctx* c, const tfl_tensor* flags) {
  if (!c || !flags || !flags->data) return nullptr;
  if (check_flags(c, "wall_plan_create", flags) != TFL_OK) return nullptr;
  const long long n = (long long)flags->B * flags->Z * flags->Y * flags->X;
  tfl_wall_plan* p = new tfl_wall_plan();
  p->flags = flags->data; p->B = flags->B; p->Z = flags->Z; p->Y = flags->Y; p->X = flags->X;
  if (hipMalloc((void**)&p->code, (size_t)n * sizeof(unsigned short)) != hipSuccess) { (void)hipGetLastError(); c->err = "wall_plan_create: hipMalloc failed"; delete p; return nullptr; }
  (void)hipDeviceSynchronize();               // one-time set-up: whoever filled the flags (any stream) is done
  p->is3d = flags->Z > 1; p->owner = c;
  // (outside a WindowScope the thread's z-window is empty: the launch covers every plane, whatever window the host has set)
  tfl::wall_code(c->stream, p->is3d, flags->B, flags->Z, flags->Y, flags->X, flags->data, p->code);
  if (hipStreamSynchronize(c->stream) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(p->code); c->err = "wall_plan_create: the code kernel failed"; delete p; return nullptr; }
  std::lock_guard<std::mutex> lock(c->wall_mu);
  for (tfl_wall_plan*& q : c->wall_plans)     // a new plan for the same array replaces the old registration (its owner still frees it)
    if (q->flags == p->flags) { q->owner = nullptr; q = p; return p; }
  c->wall_plans.push_back(p);
  return p;
}

// Take the plan out of its context's registry WITHOUT freeing it: no HIP call, safe from a finalizer (where the hipFree of
// tfl_wall_plan_destroy could invalidate a graph capture in progress). The step stops finding it at once -- which matters when the
// flags array has died and its address may be handed out again before the host gets round to tfl_wall_plan_destroy.
void tfl_wall_plan_retire(tfl_wall_plan* p) {
  if (!p) return;
  tfl_ctx* o = p->owner;
  if (!o) return;
  std::lock_guard<std::mutex> lock(o->wall_mu);
  for (size_t i = 0; i < o->wall_plans.size(); i++)
    if (o->wall_plans[i] == p) { o->wall_plans.erase(o->wall_plans.begin() + (long)i); break; }
  p->owner = nullptr;
}

void tfl_wall_plan_destroy(tfl_ctx* c, tfl_wall_plan* p) {
  if (!p) return;
  (void)c;                                    // (the plan knows its context, and whether that context still exists)
  tfl_wall_plan_retire(p);
  if (p->code) (void)hipFree(p->code);
  delete p;
}

int tfl_model_finish(tfl_ctx* c, tfl_model* m, const tfl_tensor* pDiv, const tfl_tensor* flags,
                     const tfl_tensor* pOut, const tfl_tensor* UOut, float* workspace, int64_t workspace_floats,
                     const double* stats, double count, const tfl_tensor* UBC, const tfl_tensor* UBCInvMask,
                     int doClamp, float lo, float hi) {
  TRY(check_flags(c, "model_finish", flags));
  if (!m) return fail(c, TFL_EINVAL, "model_finish: null model");
  const int is3D = m->is3d ? 1 : 0;
  TRY(check_vel(c, "model_finish", "UOut", UOut, flags, is3D));
  TRY(check_scalar(c, "model_finish", "pDiv", pDiv, flags));
  TRY(check_scalar(c, "model_finish", "pOut", pOut, flags));
  if ((UBC == nullptr) != (UBCInvMask == nullptr)) return fail(c, TFL_EINVAL, "model_finish: UBC and UBCInvMask go together");
  if (UBC) {
    TRY(check_vel(c, "model_finish", "UBC", UBC, flags, is3D));
    TRY(check_vel(c, "model_finish", "UBCInvMask", UBCInvMask, flags, is3D));
  }
  if (!(count > 1.0)) return fail(c, TFL_EINVAL, "model_finish: Sample variance requires more than one sample.");
  ModelWs w;
  TRY(model_ws(c, m, flags, workspace, workspace_floats, &w));
  const int B = flags->B, Z = flags->Z, Y = flags->Y, X = flags->X;
  const double* st_in = stats ? stats : m->d_stats;
  hipStream_t st = c->stream;
  if (m->custom) {
    // tfl_model_opts: the input scale comes from another field / function / not at all. All three reach the kernels
    // through the (stats, count) pair scale_from_stats reads -- sqrt((n s2 - s1^2) / (n (n - 1))): the l2 norm is
    // (s1, s2, n) = (0, sum x^2, 2), "no scaling" is (0, 1, 2).
    if (stats || c->stages || c->zwin.a1 > c->zwin.a0 || c->zwin.b1 > c->zwin.b0)
      return fail(c, TFL_EUNSUPPORTED, "model_finish: models with non-default tfl_model_opts run un-sharded only");
    const tfl_model_opts& o = m->opts;
    const long long cells = (long long)Z * Y * X;
    const float* field = o.norm_chan == TFL_NORM_PDIV ? pDiv->data : (o.norm_chan == TFL_NORM_DIV ? w.div : UOut->data);
    const long long nf = o.norm_chan == TFL_NORM_UDIV ? cells * (m->is3d ? 3 : 2) : cells;
    if (!o.normalize) { tfl::model_field_stats(st, B, nf, field, 2, m->d_stats); count = 2.0; }
    else if (o.norm_func == TFL_NORMFUNC_L2) { tfl::model_field_stats(st, B, nf, field, 1, m->d_stats); count = 2.0; }
    else if (o.norm_chan != TFL_NORM_UDIV) {
      if (nf < 2) return fail(c, TFL_EINVAL, "model_finish: Sample variance requires more than one sample.");
      tfl::model_field_stats(st, B, nf, field, 0, m->d_stats); count = (double)nf;
    }
  }
  WindowScope win(c);
  // tfl_set_stages (z-slab ranks run each layer under its own z-window): 1 = first conv layer, 2 = second, 4 = third
  // + the two 1x1x1 layers, 8 = velocity update / un-scale / wall BCs. Only the 3-D MFMA path is staged.
  const int stg = stages_of(c);
  if (c->stages && !m->mfma3d) return fail(c, TFL_EUNSUPPORTED, "model_finish: stage masks need the 3-D default topology");
  if (m->mfma3d && m->m16) {
    // split-operand fp16 MFMA (conv_mfma16.hip); the two activation buffers hold the "h2" form (32 B per voxel, as 8 fp32)
    // layers 1 + 2 in one launch where the whole array is computed (round 5); a z-slab rank runs them under separate windows
    const bool fused12 = (stg & 3) == 3 &&
                         tfl::conv3_m16_first2_fused(st, B, Z, Y, X, pDiv->data, w.div, flags->data, st_in, count, m->wfrag16[0],
                                                     m->layers[0].b, m->post16[0], m->wfrag16[1], m->layers[1].b, m->post16[1],
                                                     w.act[1], m->d_range_err);
    if ((stg & 1) && !fused12) {
      if (c->defer_stats)      // (tfl_model_forward only: whole array, the model's own stats buffer)
        tfl::conv3_m16_first_fused(st, B, Z, Y, X, pDiv->data, w.div, flags->data, st_in, count, m->wfrag16[0], m->layers[0].b,
                                   m->post16[0], w.act[0], m->d_range_err, w.partials, m->stat_pairs_per_plane * Z, m->d_stats);
      else
        tfl::conv3_m16_first_fused(st, B, Z, Y, X, pDiv->data, w.div, flags->data, st_in, count, m->wfrag16[0], m->layers[0].b,
                                   m->post16[0], w.act[0], m->d_range_err);
    }
    if ((stg & 2) && !fused12)
      tfl::conv3_m16_mid(st, B, Z, Y, X, w.act[0], m->wfrag16[1], m->layers[1].b, m->post16[1], w.act[1], m->d_range_err);
    if (stg & 4)
      tfl::conv3_m16_tail(st, B, Z, Y, X, w.act[1], m->wfrag16[2], m->tail_pack, m->post16[2], w.pPred, m->d_range_err);
  } else if (m->mfma3d && m->valu3d) {
    // the first layer builds {pDiv/scale, div/scale, occupancy} while staging its LDS tile
    if (stg & 1)
      tfl::conv3_valu_first_fused(st, B, Z, Y, X, pDiv->data, w.div, flags->data, st_in, count, m->wino[0], m->layers[0].b, w.act[0]);
    if (stg & 2) tfl::conv3_valu_mid(st, B, Z, Y, X, w.act[0], m->wino[1], m->layers[1].b, w.act[1]);
    if (stg & 4)
      tfl::conv3_valu_tail(st, B, Z, Y, X, w.act[1], m->wino[2], m->tail_pack, w.pPred);
  } else if (m->mfma3d) {
    // the first MFMA layer builds {pDiv/scale, div/scale, occupancy} while staging its LDS tile
    if (stg & 1)
      tfl::conv3_mfma_first_fused(st, B, Z, Y, X, pDiv->data, w.div, flags->data, st_in, count, m->bfrag[0],
                                  m->layers[0].b, w.act[0]);
    if (stg & 2) tfl::conv3_mfma_mid(st, B, Z, Y, X, w.act[0], m->bfrag[1], m->layers[1].b, w.act[1]);
    if (stg & 4)
      tfl::conv3_mfma_tail(st, B, Z, Y, X, w.act[1], m->bfrag[2], m->layers[2].b, m->tail_w4, m->layers[3].b,
                           m->tail_w5, m->layers[4].b, w.pPred);
  } else if (m->mfma2d) {
    tfl::conv2_mfma_first_fused(st, B, Y, X, pDiv->data, w.div, flags->data, st_in, count, m->bfrag2[0],
                                m->layers[0].b, w.act[0]);
    tfl::conv2_mfma_mid(st, B, Y, X, w.act[0], m->bfrag2[1], m->layers[1].b, w.act[1]);
    tfl::conv2_mfma_mid(st, B, Y, X, w.act[1], m->bfrag2[2], m->layers[2].b, w.act[0]);
    tfl::conv2_mfma_tail(st, B, Y, X, w.act[0], m->bfrag2[3], m->layers[3].b, m->tail_w5, m->layers[4].b, w.pPred);
  } else {
    if (m->multires && ((m->is3d && Z % m->max_down) || Y % m->max_down || X % m->max_down))
      return fail(c, TFL_EINVAL, "model_finish: grid %dx%dx%d is not divisible by the model's pooling factor %d", Z, Y, X, m->max_down);
    if (m->custom)
      tfl::model_net_input_gen(st, m->is3d, B, Z, Y, X, m->opts.in_pDiv, m->opts.in_UDiv, m->opts.in_div, pDiv->data,
                               UOut->data, w.div, flags->data, st_in, count, w.x3);
    else
      tfl::model_net_input(st, m->is3d, B, Z, Y, X, pDiv->data, w.div, flags->data, st_in, count, w.x3);
    const int act = 1 + m->opts.nonlin;            // conv_direct: 1 ReLU, 2 ReLU6, 3 sigmoid
    const size_t nl = m->layers.size();
    if (m->opts.pressure_skip && (m->layers[nl - 1].pool > 1 || m->layers[nl - 1].up > 1 || m->layers[nl - 2].pool > 1 ||
                                  m->layers[nl - 2].up > 1 || (nl > 2 && m->multires)))
      return fail(c, TFL_EUNSUPPORTED, "model_finish: addPressureSkip with pooling / upsampling layers");
    const float* in = w.x3;
    int Zc = Z, Yc = Y, Xc = X;     // resolution of `in`
    for (size_t l = 0; l < m->layers.size(); l++) {
      const tfl_layer& L = m->layers[l];
      const bool last = l + 1 == m->layers.size();
      const bool joins_skip = m->opts.pressure_skip && l + 2 == m->layers.size();
      float* out = last ? w.pPred : (in == w.act[0] ? w.act[1] : w.act[0]);
      const int taps = m->is3d ? L.k * L.k * L.k : L.k * L.k;
      const int S = m->is3d ? L.up * L.up * L.up : L.up * L.up;
      for (int sub = 0; sub < S; sub++)      // ConvolutionUpsample: one strided-store convolution per sub-position
        if (!tfl::conv_direct(st, m->is3d, B, Zc, Yc, Xc, L.cin, L.cout, L.k, last ? 0 : act, in,
                              L.w + (size_t)sub * taps * L.cin * L.cout, L.b + (size_t)sub * L.cout, out, L.up, sub,
                              joins_skip ? L.cout + 1 : 0))
          return fail(c, TFL_EUNSUPPORTED, "model_finish: no kernel for %d output channels", L.cout);
      // addPressureSkip: pDiv/scale becomes the last input channel of the last layer (model.lua:356-360)
      if (joins_skip) tfl::model_skip_channel(st, B, (long long)Z * Y * X, pDiv->data, st_in, count, out, L.cout + 1, L.cout);
      if (m->is3d) Zc *= L.up;
      Yc *= L.up; Xc *= L.up;
      in = out;
      if (L.pool > 1) {
        float* pooled = in == w.act[0] ? w.act[1] : w.act[0];
        tfl::avg_pool2(st, m->is3d, B * L.cout, Zc, Yc, Xc, in, pooled);
        if (m->is3d) Zc /= 2;
        Yc /= 2; Xc /= 2;
        in = pooled;
      }
    }
  }
  if (stg & 8) {
    const bool folded = tfl::model_project(st, m->is3d, B, Z, Y, X, w.pPred, flags->data, st_in, count, UOut->data, pOut->data,
                                           UBC ? UBC->data : nullptr, UBC ? UBCInvMask->data : nullptr, doClamp, lo, hi,
                                           m->d_range_host ? m->d_range_err : nullptr, m->d_range_host,
                                           c->reach_sink ? c->d_reach : nullptr, c->reach_sink ? c->d_reach_host : nullptr,
                                           c->reach_sink ? c->d_reach : nullptr, wall_code_of(c, m, flags),
                                           c->reach_sink ? c->d_reach_tick : nullptr);
    if (c->reach_sink) { c->reach_folded = folded; c->reach_issued++; }
  }
  return check_launch(c, "model_finish");
}

int tfl_model_forward(tfl_ctx* c, tfl_model* m, const tfl_tensor* pDiv, const tfl_tensor* UDiv,
                      const tfl_tensor* flags, const tfl_tensor* pOut, const tfl_tensor* UOut, float* workspace,
                      int64_t workspace_floats, const tfl_tensor* UBC, const tfl_tensor* UBCInvMask, int doClamp,
                      float lo, float hi) {
  TRY(check_flags(c, "model_forward", flags));
  if (c->stages || c->zwin.a1 > c->zwin.a0 || c->zwin.b1 > c->zwin.b0)
    return fail(c, TFL_EINVAL, "model_forward: clear the z-window / stage mask first (use tfl_model_begin / _finish)");
  // round 6: with the default 3-D conv path the first layer's blocks sum k_bcs_div_stats' partial pairs themselves (the same
  // order, the same bits) -- no k_reduce_stats launch between the two kernels (4.3 us of pure latency at 128^3)
  struct Defer { tfl_ctx* c; ~Defer() { c->defer_stats = false; } } defer{c};
  c->defer_stats = m && m->mfma3d && m->m16 && !m->custom && tfl::conv3_m16_first_sums_partials() &&
                   !tfl::conv3_m16_fuse12_requested() && !tfl::model_stats_fold_requested();
  TRY(tfl_model_begin(c, m, UDiv, flags, UOut, workspace, workspace_floats, 0, flags->Z, nullptr));
  const double count = (double)flags->Z * flags->Y * flags->X * (m->is3d ? 3 : 2);
  return tfl_model_finish(c, m, pDiv, flags, pOut, UOut, workspace, workspace_floats, nullptr, count, UBC, UBCInvMask,
                          doClamp, lo, hi);
}

double tfl_getDx(tfl_ctx* c, const tfl_tensor* flags) {
  if (!flags) return 0.0;
  if (c && c->dx_dim > 0) return 1.0 / (double)c->dx_dim;
  if (c && c->dx_override > 0.0f) return (double)c->dx_override;
  int m = flags->X > flags->Y ? flags->X : flags->Y;
  if (flags->Z > m) m = flags->Z;
  return 1.0 / (double)m;
}

int tfl_copy(tfl_ctx* c, const tfl_tensor* dst, const tfl_tensor* src) {
  if (!c) return TFL_EINVAL;
  if (!dst || !src || !dst->data || !src->data) return fail(c, TFL_EINVAL, "copy: null tensor");
  const long long n = (long long)dst->B * dst->C * dst->Z * dst->Y * dst->X;
  if (n != (long long)src->B * src->C * src->Z * src->Y * src->X) return fail(c, TFL_EINVAL, "copy: size mismatch");
  HIP_TRY(c, hipMemcpyAsync(dst->data, src->data, sizeof(float) * (size_t)n, hipMemcpyDeviceToDevice, c->stream));
  return TFL_OK;
}

int tfl_stream_copy(tfl_ctx* c, float* dst, const float* src, int64_t n) {
  if (!c) return TFL_EINVAL;
  if (!dst || !src || n < 0 || (n & 3) || (((uintptr_t)dst | (uintptr_t)src) & 15))
    return fail(c, TFL_EINVAL, "stream_copy: needs 16-byte aligned pointers and a multiple of 4 floats");
  tfl::stream_copy(c->stream, n / 4, src, dst);
  return check_launch(c, "stream_copy");
}

int tfl_applyBCs(tfl_ctx* c, const tfl_tensor* x, const tfl_tensor* bc, const tfl_tensor* invMask, int doClamp,
                 float lo, float hi) {
  if (!c) return TFL_EINVAL;
  if (!x || !x->data) return fail(c, TFL_EINVAL, "applyBCs: x is null");
  if ((bc == nullptr) != (invMask == nullptr)) return fail(c, TFL_EINVAL, "applyBCs: bc and invMask go together");
  const long long n = (long long)x->B * x->C * x->Z * x->Y * x->X;
  if (bc) {
    const long long nb = (long long)bc->B * bc->C * bc->Z * bc->Y * bc->X;
    const long long nm = (long long)invMask->B * invMask->C * invMask->Z * invMask->Y * invMask->X;
    if (nb != n || nm != n || !bc->data || !invMask->data) return fail(c, TFL_EINVAL, "applyBCs: size mismatch");
  }
  tfl::apply_bcs(c->stream, n, x->data, bc ? bc->data : nullptr, bc ? invMask->data : nullptr, doClamp, lo, hi);
  return check_launch(c, "applyBCs");
}

int tfl_applyBCsIndexed(tfl_ctx* c, const tfl_tensor* x, const tfl_tensor* bc, const tfl_tensor* invMask,
                        const int32_t* idx, int64_t n) {
  if (!c) return TFL_EINVAL;
  if (!x || !x->data || !bc || !bc->data || !invMask || !invMask->data) return fail(c, TFL_EINVAL, "applyBCsIndexed: null tensor");
  const long long nx = (long long)x->B * x->C * x->Z * x->Y * x->X;
  const long long nb = (long long)bc->B * bc->C * bc->Z * bc->Y * bc->X;
  const long long nm = (long long)invMask->B * invMask->C * invMask->Z * invMask->Y * invMask->X;
  if (nb != nx || nm != nx) return fail(c, TFL_EINVAL, "applyBCsIndexed: size mismatch");
  if (nx >= (1ll << 31)) return fail(c, TFL_EINVAL, "applyBCsIndexed: tensor too large for 32-bit indices");
  if (n < 0 || (n > 0 && !idx)) return fail(c, TFL_EINVAL, "applyBCsIndexed: bad index list");
  tfl::apply_bcs_indexed(c->stream, n, idx, x->data, bc->data, invMask->data);
  return check_launch(c, "applyBCsIndexed");
}

int tfl_applyBCsIndexedMulti(tfl_ctx* c, int count, const tfl_tensor* const* x, const tfl_tensor* const* bc,
                             const tfl_tensor* const* invMask, const int32_t* const* idx, const int64_t* n) {
  if (!c) return TFL_EINVAL;
  if (count < 1 || count > 8 || !x || !bc || !invMask || !idx || !n)
    return fail(c, TFL_EINVAL, "applyBCsIndexedMulti: 1..8 (x, bc, invMask, idx) tuples are required");
  long long nn[8];
  const int* ii[8];
  float* xx[8];
  const float *bb[8], *mm[8];
  for (int i = 0; i < count; i++) {
    const tfl_tensor *xi = x[i], *bi = bc[i], *mi = invMask[i];
    if (!xi || !xi->data || !bi || !bi->data || !mi || !mi->data) return fail(c, TFL_EINVAL, "applyBCsIndexedMulti: null tensor in tuple %d", i);
    const long long nx = (long long)xi->B * xi->C * xi->Z * xi->Y * xi->X;
    const long long nb = (long long)bi->B * bi->C * bi->Z * bi->Y * bi->X;
    const long long nm = (long long)mi->B * mi->C * mi->Z * mi->Y * mi->X;
    if (nb != nx || nm != nx) return fail(c, TFL_EINVAL, "applyBCsIndexedMulti: size mismatch in tuple %d", i);
    if (nx >= (1ll << 31)) return fail(c, TFL_EINVAL, "applyBCsIndexedMulti: tensor %d too large for 32-bit indices", i);
    if (n[i] < 0 || (n[i] > 0 && !idx[i])) return fail(c, TFL_EINVAL, "applyBCsIndexedMulti: bad index list %d", i);
    nn[i] = n[i]; ii[i] = idx[i]; xx[i] = xi->data; bb[i] = bi->data; mm[i] = mi->data;
  }
  tfl::apply_bcs_indexed_multi(c->stream, count, nn, ii, xx, bb, mm);
  return check_launch(c, "applyBCsIndexedMulti");
}

int tfl_packPlanes(tfl_ctx* c, int n, const tfl_tensor* const* fields, int zlo, int zhi, float* buf, int unpack) {
  if (!c) return TFL_EINVAL;
  if (n < 1 || n > 8 || !fields || !buf) return fail(c, TFL_EINVAL, "packPlanes: 1..8 fields and a buffer are required");
  const tfl_tensor* f0 = fields[0];
  if (!f0 || zlo < 0 || zhi > f0->Z || zlo >= zhi) return fail(c, TFL_EINVAL, "packPlanes: bad plane range [%d, %d)", zlo, zhi);
  float* ptrs[8];
  int rows[8], los[8], nps[8];
  for (int i = 0; i < n; i++) {
    const tfl_tensor* f = fields[i];
    if (!f || !f->data || f->Z != f0->Z || f->Y != f0->Y || f->X != f0->X)
      return fail(c, TFL_EINVAL, "packPlanes: field %d does not match the grid of field 0", i);
    ptrs[i] = f->data;
    rows[i] = f->B * f->C;
    los[i] = zlo; nps[i] = zhi - zlo;
  }
  const long long yx = (long long)f0->Y * f0->X;
  tfl::pack_planes(c->stream, n, ptrs, rows, los, nps, yx * f0->Z, yx, buf, unpack);
  return check_launch(c, "packPlanes");
}

int tfl_setWallBcsBackward(tfl_ctx* c, const tfl_tensor* flags, const tfl_tensor* gradOutput, int is3D, const tfl_tensor* gradU) {
  TRY(check_flags(c, "setWallBcsBackward", flags));
  TRY(check_vel(c, "setWallBcsBackward", "gradOutput", gradOutput, flags, is3D));
  TRY(check_vel(c, "setWallBcsBackward", "gradU", gradU, flags, is3D));
  if (gradU->data != gradOutput->data)
    HIP_TRY(c, hipMemcpyAsync(gradU->data, gradOutput->data, sizeof(float) * (size_t)flags->B * gradU->C * flags->Z * flags->Y * flags->X,
                              hipMemcpyDeviceToDevice, c->stream));
  tfl::set_wall_bcs(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, gradU->data, flags->data);
  return check_launch(c, "setWallBcsBackward");
}

int tfl_velocityDivergenceBackward(tfl_ctx* c, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* gradOutput,
                                   int is3D, const tfl_tensor* gradU) {
  TRY(check_flags(c, "velocityDivergenceBackward", flags));
  TRY(check_vel(c, "velocityDivergenceBackward", "U", U, flags, is3D));
  TRY(check_vel(c, "velocityDivergenceBackward", "gradU", gradU, flags, is3D));
  TRY(check_scalar(c, "velocityDivergenceBackward", "gradOutput", gradOutput, flags));
  tfl::velocity_divergence_bwd(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, flags->data,
                               gradOutput->data, gradU->data);
  return check_launch(c, "velocityDivergenceBackward");
}

int tfl_velocityUpdateBackward(tfl_ctx* c, const tfl_tensor* U, const tfl_tensor* flags, const tfl_tensor* p,
                               const tfl_tensor* gradOutput, int is3D, const tfl_tensor* gradP) {
  TRY(check_flags(c, "velocityUpdateBackward", flags));
  TRY(check_vel(c, "velocityUpdateBackward", "U", U, flags, is3D));
  TRY(check_vel(c, "velocityUpdateBackward", "gradOutput", gradOutput, flags, is3D));
  TRY(check_scalar(c, "velocityUpdateBackward", "p", p, flags));
  TRY(check_scalar(c, "velocityUpdateBackward", "gradP", gradP, flags));
  tfl::velocity_update_bwd(c->stream, is3D != 0, flags->B, flags->Z, flags->Y, flags->X, flags->data, gradOutput->data,
                           gradP->data);
  return check_launch(c, "velocityUpdateBackward");
}

int tfl_volumetricUpSamplingNearestForward(tfl_ctx* c, int ratio, const tfl_tensor* input, const tfl_tensor* output) {
  if (!c) return TFL_EINVAL;
  if (!input || !output || !input->data || !output->data) return fail(c, TFL_EINVAL, "ERROR: input and output must be dim 5");
  if (ratio < 1 || output->B != input->B || output->C != input->C || output->Z != input->Z * ratio ||
      output->Y != input->Y * ratio || output->X != input->X * ratio)
    return fail(c, TFL_EINVAL, "ERROR: input : output size mismatch.");
  tfl::upsample_nearest_fwd(c->stream, ratio, (long long)input->B * input->C, output->Z, output->Y, output->X,
                            input->data, output->data);
  return check_launch(c, "volumetricUpSamplingNearestForward");
}

int tfl_volumetricUpSamplingNearestBackward(tfl_ctx* c, int ratio, const tfl_tensor* input, const tfl_tensor* gradOutput,
                                            const tfl_tensor* gradInput) {
  if (!c) return TFL_EINVAL;
  if (!input || !gradOutput || !gradInput || !gradOutput->data || !gradInput->data)
    return fail(c, TFL_EINVAL, "ERROR: input, gradOutput and gradInput must be dim 5");
  if (ratio < 1 || gradOutput->B != input->B || gradOutput->C != input->C || gradOutput->Z != input->Z * ratio ||
      gradOutput->Y != input->Y * ratio || gradOutput->X != input->X * ratio)
    return fail(c, TFL_EINVAL, "ERROR: input : gradOutput size mismatch.");
  if (gradInput->B != input->B || gradInput->C != input->C || gradInput->Z != input->Z || gradInput->Y != input->Y ||
      gradInput->X != input->X)
    return fail(c, TFL_EINVAL, "ERROR: input : gradInput size mismatch.");
  tfl::upsample_nearest_bwd(c->stream, ratio, (long long)input->B * input->C, input->Z, input->Y, input->X,
                            gradOutput->data, gradInput->data);
  return check_launch(c, "volumetricUpSamplingNearestBackward");
}

}  // extern "C"
