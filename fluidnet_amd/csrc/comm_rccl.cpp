// comm_rccl.cpp -- the z-slab step's transport inside the library: RCCL send/recv over xGMI (BASELINE north_star:
// "one-cell halo exchange via RCCL send/recv over xGMI"; the reference itself is single-GPU and has no counterpart).
//
// tfl_simulate_step_slab talks to its neighbours through three callbacks (tfl_comm, include/tfluids_hip.h). Until
// round 3 only fluidnet_amd/dist.py implemented them (torch.distributed); a LuaJIT or C host had no multi-GPU path.
// This file implements them natively:
//   exchange_start  ncclGroupStart; ncclSend / ncclRecv with rank-1 and rank+1 (each pair rides its own xGMI link: a
//                   slab only ever talks to its two neighbours -- no ring, no all-to-all); ncclGroupEnd
//   exchange_wait   the step's stream waits for the event recorded behind that group
//   allreduce_sum   ncclAllReduce(sum, double) for the ConvNet's global std normaliser (2*B doubles)
// on a communication stream of its own, tied to the context's stream by events only: nothing blocks the host, the
// packing kernels queued before exchange_start are waited for on the device, and whatever the step enqueues next
// (interior strips) overlaps the transfer.
//
// RCCL is bound at run time (dlopen), not at link time: libtfluids_hip.so keeps depending on the HIP runtime only,
// a process that already carries an RCCL (PyTorch ships its own librccl.so) shares that copy instead of loading a
// second one, and hosts that never cut a grid never load it. Resolution order: $TFL_RCCL_LIBRARY, an RCCL already
// loaded in the process, then librccl.so.1 / librccl.so from the loader path.
#include <dlfcn.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

#include "../../include/tfluids_hip.h"
#include "tfl_ctx.hpp"

namespace {

// the slice of rccl.h this file needs (ABI-stable across NCCL 2.x / RCCL): /opt/rocm/include/rccl/rccl.h:40-43,
// 187-260, 448-467, 611-722, 923-933
struct NcclUniqueId { char internal[TFL_RCCL_UNIQUE_ID_BYTES]; };
typedef void* NcclComm;
enum { kNcclSuccess = 0, kNcclSum = 0, kNcclFloat = 7, kNcclDouble = 8 };

struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*Send)(const void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Recv)(void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  std::string origin, error;
};

RcclApi g_api;
std::once_flag g_once;

void load_api() {
  RcclApi& a = g_api;
  const char* env = getenv("TFL_RCCL_LIBRARY");
  const char* names[] = {"librccl.so.1", "librccl.so"};
  if (env && env[0]) {
    a.handle = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    a.origin = env;
    if (!a.handle) { a.error = std::string("dlopen(") + env + "): " + dlerror(); return; }
  }
  for (int pass = 0; pass < 2 && !a.handle; pass++)         // pass 0: a copy already in the process (RTLD_NOLOAD)
    for (const char* n : names) {
      a.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (a.handle) { a.origin = std::string(n) + (pass == 0 ? " (already loaded)" : ""); break; }
    }
  if (!a.handle) { a.error = "no RCCL found: set TFL_RCCL_LIBRARY or put librccl.so on the loader path"; return; }
  bool ok = true;
  auto sym = [&](const char* n) { void* p = dlsym(a.handle, n); if (!p) { ok = false; a.error = std::string("RCCL symbol missing: ") + n; } return p; };
  a.GetUniqueId = (int (*)(NcclUniqueId*))sym("ncclGetUniqueId");
  a.CommInitRank = (int (*)(NcclComm*, int, NcclUniqueId, int))sym("ncclCommInitRank");
  a.CommDestroy = (int (*)(NcclComm))sym("ncclCommDestroy");
  a.Send = (int (*)(const void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclSend");
  a.Recv = (int (*)(void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclRecv");
  a.AllReduce = (int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclAllReduce");
  a.GroupStart = (int (*)())sym("ncclGroupStart");
  a.GroupEnd = (int (*)())sym("ncclGroupEnd");
  a.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  if (!ok) { dlclose(a.handle); a.handle = nullptr; }
}

const RcclApi* api(tfl_ctx* c) {
  std::call_once(g_once, load_api);
  if (!g_api.handle) { if (c) c->err = "rccl transport: " + g_api.error; return nullptr; }
  return &g_api;
}

constexpr int kTags = 8;   // the slab step uses tags 0..3

}  // namespace

struct tfl_rccl_comm {
  tfl_ctx* ctx = nullptr;
  const RcclApi* api = nullptr;
  NcclComm comm = nullptr;
  bool owns_comm = false;
  int rank = 0, world = 1;
  hipStream_t stream = nullptr;          // communication stream
  hipEvent_t ready = nullptr;            // "the step's stream has reached this point" (re-recorded per call)
  hipEvent_t done[kTags] = {};           // behind the transfers of exchange_start(tag)
  hipEvent_t reduced = nullptr;
  bool inline_stream = false;            // tfl_rccl_comm_set_inline: every nccl* call on the context's stream, no events
  tfl_comm callbacks{};
  std::string last_error;
};

namespace {

int fail(tfl_rccl_comm* q, const char* what, int rc) {
  q->last_error = std::string(what) + ": " + (q->api && q->api->GetErrorString ? q->api->GetErrorString(rc) : "?");
  if (q->ctx) q->ctx->err = "rccl transport: " + q->last_error;
  return 1;
}
int hip_fail(tfl_rccl_comm* q, const char* what, hipError_t e) {
  q->last_error = std::string(what) + ": " + hipGetErrorString(e);
  if (q->ctx) q->ctx->err = "rccl transport: " + q->last_error;
  return 1;
}
#define RCHK(call, what) do { int rc_ = (call); if (rc_ != kNcclSuccess) return fail(q, what, rc_); } while (0)
// inside ncclGroupStart ... ncclGroupEnd: a failing call must still close the group on this thread before returning
#define GCHK(call, what) do { int rc_ = (call); if (rc_ != kNcclSuccess) { (void)a->GroupEnd(); return fail(q, what, rc_); } } while (0)
#define HCHK(call, what) do { hipError_t e_ = (call); if (e_ != hipSuccess) return hip_fail(q, what, e_); } while (0)

// the stream the nccl* calls of one callback go to: the communication stream, made to wait for what the step's stream holds so
// far -- or, inline (tfl_rccl_comm_set_inline), the step's stream itself: an event hop between two streams costs 12-15 us of
// device-side latency on this stack (tools/ubench/host_costs.hip), eight of them per step, and a thin slab has nothing for
// the transfer to overlap with
int comm_stream(tfl_rccl_comm* q, hipStream_t* st) {
  if (q->inline_stream) { *st = q->ctx->stream; return 0; }
  HCHK(hipEventRecord(q->ready, q->ctx->stream), "hipEventRecord");
  HCHK(hipStreamWaitEvent(q->stream, q->ready, 0), "hipStreamWaitEvent");
  *st = q->stream;
  return 0;
}

int cb_exchange_start(void* user, int tag, const float* send_lo, int64_t n_send_lo, float* recv_lo, int64_t n_recv_lo,
                      const float* send_hi, int64_t n_send_hi, float* recv_hi, int64_t n_recv_hi) {
  tfl_rccl_comm* q = (tfl_rccl_comm*)user;
  if (tag < 0 || tag >= kTags) { q->last_error = "tag out of range"; return 1; }
  const RcclApi* a = q->api;
  // the transfer starts behind the packing kernels already queued on the step's stream
  hipStream_t st;
  if (comm_stream(q, &st)) return 1;
  RCHK(a->GroupStart(), "ncclGroupStart");
  if (n_send_lo > 0 && q->rank > 0) GCHK(a->Send(send_lo, (size_t)n_send_lo, kNcclFloat, q->rank - 1, q->comm, st), "ncclSend(lower)");
  if (n_recv_lo > 0 && q->rank > 0) GCHK(a->Recv(recv_lo, (size_t)n_recv_lo, kNcclFloat, q->rank - 1, q->comm, st), "ncclRecv(lower)");
  if (n_send_hi > 0 && q->rank + 1 < q->world) GCHK(a->Send(send_hi, (size_t)n_send_hi, kNcclFloat, q->rank + 1, q->comm, st), "ncclSend(upper)");
  if (n_recv_hi > 0 && q->rank + 1 < q->world) GCHK(a->Recv(recv_hi, (size_t)n_recv_hi, kNcclFloat, q->rank + 1, q->comm, st), "ncclRecv(upper)");
  RCHK(a->GroupEnd(), "ncclGroupEnd");
  if (!q->inline_stream) HCHK(hipEventRecord(q->done[tag], q->stream), "hipEventRecord");
  return 0;
}

// the same exchange on the planes where they lie: several sends / receives per neighbour in ONE group (RCCL matches the
// pieces of a pair of ranks in issue order)
int cb_exchange_start_v(void* user, int tag, int n_lo, const tfl_comm_chunk* send_lo, const tfl_comm_chunk* recv_lo, int n_hi,
                        const tfl_comm_chunk* send_hi, const tfl_comm_chunk* recv_hi) {
  tfl_rccl_comm* q = (tfl_rccl_comm*)user;
  if (tag < 0 || tag >= kTags) { q->last_error = "tag out of range"; return 1; }
  const RcclApi* a = q->api;
  hipStream_t st;
  if (comm_stream(q, &st)) return 1;
  RCHK(a->GroupStart(), "ncclGroupStart");
  if (q->rank > 0)
    for (int i = 0; i < n_lo; i++) {
      if (send_lo[i].n > 0) GCHK(a->Send(send_lo[i].ptr, (size_t)send_lo[i].n, kNcclFloat, q->rank - 1, q->comm, st), "ncclSend(lower)");
      if (recv_lo[i].n > 0) GCHK(a->Recv(recv_lo[i].ptr, (size_t)recv_lo[i].n, kNcclFloat, q->rank - 1, q->comm, st), "ncclRecv(lower)");
    }
  if (q->rank + 1 < q->world)
    for (int i = 0; i < n_hi; i++) {
      if (send_hi[i].n > 0) GCHK(a->Send(send_hi[i].ptr, (size_t)send_hi[i].n, kNcclFloat, q->rank + 1, q->comm, st), "ncclSend(upper)");
      if (recv_hi[i].n > 0) GCHK(a->Recv(recv_hi[i].ptr, (size_t)recv_hi[i].n, kNcclFloat, q->rank + 1, q->comm, st), "ncclRecv(upper)");
    }
  RCHK(a->GroupEnd(), "ncclGroupEnd");
  if (!q->inline_stream) HCHK(hipEventRecord(q->done[tag], q->stream), "hipEventRecord");
  return 0;
}

int cb_exchange_wait(void* user, int tag) {
  tfl_rccl_comm* q = (tfl_rccl_comm*)user;
  if (tag < 0 || tag >= kTags) { q->last_error = "tag out of range"; return 1; }
  if (!q->inline_stream) HCHK(hipStreamWaitEvent(q->ctx->stream, q->done[tag], 0), "hipStreamWaitEvent");
  return 0;
}

int cb_allreduce_sum(void* user, double* dev, int64_t n) {
  tfl_rccl_comm* q = (tfl_rccl_comm*)user;
  // one stream per communicator: the reduction queues behind the point-to-point groups issued so far, and every rank
  // issues the same sequence (the step is deterministic), which is what RCCL requires of a communicator's callers
  hipStream_t st;
  if (comm_stream(q, &st)) return 1;
  RCHK(q->api->AllReduce(dev, dev, (size_t)n, kNcclDouble, kNcclSum, q->comm, st), "ncclAllReduce");
  if (!q->inline_stream) {
    HCHK(hipEventRecord(q->reduced, q->stream), "hipEventRecord");
    HCHK(hipStreamWaitEvent(q->ctx->stream, q->reduced, 0), "hipStreamWaitEvent");
  }
  return 0;
}

tfl_rccl_comm* make(tfl_ctx* c, const RcclApi* a, NcclComm comm, bool owns, int rank, int world) {
  tfl_rccl_comm* q = new tfl_rccl_comm();
  q->ctx = c; q->api = a; q->comm = comm; q->owns_comm = owns; q->rank = rank; q->world = world;
  bool ok = hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking) == hipSuccess &&
            hipEventCreateWithFlags(&q->ready, hipEventDisableTiming) == hipSuccess &&
            hipEventCreateWithFlags(&q->reduced, hipEventDisableTiming) == hipSuccess;
  for (int t = 0; ok && t < kTags; t++) ok = hipEventCreateWithFlags(&q->done[t], hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    c->err = "rccl transport: could not create the communication stream / events";
    q->owns_comm = false;      // the caller still owns `comm` on this path (tfl_rccl_comm_create destroys it once)
    tfl_rccl_comm_destroy(c, q);
    return nullptr;
  }
  q->callbacks.size = (int32_t)sizeof(tfl_comm);
  q->callbacks.user = q;
  q->callbacks.exchange_start = cb_exchange_start;
  q->callbacks.exchange_wait = cb_exchange_wait;
  q->callbacks.allreduce_sum = cb_allreduce_sum;
  // TFL_RCCL_PACKED=1: staged messages (one send + one receive per neighbour, pack / unpack kernels) for comparison
  q->callbacks.exchange_start_v = getenv("TFL_RCCL_PACKED") ? nullptr : cb_exchange_start_v;
  q->callbacks.capturable = 1;     // every callback above is event record / wait + nccl* calls on streams: a step records into a HIP graph
  return q;
}

}  // namespace

extern "C" {

int tfl_rccl_available(tfl_ctx* c) { return api(c) != nullptr; }

int tfl_rccl_get_unique_id(tfl_ctx* c, void* id) {
  if (!c || !id) return TFL_EINVAL;
  const RcclApi* a = api(c);
  if (!a) return TFL_EUNSUPPORTED;
  NcclUniqueId u;
  const int rc = a->GetUniqueId(&u);
  if (rc != kNcclSuccess) { c->err = std::string("rccl transport: ncclGetUniqueId: ") + a->GetErrorString(rc); return TFL_EINVAL; }
  memcpy(id, u.internal, TFL_RCCL_UNIQUE_ID_BYTES);
  return TFL_OK;
}

tfl_rccl_comm* tfl_rccl_comm_create(tfl_ctx* c, const void* id, int rank, int world) {
  if (!c || !id || world < 1 || rank < 0 || rank >= world) { if (c) c->err = "rccl transport: bad rank / world"; return nullptr; }
  const RcclApi* a = api(c);
  if (!a) return nullptr;
  if (hipSetDevice(c->device) != hipSuccess) { c->err = "rccl transport: hipSetDevice failed"; return nullptr; }
  NcclUniqueId u;
  memcpy(u.internal, id, TFL_RCCL_UNIQUE_ID_BYTES);
  NcclComm comm = nullptr;
  const int rc = a->CommInitRank(&comm, world, u, rank);
  if (rc != kNcclSuccess) { c->err = std::string("rccl transport: ncclCommInitRank: ") + a->GetErrorString(rc); return nullptr; }
  tfl_rccl_comm* q = make(c, a, comm, true, rank, world);
  if (!q) a->CommDestroy(comm);
  return q;
}

tfl_rccl_comm* tfl_rccl_comm_wrap(tfl_ctx* c, void* nccl_comm, int rank, int world) {
  if (!c || !nccl_comm || world < 1 || rank < 0 || rank >= world) { if (c) c->err = "rccl transport: bad communicator / rank / world"; return nullptr; }
  const RcclApi* a = api(c);
  if (!a) return nullptr;
  return make(c, a, (NcclComm)nccl_comm, false, rank, world);
}

const tfl_comm* tfl_rccl_comm_callbacks(tfl_rccl_comm* q) { return q ? &q->callbacks : nullptr; }

int tfl_rccl_comm_set_inline(tfl_rccl_comm* q, int on) {
  if (!q) return TFL_EINVAL;
  if (q->stream) (void)hipStreamSynchronize(q->stream);      // nothing of the old mode is left in flight
  q->inline_stream = on != 0;
  return TFL_OK;
}

const char* tfl_rccl_comm_origin(tfl_ctx* c) { return api(c) ? g_api.origin.c_str() : ""; }

void tfl_rccl_comm_destroy(tfl_ctx* c, tfl_rccl_comm* q) {
  (void)c;
  if (!q) return;
  if (q->stream) (void)hipStreamSynchronize(q->stream);
  if (q->owns_comm && q->comm && q->api) q->api->CommDestroy(q->comm);
  for (int t = 0; t < kTags; t++) if (q->done[t]) (void)hipEventDestroy(q->done[t]);
  if (q->ready) (void)hipEventDestroy(q->ready);
  if (q->reduced) (void)hipEventDestroy(q->reduced);
  if (q->stream) (void)hipStreamDestroy(q->stream);
  delete q;
}

}  // extern "C"
