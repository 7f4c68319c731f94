// stub_rccl.cpp -- TEST INFRASTRUCTURE: an in-process stand-in for librccl.so, so that the library's native transport
// (fluidnet_amd/csrc/comm_rccl.cpp) can be exercised end to end on ONE GPU. RCCL itself refuses two ranks on one
// device; here the "ranks" are threads of one process (fluidnet_amd.dist.run_virtual_ranks) and this file implements
// the nine entry points the transport binds, with RCCL's stream semantics: a send's data is read at the point of the
// sender's stream where ncclSend was issued, a receive is complete at the point of the receiver's stream where ncclRecv
// was issued, nothing blocks on the device from the host. Point-to-point ops are matched FIFO per (source, destination)
// at ncclGroupEnd; the all-reduce goes through the host (it moves 2*B doubles).
// Built by tests with: hipcc -shared -fPIC -o libstub_rccl.so stub_rccl.cpp ; selected with TFL_RCCL_LIBRARY.
//
// STUB_RCCL_NULL=1 (round 6, tools/slab_host_cost.py): ONE rank of an N-rank world with neighbours that are not there -- a send
// is dropped, a receive is a zero-fill of its buffer on the stream, the all-reduce adds what the other ranks would have
// contributed (1e5 to every sum of squares). Pure stream operations, so the rank-step -- the library's native transport
// included -- can be timed and recorded into a HIP graph on one GPU; the results are wrong by construction.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

struct Post { const void* buf; size_t bytes; hipEvent_t ready, copied; bool done; };
struct World {
  int nranks = 0, joined = 0, left = 0;
  std::mutex m;
  std::condition_variable cv;
  std::map<std::pair<int, int>, std::deque<Post*>> box;    // (src, dst) -> posted sends
  std::vector<std::vector<double>> red;                    // all-reduce contributions
  int red_arrived = 0, red_read = 0; long red_gen = 0;
};
struct Comm { World* w; int rank; };
struct Op { bool send; const void* sbuf; void* rbuf; size_t bytes; int peer; Comm* c; hipStream_t st; Post* post; };

std::mutex g_m;
std::map<std::string, World*> g_worlds;
int g_ids = 0;
thread_local int t_depth = 0;
thread_local std::vector<Op> t_ops;

size_t dtype_bytes(int dt) { return dt == 8 ? 8 : 4; }

bool null_mode() { static const bool on = getenv("STUB_RCCL_NULL") && atoi(getenv("STUB_RCCL_NULL")) != 0; return on; }
__global__ void k_null_allreduce(double* x, size_t n) { const size_t i = 2 * (size_t)threadIdx.x + 1; if (i < n) x[i] += 1.0e5; }
// RCCL runs all the sends / receives of a group as ONE kernel: the null mode's stand-in is one zero-fill kernel over the
// group's receive buffers (the sends are dropped)
struct NullGroup { float* p[32]; unsigned n4[32]; int count; };
__global__ void k_null_group(NullGroup g) {
  float4* q = (float4*)g.p[blockIdx.y];
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < g.n4[blockIdx.y]; i += gridDim.x * blockDim.x) q[i] = float4{0.f, 0.f, 0.f, 0.f};
}
thread_local NullGroup t_null = {{}, {}, 0};
thread_local hipStream_t t_null_st = nullptr;
int null_flush() {
  if (t_null.count == 0) return 0;
  hipLaunchKernelGGL(k_null_group, dim3(32, t_null.count), dim3(256), 0, t_null_st, t_null);
  t_null.count = 0;
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

int flush() {
  // 1. publish the sends
  for (Op& o : t_ops) if (o.send) {
    Post* p = new Post{o.sbuf, o.bytes, nullptr, nullptr, false};
    if (hipEventCreateWithFlags(&p->ready, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&p->copied, hipEventDisableTiming) != hipSuccess) return 1;
    if (hipEventRecord(p->ready, o.st) != hipSuccess) return 1;
    o.post = p;
    World* w = o.c->w;
    { std::lock_guard<std::mutex> l(w->m); w->box[{o.c->rank, o.peer}].push_back(p); }
    w->cv.notify_all();
  }
  // 2. receives: wait (host) for the matching post, then copy on the receiver's stream behind the sender's data
  for (Op& o : t_ops) if (!o.send) {
    World* w = o.c->w;
    Post* p = nullptr;
    {
      std::unique_lock<std::mutex> l(w->m);
      auto& q = w->box[{o.peer, o.c->rank}];
      w->cv.wait(l, [&] { return !q.empty(); });
      p = q.front(); q.pop_front();
    }
    if (p->bytes != o.bytes) return 2;
    if (hipStreamWaitEvent(o.st, p->ready, 0) != hipSuccess) return 1;
    if (hipMemcpyAsync(o.rbuf, p->buf, o.bytes, hipMemcpyDeviceToDevice, o.st) != hipSuccess) return 1;
    if (hipEventRecord(p->copied, o.st) != hipSuccess) return 1;
    { std::lock_guard<std::mutex> l(w->m); p->done = true; }
    w->cv.notify_all();
  }
  // 3. sends complete (stream-wise) once the receiver's copy has been queued and run
  for (Op& o : t_ops) if (o.send) {
    World* w = o.c->w;
    { std::unique_lock<std::mutex> l(w->m); w->cv.wait(l, [&] { return o.post->done; }); }
    if (hipStreamWaitEvent(o.st, o.post->copied, 0) != hipSuccess) return 1;
    // the events may still be referenced by queued stream waits: HIP defers the release until they have passed
    (void)hipEventDestroy(o.post->ready); (void)hipEventDestroy(o.post->copied);
    delete o.post;
  }
  t_ops.clear();
  return 0;
}

}  // namespace

extern "C" {

int ncclGetUniqueId(char* id) {
  std::lock_guard<std::mutex> l(g_m);
  memset(id, 0, 128);
  snprintf(id, 128, "stub-rccl-%d", ++g_ids);
  return 0;
}

struct IdByValue { char b[128]; };
int ncclCommInitRank(void** comm, int nranks, IdByValue id, int rank) {
  World* w;
  {
    std::lock_guard<std::mutex> l(g_m);
    std::string key(id.b, strnlen(id.b, 128));
    auto it = g_worlds.find(key);
    if (it == g_worlds.end()) { w = new World(); w->nranks = nranks; w->red.resize(nranks); g_worlds[key] = w; }
    else w = it->second;
  }
  if (!null_mode()) {
    std::unique_lock<std::mutex> l(w->m);
    w->joined++;
    w->cv.notify_all();
    w->cv.wait(l, [&] { return w->joined >= w->nranks; });
  }
  *comm = new Comm{w, rank};
  return 0;
}

int ncclCommDestroy(void* comm) { delete (Comm*)comm; return 0; }
int ncclGroupStart() { t_depth++; return 0; }
int ncclGroupEnd() { if (--t_depth == 0) return null_mode() ? null_flush() : flush(); return 0; }

int ncclSend(const void* buf, size_t count, int dt, int peer, void* comm, hipStream_t st) {
  if (null_mode()) return 0;
  t_ops.push_back(Op{true, buf, nullptr, count * dtype_bytes(dt), peer, (Comm*)comm, st, nullptr});
  return t_depth == 0 ? flush() : 0;
}
int ncclRecv(void* buf, size_t count, int dt, int peer, void* comm, hipStream_t st) {
  if (null_mode()) {
    const size_t bytes = count * dtype_bytes(dt);
    if (t_null.count >= 32 || (bytes & 15) || ((size_t)buf & 15) || bytes >= (1ull << 34)) return 3;
    t_null.p[t_null.count] = (float*)buf; t_null.n4[t_null.count] = (unsigned)(bytes / 16); t_null.count++; t_null_st = st;
    return t_depth == 0 ? null_flush() : 0;
  }
  t_ops.push_back(Op{false, nullptr, buf, count * dtype_bytes(dt), peer, (Comm*)comm, st, nullptr});
  return t_depth == 0 ? flush() : 0;
}

int ncclAllReduce(const void* sbuf, void* rbuf, size_t count, int dt, int op, void* comm, hipStream_t st) {
  if (dt != 8 || op != 0) return 3;
  if (null_mode()) {
    if (sbuf != rbuf || count > 2048) return 3;
    hipLaunchKernelGGL(k_null_allreduce, dim3(1), dim3(1024), 0, st, (double*)rbuf, count);
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }
  Comm* c = (Comm*)comm; World* w = c->w;
  std::vector<double> mine(count);
  if (hipMemcpyAsync(mine.data(), sbuf, count * 8, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
  std::vector<double> sum(count, 0.0);
  {
    std::unique_lock<std::mutex> l(w->m);
    const long gen = w->red_gen;
    w->red[c->rank] = mine;
    if (++w->red_arrived == w->nranks) w->cv.notify_all();
    w->cv.wait(l, [&] { return w->red_arrived == w->nranks || w->red_gen != gen; });
    for (int r = 0; r < w->nranks; r++) for (size_t i = 0; i < count; i++) sum[i] += w->red[r][i];   // rank order: every rank gets the same bits
    if (++w->red_read == w->nranks) { w->red_arrived = 0; w->red_read = 0; w->red_gen++; w->cv.notify_all(); }
    else w->cv.wait(l, [&] { return w->red_gen != gen; });
  }
  if (hipMemcpyAsync(rbuf, sum.data(), count * 8, hipMemcpyHostToDevice, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) return 1;
  return 0;
}

const char* ncclGetErrorString(int rc) { return rc == 0 ? "success" : (rc == 2 ? "stub: send/recv size mismatch" : "stub: error"); }

}  // extern "C"
