// What a rank-step of 12 small launches can cost at best on this stack (round 6): host time and GPU-side
// spacing of (1) back-to-back tiny kernels on one stream, (2) a kernel chain that hops between two streams through events,
// (3) the same chains replayed as HIP graphs, (4) tiny memsets / a 4-byte D2H copy into pinned memory.
// build: hipcc --offload-arch=gfx950 -O3 host_costs.hip -o host_costs
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_tiny(float* x) { if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0f; }
// a kernel that takes ~T us: 256 blocks x 256 threads spinning on the clock
__global__ void k_spin(float* x, long long clocks) {
  const long long t0 = clock64();
  while (clock64() - t0 < clocks) {}
  if (threadIdx.x == 0 && blockIdx.x == 0) x[0] += 1.0f;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
  float* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  float* h; CK(hipHostMalloc(&h, 64, hipHostMallocDefault));
  hipStream_t s0, s1; CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  hipEvent_t ev[64]; for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  const int N = 2000;
  for (int w = 0; w < 100; w++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s0, d);
  CK(hipStreamSynchronize(s0));
  for (long long spin : {0ll, 10000ll}) {   // 0: empty kernels; 10000 clocks ~ 5 us of work over 256 blocks
    auto launch = [&](hipStream_t st) { if (spin) hipLaunchKernelGGL(k_spin, dim3(256), dim3(256), 0, st, d, spin); else hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, st, d); };
    // 1. one stream
    double t0 = now();
    for (int i = 0; i < N; i++) launch(s0);
    double t1 = now(); CK(hipStreamSynchronize(s0)); double t2 = now();
    printf("[spin %lld] one stream: host %.2f us per launch, drained %.2f us per kernel\n", spin, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
    // 2. alternating streams through events
    t0 = now();
    for (int i = 0; i < N; i++) {
      hipStream_t a = (i & 1) ? s1 : s0, b = (i & 1) ? s0 : s1;
      launch(a);
      hipEventRecord(ev[i & 63], a); hipStreamWaitEvent(b, ev[i & 63], 0);
    }
    t1 = now(); CK(hipStreamSynchronize(s0)); CK(hipStreamSynchronize(s1)); t2 = now();
    printf("[spin %lld] two streams, an event hop behind every kernel: host %.2f us per (launch + record + wait), drained %.2f us per kernel\n", spin, (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
    // 3. graphs: a chain of 12 kernels on one stream; the same with 4 hops
    for (int hops : {0, 4}) {
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(s0, hipStreamCaptureModeRelaxed));
      for (int i = 0; i < 12; i++) {
        if (hops && i % 3 == 1) { hipEventRecord(ev[0], s0); hipStreamWaitEvent(s1, ev[0], 0); launch(s1); hipEventRecord(ev[1], s1); hipStreamWaitEvent(s0, ev[1], 0); }
        else launch(s0);
      }
      CK(hipStreamEndCapture(s0, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      size_t nn = 0; hipGraphGetNodes(g, nullptr, &nn);
      for (int w = 0; w < 5; w++) hipGraphLaunch(ge, s0);
      CK(hipStreamSynchronize(s0));
      const int M = 300;
      t0 = now();
      for (int i = 0; i < M; i++) hipGraphLaunch(ge, s0);
      t1 = now(); CK(hipStreamSynchronize(s0)); t2 = now();
      printf("[spin %lld] graph of 12 kernels (%zu nodes, %d side-stream hops): host %.2f us per graph launch, drained %.2f us per graph = %.2f per kernel\n",
             spin, nn, hops, (t1 - t0) / M * 1e6, (t2 - t0) / M * 1e6, (t2 - t0) / M / 12 * 1e6);
      hipGraphExecDestroy(ge); hipGraphDestroy(g);
    }
  }
  // 4. memset / small copies
  double t0 = now();
  for (int i = 0; i < N; i++) hipMemsetAsync(d + 1024, 0, 65536, s0);
  double t1 = now(); CK(hipStreamSynchronize(s0)); double t2 = now();
  printf("hipMemsetAsync 64 KB: host %.2f us, drained %.2f us each\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
  t0 = now();
  for (int i = 0; i < N; i++) hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, s0);
  t1 = now(); CK(hipStreamSynchronize(s0)); t2 = now();
  printf("hipMemcpyAsync 4 B D2H pinned: host %.2f us, drained %.2f us each\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
  t0 = now();
  for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, s0, d); hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, s0); }
  t1 = now(); CK(hipStreamSynchronize(s0)); t2 = now();
  printf("kernel + 4 B D2H alternating: host %.2f us, drained %.2f us per pair\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
  // 4b. hipExtAnyOrderLaunch: may the second of two independent kernels start before the first has finished? (hip_ext.h says the
  // flag is not supported on gfx9xx; measured anyway: a pair of 64-block spin kernels, each a quarter of the chip)
  for (int flag : {0, 1}) {
    t0 = now();
    for (int i = 0; i < 500; i++) {
      hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, nullptr, nullptr, 0, d, 20000ll);
      hipExtLaunchKernelGGL(k_spin, dim3(64), dim3(256), 0, s0, nullptr, nullptr, flag, d + 64, 20000ll);
    }
    t1 = now(); CK(hipStreamSynchronize(s0)); t2 = now();
    printf("pairs of quarter-chip 10 us kernels, second one with flags = %d: host %.2f us, drained %.2f us per pair\n", flag, (t1 - t0) / 500 * 1e6, (t2 - t0) / 500 * 1e6);
  }
  // 5. the null stream (what a torch host hands the library by default)
  t0 = now();
  for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_tiny, dim3(1), dim3(64), 0, 0, d);
  t1 = now(); CK(hipDeviceSynchronize()); t2 = now();
  printf("null stream: host %.2f us per launch, drained %.2f us per kernel\n", (t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6);
  return 0;
}
