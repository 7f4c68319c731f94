// Round trip of an 8-byte {value, tag} pair between two workgroups through global memory (development micro-benchmark for
// the hand-off between sub-boxes of the PCG sweeps, pcg.hip). Block A stores pair n, block B waits for it and stores its
// own pair n, A waits for that: time / iterations = one round trip = two one-way latencies. Variants: which two blocks
// of the launch play (consecutive workgroup ids sit on different XCDs, ids 8 apart on the same one) and the cache policy
// bits of the accesses.
// build: hipcc --offload-arch=gfx950 -O3 pingpong.hip -o pingpong
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__device__ __forceinline__ unsigned long long ld(const unsigned long long* p) {
  unsigned long long v;
  if (MODE == 0) asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  if (MODE == 1) asm volatile("global_load_dwordx2 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  if (MODE == 2) asm volatile("global_load_dwordx2 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  if (MODE == 3) asm volatile("global_load_dwordx2 %0, %1, off nt sc1\n\ts_waitcnt vmcnt(0)" : "=&v"(v) : "v"(p) : "memory");
  return v;
}
template <int MODE>
__device__ __forceinline__ void st(unsigned long long* p, unsigned long long v) {
  if (MODE == 0) asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
  if (MODE == 1) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" : : "v"(p), "v"(v) : "memory");
  if (MODE == 2) asm volatile("global_store_dwordx2 %0, %1, off sc0" : : "v"(p), "v"(v) : "memory");
  if (MODE == 3) asm volatile("global_store_dwordx2 %0, %1, off nt sc1" : : "v"(p), "v"(v) : "memory");
}

template <int MODE>
__global__ void k(unsigned long long* a, unsigned long long* b, int iters, int ida, int idb, int* xcc, long long* fail) {
  if (threadIdx.x != 0) return;
  const int me = blockIdx.x;
  if (me != ida && me != idb) return;
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  xcc[me == ida ? 0 : 1] = (int)(id & 0xf);
  for (int n = 1; n <= iters; n++) {
    const unsigned long long want = ((unsigned long long)n << 32) | (unsigned)n;
    if (me == ida) {
      st<MODE>(a, want);
      long long spin = 0;
      while (ld<MODE>(b) != want) if (++spin > (1ll << 22)) { *fail = n; return; }
    } else {
      long long spin = 0;
      while (ld<MODE>(a) != want) if (++spin > (1ll << 22)) { *fail = n; return; }
      st<MODE>(b, want);
    }
  }
}

template <int MODE>
void run(const char* name, int ida, int idb) {
  unsigned long long *a, *b; int* xcc; long long* fail;
  hipMalloc(&a, 4096); hipMalloc(&b, 4096); hipMalloc(&xcc, 64); hipMalloc(&fail, 8);
  hipMemset(a, 0, 4096); hipMemset(b, 0, 4096); hipMemset(fail, 0, 8);
  const int iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  k<MODE><<<64, 64>>>(a, b, iters, ida, idb, xcc, fail);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  int hx[2]; long long hf;
  hipMemcpy(hx, xcc, 8, hipMemcpyDeviceToHost); hipMemcpy(&hf, fail, 8, hipMemcpyDeviceToHost);
  printf("%-22s blocks %2d,%2d (XCC %d,%d): round trip %7.0f ns%s\n", name, ida, idb, hx[0], hx[1], ms * 1e6 / iters, hf ? "  TIMED OUT (not coherent)" : "");
  hipFree(a); hipFree(b); hipFree(xcc); hipFree(fail);
}

int main() {
  run<0>("sc1 (agent)", 0, 1); run<0>("sc1 (agent)", 0, 8); run<0>("sc1 (agent)", 0, 16);
  run<1>("sc0 sc1 (system)", 0, 1); run<1>("sc0 sc1 (system)", 0, 8);
  run<3>("nt sc1", 0, 1); run<3>("nt sc1", 0, 8);
  run<2>("sc0 (workgroup)", 0, 8); run<2>("sc0 (workgroup)", 0, 1);
  return 0;
}
