/* step_from_c.c -- calling libtfluids_hip.so from plain C through include/tfluids_hip.h: no torch, no Python.
 * This is what a cgo / JNI / LuaJIT-FFI binding does. It builds a 32^3 box, runs
 * setWallBcs + a MacCormack velocity advection + divergence + 20 Jacobi iterations + velocityUpdate on the GPU
 * and checks that the projection reduced the divergence.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ examples/step_from_c.c -Iinclude -I/opt/rocm/include \
 *       -Lfluidnet_amd -ltfluids_hip -L/opt/rocm/lib -lamdhip64 -lm \
 *       -Wl,-rpath,$PWD/fluidnet_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/step_from_c && /tmp/step_from_c
 * (the HIP runtime is linked only because the example itself uses hipMalloc/hipMemcpy for its buffers; the
 *  platform define is what hip_runtime_api.h needs when the compiler is not hipcc.)
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "tfluids_hip.h"

#define CHECK(call)                                                                       \
  do {                                                                                    \
    int rc_ = (call);                                                                     \
    if (rc_ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, tfl_last_error(ctx)); return 1; } \
  } while (0)

static float* dev_alloc(size_t n) {
  float* p = NULL;
  if (hipMalloc((void**)&p, n * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(2); }
  (void)hipMemset(p, 0, n * sizeof(float));
  return p;
}

static double l2(const float* dev, size_t n) {
  float* h = (float*)malloc(n * sizeof(float));
  double s = 0.0;
  (void)hipMemcpy(h, dev, n * sizeof(float), hipMemcpyDeviceToHost);
  for (size_t i = 0; i < n; i++) s += (double)h[i] * h[i];
  free(h);
  return sqrt(s);
}

int main(void) {
  const int R = 32;
  const size_t N = (size_t)R * R * R;
  tfl_ctx* ctx = tfl_create(0);
  if (!ctx) { fprintf(stderr, "tfl_create failed (no GPU?)\n"); return 3; }
  if (tfl_abi_version() != TFL_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 4; }

  tfl_tensor flags = {dev_alloc(N), 1, 1, R, R, R}, p = {dev_alloc(N), 1, 1, R, R, R}, div = {dev_alloc(N), 1, 1, R, R, R};
  tfl_tensor pPrev = {dev_alloc(N), 1, 1, R, R, R}, pDelta = {dev_alloc(N), 1, 1, R, R, R}, pNorm = {dev_alloc(1), 1, 1, 1, 1, 1};
  tfl_tensor U = {dev_alloc(3 * N), 1, 3, R, R, R}, fwd = {dev_alloc(3 * N), 1, 3, R, R, R};
  tfl_tensor bwd = {dev_alloc(3 * N), 1, 3, R, R, R}, Unew = {dev_alloc(3 * N), 1, 3, R, R, R};

  /* a swirling initial velocity, uploaded from the host */
  float* h = (float*)malloc(3 * N * sizeof(float));
  for (int k = 0; k < R; k++)
    for (int j = 0; j < R; j++)
      for (int i = 0; i < R; i++) {
        const size_t o = ((size_t)k * R + j) * R + i;
        h[o] = 4.0f * sinf(0.3f * j) * cosf(0.2f * k);
        h[N + o] = 4.0f * sinf(0.25f * i + 0.1f * k);
        h[2 * N + o] = 3.0f * cosf(0.2f * i) * sinf(0.3f * j);
      }
  (void)hipMemcpy(U.data, h, 3 * N * sizeof(float), hipMemcpyHostToDevice);
  free(h);

  CHECK(tfl_emptyDomain(ctx, &flags, 1, 1));
  CHECK(tfl_setWallBcsForward(ctx, &U, &flags, 1));
  CHECK(tfl_advectVel(ctx, 0.1f, &U, &flags, &fwd, &bwd, 1, "maccormackOurs", 1, 0.75f, &Unew));
  CHECK(tfl_setWallBcsForward(ctx, &Unew, &flags, 1));
  CHECK(tfl_velocityDivergenceForward(ctx, &Unew, &flags, &div, 1));
  const double div0 = l2(div.data, N);
  float residual = -1.0f;
  CHECK(tfl_solveLinearSystemJacobi(ctx, &p, &flags, &div, &pPrev, &pDelta, &pNorm, 1, 0.0f, 200, 0, &residual));
  CHECK(tfl_velocityUpdateForward(ctx, &Unew, &flags, &p, 1));
  CHECK(tfl_velocityDivergenceForward(ctx, &Unew, &flags, &div, 1));
  CHECK(tfl_synchronize(ctx));
  const double div1 = l2(div.data, N);
  const long long terr = (long long)tfl_trace_errors(ctx);
  printf("||div|| before %.4f  after 200 Jacobi iterations %.4f  (residual %.3e, trace errors %lld)\n", div0, div1,
         (double)residual, terr);
  tfl_destroy(ctx);
  if (!(div1 < 0.5 * div0) || terr != 0) { printf("FAILED\n"); return 5; }
  printf("OK\n");
  return 0;
}
