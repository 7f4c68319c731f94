/* step_from_c.c -- calling libtfluids_hip.so from plain C through include/tfluids_hip.h: no torch, no Python.
 * This is what a cgo / JNI / LuaJIT-FFI binding does. It builds a 32^3 box, runs
 * setWallBcs + a MacCormack velocity advection + divergence + 20 Jacobi iterations + velocityUpdate on the GPU
 * and checks that the projection reduced the divergence; then it runs 40 whole tfluids.simulate() steps of a
 * buoyant smoke plume through tfl_simulate_step (one call per step) and checks that the smoke rose.
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
  if (!(div1 < 0.5 * div0) || terr != 0) { printf("FAILED\n"); tfl_destroy(ctx); return 5; }

  /* ---- the whole tfluids.simulate() step as ONE call: a buoyant plume, Jacobi projection --------------------
   * BC tensors as lib/simulate.lua's createPlumeBCs builds them: density 1 and upward velocity on rows 1..3 of a
   * disc under the floor, masks 0 there. tfl_bc_plan_create scans each (BC, invMask) pair once. */
  {
    tfl_tensor rho = {dev_alloc(N), 1, 1, R, R, R}, rhoBC = {dev_alloc(N), 1, 1, R, R, R}, rhoMask = {dev_alloc(N), 1, 1, R, R, R};
    tfl_tensor UBC = {dev_alloc(3 * N), 1, 3, R, R, R}, UMask = {dev_alloc(3 * N), 1, 3, R, R, R};
    float* hb = (float*)calloc(N, sizeof(float));
    float* hm = (float*)malloc(N * sizeof(float));
    float* hub = (float*)calloc(3 * N, sizeof(float));
    float* hum = (float*)malloc(3 * N * sizeof(float));
    for (size_t t = 0; t < N; t++) hm[t] = 1.0f;
    for (size_t t = 0; t < 3 * N; t++) hum[t] = 1.0f;
    for (int k = 0; k < R; k++)
      for (int j = 1; j < 4; j++)
        for (int i = 0; i < R; i++) {
          const size_t o = ((size_t)k * R + j) * R + i;
          const int in = (i - R / 2) * (i - R / 2) + (k - R / 2) * (k - R / 2) <= (R / 5) * (R / 5);
          hm[o] = in ? 0.0f : 1.0f; hb[o] = in ? 1.0f : 0.0f;
          for (int c = 0; c < 3; c++) hum[c * N + o] = 0.0f;
          hub[N + o] = in ? 1.0f : 0.0f;
        }
    (void)hipMemcpy(rhoBC.data, hb, N * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemcpy(rhoMask.data, hm, N * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemcpy(UBC.data, hub, 3 * N * sizeof(float), hipMemcpyHostToDevice);
    (void)hipMemcpy(UMask.data, hum, 3 * N * sizeof(float), hipMemcpyHostToDevice);
    free(hb); free(hm); free(hub); free(hum);
    (void)hipMemset(Unew.data, 0, 3 * N * sizeof(float));
    (void)hipMemset(p.data, 0, N * sizeof(float));
    CHECK(tfl_emptyDomain(ctx, &flags, 1, 1));
    tfl_bc_plan* planU = tfl_bc_plan_create(ctx, &UBC, &UMask);
    tfl_bc_plan* planR = tfl_bc_plan_create(ctx, &rhoBC, &rhoMask);
    if (!planU || !planR) { fprintf(stderr, "tfl_bc_plan_create failed\n"); return 6; }
    tfl_sim_params prm = {0};
    prm.dt = 0.1f; prm.maccormackStrength = 0.6f; prm.buoyancyScale = 1.0f; prm.gravity[1] = 1.0f;
    prm.vorticityConfinementAmp = 0.5f; prm.simMethod = "jacobi"; prm.maxIter = 30;
    tfl_sim_state st = {0};
    st.p = &p; st.U = &Unew; st.flags = &flags; st.n_density = 1; st.density[0] = &rho; st.UBC = planU; st.densityBC[0] = planR;
    const long long nws = (long long)tfl_simulate_workspace_floats(ctx, &prm, &st);
    float* ws = dev_alloc((size_t)nws);
    for (int step = 0; step < 40; step++) CHECK(tfl_simulate_step(ctx, &prm, &st, ws, nws));
    CHECK(tfl_synchronize(ctx));
    float* hr = (float*)malloc(N * sizeof(float));
    (void)hipMemcpy(hr, rho.data, N * sizeof(float), hipMemcpyDeviceToHost);
    double above = 0.0;
    for (int k = 0; k < R; k++)
      for (int j = 8; j < R; j++)
        for (int i = 0; i < R; i++) above += hr[((size_t)k * R + j) * R + i];
    free(hr);
    printf("40 tfl_simulate_step calls: smoke above row 8 = %.2f, |U| = %.3f\n", above, l2(Unew.data, 3 * N));
    tfl_bc_plan_destroy(ctx, planU); tfl_bc_plan_destroy(ctx, planR);
    if (!(above > 1.0)) { printf("FAILED (plume did not rise)\n"); tfl_destroy(ctx); return 7; }
  }
  tfl_destroy(ctx);
  printf("OK\n");
  return 0;
}
