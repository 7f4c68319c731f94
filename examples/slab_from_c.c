/* slab_from_c.c -- the z-slab step of libtfluids_hip.so driven from plain C: what a LuaJIT / cgo host with no communication
 * layer of its own does. Two ranks cut a 32 x 24 x 32 (z, y, x) plume grid along z; each rank is a
 * THREAD of this process with its own tfl_ctx and HIP stream (on a real node: one process per GPU), the halo messages go through
 * the library's OWN transport (tfl_rccl_comm_create: ncclSend / ncclRecv / ncclAllReduce, dlopen'ed -- here from the in-process
 * stand-in tests/stub_rccl.cpp named by TFL_RCCL_LIBRARY, because RCCL itself refuses two ranks on one device), the ConvNet
 * projection uses seeded weights built through tfl_model_create. Checked at the end: the owned planes of both ranks equal the
 * un-cut tfl_simulate_step of the same state (bit for bit up to the summation order of the all-reduce: rel-L2 <= 1e-7).
 * Then the round-6 entry points on a slab WITHOUT neighbours: the step recorded into a HIP graph (tfl_slab_graph_create /
 * _step) must reproduce the eager step exactly, and check_reach = 2 must refuse a too fast flow BEFORE the step (TFL_EREACH,
 * tfl_slab_needed_reach) with the state untouched.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ examples/slab_from_c.c -Iinclude -I/opt/rocm/include -Lfluidnet_amd -ltfluids_hip \
 *       -L/opt/rocm/lib -lamdhip64 -lm -lpthread -Wl,-rpath,$PWD/fluidnet_amd -Wl,-rpath,/opt/rocm/lib -o /tmp/slab_from_c
 *   TFL_RCCL_LIBRARY=/path/to/libstub_rccl.so /tmp/slab_from_c
 *
 * `slab_from_c --processes` (round 6) is the same two-rank run as a real node does it: one PROCESS per rank, the REAL librccl
 * (no TFL_RCCL_LIBRARY), the communicator's unique id handed from rank 0 to rank 1 through a file. With two GPUs in the box each
 * rank takes its own; with one, both ranks share device 0 and each process is started under its own NCCL_HOSTID, so that RCCL
 * takes them for two hosts (its "Duplicate GPU detected" test compares host hash and bus id) and carries the messages through
 * its socket transport over the loopback interface. Every rank checks its owned planes against the un-cut step itself.
 */
#define _DEFAULT_SOURCE      /* mkdtemp, usleep, setenv under -std=c99 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include "tfluids_hip.h"

enum { Z = 32, Y = 24, X = 32, WORLD = 2, STEPS = 4 };
#define YX ((size_t)Y * X)

static float* dev_alloc(size_t n) {
  float* p = NULL;
  if (hipMalloc((void**)&p, n * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(2); }
  (void)hipMemset(p, 0, n * sizeof(float));
  (void)hipDeviceSynchronize();      /* the ranks' streams are non-blocking: they do not wait for the null stream's memset */
  return p;
}
static float* dev_from(const float* h, size_t n) {
  float* p = dev_alloc(n);
  (void)hipMemcpy(p, h, n * sizeof(float), hipMemcpyHostToDevice);
  return p;
}

/* the global initial state on the host: [C][Z][Y][X] fields */
typedef struct { float *p, *U, *flags, *rho, *UBC, *UMask, *rhoBC, *rhoMask; } HostState;

static unsigned lcg(unsigned* s) { *s = *s * 1664525u + 1013904223u; return *s >> 8; }
static float unif(unsigned* s) { return (float)lcg(s) / 16777216.0f - 0.5f; }

static void make_state(HostState* h) {
  const size_t N = (size_t)Z * YX;
  h->p = calloc(N, 4); h->U = calloc(3 * N, 4); h->flags = malloc(N * 4); h->rho = calloc(N, 4);
  h->UBC = calloc(3 * N, 4); h->UMask = malloc(3 * N * 4); h->rhoBC = calloc(N, 4); h->rhoMask = malloc(N * 4);
  for (size_t t = 0; t < N; t++) h->rhoMask[t] = 1.0f;
  for (size_t t = 0; t < 3 * N; t++) h->UMask[t] = 1.0f;
  for (int k = 0; k < Z; k++)
    for (int j = 0; j < Y; j++)
      for (int i = 0; i < X; i++) {
        const size_t o = ((size_t)k * Y + j) * X + i;
        const int border = i == 0 || i == X - 1 || j == 0 || j == Y - 1 || k == 0 || k == Z - 1;
        const int box = i >= 12 && i < 18 && j >= 12 && j < 16 && k >= 13 && k < 19;      /* an obstacle across the cut */
        h->flags[o] = (border || box) ? 2.0f : 1.0f;
        if (!border && !box) {                                                          /* a gentle swirl: < 1 cell per step */
          h->U[o] = 2.0f * sinf(0.3f * j) * cosf(0.2f * k);
          h->U[N + o] = 2.0f * sinf(0.25f * i + 0.1f * k);
          h->U[2 * N + o] = 3.0f * cosf(0.2f * i) * sinf(0.3f * j);
        }
        if (j >= 1 && j < 4) {                                                          /* createPlumeBCs: a disc on rows 1..3 */
          const int in = (i - X / 2) * (i - X / 2) + (k - Z / 2) * (k - Z / 2) <= 36;
          h->rhoMask[o] = in ? 0.0f : 1.0f; h->rhoBC[o] = in ? 1.0f : 0.0f;
          for (int c = 0; c < 3; c++) h->UMask[c * N + o] = 0.0f;
          h->UBC[N + o] = in ? 1.0f : 0.0f;
        }
      }
}

/* planes [lo, hi) of a C-channel global field as a device tensor */
static tfl_tensor cut(const float* g, int C, int lo, int hi) {
  const size_t N = (size_t)Z * YX, n = (size_t)(hi - lo) * YX;
  float* h = malloc(C * n * 4);
  for (int c = 0; c < C; c++) memcpy(h + c * n, g + c * N + (size_t)lo * YX, n * 4);
  tfl_tensor t = {dev_from(h, C * n), 1, C, hi - lo, Y, X};
  free(h);
  return t;
}

/* the 3-D `default` topology (lib/model.lua:219-226) with seeded He-scaled weights: 3 -> 8 -> 8 -> 8 (k 3), 8 -> 8, 8 -> 1 (k 1) */
static tfl_model* make_model(tfl_ctx* ctx) {
  const int32_t cin[5] = {3, 8, 8, 8, 8}, cout[5] = {8, 8, 8, 8, 1}, ks[5] = {3, 3, 3, 1, 1};
  float* w[5]; float* b[5];
  unsigned seed = 12345u;
  for (int l = 0; l < 5; l++) {
    const int taps = ks[l] * ks[l] * ks[l], nw = cout[l] * cin[l] * taps;
    const float sc = sqrtf(2.0f / (float)(cin[l] * taps)) * 1.7f;
    w[l] = malloc(nw * 4); b[l] = malloc(cout[l] * 4);
    for (int t = 0; t < nw; t++) w[l][t] = unif(&seed) * sc;
    for (int t = 0; t < cout[l]; t++) b[l][t] = unif(&seed) * 0.05f;
  }
  tfl_model* m = tfl_model_create(ctx, 1, 5, cin, cout, ks, (const float* const*)w, (const float* const*)b);
  for (int l = 0; l < 5; l++) { free(w[l]); free(b[l]); }
  return m;
}

static void set_params(tfl_sim_params* prm) {
  memset(prm, 0, sizeof(*prm));
  prm->dt = 0.1f; prm->maccormackStrength = 0.6f; prm->buoyancyScale = 1.0; prm->gravity[1] = 1.0f;
  prm->vorticityConfinementAmp = 1.0; prm->simMethod = "convnet";
}

typedef struct { tfl_tensor p, U, flags, rho, UBC, UMask, rhoBC, rhoMask; tfl_bc_plan *planU, *planR; tfl_wall_plan* wall; tfl_sim_state st; } DevState;
static int make_dev(tfl_ctx* ctx, const HostState* h, int lo, int hi, tfl_model* model, DevState* d) {
  d->p = cut(h->p, 1, lo, hi); d->U = cut(h->U, 3, lo, hi); d->flags = cut(h->flags, 1, lo, hi); d->rho = cut(h->rho, 1, lo, hi);
  d->UBC = cut(h->UBC, 3, lo, hi); d->UMask = cut(h->UMask, 3, lo, hi); d->rhoBC = cut(h->rhoBC, 1, lo, hi); d->rhoMask = cut(h->rhoMask, 1, lo, hi);
  d->planU = tfl_bc_plan_create(ctx, &d->UBC, &d->UMask);
  d->planR = tfl_bc_plan_create(ctx, &d->rhoBC, &d->rhoMask);
  if (!d->planU || !d->planR) return 1;
  d->wall = tfl_wall_plan_create(ctx, &d->flags);      /* optional: the scene's wall decisions as one byte per cell, found by the step through the flags' address */
  memset(&d->st, 0, sizeof(d->st));
  d->st.p = &d->p; d->st.U = &d->U; d->st.flags = &d->flags; d->st.n_density = 1; d->st.density[0] = &d->rho;
  d->st.UBC = d->planU; d->st.densityBC[0] = d->planR; d->st.model = model;
  return 0;
}

typedef struct { int rank; const HostState* h; const char* uid; float* owned[3]; int rc; char err[512]; int device; } RankArg;

static void* rank_main(void* vp) {
  RankArg* a = vp;
  a->rc = 1;
  tfl_ctx* ctx = tfl_create(a->device);
  hipStream_t st = NULL;
  if (!ctx || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { snprintf(a->err, sizeof a->err, "tfl_create / stream"); return NULL; }
  tfl_set_stream(ctx, st);
  const int per = Z / WORLD, H = tfl_slab_halo(1);
  const int z0 = a->rank * per, z1 = z0 + per, lo = z0 - H < 0 ? 0 : z0 - H, hi = z1 + H > Z ? Z : z1 + H;
  tfl_model* model = make_model(ctx);
  DevState d;
  if (!model || make_dev(ctx, a->h, lo, hi, model, &d)) { snprintf(a->err, sizeof a->err, "model / plans: %s", tfl_last_error(ctx)); return NULL; }
  tfl_slab slab = {Z, lo, z0 - lo, z1 - lo, 1, 0, 1, 0};
  tfl_sim_params prm; set_params(&prm);
  tfl_rccl_comm* rc = tfl_rccl_comm_create(ctx, a->uid, a->rank, WORLD);      /* collective */
  if (!rc) { snprintf(a->err, sizeof a->err, "tfl_rccl_comm_create: %s", tfl_last_error(ctx)); return NULL; }
  tfl_rccl_comm_set_inline(rc, 1);                                            /* overlap = 0: RCCL on the step's own stream */
  const tfl_comm* comm = tfl_rccl_comm_callbacks(rc);
  const long long nws = (long long)tfl_simulate_slab_workspace_floats(ctx, &prm, &d.st, &slab);
  float* ws = dev_alloc((size_t)nws);
  for (int s = 0; s < STEPS; s++)
    if (tfl_simulate_step_slab(ctx, &prm, &d.st, &slab, comm, ws, nws) != 0) { snprintf(a->err, sizeof a->err, "step %d: %s", s, tfl_last_error(ctx)); return NULL; }
  if (tfl_slab_drain(ctx, &d.st, &slab, comm, ws, nws) != 0 || tfl_synchronize(ctx) != 0) { snprintf(a->err, sizeof a->err, "drain: %s", tfl_last_error(ctx)); return NULL; }
  /* owned planes of p, U, rho back to the host */
  const size_t n = (size_t)per * YX, nl = (size_t)(hi - lo) * YX;
  a->owned[0] = malloc(n * 4); a->owned[1] = malloc(3 * n * 4); a->owned[2] = malloc(n * 4);
  (void)hipMemcpy(a->owned[0], d.p.data + (size_t)(z0 - lo) * YX, n * 4, hipMemcpyDeviceToHost);
  for (int c = 0; c < 3; c++) (void)hipMemcpy(a->owned[1] + c * n, d.U.data + c * nl + (size_t)(z0 - lo) * YX, n * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(a->owned[2], d.rho.data + (size_t)(z0 - lo) * YX, n * 4, hipMemcpyDeviceToHost);
  tfl_rccl_comm_destroy(ctx, rc);
  tfl_bc_plan_destroy(ctx, d.planU); tfl_bc_plan_destroy(ctx, d.planR); tfl_wall_plan_destroy(ctx, d.wall); tfl_model_destroy(ctx, model);
  tfl_destroy(ctx);
  a->rc = 0;
  return NULL;
}

static double rel_l2(const float* a, const float* b, size_t n) {
  double num = 0.0, den = 0.0;
  for (size_t i = 0; i < n; i++) { num += ((double)a[i] - b[i]) * ((double)a[i] - b[i]); den += (double)b[i] * b[i]; }
  return sqrt(num) / (sqrt(den) > 1e-30 ? sqrt(den) : 1e-30);
}

/* the un-cut run: STEPS steps of tfl_simulate_step on the whole grid; p, U, rho back on the host */
static int run_uncut(tfl_ctx* ctx, const HostState* h, tfl_model** model, DevState* g, float* rp, float* rU, float* rr) {
  *model = make_model(ctx);
  if (!*model || make_dev(ctx, h, 0, Z, *model, g)) { fprintf(stderr, "set-up: %s\n", tfl_last_error(ctx)); return 5; }
  tfl_sim_params prm; set_params(&prm);
  const long long nws = (long long)tfl_simulate_workspace_floats(ctx, &prm, &g->st);
  float* ws = dev_alloc((size_t)nws);
  for (int s = 0; s < STEPS; s++)
    if (tfl_simulate_step(ctx, &prm, &g->st, ws, nws) != 0) { fprintf(stderr, "tfl_simulate_step: %s\n", tfl_last_error(ctx)); return 6; }
  tfl_synchronize(ctx);
  const size_t N = (size_t)Z * YX;
  (void)hipMemcpy(rp, g->p.data, N * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(rU, g->U.data, 3 * N * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(rr, g->rho.data, N * 4, hipMemcpyDeviceToHost);
  return 0;
}

/* worst rel-L2 of a rank's owned planes (p, the three components of U, rho) against the un-cut fields */
static double owned_error(const RankArg* a, const float* rp, const float* rU, const float* rr) {
  const int per = Z / WORLD;
  const size_t N = (size_t)Z * YX, n = (size_t)per * YX, off = (size_t)a->rank * per * YX;
  double e = rel_l2(a->owned[0], rp + off, n);
  for (int c = 0; c < 3; c++) { const double ec = rel_l2(a->owned[1] + c * n, rU + c * N + off, n); if (ec > e) e = ec; }
  const double er = rel_l2(a->owned[2], rr + off, n);
  return er > e ? er : e;
}

/* ---- `--rank r dir`: one rank of the process mode -------------------------------------------------------------------- */
static int rank_process(int rank, const char* dir) {
  HostState h; make_state(&h);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { fprintf(stderr, "rank %d: no GPU\n", rank); return 3; }
  const int dev = ndev >= WORLD ? rank : 0;
  tfl_ctx* ctx = tfl_create(dev);
  if (!ctx) { fprintf(stderr, "rank %d: tfl_create failed\n", rank); return 3; }
  char uid[TFL_RCCL_UNIQUE_ID_BYTES], path[1024], tmp[1100];
  snprintf(path, sizeof path, "%s/uid", dir); snprintf(tmp, sizeof tmp, "%s.tmp", path);
  if (rank == 0) {                      /* the id is made by ONE rank and handed to the others: here through a file */
    if (tfl_rccl_get_unique_id(ctx, uid) != 0) { fprintf(stderr, "tfl_rccl_get_unique_id: %s\n", tfl_last_error(ctx)); return 7; }
    FILE* f = fopen(tmp, "wb");
    if (!f || fwrite(uid, 1, sizeof uid, f) != sizeof uid) { fprintf(stderr, "rank 0: cannot write %s\n", tmp); return 7; }
    fclose(f); rename(tmp, path);
  } else {
    FILE* f = NULL;
    for (int t = 0; t < 1200 && !(f = fopen(path, "rb")); t++) usleep(50000);
    if (!f || fread(uid, 1, sizeof uid, f) != sizeof uid) { fprintf(stderr, "rank %d: no unique id from rank 0\n", rank); return 7; }
    fclose(f);
  }
  RankArg a; memset(&a, 0, sizeof a);
  a.rank = rank; a.h = &h; a.uid = uid; a.device = dev;
  rank_main(&a);
  if (a.rc) { fprintf(stderr, "rank %d failed: %s\n", rank, a.err); return 8; }
  const size_t N = (size_t)Z * YX;
  float *rp = malloc(N * 4), *rU = malloc(3 * N * 4), *rr = malloc(N * 4);
  tfl_model* model; DevState g;
  const int rc = run_uncut(ctx, &h, &model, &g, rp, rU, rr);
  if (rc) return rc;
  const double e = owned_error(&a, rp, rU, rr);
  printf("process %d of %d (device %d, %s): owned planes against the un-cut step after %d steps over %s: rel-L2 %.3e\n", rank, WORLD, dev,
         ndev >= WORLD ? "one GPU per rank" : "ranks share the GPU", STEPS, tfl_rccl_comm_origin(ctx), e);
  return e <= 1e-7 ? 0 : 9;
}

/* ---- `--processes`: start the ranks, wait for them ------------------------------------------------------------------- */
static int run_processes(const char* self) {
  char dir[] = "/tmp/slab_from_c.XXXXXX";
  if (!mkdtemp(dir)) { perror("mkdtemp"); return 3; }
  int ndev = 0;
  pid_t pid[WORLD];
  for (int r = 0; r < WORLD; r++) {
    pid[r] = fork();                      /* (no HIP call has been made in this process: the children start clean) */
    if (pid[r] == 0) {
      char rs[16], host[64];
      snprintf(rs, sizeof rs, "%d", r); snprintf(host, sizeof host, "slab-from-c-rank%d", r);
      unsetenv("TFL_RCCL_LIBRARY");       /* the real library */
      setenv("HSA_ENABLE_IPC_MODE_LEGACY", "0", 0);
      if (!getenv("SLAB_FROM_C_ONE_GPU_PER_RANK")) {      /* harmless with one GPU per rank too: RCCL then uses its network transport */
        setenv("NCCL_HOSTID", host, 1); setenv("NCCL_SOCKET_IFNAME", "lo", 0); setenv("NCCL_IB_DISABLE", "1", 0);
      }
      execl(self, self, "--rank", rs, dir, (char*)NULL);
      perror("execl"); _exit(127);
    }
  }
  (void)ndev;
  int bad = 0;
  for (int r = 0; r < WORLD; r++) {
    int stw = 0;
    if (waitpid(pid[r], &stw, 0) < 0 || !WIFEXITED(stw) || WEXITSTATUS(stw) != 0) { fprintf(stderr, "process of rank %d ended with status 0x%x\n", r, stw); bad = 1; }
  }
  printf(bad ? "FAILED\n" : "OK (two processes over the real RCCL)\n");
  return bad ? 9 : 0;
}

int main(int argc, char** argv) {
  if (argc >= 2 && strcmp(argv[1], "--processes") == 0) return run_processes(argv[0]);
  if (argc >= 4 && strcmp(argv[1], "--rank") == 0) return rank_process(atoi(argv[2]), argv[3]);
  if (!getenv("TFL_RCCL_LIBRARY")) { fprintf(stderr, "set TFL_RCCL_LIBRARY to tests/stub_rccl.cpp built as a shared library (two ranks share one GPU here), or run `--processes`\n"); return 3; }
  HostState h; make_state(&h);
  tfl_ctx* ctx = tfl_create(0);
  if (!ctx) { fprintf(stderr, "tfl_create failed (no GPU?)\n"); return 3; }
  if (tfl_abi_version() != TFL_ABI_VERSION) { fprintf(stderr, "ABI mismatch\n"); return 4; }
  /* ---- the un-cut run ------------------------------------------------------------------------------------------ */
  const size_t N = (size_t)Z * YX;
  float *rp = malloc(N * 4), *rU = malloc(3 * N * 4), *rr = malloc(N * 4);
  tfl_model* model; DevState g;
  { const int rc0 = run_uncut(ctx, &h, &model, &g, rp, rU, rr); if (rc0) return rc0; }
  tfl_sim_params prm; set_params(&prm);
  /* ---- the two ranks ---------------------------------------------------------------------------------------------- */
  char uid[TFL_RCCL_UNIQUE_ID_BYTES];
  if (tfl_rccl_get_unique_id(ctx, uid) != 0) { fprintf(stderr, "tfl_rccl_get_unique_id: %s\n", tfl_last_error(ctx)); return 7; }
  RankArg ra[WORLD]; pthread_t th[WORLD];
  for (int r = 0; r < WORLD; r++) { memset(&ra[r], 0, sizeof ra[r]); ra[r].rank = r; ra[r].h = &h; ra[r].uid = uid; pthread_create(&th[r], NULL, rank_main, &ra[r]); }
  for (int r = 0; r < WORLD; r++) pthread_join(th[r], NULL);
  const int per = Z / WORLD;
  double worst = 0.0;
  for (int r = 0; r < WORLD; r++) {
    if (ra[r].rc) { fprintf(stderr, "rank %d failed: %s\n", r, ra[r].err); return 8; }
    const double e = owned_error(&ra[r], rp, rU, rr);
    printf("rank %d of %d: owned planes [%d, %d) against the un-cut step after %d steps: rel-L2 %.3e\n", r, WORLD, r * per, (r + 1) * per, STEPS, e);
    if (e > worst) worst = e;
  }
  if (!(worst <= 1e-7) || !(rel_l2(rU, rU, 8) == 0.0)) { printf("FAILED (cut run differs)\n"); return 9; }
  /* ---- a slab without neighbours: the recorded step, and the exact reach mode --------------------------------------- */
  {
    tfl_slab slab = {Z, 0, 0, Z, 1, 0, 1, 0};
    const long long ns = (long long)tfl_simulate_slab_workspace_floats(ctx, &prm, &g.st, &slab);
    float* ws2 = dev_alloc((size_t)ns);
    for (int s = 0; s < 2; s++)
      if (tfl_simulate_step_slab(ctx, &prm, &g.st, &slab, NULL, ws2, ns) != 0) { fprintf(stderr, "slab step: %s\n", tfl_last_error(ctx)); return 10; }
    tfl_slab_graph* gr = tfl_slab_graph_create(ctx, &prm, &g.st, &slab, NULL, ws2, ns);
    if (!gr) { fprintf(stderr, "tfl_slab_graph_create: %s\n", tfl_last_error(ctx)); return 11; }
    /* eager on a copy of the state vs replay on the state: the same bits */
    float* Ucopy = dev_alloc(3 * N); float* pcopy = dev_alloc(N); float* rcopy = dev_alloc(N);
    (void)hipMemcpy(Ucopy, g.U.data, 3 * N * 4, hipMemcpyDeviceToDevice); (void)hipMemcpy(pcopy, g.p.data, N * 4, hipMemcpyDeviceToDevice);
    (void)hipMemcpy(rcopy, g.rho.data, N * 4, hipMemcpyDeviceToDevice);
    for (int s = 0; s < 3; s++) if (tfl_slab_graph_step(ctx, gr) != 0) { fprintf(stderr, "graph step: %s\n", tfl_last_error(ctx)); return 12; }
    tfl_synchronize(ctx);
    float* Ug = malloc(3 * N * 4);
    (void)hipMemcpy(Ug, g.U.data, 3 * N * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(g.U.data, Ucopy, 3 * N * 4, hipMemcpyDeviceToDevice); (void)hipMemcpy(g.p.data, pcopy, N * 4, hipMemcpyDeviceToDevice);
    (void)hipMemcpy(g.rho.data, rcopy, N * 4, hipMemcpyDeviceToDevice);
    for (int s = 0; s < 3; s++) if (tfl_simulate_step_slab(ctx, &prm, &g.st, &slab, NULL, ws2, ns) != 0) { fprintf(stderr, "slab step: %s\n", tfl_last_error(ctx)); return 13; }
    tfl_synchronize(ctx);
    float* Ue = malloc(3 * N * 4);
    (void)hipMemcpy(Ue, g.U.data, 3 * N * 4, hipMemcpyDeviceToHost);
    const int same = memcmp(Ug, Ue, 3 * N * 4) == 0;
    printf("recorded rank-step (%lld graph nodes) against the eager one after 3 steps: %s\n", (long long)tfl_slab_graph_nodes(gr), same ? "identical" : "DIFFERENT");
    tfl_slab_graph_destroy(ctx, gr);
    if (!same) { printf("FAILED\n"); return 14; }
    /* check_reach = 2: a flow of 2.5 cells per step along z is refused before anything is written */
    float* fast = malloc(N * 4);
    for (size_t t = 0; t < N; t++) fast[t] = 25.0f;
    (void)hipMemcpy(g.U.data + 2 * N, fast, N * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(Ucopy, g.U.data, 3 * N * 4, hipMemcpyDeviceToDevice);
    tfl_slab exact = {Z, 0, 0, Z, 1, 0, 2, 0};
    const int rc = tfl_simulate_step_slab(ctx, &prm, &g.st, &exact, NULL, ws2, ns);
    tfl_synchronize(ctx);
    (void)hipMemcpy(Ug, g.U.data, 3 * N * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(Ue, Ucopy, 3 * N * 4, hipMemcpyDeviceToHost);
    printf("check_reach = 2 on a 2.5-cells-per-step flow: rc %d (TFL_EREACH = %d), needed reach %d, state %s\n", rc, (int)TFL_EREACH,
           (int)tfl_slab_needed_reach(ctx), memcmp(Ug, Ue, 3 * N * 4) == 0 ? "untouched" : "MODIFIED");
    if (rc != TFL_EREACH || tfl_slab_needed_reach(ctx) != 3 || memcmp(Ug, Ue, 3 * N * 4) != 0) { printf("FAILED\n"); return 15; }
    free(fast); free(Ug); free(Ue);
  }
  tfl_bc_plan_destroy(ctx, g.planU); tfl_bc_plan_destroy(ctx, g.planR); tfl_wall_plan_destroy(ctx, g.wall); tfl_model_destroy(ctx, model);
  tfl_destroy(ctx);
  printf("OK\n");
  return 0;
}
