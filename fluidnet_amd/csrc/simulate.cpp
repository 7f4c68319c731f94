// simulate.cpp -- tfluids.simulate() (torch/lib/simulate.lua:175-327) as one native call, for hosts that are not
// Python. Pure orchestration of the operators behind include/tfluids_hip.h (nothing here launches a kernel of its
// own except the one-time BC scan): the same sequence, temp layout and launch-saving rules as
// fluidnet_amd/simulate.py, which it is tested against bit for bit (tests/test_hip_simulate.py).
#include "../../include/tfluids_hip.h"
#include "tfl_ctx.hpp"
#include "tfl_host.hpp"

#include <cstddef>
#include <algorithm>
#include <vector>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <thread>
#include <string>

namespace tfl { const unsigned long long* model_range_counter(const tfl_model* m); }      // abi.cpp (library-internal)
namespace tfl {      // abi.cpp (library-internal): both advections' passes A / passes B as one launch each (advect_pair3.hip)
int advect_pair(tfl_ctx* c, float dt, float strength, const tfl_tensor* s, const tfl_tensor* U, const tfl_tensor* flags,
                const tfl_tensor* sfwd, const tfl_tensor* sbounds, const tfl_tensor* sDst, const tfl_tensor* vfwd, const tfl_tensor* UDst,
                const BcFoldArg& fold_s, const BcFoldArg& fold_v);
}

struct tfl_bc_plan {
  tfl_tensor bc, inv;      // the dense pair (pointers kept, not owned)
  int* d_idx = nullptr;    // device list of non-identity elements
  long long n_idx = 0, numel = 0;
  bool sparse = false;     // worth using the list (fewer than a quarter of the elements)
  bool idem = false;       // every listed element has invMask == 0 and |bc| <= 1e6
  bool boxed = false;      // box[] = the listed elements' bounding box in (x, y, z), inclusive (any batch item / channel)
  int box[6] = {0, 0, 0, 0, 0, 0};
  tfl::BcFold* d_fold = nullptr;   // the device descriptor {bc, inv, box} a producing kernel reads (tfl_host.hpp BcFold)
};

namespace {

long long numel_of(const tfl_tensor* t) { return (long long)t->B * t->C * t->Z * t->Y * t->X; }

struct Todo { const tfl_tensor* x; const tfl_bc_plan* plan; };

// setConstVals (simulate.lua:130-160) over the fields in `todo`: sparse plans in one launch, dense ones one each.
int apply_bcs(tfl_ctx* c, const Todo* todo, int n) {
  const tfl_tensor *xs[8], *bs[8], *ms[8];
  const int32_t* is[8];
  int64_t ns[8];
  int k = 0;
  for (int i = 0; i < n; i++) {
    const tfl_bc_plan* p = todo[i].plan;
    if (p->sparse) {
      if (p->n_idx == 0) continue;
      xs[k] = todo[i].x; bs[k] = &p->bc; ms[k] = &p->inv; is[k] = p->d_idx; ns[k] = p->n_idx; k++;
      if (k == 8) { int rc = tfl_applyBCsIndexedMulti(c, k, xs, bs, ms, is, ns); if (rc) return rc; k = 0; }
    } else {
      int rc = tfl_applyBCs(c, todo[i].x, &p->bc, &p->inv, 0, 0.0f, 0.0f);
      if (rc) return rc;
    }
  }
  if (k == 1) return tfl_applyBCsIndexed(c, xs[0], bs[0], ms[0], is[0], ns[0]);
  if (k > 1) return tfl_applyBCsIndexedMulti(c, k, xs, bs, ms, is, ns);
  return TFL_OK;
}

// which fields of the state a setConstVals call may skip: those nothing has written since the previous call AND
// whose pair is idempotent
struct Unchanged { bool p, U, density; };

// density_done: bit i = the pair of density channel i has been applied already (by the kernel that produced the field)
int set_const_vals(tfl_ctx* c, const tfl_sim_state* s, const tfl_tensor* U_now, bool with_U, Unchanged un, unsigned density_done = 0) {
  Todo todo[10];
  int n = 0;
  auto add = [&](const tfl_tensor* x, const tfl_bc_plan* p, bool unchanged) {
    if (!p) return;
    if (unchanged && p->sparse && p->idem) return;
    todo[n].x = x; todo[n].plan = p; n++;
  };
  add(s->p, s->pBC, un.p);
  if (with_U) add(U_now, s->UBC, un.U);
  for (int i = 0; i < s->n_density; i++)
    if (!((density_done >> i) & 1u)) add(s->density[i], s->densityBC[i], un.density);
  return apply_bcs(c, todo, n);
}

// Folding a sparse pair into the kernel that produces the field (tfl_host.hpp BcFold): fold_ask before the operator, fold_took
// after it. TFL_BC_FOLD=0 keeps every pair in its own launch (A/B switch).
bool fold_ask(tfl_ctx* c, const tfl_bc_plan* p) {
  const char* e = getenv("TFL_BC_FOLD");       // (read per call: the A/B parity test switches it inside one process)
  const bool off = e && atoi(e) == 0;
  c->fold_done = false;
  if (off || !p || !p->sparse || !p->boxed || !p->d_fold || p->n_idx == 0) { c->fold = tfl::no_fold(); return false; }
  c->fold = tfl::BcFoldArg{p->d_fold, (unsigned)p->box[2] | ((unsigned)p->box[4] << 16), (unsigned)p->box[3] | ((unsigned)p->box[5] << 16)};
  return true;
}
// the same request as a value (the pair kernels of advect_pair3.hip take two of them as arguments)
tfl::BcFoldArg fold_arg(const tfl_bc_plan* p) {
  const char* e = getenv("TFL_BC_FOLD");
  if ((e && atoi(e) == 0) || !p || !p->sparse || !p->boxed || !p->d_fold || p->n_idx == 0) return tfl::no_fold();
  return tfl::BcFoldArg{p->d_fold, (unsigned)p->box[2] | ((unsigned)p->box[4] << 16), (unsigned)p->box[3] | ((unsigned)p->box[5] << 16)};
}
bool fold_took(tfl_ctx* c) {
  const bool d = c->fold_done;
  c->fold = tfl::no_fold();
  c->fold_done = false;
  return d;
}

struct InStep {              // tfl_ctx::in_step for the length of a step, whatever way it ends
  tfl_ctx* c;
  explicit InStep(tfl_ctx* ctx) : c(ctx) { c->in_step = true; }
  ~InStep() { c->in_step = false; }
};

struct Sizes { long long N, C; int B, Z, Y, X; bool is3d; };
Sizes sizes_of(const tfl_sim_state* s) {
  Sizes z;
  z.B = s->flags->B; z.Z = s->flags->Z; z.Y = s->flags->Y; z.X = s->flags->X;
  z.N = (long long)z.B * z.Z * z.Y * z.X; z.C = s->U->C; z.is3d = s->U->C == 3;
  return z;
}
std::string method_of(const tfl_sim_params* p) { return p->simMethod && p->simMethod[0] ? p->simMethod : "convnet"; }

}  // namespace

extern "C" {

tfl_bc_plan* tfl_bc_plan_create(tfl_ctx* c, const tfl_tensor* bc, const tfl_tensor* invMask) {
  if (!c || !bc || !invMask || !bc->data || !invMask->data) return nullptr;
  if (numel_of(bc) != numel_of(invMask) || numel_of(bc) >= (1ll << 31)) return nullptr;
  tfl_bc_plan* p = new tfl_bc_plan();
  p->bc = *bc; p->inv = *invMask; p->numel = numel_of(bc);
  int* d_cnt = nullptr;
  int h_cnt[2] = {0, 0};
  (void)hipDeviceSynchronize();   // one-time set-up: whoever filled the BC tensors (any stream) is done
  if (hipMalloc((void**)&d_cnt, 2 * sizeof(int)) != hipSuccess) { delete p; return nullptr; }
  (void)hipMemset(d_cnt, 0, 2 * sizeof(int));
  tfl::bc_scan(nullptr, p->numel, bc->data, invMask->data, d_cnt, nullptr);
  (void)hipMemcpy(h_cnt, d_cnt, 2 * sizeof(int), hipMemcpyDeviceToHost);
  p->n_idx = h_cnt[0];
  p->idem = h_cnt[1] == 0;
  p->sparse = p->n_idx * 4 < p->numel;
  if (p->sparse && p->n_idx > 0) {
    if (hipMalloc((void**)&p->d_idx, sizeof(int) * (size_t)p->n_idx) != hipSuccess) { (void)hipFree(d_cnt); delete p; return nullptr; }
    (void)hipMemset(d_cnt, 0, 2 * sizeof(int));
    tfl::bc_scan(nullptr, p->numel, bc->data, invMask->data, d_cnt, p->d_idx);
    (void)hipDeviceSynchronize();
    // the listed elements' bounding box (one-time, on the host): what lets a producing kernel apply the pair itself
    std::vector<int> h((size_t)p->n_idx);
    if (hipMemcpy(h.data(), p->d_idx, sizeof(int) * h.size(), hipMemcpyDeviceToHost) == hipSuccess) {
      int lo[3] = {bc->X, bc->Y, bc->Z}, hi[3] = {-1, -1, -1};
      for (int e : h) {
        const int x = e % bc->X, y = (e / bc->X) % bc->Y, z = (int)((e / ((long long)bc->X * bc->Y)) % bc->Z);
        lo[0] = std::min(lo[0], x); hi[0] = std::max(hi[0], x);
        lo[1] = std::min(lo[1], y); hi[1] = std::max(hi[1], y);
        lo[2] = std::min(lo[2], z); hi[2] = std::max(hi[2], z);
      }
      for (int a = 0; a < 3; a++) { p->box[2 * a] = lo[a]; p->box[2 * a + 1] = hi[a]; }
      p->boxed = true;
      // the device descriptor (vector loads of the pair need 16-byte aligned tensors)
      const tfl::BcFold hf = {bc->data, invMask->data, lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]};
      if ((((uintptr_t)bc->data | (uintptr_t)invMask->data) & 15) == 0 && bc->Y < 65536 && bc->Z < 65536 && hipMalloc((void**)&p->d_fold, sizeof(hf)) == hipSuccess &&
          hipMemcpy(p->d_fold, &hf, sizeof(hf), hipMemcpyHostToDevice) != hipSuccess) { (void)hipFree(p->d_fold); p->d_fold = nullptr; }
    }
  }
  (void)hipFree(d_cnt);
  return p;
}

void tfl_bc_plan_destroy(tfl_ctx* c, tfl_bc_plan* p) {
  (void)c;
  if (!p) return;
  if (p->d_idx) (void)hipFree(p->d_idx);
  if (p->d_fold) (void)hipFree(p->d_fold);
  delete p;
}

int64_t tfl_simulate_workspace_floats(tfl_ctx* c, const tfl_sim_params* prm, const tfl_sim_state* s) {
  if (!c || !prm || !s || !s->flags || !s->U) return 0;
  const Sizes z = sizes_of(s);
  long long need = std::max((3 + 2 * z.C) * z.N, 3 * z.C * z.N);       // advectScalar / advectVel temps (init.lua)
  need = std::max(need, (2 * z.C + 4) * z.N);                           // vorticityConfinement
  const std::string sm = method_of(prm);
  if (sm == "convnet" && s->model) need = std::max<long long>(need, tfl_model_workspace_floats(s->model, z.B, z.Z, z.Y, z.X));
  if (sm == "jacobi") need = std::max(need, 3 * z.N + z.B);             // div + pPrev + pDelta + norms
  if (sm == "pcg") need = std::max<long long>(need, z.N + 2 + tfl_pcg_workspace_floats(z.Z, z.Y, z.X));
  return need + 2;
}

int tfl_simulate_step(tfl_ctx* c, const tfl_sim_params* prm, const tfl_sim_state* s, float* ws, int64_t ws_floats) {
  if (!c || !prm || !s || !s->p || !s->U || !s->flags) return TFL_EINVAL;
  if (s->n_density < 0 || s->n_density > 8) return TFL_EINVAL;
  if (!ws || ((uintptr_t)ws & 15) != 0 || ws_floats < tfl_simulate_workspace_floats(c, prm, s)) return TFL_EINVAL;
  const Sizes z = sizes_of(s);
  const int is3D = z.is3d ? 1 : 0;
  const char* method = (prm->advectionMethod && prm->advectionMethod[0]) ? prm->advectionMethod : "maccormackOurs";
  const bool ours = std::strcmp(method, "maccormackOurs") == 0;
  auto view = [&](float* base, int C) { tfl_tensor t = *s->flags; t.data = base; t.C = C; return t; };
  int rc;
  // an earlier step's projection left the fp16 range of the default 3-D conv path (no sync: the word the projection kernel
  // wrote to pinned memory): refuse to step on from a clamped pressure, before anything of this step has been written
  if (s->model && method_of(prm) == "convnet" && tfl_model_range_flag(c, s->model) > 0) {
    c->err = "simulate_step: an earlier step's ConvNet projection clamped activations at the fp16 range (a blown-up simulation); "
             "read the count with tfl_model_range_errors, or create the model under TFL_CONV_PATH=winograd (strict fp32)";
    return TFL_ERANGE;
  }
  InStep in_step(c);      // the gate above is the step's only one: nothing below refuses half-way through

  // ---- advection (simulate.lua:183-200): every density channel with the pre-advection U, then U ----------------
  unsigned density_done = 0;
  for (int i = 0; i < s->n_density; i++) {
    tfl_tensor fwd = view(ws, 1), bwd = view(ws + z.N, 1), fwdPos = view(ws + 2 * z.N, (int)z.C),
               bwdPos = view(ws + (2 + z.C) * z.N, (int)z.C), out = view(ws + (2 + 2 * z.C) * z.N, 1);
    const tfl_tensor* dst = ours ? s->density[i] : &out;       // maccormackOurs runs in place (no copy back)
    // the first setConstVals (below) follows the advection directly: the kernel that writes the advected field applies
    // the field's pair itself where it can (fold_ask / fold_took; the sparse launch covers the rest)
    fold_ask(c, s->densityBC[i]);
    rc = tfl_advectScalar(c, prm->dt, s->density[i], s->U, s->flags, &fwd, &bwd, is3D, method, &fwdPos, &bwdPos, 1, 0,
                          prm->maccormackStrength, dst);
    if (fold_took(c)) density_done |= 1u << i;
    if (rc) return rc;
    if (!ours) { rc = tfl_copy(c, s->density[i], &out); if (rc) return rc; }
  }
  tfl_tensor vfwd = view(ws, (int)z.C), vbwd = view(ws + z.C * z.N, (int)z.C), Uadv = view(ws + 2 * z.C * z.N, (int)z.C);
  const bool buoyant = s->n_density > 0 && prm->buoyancyScale > 0.0;
  const double dx = tfl_getDx(c, s->flags);
  const float bsc = (float)(-(dx / 4.0) * prm->buoyancyScale);
  const float bg[3] = {prm->gravity[0] * bsc, prm->gravity[1] * bsc, prm->gravity[2] * bsc};
  const bool U_pair_asked = fold_ask(c, s->UBC);
  // addBuoyancy follows advectVel and the first setConstVals directly and needs nothing but the advected density, which is
  // final by now: pass B of advectVel may add the force itself behind the folded U pair (tfl_host.hpp BuoyFold; round 5).
  // Asked for only when the density's own pair has been applied already (or there is none) and the U pair travels with the
  // same kernel (or there is none); TFL_BUOY_FOLD=0 keeps the force in its own launch (A/B switch).
  bool buoy_folded = false;
  {
    const char* e = getenv("TFL_BUOY_FOLD");
    const bool rho_final = buoyant && (!s->densityBC[0] || (density_done & 1u));
    if (rho_final && ours && is3D && !(e && atoi(e) == 0) && (!s->UBC || U_pair_asked)) {
      // strength = -gravity * (dt / dx) in float, exactly as tfl_addBuoyancyFrom forms it (abi.cpp get_dx; tfluids.cc:1190-1192)
      float dxf;
      if (c->dx_dim > 0) dxf = 1.0f / (float)c->dx_dim;
      else if (c->dx_override > 0.0f) dxf = c->dx_override;
      else dxf = 1.0f / (float)std::max(std::max(s->flags->X, s->flags->Y), s->flags->Z);
      const float bs = prm->dt / dxf;
      c->buoy = tfl::BuoyFold{s->density[0]->data, -bg[0] * bs, -bg[1] * bs, -bg[2] * bs};
    }
    c->buoy_done = false;
  }
  rc = tfl_advectVel(c, prm->dt, s->U, s->flags, &vfwd, &vbwd, is3D, method, 1, prm->maccormackStrength, &Uadv);
  const bool Uadv_done = fold_took(c);
  buoy_folded = c->buoy_done;
  c->buoy = tfl::no_buoy(); c->buoy_done = false;
  if (rc) return rc;
  // U:copy(advected) (init.lua:216-218) costs nothing: the velocity stays in the advection's scratch array (`cur`) until an
  // operator that writes every cell of its output delivers it into U -- addBuoyancy (tfl_addBuoyancyFrom), the vorticity
  // confinement (tfl_vorticityConfinementFrom: the fused kernel cannot run in place anyway, the two-launch form reads one
  // array and writes the other since round 5) or the ConvNet projection (UDiv = cur, UOut = U); a copy only where none does.
  const bool vort = prm->vorticityConfinementAmp > 0.0;
  const bool vfused = vort && tfl::vorticity_confinement_fused_ok(is3D != 0, (int)z.Z, (int)z.Y, (int)z.X);
  // scratch of the vorticity operator (curl[3] | cnorm) = the advectVel `fwd` / `bwd` planes, dead after the advection. It must
  // not touch `Uadv` (planes 2C .. 3C of the workspace): the two-launch confinement may still read its velocity from there.
  // 3-D: curl on the `bwd` planes and |curl| behind `Uadv` -- disjoint from the scratch velocity `Utmp` below as well,
  // so that the two-launch confinement is safe whichever array the velocity sits in (the workspace holds (2C + 4) N floats)
  tfl_tensor curl = z.is3d ? view(ws + 3 * z.N, 3) : view(ws, 3), cnorm = z.is3d ? view(ws + 9 * z.N, 1) : view(ws + 3 * z.N, 1);
  tfl_tensor Utmp = view(ws, (int)z.C);                    // the `fwd` planes as a scratch velocity (fused confinement: no curl arrays)
  const tfl_tensor* cur = &Uadv;                           // where the velocity of the step currently lives
  rc = set_const_vals(c, s, cur, !Uadv_done, Unchanged{false, false, false}, density_done);
  if (rc) return rc;

  // ---- forces (simulate.lua:204-239) -------------------------------------------------------------------------
  // The setConstVals that follows the forces (below) touches only U (nothing else has been written since the first one):
  // with the ConvNet projection nothing comes between the last force and it, so the LAST force's kernel applies the U pair
  // itself where it can (fold_ask / fold_took; the other projections run setWallBcs first, outputDiv skips the call).
  const std::string sm = method_of(prm);
  const bool fold_forces = !prm->outputDiv && sm == "convnet";
  const bool gravity_on = prm->gravityScale > 0.0;
  bool U_folded = false;
  if (buoyant && !buoy_folded) {
    // (with the fused confinement next, or the vorticity operator / the ConvNet projection to move it later, the force may
    // stay in scratch: in place on `cur` when nothing needs the velocity in U yet -- but every cell of U must be written by
    // SOMEONE, and addBuoyancyFrom does that for free, so it delivers into U unless the fused confinement reads next)
    const tfl_tensor* dst = vfused ? &Utmp : s->U;
    const bool last = fold_forces && !gravity_on && !vort;
    if (last) fold_ask(c, s->UBC);
    rc = tfl_addBuoyancyFrom(c, cur, dst, s->flags, s->density[0], bg, prm->dt, is3D);
    if (last) U_folded = fold_took(c);
    if (rc) return rc;
    cur = dst;
  }
  if (gravity_on) {
    const float sc = (float)((-dx / 4.0) * prm->gravityScale);
    const float g[3] = {prm->gravity[0] * sc, prm->gravity[1] * sc, prm->gravity[2] * sc};
    rc = tfl_addGravity(c, cur, s->flags, g, prm->dt, is3D, nullptr);
    if (rc) return rc;
  }
  if (vort) {
    const float strength = (float)(dx * prm->vorticityConfinementAmp);
    if (fold_forces) fold_ask(c, s->UBC);
    if (cur != s->U) {
      c->vort_from_two_launch = !vfused;     // below the fused kernel's size the two launches read `cur` and write U
      rc = tfl_vorticityConfinementFrom(c, cur, s->U, s->flags, strength, &curl, &cnorm, is3D);
      c->vort_from_two_launch = false;
    } else {
      tfl_tensor centered = view(ws + 4 * z.N, (int)z.C), force = view(ws + (4 + z.C) * z.N, (int)z.C);   // (unused by the kernels)
      rc = tfl_vorticityConfinement(c, s->U, s->flags, strength, &centered, &curl, &cnorm, &force, is3D);
    }
    if (fold_forces) U_folded = fold_took(c);
    if (rc) return rc;
    cur = s->U;
  }
  // the ConvNet projection reads its velocity from one array and writes the other; everything else wants it in U now
  if (cur != s->U && (prm->outputDiv || sm != "convnet")) {
    rc = tfl_copy(c, s->U, cur);
    if (rc) return rc;
    cur = s->U;
  }
  if (prm->outputDiv) return TFL_OK;

  // ---- projection (simulate.lua:247-304) ---------------------------------------------------------------------
  if (sm != "convnet") {
    rc = tfl_setWallBcsForward(c, s->U, s->flags, is3D);
    if (rc) return rc;
  }
  // only U has been written since the first setConstVals
  rc = set_const_vals(c, s, cur, !U_folded, Unchanged{true, false, true});
  if (rc) return rc;
  const int max_iter = prm->maxIter > 0 ? prm->maxIter : 100;
  if (sm == "convnet") {
    if (!s->model) return TFL_EINVAL;
    // sparse idempotent U BCs go after the projection by index list instead of as two dense fields inside it
    // (round 4: the projection kernel applies them itself on the rows inside their box -- fold_ask -- and the launch goes too)
    const bool late_ubc = s->UBC && s->UBC->sparse && s->UBC->idem;
    const tfl_tensor* ubc = (s->UBC && !late_ubc) ? &s->UBC->bc : nullptr;
    const tfl_tensor* umask = (s->UBC && !late_ubc) ? &s->UBC->inv : nullptr;
    if (late_ubc) fold_ask(c, s->UBC);
    rc = tfl_model_forward(c, s->model, s->p, cur, s->flags, s->p, s->U, ws, ws_floats, ubc, umask, 1, -1e6f, 1e6f);
    const bool folded = fold_took(c);
    if (rc) return rc;
    // p was rewritten by the model, density has not changed since the second setConstVals
    return set_const_vals(c, s, s->U, late_ubc && !folded, Unchanged{false, false, true});
  }
  tfl_tensor div = view(ws, 1);
  rc = tfl_velocityDivergenceForward(c, s->U, s->flags, &div, is3D);
  if (rc) return rc;
  if (sm == "jacobi") {
    tfl_tensor pPrev = view(ws + z.N, 1), pDelta = view(ws + 2 * z.N, 1), norm = view(ws + 3 * z.N, 1);
    norm.B = z.B; norm.Z = norm.Y = norm.X = 1;
    rc = tfl_solveLinearSystemJacobi(c, s->p, s->flags, &div, &pPrev, &pDelta, &norm, is3D, 0.0f, max_iter, 0, nullptr);
  } else if (sm == "pcg") {
    float* pws = ws + ((z.N + 1) & ~1ll);    // 8-byte aligned
    float res = 0.0f;
    // simulate.lua:283 hard-codes 'ic0', and so does this step unless pcgPrecond says otherwise. The lexicographic
    // IC(0) / ILU(0) solves run as pipelined wavefronts (pcg.hip, two launches per application); the preconditioned
    // solve takes ~1.7x the time of the unpreconditioned one at 128^3 on this machine and 0.7x at 256^3 (a third of the
    // iterations, each with two latency-bound sweeps): pcgPrecond = "none" trades that for more iterations within maxIter.
    rc = tfl_solveLinearSystemPCG(c, s->p, s->flags, &div, is3D, prm->pcgPrecond && prm->pcgPrecond[0] ? prm->pcgPrecond : "ic0",
                                  1e-4f, max_iter, 0, pws, ws_floats - (pws - ws), &res);
  } else {
    return TFL_EINVAL;   // mconf.simMethod is not a valid option
  }
  if (rc) return rc;
  rc = tfl_velocityUpdateForward(c, s->U, s->flags, s->p, is3D);
  if (rc) return rc;
  rc = set_const_vals(c, s, s->U, true, Unchanged{false, false, true});
  if (rc) return rc;
  return tfl_applyBCs(c, s->U, nullptr, nullptr, 1, -1e6f, 1e6f);
}


// ================================================================================================================
// z-slab decomposition: tfl_simulate_step on one slab of a grid cut along z (include/tfluids_hip.h, "z-slab" section).
//
// Valid-plane bookkeeping. Write (a, b) for "planes [own_lo - a, own_hi + b)", clipped to the local array (at a domain
// end the slab simply ends there). With R = back-trace reach in cells (max|u_z|*dt < R) the per-phase z-dependencies are
//   3^3 min/max grid of rho  +-1      pass A of MacCormack  rho +-(R+1) incl. the min/max lookup, U +-(R+1)
//   pass B                   fwd +-R, U +-(R+1)             buoyancy  rho -1         curl  U -1..+2
//   confinement              curl -1, |curl| -2..+1         divergence  U +1 (flags +2)
//   3 conv layers            +-1 each                       velocity update  p -1
// so, working backwards from "owned planes exact":
//   project (0,0) <- conv3 (1,0) <- conv2 (2,1) <- conv1 (3,2) <- {p, div} (4,3)            [T1: p, T3: div]
//   divergence (0,0) <- confine (0,1) <- curl (2,2) <- buoyancy/gravity (3,4) <- {U_adv (3,4), rho (4,4)}   [T2]
//   pass B (0,0) <- pass A (R,R) <- min/max (2R,2R) <- {rho (2R+1,2R+1), U (max(R+1,2R), same): trace velocity R+1,
//                                                       advected field 2R}                    [T0: U; rho is still
//                                                                                             valid from T2]
// Local array ends are treated by the kernels as the domain's border shell (zeros); with halo >= max(4, 2R+1) no window
// above ever evaluates a tap there except through data that is exchanged instead (div, p).
namespace {

struct Halo { const tfl_tensor* t; int below, above; };   // planes of `t` refreshed below / above the owned range

struct Msg {                 // one neighbour exchange: buffers inside the workspace
  int tag, n;
  Halo f[8];
  float *send_lo, *recv_lo, *send_hi, *recv_hi;
  long long n_send_lo, n_recv_lo, n_send_hi, n_recv_hi;   // floats
};

struct SlabGeom {
  int Zl, o0, o1, R, H;
  bool lower, upper;
  long long yx, N;           // Y*X, B*Zl*Y*X
  int B;
};

int slab_geom(tfl_ctx* c, const tfl_sim_state* s, const tfl_slab* sl, SlabGeom* g) {
  if (!s || !s->flags || !s->U || !sl) return TFL_EINVAL;
  g->Zl = s->flags->Z; g->B = s->flags->B;
  g->o0 = sl->own_lo; g->o1 = sl->own_hi;
  g->R = sl->reach > 0 ? sl->reach : 1;
  g->H = tfl_slab_halo(g->R);
  g->yx = (long long)s->flags->Y * s->flags->X;
  g->N = (long long)g->B * g->Zl * g->yx;
  g->lower = sl->z_first + sl->own_lo > 0;
  g->upper = sl->z_first + sl->own_hi < sl->z_total;
  const bool ok = g->o0 >= 0 && g->o1 > g->o0 && g->o1 <= g->Zl && sl->z_first >= 0 && sl->z_first + g->Zl <= sl->z_total &&
                  (g->lower ? g->o0 >= g->H : (g->o0 == 0 && sl->z_first == 0)) &&
                  (g->upper ? g->Zl - g->o1 >= g->H : (g->o1 == g->Zl && sl->z_first + g->Zl == sl->z_total)) &&
                  ((!g->lower && !g->upper) || g->o1 - g->o0 >= g->H);
  if (!ok) { if (c) c->err = "simulate_step_slab: inconsistent slab description (halo too thin, or owned range thinner than the halo)"; return TFL_EINVAL; }
  return TFL_OK;
}

long long msg_floats(const SlabGeom& g, const Halo* f, int n, bool to_lower, bool send) {
  // to lower neighbour: I send what it needs ABOVE its range (my lowest `above` planes) and receive my `below` planes
  long long planes = 0;
  for (int i = 0; i < n; i++) {
    const int cnt = (to_lower == send) ? f[i].above : f[i].below;
    planes += (long long)f[i].t->B * f[i].t->C * cnt;
  }
  return planes * g.yx;
}

// carve the four buffers of each message out of `base`; returns the floats used
long long msg_layout(const SlabGeom& g, Msg* m, int count, float* base) {
  long long off = 0;
  auto take = [&](long long n) { float* p = base ? base + off : nullptr; off += (n + 3) & ~3ll; return p; };
  for (int i = 0; i < count; i++) {
    Msg& q = m[i];
    q.n_send_lo = g.lower ? msg_floats(g, q.f, q.n, true, true) : 0;
    q.n_recv_lo = g.lower ? msg_floats(g, q.f, q.n, true, false) : 0;
    q.n_send_hi = g.upper ? msg_floats(g, q.f, q.n, false, true) : 0;
    q.n_recv_hi = g.upper ? msg_floats(g, q.f, q.n, false, false) : 0;
    q.send_lo = take(q.n_send_lo); q.recv_lo = take(q.n_recv_lo); q.send_hi = take(q.n_send_hi); q.recv_hi = take(q.n_recv_hi);
  }
  return off;
}

// the planes of one message that leave (unpack = false: owned planes -> send buffers) or arrive (unpack = true: receive
// buffers -> halo planes), BOTH neighbours in one launch
void pack_msg(tfl_ctx* c, const SlabGeom& g, const Msg& q, bool unpack) {
  float* ptrs[8]; float* bufs[8]; int rows[8], zlo[8], np[8];
  int n = 0;
  for (int side = 0; side < 2; side++) {
    const bool lower = side == 0;
    if (lower ? !g.lower : !g.upper) continue;
    float* buf = unpack ? (lower ? q.recv_lo : q.recv_hi) : (lower ? q.send_lo : q.send_hi);
    for (int i = 0; i < q.n; i++) {
      const Halo& h = q.f[i];
      ptrs[n] = h.t->data; rows[n] = h.t->B * h.t->C;
      if (!unpack) { np[n] = lower ? h.above : h.below; zlo[n] = lower ? g.o0 : g.o1 - h.below; }     // owned planes out
      else { np[n] = lower ? h.below : h.above; zlo[n] = lower ? g.o0 - h.below : g.o1; }              // halo planes in
      bufs[n] = buf;
      buf += (long long)rows[n] * np[n] * g.yx;
      n++;
    }
  }
  if (n) tfl::pack_planes(c->stream, n, ptrs, rows, zlo, np, g.yx * g.Zl, g.yx, nullptr, unpack ? 1 : 0, bufs);
}

// the same planes as contiguous runs of the fields themselves, one per (batch item, channel): what exchange_start_v moves
int msg_chunks(const SlabGeom& g, const Msg& q, bool lower, bool send, tfl_comm_chunk* out) {
  if (lower ? !g.lower : !g.upper) return 0;
  int n = 0;
  for (int i = 0; i < q.n; i++) {
    const Halo& h = q.f[i];
    int np, zlo;
    if (send) { np = lower ? h.above : h.below; zlo = lower ? g.o0 : g.o1 - h.below; }      // owned planes out
    else { np = lower ? h.below : h.above; zlo = lower ? g.o0 - h.below : g.o1; }           // halo planes in
    for (int row = 0; row < h.t->B * h.t->C; row++) {
      out[n].ptr = h.t->data + ((long long)row * g.Zl + zlo) * g.yx;
      out[n].n = (long long)np * g.yx;
      n++;
    }
  }
  return n;
}
constexpr int kMaxChunks = 64;     // per neighbour and message: (B * C) of every field of the message

// the optional in-place transport, read only when the host's struct is large enough to hold it (tfl_comm::size)
static inline decltype(tfl_comm::exchange_start_v) start_v_of(const tfl_comm* comm) {
  return comm->size >= (int32_t)(offsetof(tfl_comm, exchange_start_v) + sizeof(comm->exchange_start_v)) ? comm->exchange_start_v : nullptr;
}

int msg_start(tfl_ctx* c, const SlabGeom& g, const tfl_comm* comm, const Msg& q) {
  if ((!g.lower && !g.upper) || q.n == 0) return TFL_OK;
  if (auto start_v = start_v_of(comm)) {
    int rows = 0;
    for (int i = 0; i < q.n; i++) rows += q.f[i].t->B * q.f[i].t->C;
    if (q.n > 0 && rows <= kMaxChunks) {
      tfl_comm_chunk slo[kMaxChunks], rlo[kMaxChunks], shi[kMaxChunks], rhi[kMaxChunks];
      const int n_lo = msg_chunks(g, q, true, true, slo), n_hi = msg_chunks(g, q, false, true, shi);
      msg_chunks(g, q, true, false, rlo); msg_chunks(g, q, false, false, rhi);
      if (start_v(comm->user, q.tag, n_lo, slo, rlo, n_hi, shi, rhi) != 0) { c->err = "simulate_step_slab: comm callback failed (exchange_start_v)"; return TFL_EINVAL; }
      return TFL_OK;
    }
  }
  pack_msg(c, g, q, false);
  if (comm->exchange_start(comm->user, q.tag, q.send_lo, q.n_send_lo, q.recv_lo, q.n_recv_lo, q.send_hi, q.n_send_hi,
                           q.recv_hi, q.n_recv_hi) != 0) { c->err = "simulate_step_slab: comm callback failed (exchange_start)"; return TFL_EINVAL; }
  return TFL_OK;
}
int msg_finish(tfl_ctx* c, const SlabGeom& g, const tfl_comm* comm, const Msg& q) {
  if ((!g.lower && !g.upper) || q.n == 0) return TFL_OK;
  if (comm->exchange_wait(comm->user, q.tag) != 0) { c->err = "simulate_step_slab: comm callback failed (exchange_wait)"; return TFL_EINVAL; }
  int rows = 0;
  for (int i = 0; i < q.n; i++) rows += q.f[i].t->B * q.f[i].t->C;
  if (start_v_of(comm) && q.n > 0 && rows <= kMaxChunks) return TFL_OK;      // delivered in place
  pack_msg(c, g, q, true);
  return TFL_OK;
}

// the three messages of a step (slot 1 is empty); `div` / `Uadv` may be null tensors when only the layout of T0 is wanted
void slab_messages(const SlabGeom& g, const tfl_sim_state* s, const tfl_tensor* Uadv, const tfl_tensor* div, Msg m[4]) {
  const int rr = 2 * g.R + 1;
  // U feeds pass A twice: as the trace velocity (window (R,R) reads it out to +-(R+1)) and as the ADVECTED field of the
  // velocity's self-advection, sampled at positions up to R cells away from the window: +-2R. The projection only
  // rewrites the owned planes, so everything out to max(R+1, 2R) must be refreshed (R = 1: 2 planes either way).
  const int ur = std::max(g.R + 1, 2 * g.R);
  // U and p leave together at the end of a step (ONE message, one packing launch) and are consumed at the start of the
  // next one: nothing writes either field's halo planes in between (p is only read by the first conv layer)
  m[0].tag = 0; m[0].n = 2; m[0].f[0] = Halo{s->U, ur, ur}; m[0].f[1] = Halo{s->p, 4, 3};
  m[1].tag = 1; m[1].n = 0;
  m[2].tag = 2; m[2].n = s->n_density > 0 ? 2 : 1; m[2].f[0] = Halo{Uadv, 3, 4};
  if (s->n_density > 0) m[2].f[1] = Halo{s->density[0], rr > 4 ? rr : 4, rr > 4 ? rr : 4};
  m[3].tag = 3; m[3].n = 1; m[3].f[0] = Halo{div, 4, 3};
}

constexpr int kReachPrimed = 0x100;  // tfl_slab::in_flight, beside the message bits: this run has reset the context's sticky reach word
constexpr int kReachFlags = 8;       // check_reach = 2 resolves reaches 1 .. 8 (a 16-plane slab can hold the halo of 7)
struct SlabWs { float* msg; double* stats; float* compute; long long compute_floats; };

// workspace = [message buffers][stats: 2*B doubles][compute region]
long long slab_ws(const SlabGeom& g, const tfl_sim_state* s, float* ws, SlabWs* out) {
  tfl_tensor u3 = *s->U, s1 = *s->flags;
  Msg m[4];
  slab_messages(g, s, &u3, &s1, m);
  long long off = msg_layout(g, m, 4, nullptr);
  off = (off + 3) & ~3ll;
  const long long stats_off = off;
  off += 4ll * g.B + 2ll * (kReachFlags + 1);         // 2*B doubles + the reach / range flags of check_reach = 2 (behind the stats)
  off = (off + 3) & ~3ll;
  long long model = s->model ? tfl_model_workspace_floats(s->model, g.B, g.Zl, s->flags->Y, s->flags->X) : 0;
  const long long comp = std::max<long long>(13 * g.N, model) + 4;
  if (out) { out->msg = ws; out->stats = ws ? (double*)(ws + stats_off) : nullptr; out->compute = ws ? ws + off : nullptr; out->compute_floats = comp; }
  return off + comp;
}

struct Win { int a, b; };
Win ext(const SlabGeom& g, int below, int above) {
  Win w;
  static const int widen = tfl::exp_env("TFL_SLAB_WIDEN") ? atoi(tfl::exp_env("TFL_SLAB_WIDEN")) : 0;   // development aid
  below += widen; above += widen; w.a = g.o0 - below < 0 ? 0 : g.o0 - below; w.b = g.o1 + above > g.Zl ? g.Zl : g.o1 + above; return w;
}
int set_win(tfl_ctx* c, Win w) { return tfl_set_z_window(c, w.a, w.b, 0, 0); }
#define WIN(call) do { int wrc_ = (call); if (wrc_) return wrc_; } while (0)

// boundary strips of the owned range (k planes next to each existing neighbour) and what is left between them
struct Split { int a0, a1, b0, b1, i0, i1; bool has_strips, has_interior; };
Split split_owned(const SlabGeom& g, int k) {
  Split sp;
  sp.a0 = g.o0; sp.a1 = g.lower ? std::min(g.o0 + k, g.o1) : g.o0;
  sp.b1 = g.o1; sp.b0 = g.upper ? std::max(g.o1 - k, sp.a1) : g.o1;
  sp.i0 = sp.a1; sp.i1 = sp.b0;
  sp.has_strips = sp.a1 > sp.a0 || sp.b1 > sp.b0;
  sp.has_interior = sp.i1 > sp.i0;
  return sp;
}

struct StageGuard {          // whatever happens, leave the context with no window / stage mask / dx override
  tfl_ctx* c;
  explicit StageGuard(tfl_ctx* ctx) : c(ctx) {}
  ~StageGuard() { (void)tfl_set_z_window(c, 0, 0, 0, 0); (void)tfl_set_stages(c, 0); c->dx_dim = 0; (void)tfl_set_z_origin(c, 0, 0); c->reach_sink = false; }
};

// the pinned publication count has reached `target` (wrap-safe); spins without an API call, gives the core away after a while,
// and gives up after 30 s (a dead device must not hang the host here: the next HIP call reports it)
void reach_wait(tfl_ctx* c, unsigned target) {
  volatile unsigned* tick = reinterpret_cast<volatile unsigned*>(c->h_reach) + 1;
  if ((int)(*tick - target) >= 0) return;
  const auto t0 = std::chrono::steady_clock::now();
  for (unsigned long i = 0; (int)(*tick - target) < 0; i++) {
    if (i < 4096) { __builtin_ia32_pause(); continue; }
    std::this_thread::yield();
    if ((i & 1023) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::seconds(30)) return;
  }
}
// check_reach = 1, host side: has a step that the device has certainly started (two calls back) -- or any later one it has got
// to since -- found max|u_z| dt >= R? Waits only when the host is more than two steps ahead of the device.
bool reach_violated(tfl_ctx* c, float dt, int R, char* msg, size_t msg_len) {
  if (c->reach_hist[0]) reach_wait(c, c->reach_hist[0]);
  const float v = *(volatile float*)c->h_reach;
  if (!(v * dt >= (float)R)) return false;
  snprintf(msg, msg_len, "simulate_step_slab: max|u_z|*dt = %.3f cells reached the slab's back-trace reach %d (found up to three steps after the fact: "
                         "check_reach = 2 refuses such a step BEFORE it runs)", v * dt, R);
  // acknowledged: the sticky maximum starts again (the state is past saving; a host that carries on gets the next report afresh)
  c->h_reach[0] = 0.0f;
  (void)hipMemsetAsync(c->d_reach, 0, sizeof(float), c->stream);
  return true;
}
// ... device side: the sticky word and the publication count reach the host through the step's LAST kernel (k_project writes
// them into the mapped pinned mirror: tfl_ctx::reach_sink) -- not through hipMemcpyAsync (a 4-byte D2H copy on the stream makes
// the HOST wait until the stream has drained on this stack, tools/ubench/host_costs.hip) and not behind an event (4 us of
// device time per hipEventRecord): the host counts the publishing launches it has enqueued (tfl_model_finish) and keeps the
// count as it stood at the end of the last two steps
void reach_mark(tfl_ctx* c) {      // at the end of a step whose projection published (eager, or behind the launch of a recorded step)
  c->reach_hist[0] = c->reach_hist[1]; c->reach_hist[1] = c->reach_issued;
}
void reach_quiesce(tfl_ctx* c) {   // every publication enqueued so far has landed
  if (c->reach_issued) reach_wait(c, c->reach_issued);
}

}  // namespace

int32_t tfl_slab_halo(int32_t reach) {
  const int r = reach > 0 ? reach : 1;
  return 2 * r + 1 > 4 ? 2 * r + 1 : 4;
}

int64_t tfl_simulate_slab_workspace_floats(tfl_ctx* c, const tfl_sim_params* prm, const tfl_sim_state* s, const tfl_slab* sl) {
  (void)prm;
  SlabGeom g;
  if (slab_geom(c, s, sl, &g) != TFL_OK) return 0;
  return slab_ws(g, s, nullptr, nullptr);
}

int tfl_slab_drain(tfl_ctx* c, const tfl_sim_state* s, tfl_slab* sl, const tfl_comm* comm, float* ws, int64_t ws_floats) {
  if (!c || !s || !sl || !comm || !ws) return TFL_EINVAL;
  SlabGeom g;
  int rc = slab_geom(c, s, sl, &g);
  if (rc) return rc;
  if (ws_floats < slab_ws(g, s, nullptr, nullptr)) return TFL_EINVAL;
  Msg m[4];
  tfl_tensor u3 = *s->U, s1 = *s->flags;
  slab_messages(g, s, &u3, &s1, m);
  msg_layout(g, m, 4, ws);
  for (int t = 0; t < 2; t++)
    if (sl->in_flight & (1 << t)) { rc = msg_finish(c, g, comm, m[t]); if (rc) return rc; sl->in_flight &= ~(1 << t); }
  return TFL_OK;
}

int tfl_simulate_step_slab(tfl_ctx* c, const tfl_sim_params* prm, const tfl_sim_state* s, tfl_slab* sl, const tfl_comm* comm,
                           float* ws, int64_t ws_floats) {
  if (!c || !prm || !s || !s->p || !s->U || !s->flags || !sl) return TFL_EINVAL;
  auto bad = [&](const char* m) { c->err = std::string("simulate_step_slab: ") + m; return TFL_EUNSUPPORTED; };
  SlabGeom g;
  int rc = slab_geom(c, s, sl, &g);
  if (rc) return rc;
  const bool multi = g.lower || g.upper;
  if (multi && (!comm || comm->size < (int32_t)(offsetof(tfl_comm, allreduce_sum) + sizeof(comm->allreduce_sum)) ||
                !comm->exchange_start || !comm->exchange_wait || !comm->allreduce_sum)) {
    c->err = "simulate_step_slab: tfl_comm is null, has no size (ABI 3: size = sizeof(tfl_comm)) or lacks a required callback";
    return TFL_EINVAL;
  }
  const Sizes z = sizes_of(s);
  if (!z.is3d) return bad("2-D grids have no z to cut (run replicas)");
  const char* method = (prm->advectionMethod && prm->advectionMethod[0]) ? prm->advectionMethod : "maccormackOurs";
  if (std::strcmp(method, "maccormackOurs") != 0) return bad("only advectionMethod maccormackOurs");
  if (method_of(prm) != "convnet" || !s->model) return bad("only the ConvNet projection");
  if (s->n_density < 0 || s->n_density > 1) return bad("at most one density channel");
  if (!ws || ((uintptr_t)ws & 15) != 0 || ws_floats < slab_ws(g, s, nullptr, nullptr)) { c->err = "simulate_step_slab: workspace too small or misaligned"; return TFL_EINVAL; }
  SlabWs W;
  slab_ws(g, s, ws, &W);
  StageGuard guard(c);
  const int is3D = 1;
  const long long N = g.N;
  float* cw = W.compute;
  auto view = [&](float* base, int C) { tfl_tensor t = *s->flags; t.data = base; t.C = C; return t; };
  // compute region: scalar temps [fwd N][fwdPos 3N][bwdPos 3N] | velocity temps [vfwd 3N][Uadv 3N]; the vorticity temps
  // and the model workspace re-use the region once the advection is done (Uadv lives until addBuoyancy)
  tfl_tensor fwd = view(cw, 1), fwdPos = view(cw + N, 3), bwdPos = view(cw + 4 * N, 3);
  tfl_tensor vfwd = view(cw + 7 * N, 3), Uadv = view(cw + 10 * N, 3);
  tfl_tensor curl = view(cw, 3), cnorm = view(cw + 3 * N, 1);
  tfl_tensor div = view(tfl_model_div(s->model, g.B, g.Zl, s->flags->Y, s->flags->X, cw), 1);
  Msg m[4];
  slab_messages(g, s, &Uadv, &div, m);
  msg_layout(g, m, 4, W.msg);

  // ---- reach check of the PREVIOUS step's velocity (no host sync: the word was copied back behind that step) ---------
  if (sl->check_reach == 1 && !c->capturing) {  // (a captured step: tfl_slab_graph_step makes both checks before it launches)
    char buf[256];
    if (reach_violated(c, prm->dt, g.R, buf, sizeof(buf))) {
      // the neighbours' matching receives of the U / p messages are already posted: finish ours before giving up
      for (int t = 0; t < 2; t++)
        if (multi && (sl->in_flight & (1 << t))) { (void)msg_finish(c, g, comm, m[t]); sl->in_flight &= ~(1 << t); }
      c->err = buf;
      return TFL_EINVAL;
    }
  }
  // The fp16 range gate of tfl_simulate_step -- only where refusing cannot strand a neighbour: a slab WITHOUT
  // neighbours, or (below) collectively under check_reach = 2. A rank that refused on its own word would leave the others
  // waiting for its halos; the host of a cut run polls tfl_model_range_flag / tfl_model_range_errors and stops every rank.
  if (!c->capturing && !multi && s->model && tfl_model_range_flag(c, s->model) > 0) {
    for (int t = 0; t < 2; t++)
      if (multi && (sl->in_flight & (1 << t))) { (void)msg_finish(c, g, comm, m[t]); sl->in_flight &= ~(1 << t); }
    c->err = "simulate_step_slab: an earlier step's ConvNet projection clamped activations at the fp16 range (a blown-up simulation); "
             "read the count with tfl_model_range_errors, or create the model under TFL_CONV_PATH=winograd (strict fp32)";
    return TFL_ERANGE;
  }
  InStep in_step(c);      // the range gate above is the step's only one
  if (sl->in_flight & 1) { rc = msg_finish(c, g, comm, m[0]); if (rc) return rc; sl->in_flight &= ~1; }     // U and p halos
  if (sl->check_reach == 1 && !(sl->in_flight & kReachPrimed)) {
    // the first step of a run on this context (the host initialised in_flight to 0): the sticky maximum of whatever ran on the
    // context before must not speak for this run
    c->h_reach[0] = 0.0f;
    (void)hipMemsetAsync(c->d_reach, 0, sizeof(float), c->stream);
    sl->in_flight |= kReachPrimed;
    c->reach_folded = false;               // the state this run starts from has been through no projection of ours
  }
  // mode 1, from the second step on: the projection kernel of the step before has folded max |u_z| of the planes it wrote -- this
  // rank's OWNED planes; the halo planes are their owners' to report -- into the sticky word (k_project_v4, reach_acc): no launch
  // of our own (round 6: k_absmax was 5-6 us of a 0.1 ms rank-step). Mode 2 needs THIS step's exact maximum and keeps it.
  const bool reach_known = sl->check_reach == 1 && c->reach_folded;
  c->reach_folded = false;
  if (sl->check_reach && !reach_known) {
    for (int b = 0; b < g.B; b++)                                                                            // u_z of every batch item
      tfl::absmax(c->stream, (long long)g.Zl * g.yx, s->U->data + (3ll * b + 2) * g.Zl * g.yx, c->d_reach,
                  sl->check_reach == 2 && b == 0);      // (mode 1: a sticky maximum over the steps, see tfl_ctx.hpp)
  }
  if (sl->check_reach == 2) {
    // "exact" (round 6): the reach THIS step needs, agreed by all ranks, before anything is written. One-hot flags
    // (max|u_z| dt >= r, r = 1 .. 8) so that the transport's SUM all-reduce can combine them; one host synchronisation.
    if (c->capturing) { c->err = "simulate_step_slab: check_reach = 2 synchronises with the host and cannot be recorded into a graph"; return TFL_EUNSUPPORTED; }
    if (!c->h_reach_flags && hipHostMalloc((void**)&c->h_reach_flags, sizeof(double) * (kReachFlags + 1), hipHostMallocDefault) != hipSuccess) {
      c->h_reach_flags = nullptr; c->err = "simulate_step_slab: hipHostMalloc failed"; return TFL_EHIP;
    }
    double* d_flags = W.stats + 2 * g.B;
    tfl::reach_flags(c->stream, c->d_reach, prm->dt, kReachFlags, d_flags, tfl::model_range_counter(s->model));
    if (multi && comm->allreduce_sum(comm->user, d_flags, kReachFlags + 1) != 0) { c->err = "simulate_step_slab: comm callback failed (allreduce_sum, reach)"; return TFL_EINVAL; }
    if (hipMemcpyAsync(c->h_reach_flags, d_flags, sizeof(double) * (kReachFlags + 1), hipMemcpyDeviceToHost, c->stream) != hipSuccess ||
        hipStreamSynchronize(c->stream) != hipSuccess) { c->err = "simulate_step_slab: reading the reach flags failed"; (void)hipGetLastError(); return TFL_EHIP; }
    if (c->h_reach_flags[kReachFlags] > 0.0) {      // some rank's conv stack left the fp16 range: EVERY rank refuses this step
      c->err = "simulate_step_slab: an earlier step's ConvNet projection clamped activations at the fp16 range on some rank (a blown-up "
               "simulation); read the counts with tfl_model_range_errors, or create the model under TFL_CONV_PATH=winograd (strict fp32)";
      return TFL_ERANGE;
    }
    int need = 1;
    for (int r = 0; r < kReachFlags; r++) if (c->h_reach_flags[r] > 0.0) need = r + 2;
    if (need > g.R) {
      char buf[200];
      snprintf(buf, sizeof(buf), "simulate_step_slab: this step's back-traces reach %d cells along z%s, the slab is laid out for %d; nothing has been written",
               need, need > kReachFlags ? " or more" : "", g.R);
      c->err = buf; c->needed_reach = need;
      return TFL_EREACH;
    }
  } else if (sl->check_reach) {
    c->reach_sink = true;                  // the projection kernel of THIS step publishes the word (cleared by the guard below)
  }
  c->dx_dim = std::max(std::max(s->flags->X, s->flags->Y), sl->z_total);   // tfluids.getDx of the WHOLE grid
  (void)tfl_set_z_origin(c, sl->z_first, sl->z_total);
  const bool buoyant = s->n_density > 0 && prm->buoyancyScale > 0.0;
  const tfl_tensor* rho = s->n_density > 0 ? s->density[0] : nullptr;

  // ---- advection -------------------------------------------------------------------------------------------------
  auto adv_scalar = [&]() { return tfl_advectScalar(c, prm->dt, rho, s->U, s->flags, &fwd, &fwd, is3D, method, &fwdPos, &bwdPos, 1, 0,
                                                    prm->maccormackStrength, rho); };
  auto adv_vel = [&]() { return tfl_advectVel(c, prm->dt, s->U, s->flags, &vfwd, &vfwd, is3D, method, 1, prm->maccormackStrength, &Uadv); };
  // pass B (the launches that write the advected fields) applies the pairs of the setConstVals that follows on the planes
  // it writes -- the owned ones, which is what message T2 carries to the neighbours' halos (fold_ask / fold_took)
  bool rho_done = true, Uadv_done = true;
  auto adv_scalar_b = [&]() { fold_ask(c, s->densityBC[0]); const int r = adv_scalar(); rho_done = fold_took(c) && rho_done; return r; };
  auto adv_vel_b = [&]() { fold_ask(c, s->UBC); const int r = adv_vel(); Uadv_done = fold_took(c) && Uadv_done; return r; };
  // (Round 6 measured the scalar advection on a second stream beside the velocity's -- they are independent, simulate.lua:183-200
  // -- and dropped it: an event hop between two streams costs 12-15 us of GPU-side latency on this stack
  // (tools/ubench/host_costs.hip), more than the overlap of two 8 us kernels returns; profiles/r06_slab_host_cost.txt.)
  // Round 6: the two advections' passes as PAIRS -- passes A in one launch, passes B in one launch (advect_pair3.hip; they are
  // independent here: the slab step does not fold the buoyancy force into pass B). false once = the shape is not the pair
  // kernels': the four launches as before.
  bool paired = rho != nullptr;
  auto adv_pair = [&](bool with_folds) {
    const tfl::BcFoldArg fs = with_folds ? fold_arg(s->densityBC[0]) : tfl::no_fold(), fv = with_folds ? fold_arg(s->UBC) : tfl::no_fold();
    const int r = tfl::advect_pair(c, prm->dt, prm->maccormackStrength, rho, s->U, s->flags, &fwd, &fwdPos, rho, &vfwd, &Uadv, fs, fv);
    if (r == TFL_OK && with_folds) { rho_done = rho_done && fs.dev != nullptr; Uadv_done = Uadv_done && fv.dev != nullptr; }
    return r;
  };
  if (rho && !paired) {
    (void)tfl_set_stages(c, 1); WIN(set_win(c, ext(g, 2 * g.R, 2 * g.R)));
    rc = adv_scalar(); if (rc) return rc;
  }
  (void)tfl_set_stages(c, 2); WIN(set_win(c, ext(g, g.R, g.R)));
  if (paired) {
    rc = adv_pair(false);
    if (rc == TFL_EUNSUPPORTED) {
      paired = false;
      (void)tfl_set_stages(c, 1); WIN(set_win(c, ext(g, 2 * g.R, 2 * g.R)));
      rc = adv_scalar(); if (rc) return rc;
      (void)tfl_set_stages(c, 2); WIN(set_win(c, ext(g, g.R, g.R)));
    } else if (rc) return rc;
  }
  if (!paired) {
    if (rho) { rc = adv_scalar(); if (rc) return rc; }
    rc = adv_vel(); if (rc) return rc;
  }
  (void)tfl_set_stages(c, 4);
  const int strip = 4 > 2 * g.R + 1 ? 4 : 2 * g.R + 1;        // deepest plane count message T2 sends
  const Split spB = split_owned(g, strip);
  const bool ovl = sl->overlap && multi;
  auto pass_b = [&]() -> int {      // the passes B of both operators on the current window
    if (paired) { const int r = adv_pair(true); if (r != TFL_EUNSUPPORTED) return r; paired = false; }
    if (rho) { const int r = adv_scalar_b(); if (r) return r; }
    return adv_vel_b();
  };
  if (ovl && spB.has_strips && spB.has_interior) {
    WIN(tfl_set_z_window(c, spB.a0, spB.a1, spB.b0, spB.b1));
    rc = pass_b(); if (rc) return rc;
    rc = msg_start(c, g, comm, m[2]); if (rc) return rc;
    WIN(tfl_set_z_window(c, spB.i0, spB.i1, 0, 0));
    rc = pass_b(); if (rc) return rc;
  } else {
    WIN(set_win(c, ext(g, 0, 0)));
    rc = pass_b(); if (rc) return rc;
    rc = msg_start(c, g, comm, m[2]); if (rc) return rc;
  }
  rc = msg_finish(c, g, comm, m[2]); if (rc) return rc;
  (void)tfl_set_stages(c, 0); (void)tfl_set_z_window(c, 0, 0, 0, 0);
  // as tfl_simulate_step: the fused vorticity confinement delivers into U, so buoyancy / gravity work on a scratch velocity
  const bool vort = prm->vorticityConfinementAmp > 0.0;
  const bool vfused = vort && tfl::vorticity_confinement_fused_ok(true, g.Zl, s->flags->Y, s->flags->X);
  tfl_tensor Utmp = view(cw + 4 * N, 3);                   // the scalar advection's bwdPos planes, dead by now
  const tfl_tensor* cur = &Uadv;
  if (!buoyant && !vfused) { rc = tfl_copy(c, s->U, &Uadv); if (rc) return rc; cur = s->U; }
  rc = set_const_vals(c, s, cur, !Uadv_done, Unchanged{false, false, false}, (rho && rho_done) ? 1u : 0u);
  if (rc) return rc;

  // ---- forces --------------------------------------------------------------------------------------------------------
  const double dx = tfl_getDx(c, s->flags);
  WIN(set_win(c, ext(g, 3, 4)));
  // as tfl_simulate_step: the last force's kernel applies the U pair of the setConstVals that follows, on the planes it writes
  // (they include every plane the projection reads: (0, 1)); the launch below then skips U
  const bool fold_forces = !prm->outputDiv;
  const bool gravity_on = prm->gravityScale > 0.0;
  bool U_folded = false;
  if (buoyant) {
    const float sc = (float)(-(dx / 4.0) * prm->buoyancyScale);
    const float gv[3] = {prm->gravity[0] * sc, prm->gravity[1] * sc, prm->gravity[2] * sc};
    const tfl_tensor* dst = vfused ? &Utmp : s->U;
    const bool last = fold_forces && !gravity_on && !vort;
    if (last) fold_ask(c, s->UBC);
    rc = tfl_addBuoyancyFrom(c, cur, dst, s->flags, rho, gv, prm->dt, is3D);
    if (last) U_folded = fold_took(c);
    if (rc) return rc;
    cur = dst;
  }
  if (gravity_on) {
    const float sc = (float)((-dx / 4.0) * prm->gravityScale);
    const float gv[3] = {prm->gravity[0] * sc, prm->gravity[1] * sc, prm->gravity[2] * sc};
    rc = tfl_addGravity(c, cur, s->flags, gv, prm->dt, is3D, nullptr); if (rc) return rc;
  }
  if (vort) {
    const float strength = (float)(dx * prm->vorticityConfinementAmp);
    if (vfused) {
      // one launch: curl (2,2) and |curl| live in LDS, the planes (0,1) of U are written (inputs: planes (3,3) of `cur`)
      (void)tfl_set_stages(c, 0); WIN(set_win(c, ext(g, 0, 1)));
      if (fold_forces) fold_ask(c, s->UBC);
      rc = tfl_vorticityConfinementFrom(c, cur, s->U, s->flags, strength, &curl, &cnorm, is3D);
      if (fold_forces) U_folded = fold_took(c);
      if (rc) return rc;
    } else {
      (void)tfl_set_stages(c, 2); WIN(set_win(c, ext(g, 2, 2)));
      rc = tfl_vorticityConfinement(c, s->U, s->flags, strength, &curl, &curl, &cnorm, &curl, is3D); if (rc) return rc;
      (void)tfl_set_stages(c, 4); WIN(set_win(c, ext(g, 0, 1)));
      if (fold_forces) fold_ask(c, s->UBC);
      rc = tfl_vorticityConfinement(c, s->U, s->flags, strength, &curl, &curl, &cnorm, &curl, is3D);
      if (fold_forces) U_folded = fold_took(c);
      if (rc) return rc;
    }
  }
  (void)tfl_set_stages(c, 0); (void)tfl_set_z_window(c, 0, 0, 0, 0);
  if (prm->outputDiv) return TFL_OK;
  rc = set_const_vals(c, s, s->U, !U_folded, Unchanged{true, false, true});
  if (rc) return rc;

  // ---- projection ----------------------------------------------------------------------------------------------------
  const long long mws = ws_floats - (cw - ws);
  (void)tfl_set_stages(c, 2);
  const Split spD = split_owned(g, 4);
  if (ovl && spD.has_strips && spD.has_interior) {
    WIN(tfl_set_z_window(c, spD.a0, spD.a1, spD.b0, spD.b1));
    rc = tfl_model_begin(c, s->model, s->U, s->flags, s->U, cw, mws, g.o0, g.o1, W.stats); if (rc) return rc;
    rc = msg_start(c, g, comm, m[3]); if (rc) return rc;
    WIN(tfl_set_z_window(c, spD.i0, spD.i1, 0, 0));
    rc = tfl_model_begin(c, s->model, s->U, s->flags, s->U, cw, mws, g.o0, g.o1, W.stats); if (rc) return rc;
  } else {
    WIN(set_win(c, ext(g, 0, 0)));
    rc = tfl_model_begin(c, s->model, s->U, s->flags, s->U, cw, mws, g.o0, g.o1, W.stats); if (rc) return rc;
    rc = msg_start(c, g, comm, m[3]); if (rc) return rc;
  }
  (void)tfl_set_stages(c, 4);
  rc = tfl_model_begin(c, s->model, s->U, s->flags, s->U, cw, mws, g.o0, g.o1, W.stats); if (rc) return rc;
  if (multi && comm->allreduce_sum(comm->user, W.stats, 2ll * g.B) != 0) { c->err = "simulate_step_slab: comm callback failed (allreduce_sum)"; return TFL_EINVAL; }
  const double count = 3.0 * (double)sl->z_total * (double)g.yx;
  const bool late_ubc = s->UBC && s->UBC->sparse && s->UBC->idem;
  const tfl_tensor* ubc = (s->UBC && !late_ubc) ? &s->UBC->bc : nullptr;
  const tfl_tensor* umask = (s->UBC && !late_ubc) ? &s->UBC->inv : nullptr;
  auto finish = [&]() { return tfl_model_finish(c, s->model, s->p, s->flags, s->p, s->U, cw, mws, W.stats, count, ubc, umask, 1, -1e6f, 1e6f); };
  (void)tfl_set_stages(c, 1);
  const Win w1 = ext(g, 3, 2);
  if (ovl && g.o1 - 1 > g.o0 + 1) {
    // interior of conv 1 needs only owned planes of div / p: it runs while the div halos travel
    WIN(tfl_set_z_window(c, g.lower ? g.o0 + 1 : w1.a, g.upper ? g.o1 - 1 : w1.b, 0, 0));
    rc = finish(); if (rc) return rc;
    rc = msg_finish(c, g, comm, m[3]); if (rc) return rc;
    WIN(tfl_set_z_window(c, g.lower ? w1.a : 0, g.lower ? g.o0 + 1 : 0, g.upper ? g.o1 - 1 : 0, g.upper ? w1.b : 0));
    rc = finish(); if (rc) return rc;
  } else {
    rc = msg_finish(c, g, comm, m[3]); if (rc) return rc;
    WIN(set_win(c, w1));
    rc = finish(); if (rc) return rc;
  }
  (void)tfl_set_stages(c, 2); WIN(set_win(c, ext(g, 2, 1))); rc = finish(); if (rc) return rc;
  (void)tfl_set_stages(c, 4); WIN(set_win(c, ext(g, 1, 0))); rc = finish(); if (rc) return rc;
  (void)tfl_set_stages(c, 8); WIN(set_win(c, ext(g, 0, 0)));
  if (late_ubc) fold_ask(c, s->UBC);          // the projection kernel applies the sparse U pair itself (owned planes: what T0 sends)
  rc = finish();
  const bool U_late_folded = fold_took(c);
  if (rc) return rc;
  (void)tfl_set_stages(c, 0); (void)tfl_set_z_window(c, 0, 0, 0, 0);
  rc = set_const_vals(c, s, s->U, late_ubc && !U_late_folded, Unchanged{false, false, true});
  if (rc) return rc;
  if (sl->check_reach == 1 && !c->capturing) reach_mark(c);      // behind the kernel that published this step's reach word
  // the next step's U and p halos leave now; they are consumed at its start / before its first conv layer
  if (multi) {
    rc = msg_start(c, g, comm, m[0]); if (rc) return rc;
    if (c->capturing) {
      // a captured step must end with every stream joined: the message is finished here instead of by the next call (nothing
      // of the next step could have run beside it anyway -- its first kernels read these halos)
      rc = msg_finish(c, g, comm, m[0]); if (rc) return rc;
    } else {
      sl->in_flight |= 1;
    }
  }
  return TFL_OK;
}

int32_t tfl_slab_needed_reach(const tfl_ctx* c) { return c ? c->needed_reach : 0; }

namespace {
// geometry + message of a stand-alone halo exchange (tfl_slab_exchange): every field has the slab's local depth
int exchange_setup(const tfl_ctx* c, int n, const tfl_tensor* const* fields, const int32_t* below, const int32_t* above, const tfl_slab* sl,
                   SlabGeom* g, Msg* q) {
  (void)c;
  if (n < 1 || n > 4 || !fields || !below || !above || !sl || !fields[0]) return TFL_EINVAL;     // (4 fields x 2 sides = the 8 plane runs of one packing launch)
  g->Zl = fields[0]->Z; g->B = fields[0]->B;
  g->o0 = sl->own_lo; g->o1 = sl->own_hi; g->R = sl->reach > 0 ? sl->reach : 1; g->H = 0;
  g->yx = (long long)fields[0]->Y * fields[0]->X;
  g->N = (long long)g->B * g->Zl * g->yx;
  g->lower = sl->z_first + sl->own_lo > 0;
  g->upper = sl->z_first + sl->own_hi < sl->z_total;
  if (g->o0 < 0 || g->o1 <= g->o0 || g->o1 > g->Zl) return TFL_EINVAL;
  q->tag = 4; q->n = n;
  for (int i = 0; i < n; i++) {
    const tfl_tensor* t = fields[i];
    if (!t || !t->data || t->Z != g->Zl || (long long)t->Y * t->X != g->yx || below[i] < 0 || above[i] < 0) return TFL_EINVAL;
    // a neighbour sends out of its OWNED planes: no deeper than the thinnest slab (the caller keeps own >= halo, as slab_geom demands)
    if ((g->lower && (below[i] > g->o0 || above[i] > g->o1 - g->o0)) || (g->upper && (above[i] > g->Zl - g->o1 || below[i] > g->o1 - g->o0))) return TFL_EINVAL;
    q->f[i] = Halo{t, below[i], above[i]};
  }
  return TFL_OK;
}
}  // namespace

int64_t tfl_slab_exchange_floats(int n, const tfl_tensor* const* fields, const int32_t* below, const int32_t* above, const tfl_slab* sl) {
  SlabGeom g; Msg q;
  if (exchange_setup(nullptr, n, fields, below, above, sl, &g, &q) != TFL_OK) return 0;
  return msg_layout(g, &q, 1, nullptr) + 4;
}

int tfl_slab_exchange(tfl_ctx* c, int n, const tfl_tensor* const* fields, const int32_t* below, const int32_t* above, const tfl_slab* sl,
                      const tfl_comm* comm, float* scratch, int64_t scratch_floats) {
  if (!c) return TFL_EINVAL;
  SlabGeom g; Msg q;
  if (exchange_setup(c, n, fields, below, above, sl, &g, &q) != TFL_OK) { c->err = "slab_exchange: bad fields / plane counts"; return TFL_EINVAL; }
  if (!g.lower && !g.upper) return TFL_OK;
  if (!comm || comm->size < (int32_t)(offsetof(tfl_comm, allreduce_sum) + sizeof(comm->allreduce_sum)) || !comm->exchange_start || !comm->exchange_wait) {
    c->err = "slab_exchange: tfl_comm is null or lacks a required callback"; return TFL_EINVAL;
  }
  if (!scratch || ((uintptr_t)scratch & 15) != 0 || scratch_floats < msg_layout(g, &q, 1, nullptr)) { c->err = "slab_exchange: scratch too small or misaligned"; return TFL_EINVAL; }
  msg_layout(g, &q, 1, scratch);
  // always staged (pack -> send | receive -> unpack): the in-place form needs the transport's chunk lists, and this is a one-off
  pack_msg(c, g, q, false);
  if (comm->exchange_start(comm->user, q.tag, q.send_lo, q.n_send_lo, q.recv_lo, q.n_recv_lo, q.send_hi, q.n_send_hi, q.recv_hi, q.n_recv_hi) != 0) {
    c->err = "slab_exchange: comm callback failed (exchange_start)"; return TFL_EINVAL;
  }
  if (comm->exchange_wait(comm->user, q.tag) != 0) { c->err = "slab_exchange: comm callback failed (exchange_wait)"; return TFL_EINVAL; }
  pack_msg(c, g, q, true);
  return TFL_OK;
}

// ---- the rank-step as ONE host call: a HIP graph of tfl_simulate_step_slab (round 6; include/tfluids_hip.h) -------------
struct tfl_slab_graph {
  hipGraphExec_t exec = nullptr;
  hipStream_t cap = nullptr;          // the stream the step was recorded on (kept: RCCL's captured work refers to it)
  const tfl_sim_params* prm = nullptr;
  const tfl_sim_state* s = nullptr;
  tfl_slab* sl = nullptr;
  int reach = 1;
  bool multi = false;                 // the slab has neighbours
  unsigned reach_pubs = 0;            // reach publications one replay makes (counted while recording)
  size_t nodes = 0;
};

tfl_slab_graph* tfl_slab_graph_create(tfl_ctx* c, const tfl_sim_params* prm, const tfl_sim_state* s, tfl_slab* sl, const tfl_comm* comm,
                                      float* ws, int64_t ws_floats) {
  if (!c || !prm || !s || !sl) return nullptr;
  SlabGeom g;
  if (slab_geom(c, s, sl, &g) != TFL_OK) return nullptr;
  const bool multi = g.lower || g.upper;
  if (multi) {
    const bool can = comm && comm->size >= (int32_t)(offsetof(tfl_comm, capturable) + sizeof(comm->capturable)) && comm->capturable != 0;
    if (!can) { c->err = "slab_graph_create: this transport's calls are not stream operations (tfl_comm.capturable = 0): step eagerly"; return nullptr; }
  }
  if (sl->check_reach == 2) { c->err = "slab_graph_create: check_reach = 2 synchronises with the host every step: step eagerly"; return nullptr; }
  // the U / p message the last eager step left in flight is consumed now: a captured step starts and ends with none
  if ((sl->in_flight & 0xF) && tfl_slab_drain(c, s, sl, comm, ws, ws_floats) != TFL_OK) return nullptr;
  if (sl->check_reach == 1 && !(sl->in_flight & kReachPrimed)) {
    // (a host that records before its first eager step: the reset of the sticky reach word must not become a node of the graph)
    c->h_reach[0] = 0.0f;
    (void)hipMemsetAsync(c->d_reach, 0, sizeof(float), c->stream);
    sl->in_flight |= kReachPrimed;
  }
  tfl_slab_graph* G = new tfl_slab_graph();
  G->prm = prm; G->s = s; G->sl = sl; G->reach = g.R; G->multi = multi;
  if (hipStreamCreateWithFlags(&G->cap, hipStreamNonBlocking) != hipSuccess) { c->err = "slab_graph_create: hipStreamCreate failed"; delete G; return nullptr; }
  hipStream_t user = c->stream;
  // everything queued so far on the caller's stream happens before the recording stream is used at all (warm-up steps)
  (void)hipStreamSynchronize(user);
  reach_quiesce(c);
  hipGraph_t graph = nullptr;
  // relaxed: the transport's library may call into the runtime from its own threads while we record
  if (hipStreamBeginCapture(G->cap, hipStreamCaptureModeRelaxed) != hipSuccess) {
    c->err = "slab_graph_create: hipStreamBeginCapture failed"; (void)hipStreamDestroy(G->cap); delete G; return nullptr;
  }
  c->stream = G->cap; c->capturing = true;
  const unsigned issued0 = c->reach_issued;
  const int rc = tfl_simulate_step_slab(c, prm, s, sl, comm, ws, ws_floats);
  G->reach_pubs = c->reach_issued - issued0; c->reach_issued = issued0;      // recorded, not run: every REPLAY makes them
  c->capturing = false; c->stream = user;
  const hipError_t ec = hipStreamEndCapture(G->cap, &graph);
  std::string why;
  if (rc != TFL_OK) why = "the step failed while being recorded: " + c->err;
  else if (ec != hipSuccess || !graph) why = std::string("hipStreamEndCapture: ") + hipGetErrorString(ec);
  else if (hipGraphInstantiate(&G->exec, graph, nullptr, nullptr, 0) != hipSuccess) why = "hipGraphInstantiate failed";
  if (graph) { (void)hipGraphGetNodes(graph, nullptr, &G->nodes); (void)hipGraphDestroy(graph); }
  (void)hipGetLastError();
  if (!why.empty()) {
    c->err = "slab_graph_create: " + why;
    if (G->exec) (void)hipGraphExecDestroy(G->exec);
    (void)hipStreamDestroy(G->cap);
    delete G;
    return nullptr;
  }
  return G;
}

int64_t tfl_slab_graph_nodes(const tfl_slab_graph* G) { return G ? (int64_t)G->nodes : 0; }

int tfl_slab_graph_step(tfl_ctx* c, tfl_slab_graph* G) {
  if (!c || !G || !G->exec) return TFL_EINVAL;
  // the two gates of the eager call, read from the words the PREVIOUS step left in pinned memory
  if (G->sl->check_reach) {
    char buf[256];
    if (reach_violated(c, G->prm->dt, G->reach, buf, sizeof(buf))) { c->err = buf; return TFL_EINVAL; }
  }
  if (!G->multi && G->s->model && tfl_model_range_flag(c, G->s->model) > 0) {     // (a slab with neighbours never refuses on its own word: see the eager step)
    c->err = "simulate_step_slab: an earlier step's ConvNet projection clamped activations at the fp16 range (a blown-up simulation); "
             "read the count with tfl_model_range_errors, or create the model under TFL_CONV_PATH=winograd (strict fp32)";
    return TFL_ERANGE;
  }
  if (hipGraphLaunch(G->exec, c->stream) != hipSuccess) { c->err = "slab_graph_step: hipGraphLaunch failed"; (void)hipGetLastError(); return TFL_EHIP; }
  if (G->sl->check_reach) { c->reach_issued += G->reach_pubs; reach_mark(c); }
  return TFL_OK;
}

void tfl_slab_graph_destroy(tfl_ctx* c, tfl_slab_graph* G) {
  (void)c;
  if (!G) return;
  if (G->exec) (void)hipGraphExecDestroy(G->exec);
  if (G->cap) (void)hipStreamDestroy(G->cap);
  delete G;
}

}  // extern "C"
