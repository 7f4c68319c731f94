// simulate.cpp -- tfluids.simulate() (torch/lib/simulate.lua:175-327) as one native call, for hosts that are not
// Python. Pure orchestration of the operators behind include/tfluids_hip.h (nothing here launches a kernel of its
// own except the one-time BC scan): the same sequence, temp layout and launch-saving rules as
// fluidnet_amd/simulate.py, which it is tested against bit for bit (tests/test_hip_simulate.py).
#include "../../include/tfluids_hip.h"
#include "tfl_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>

struct tfl_bc_plan {
  tfl_tensor bc, inv;      // the dense pair (pointers kept, not owned)
  int* d_idx = nullptr;    // device list of non-identity elements
  long long n_idx = 0, numel = 0;
  bool sparse = false;     // worth using the list (fewer than a quarter of the elements)
  bool idem = false;       // every listed element has invMask == 0 and |bc| <= 1e6
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

int set_const_vals(tfl_ctx* c, const tfl_sim_state* s, const tfl_tensor* U_now, bool with_U, Unchanged un) {
  Todo todo[10];
  int n = 0;
  auto add = [&](const tfl_tensor* x, const tfl_bc_plan* p, bool unchanged) {
    if (!p) return;
    if (unchanged && p->sparse && p->idem) return;
    todo[n].x = x; todo[n].plan = p; n++;
  };
  add(s->p, s->pBC, un.p);
  if (with_U) add(U_now, s->UBC, un.U);
  for (int i = 0; i < s->n_density; i++) add(s->density[i], s->densityBC[i], un.density);
  return apply_bcs(c, todo, n);
}

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
  }
  (void)hipFree(d_cnt);
  return p;
}

void tfl_bc_plan_destroy(tfl_ctx* c, tfl_bc_plan* p) {
  (void)c;
  if (!p) return;
  if (p->d_idx) (void)hipFree(p->d_idx);
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

  // ---- advection (simulate.lua:183-200): every density channel with the pre-advection U, then U ----------------
  for (int i = 0; i < s->n_density; i++) {
    tfl_tensor fwd = view(ws, 1), bwd = view(ws + z.N, 1), fwdPos = view(ws + 2 * z.N, (int)z.C),
               bwdPos = view(ws + (2 + z.C) * z.N, (int)z.C), out = view(ws + (2 + 2 * z.C) * z.N, 1);
    const tfl_tensor* dst = ours ? s->density[i] : &out;       // maccormackOurs runs in place (no copy back)
    rc = tfl_advectScalar(c, prm->dt, s->density[i], s->U, s->flags, &fwd, &bwd, is3D, method, &fwdPos, &bwdPos, 1, 0,
                          prm->maccormackStrength, dst);
    if (rc) return rc;
    if (!ours) { rc = tfl_copy(c, s->density[i], &out); if (rc) return rc; }
  }
  tfl_tensor vfwd = view(ws, (int)z.C), vbwd = view(ws + z.C * z.N, (int)z.C), Uadv = view(ws + 2 * z.C * z.N, (int)z.C);
  rc = tfl_advectVel(c, prm->dt, s->U, s->flags, &vfwd, &vbwd, is3D, method, 1, prm->maccormackStrength, &Uadv);
  if (rc) return rc;
  // U:copy(advected) (init.lua:216-218) is folded into addBuoyancy when buoyancy is on
  const bool buoyant = s->n_density > 0 && prm->buoyancyScale > 0.0;
  if (!buoyant) {
    rc = tfl_copy(c, s->U, &Uadv);
    if (rc) return rc;
  }
  rc = set_const_vals(c, s, buoyant ? &Uadv : s->U, true, Unchanged{false, false, false});
  if (rc) return rc;

  // ---- forces (simulate.lua:204-239) -------------------------------------------------------------------------
  const double dx = tfl_getDx(c, s->flags);
  if (buoyant) {
    const float sc = (float)(-(dx / 4.0) * prm->buoyancyScale);
    const float g[3] = {prm->gravity[0] * sc, prm->gravity[1] * sc, prm->gravity[2] * sc};
    rc = tfl_addBuoyancyFrom(c, &Uadv, s->U, s->flags, s->density[0], g, prm->dt, is3D);
    if (rc) return rc;
  }
  if (prm->gravityScale > 0.0) {
    const float sc = (float)((-dx / 4.0) * prm->gravityScale);
    const float g[3] = {prm->gravity[0] * sc, prm->gravity[1] * sc, prm->gravity[2] * sc};
    rc = tfl_addGravity(c, s->U, s->flags, g, prm->dt, is3D, nullptr);
    if (rc) return rc;
  }
  if (prm->vorticityConfinementAmp > 0.0) {
    tfl_tensor centered = view(ws, (int)z.C), curl = view(ws + z.C * z.N, 3), cnorm = view(ws + (z.C + 3) * z.N, 1),
               force = view(ws + (z.C + 4) * z.N, (int)z.C);
    rc = tfl_vorticityConfinement(c, s->U, s->flags, (float)(dx * prm->vorticityConfinementAmp), &centered, &curl,
                                  &cnorm, &force, is3D);
    if (rc) return rc;
  }
  if (prm->outputDiv) return TFL_OK;

  // ---- projection (simulate.lua:247-304) ---------------------------------------------------------------------
  const std::string sm = method_of(prm);
  if (sm != "convnet") {
    rc = tfl_setWallBcsForward(c, s->U, s->flags, is3D);
    if (rc) return rc;
  }
  // only U has been written since the first setConstVals
  rc = set_const_vals(c, s, s->U, true, Unchanged{true, false, true});
  if (rc) return rc;
  const int max_iter = prm->maxIter > 0 ? prm->maxIter : 100;
  if (sm == "convnet") {
    if (!s->model) return TFL_EINVAL;
    // sparse idempotent U BCs go after the projection by index list instead of as two dense fields inside it
    const bool late_ubc = s->UBC && s->UBC->sparse && s->UBC->idem;
    const tfl_tensor* ubc = (s->UBC && !late_ubc) ? &s->UBC->bc : nullptr;
    const tfl_tensor* umask = (s->UBC && !late_ubc) ? &s->UBC->inv : nullptr;
    rc = tfl_model_forward(c, s->model, s->p, s->U, s->flags, s->p, s->U, ws, ws_floats, ubc, umask, 1, -1e6f, 1e6f);
    if (rc) return rc;
    // p was rewritten by the model, density has not changed since the second setConstVals
    return set_const_vals(c, s, s->U, late_ubc, Unchanged{false, false, true});
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

}  // extern "C"
