// oracle/_ref trampoline -- TEST INFRASTRUCTURE ONLY; never linked into the product.
//
// Compiles the REFERENCE's own CPU tfluids sources where they lie (/root/reference/torch/tfluids,
// passed with -I; nothing is copied into this repo) against the fake TH/luaT shim in ref_shim/,
// and re-exports its Lua-C entry points through one plain C function. The reference registers
// its ops in the table `tfluids_<Real>Main__` (torch/tfluids/generic/tfluids.cc:927-951); we look
// the op up there by name, build the positional "Lua stack" and call it.
//
// Built by oracle/Makefile into oracle/_ref/libtfluids_ref.so (git-ignored).
#include "init.cu"                    // reference TU: instantiates Float and Double ops
#include "generic/advect_type.cc"     // StringToAdvectMethod

#include <memory>

extern "C" {

// kind: 0 = number, 1 = boolean, 2 = string, 3 = tensor (contiguous, ndim <= 5)
struct RefArg {
  int kind;
  double num;
  const char* str;
  void* data;
  int ndim;
  long size[5];
};

// dtype: 0 = float entry points (tfluids_FloatMain_*), 1 = double.
// Returns 0 on success, -1 unknown op, -2 reference raised (message in err).
int tfluids_ref_call(const char* op, int dtype, int nargs, const RefArg* args,
                     double* ret, int* nret, char* err, int errlen) {
  const luaL_Reg* tbl = dtype == 0 ? tfluids_FloatMain__ : tfluids_DoubleMain__;
  lua_CFunction fn = nullptr;
  for (const luaL_Reg* r = tbl; r->name != nullptr; r++) {
    if (std::string(r->name) == op) { fn = r->func; break; }
  }
  if (!fn) return -1;

  lua_State L;
  std::vector<std::unique_ptr<THFloatTensor>> ft;
  std::vector<std::unique_ptr<THDoubleTensor>> dt;
  std::vector<std::unique_ptr<THIntTensor>> it;
  for (int i = 0; i < nargs; i++) {
    ShimArg a;
    const RefArg& r = args[i];
    if (r.kind == 0) { a.num = r.num; }
    else if (r.kind == 1) { a.num = r.num; a.is_bool = true; }
    else if (r.kind == 2) { a.str = r.str; }
    else if (r.kind == 3 || r.kind == 4) {
      if (r.kind == 4) {  // int tensor (normalizePressureMean scratch)
        it.emplace_back(new THIntTensor());
        shim_set_contig(it.back().get(), r.ndim, r.size);
        it.back()->data = (int*)r.data;
        a.ptr = it.back().get();
      } else if (dtype == 0) {
        ft.emplace_back(new THFloatTensor());
        shim_set_contig(ft.back().get(), r.ndim, r.size);
        ft.back()->data = (float*)r.data;
        a.ptr = ft.back().get();
      } else {
        dt.emplace_back(new THDoubleTensor());
        shim_set_contig(dt.back().get(), r.ndim, r.size);
        dt.back()->data = (double*)r.data;
        a.ptr = dt.back().get();
      }
    }
    L.a.push_back(a);
  }
  try {
    fn(&L);
  } catch (const std::exception& e) {
    if (err && errlen > 0) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
    return -2;
  }
  if (nret) *nret = (int)L.ret.size();
  if (ret && !L.ret.empty()) *ret = L.ret[0];
  return 0;
}

// calcLineTrace exposed directly so the known-answer cases of generic/CalcLineTraceTest.m can
// be replayed (the reference does the same for its Matlab mex: generic/CalcLineTrace.cc:46-54).
int tfluids_ref_calcLineTrace(const float* pos, const float* delta, float* flags_data,
                              int zs, int ys, int xs, int is3d, float* new_pos, int* hit,
                              char* err, int errlen) {
  THFloatTensor t;
  long sz[5] = {1, 1, zs, ys, xs};
  shim_set_contig(&t, 5, sz);
  t.data = flags_data;
  try {
    tfluids_FloatFlagGrid flags(&t, is3d != 0);
    tfluids_Floatvec3 p(pos[0], pos[1], pos[2]), d(delta[0], delta[1], delta[2]), np;
    *hit = calcLineTrace(p, d, flags, 0, &np, true) ? 1 : 0;
    new_pos[0] = np.x; new_pos[1] = np.y; new_pos[2] = np.z;
  } catch (const std::exception& e) {
    if (err && errlen > 0) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
    return -2;
  }
  return 0;
}

}  // extern "C"
