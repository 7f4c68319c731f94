// oracle/_ref PCG trampoline -- TEST INFRASTRUCTURE ONLY; never linked into the product.
//
// solveLinearSystemPCG exists only as CUDA in the reference (generic/tfluids.cu:864-1759; init.lua:674-676 asserts a
// CudaTensor). This translation unit compiles the reference's OWN host code for the host:
//   * the CPU grid classes and findConnectedFluidComponents come from init.cu, compiled where it lies (like ref_wrap.cc);
//   * the Makefile extracts GetKernelIndices (generic/tfluids.cu:182-212) and createReducedSystemIndices ..
//     tfluids_CudaMain_solveLinearSystemPCG (:864-1759: reduced system indices, setupLaplacian, the temp-tensor helpers,
//     clampToEpsilon, the two copy kernels, the component loop and the CG iteration with its termination rule and mean
//     subtraction) into a temporary include (deleted after the compile; nothing is copied into this repo);
//   * what those lines call and /root/reference does not hold -- cuSPARSE's legacy csrilu0 / csric0 / csrsv / csrmv, three
//     cuBLAS level-1 routines, a few THC tensor calls, the Lua table of temporaries -- is ref_shim/cusparse_host.h: the
//     published algorithms of those entry points on host memory (summation orders inside cuSPARSE / cuBLAS are unspecified;
//     the shim's are sequential, fp64 for the dot products).
// So the system assembly, the component handling, the iteration logic and the clamp / termination / mean rules are the
// reference's code; only the five library primitives are restated.
//
// Built by oracle/Makefile (target ref_pcg) into oracle/_ref/libtfluids_ref_pcg.so (git-ignored).
#include "init.cu"                 // reference TU (CPU): tfluids_FloatFlagGrid, findConnectedFluidComponents, Int3
#include "cusparse_host.h"         // cuda_host.h + the cuSPARSE / cuBLAS / THC / Lua-table stand-ins

#include "third_party/grid.cu.h"   // CudaRealGrid, toCudaRealGrid (the reference's device grid classes, on host memory)

// LaunchKernel with explicit sizes (generic/tfluids.cu:92-108): grid (ceil(Z*Y*X / 512), C, B), walked serially.
template <typename TFuncPtr, typename... Args>
static void LaunchKernel(lua_State*, TFuncPtr func, const int bsize, const int csize, const int zsize, const int ysize,
                         const int xsize, Args... args) {
  const long nplane = (long)xsize * ysize * zsize;
  const long tpb = nplane > 512 ? 512 : nplane;
  const long nblk = (nplane + tpb - 1) / tpb;
  for (long bz = 0; bz < bsize; bz++)
    for (long by = 0; by < csize; by++)
      for (long bx = 0; bx < nblk; bx++) {
        gridDim = dim3((unsigned)nblk, (unsigned)csize, (unsigned)bsize);
        blockDim = dim3((unsigned)tpb, 1, 1);
        blockIdx = dim3((unsigned)bx, (unsigned)by, (unsigned)bz);
        for (long t = 0; t < tpb; t++) {
          threadIdx = dim3((unsigned)t, 0, 0);
          func(args...);
        }
      }
}

#define DEV_PTR(tensor) THCudaTensor_data(state, tensor)          // generic/tfluids.cu:861-862
#define DEV_INT_PTR(tensor) THCudaIntTensor_data(state, tensor)

#include TFL_PCG_EXTRACT   // the reference's GetKernelIndices + generic/tfluids.cu:864-1759

extern "C" {

// tfluids.solveLinearSystemPCG(tmpPCG, p, flags, div, is3D, precondType, tol, maxIter, verbose), init.lua:674-676.
// Returns 0 on success, -2 when the reference raised (message in err). p / flags / div: [B][1][Z][Y][X].
int tfluids_ref_pcg(float* p, float* flags, float* div, int B, int Z, int Y, int X, int is3d, const char* precond,
                    float tol, int max_iter, int verbose, double* residual, char* err, int errlen) {
  THCudaTensor t[3];
  float* ptr[3] = {p, flags, div};
  const long sz[5] = {B, 1, Z, Y, X};
  for (int i = 0; i < 3; i++) { shim_set_contig(&t[i], 5, sz); t[i].data = ptr[i]; }
  lua_State L;
  { ShimArg a; L.a.push_back(a); }                                   // 1: the table of temporaries
  for (int i = 0; i < 3; i++) { ShimArg a; a.ptr = &t[i]; L.a.push_back(a); }
  { ShimArg a; a.num = is3d ? 1.0 : 0.0; a.is_bool = true; L.a.push_back(a); }
  { ShimArg a; a.str = precond; L.a.push_back(a); }
  { ShimArg a; a.num = tol; L.a.push_back(a); }
  { ShimArg a; a.num = max_iter; L.a.push_back(a); }
  { ShimArg a; a.num = verbose ? 1.0 : 0.0; a.is_bool = true; L.a.push_back(a); }
  int rc = 0;
  try {
    tfluids_CudaMain_solveLinearSystemPCG(&L);
  } catch (const std::exception& e) {
    if (err && errlen > 0) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
    rc = -2;
  }
  shim_free_temporaries();
  if (rc == 0 && residual && !L.ret.empty()) *residual = L.ret[0];
  return rc;
}

// The five restated library primitives, exported so that tests can hold them to their DEFINING properties with an
// independent implementation (numpy / scipy in tests/test_oracle.py): (L U)_ij = A_ij and (R^T R)_ij = A_ij on A's pattern,
// op(T) x = alpha f, y = A x for a SYMMETRIC descriptor. what: 0 csrilu0 (val in place), 1 csric0 (upper storage, in place),
// 2 csrsv_solve (descr: type 0 general / 3 triangular, fill 0 lower / 1 upper, diag 0 non-unit / 1 unit; op 0 N / 1 T; f -> x),
// 3 csrmv (type 0 general / 1 symmetric; f -> x). Returns the primitive's status (0 = success).
int tfluids_ref_cusparse_primitive(int what, int n, int nz, const int* row, const int* col, float* val, int type, int fill,
                                   int diag, int op, const float* f, float* x) {
  cusparseMatDescr d;
  d.type = (cusparseMatrixType_t)type; d.fill = (cusparseFillMode_t)fill; d.diag = (cusparseDiagType_t)diag;
  const float one = 1.0f, zero = 0.0f;
  switch (what) {
    case 0: return cusparseScsrilu0(0, CUSPARSE_OPERATION_NON_TRANSPOSE, n, &d, val, row, col, nullptr);
    case 1: return cusparseScsric0(0, CUSPARSE_OPERATION_NON_TRANSPOSE, n, &d, val, row, col, nullptr);
    case 2: return cusparseScsrsv_solve(0, (cusparseOperation_t)op, n, &one, &d, val, row, col, nullptr, f, x);
    case 3: return cusparseScsrmv(0, CUSPARSE_OPERATION_NON_TRANSPOSE, n, n, nz, &one, &d, val, row, col, f, &zero, x);
    default: return -1;
  }
}

}  // extern "C"
