// oracle/_ref Jacobi trampoline -- TEST INFRASTRUCTURE ONLY; never linked into the product.
//
// solveLinearSystemJacobi exists only as CUDA in the reference (generic/tfluids.cu:1765-1927). This
// translation unit compiles the reference's OWN kernel body and host loop for the host: the Makefile
// extracts GetKernelIndices (generic/tfluids.cu:182-212), kernel_jacobiIteration (:1765-1821) and
// tfluids_CudaMain_solveLinearSystemJacobi (:1823-1927) from the file where it lies into a temporary
// include (deleted after the compile; nothing is copied into this repo), and ref_shim/cuda_host.h
// supplies threadIdx/blockIdx, THCDeviceTensor and the handful of THC calls on host memory. The grid
// classes come from the reference's third_party/grid.cu.h, included in place.
//
// Built by oracle/Makefile (target ref_jacobi) into oracle/_ref/libtfluids_ref_jacobi.so (git-ignored).
#include "cuda_host.h"

#include "third_party/cell_type.h"
#include "third_party/grid.cu.h"

// LaunchKernel (generic/tfluids.cu:92-130) walks the same launch geometry serially per "block":
// grid (ceil(Z*Y*X / tpb), C, B), block min(tpb, Z*Y*X); the blocks are spread over OpenMP threads
// (threadIdx/blockIdx are thread-local).
template <typename TFuncPtr, typename... Args>
static void LaunchKernel(lua_State*, TFuncPtr func, const CudaGridBase& domain, Args... args) {
  const long nplane = domain.xsize() * domain.ysize() * domain.zsize();
  const long tpb = nplane > 512 ? 512 : nplane;
  const long nblk = (nplane + tpb - 1) / tpb;
  const long csize = domain.nchan(), bsize = domain.nbatch();
#pragma omp parallel for collapse(3) schedule(static)
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

#include TFL_JACOBI_EXTRACT   // the reference's GetKernelIndices, kernel_jacobiIteration and host loop

extern "C" {

// tfluids.solveLinearSystemJacobi(p, flags, div, pPrev, pDelta, pDeltaNorm, is3D, pTol, maxIter, verbose),
// init.lua:726-727. Returns 0 on success, -2 when the reference raised (message in err).
int tfluids_ref_jacobi(float* p, float* flags, float* div, float* p_prev, float* p_delta, float* p_delta_norm,
                       int B, int Z, int Y, int X, int is3d, float p_tol, int max_iter, double* residual,
                       char* err, int errlen) {
  THCudaTensor t[6];
  float* ptr[5] = {p, flags, div, p_prev, p_delta};
  const long sz[5] = {B, 1, Z, Y, X};
  for (int i = 0; i < 5; i++) { shim_set_contig(&t[i], 5, sz); t[i].data = ptr[i]; }
  const long nb[1] = {B};
  shim_set_contig(&t[5], 1, nb); t[5].data = p_delta_norm;
  lua_State L;
  for (int i = 0; i < 6; i++) { ShimArg a; a.ptr = &t[i]; L.a.push_back(a); }
  { ShimArg a; a.num = is3d ? 1.0 : 0.0; a.is_bool = true; L.a.push_back(a); }
  { ShimArg a; a.num = p_tol; L.a.push_back(a); }
  { ShimArg a; a.num = max_iter; L.a.push_back(a); }
  { ShimArg a; a.num = 0.0; a.is_bool = true; L.a.push_back(a); }
  try {
    tfluids_CudaMain_solveLinearSystemJacobi(&L);
  } catch (const std::exception& e) {
    if (err && errlen > 0) { strncpy(err, e.what(), errlen - 1); err[errlen - 1] = 0; }
    return -2;
  }
  if (residual && !L.ret.empty()) *residual = L.ret[0];
  return 0;
}

}  // extern "C"
