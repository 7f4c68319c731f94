// Host stand-ins for the cuSPARSE / cuBLAS / THC / Lua-table calls of the reference's PCG solver -- TEST INFRASTRUCTURE ONLY
// (see TH.h in this directory; nothing here is product code, fluidnet_amd/csrc never includes it).
//
// solveLinearSystemPCG exists only as CUDA in the reference (generic/tfluids.cu:864-1759; generic/tfluids.cc raises for
// CPU tensors). ../ref_pcg.cc compiles the reference's OWN host function -- component loop, createReducedSystemIndices,
// setupLaplacian, the CG loop with its termination rule and clampToEpsilon, the mean subtraction, the two copy kernels --
// for the host. What that function calls into and what is NOT in /root/reference is provided here:
//
//   * cuSPARSE, legacy (pre-CUDA-11) API as the reference calls it (CUDA toolkit 7.5 / 8.0 era; the toolkit is not pinned
//     anywhere in the reference tree -- torch/tfluids/CMakeLists.txt only does FIND_PACKAGE(CUDA 6.5 REQUIRED)):
//       cusparseScsrsv_analysis   -- dependency analysis; numerically a no-op
//       cusparseScsrilu0          -- in-place ILU(0), no pivoting, on the sparsity pattern of A (unit lower L, upper U)
//       cusparseScsric0           -- in-place IC(0) of a symmetric matrix of which the descriptor's fill-mode triangle
//                                    is stored: A ~ R^T R (upper) -- the reference stores the upper triangle only
//       cusparseScsrsv_solve      -- op(T) x = alpha f with T the descriptor's triangle of the stored matrix
//                                    (unit or non-unit diagonal per the descriptor), op = N or T
//       cusparseScsrmv            -- y = alpha op(A) x + beta y; a SYMMETRIC descriptor multiplies by the full
//                                    symmetric matrix although one triangle is stored
//     These are the algorithms the cuSPARSE documentation publishes for those entry points (and the ones NVIDIA's
//     conjugateGradientPrecond sample, which the reference's loop follows, relies on). cuSPARSE's internal summation
//     order (level scheduling) is unspecified; here every row is accumulated sequentially in column order.
//   * cuBLAS: Sdot (fp64 accumulate, one rounding -- cuBLAS' tree order is unspecified), Sscal, Saxpy.
//   * THC tensor calls on host memory, and the Lua table of temporaries (tfluids._tmpPCG): every lookup misses, the
//     tensors created for one call are freed by ref_pcg.cc after it.
#pragma once
#include <math.h>
#include <algorithm>
#include <limits>
#include <vector>

#include "cuda_host.h"

// ---- Lua table of temporaries ------------------------------------------------------------------
#define LUA_TTABLE 5
static thread_local std::vector<THFloatTensor*> shim_tmp_float;
static thread_local std::vector<THIntTensor*> shim_tmp_int;
static thread_local int shim_last_new_kind = 0;   // 1 float, 2 int: what the next luaT_pushudata registers
static inline void luaL_checktype(lua_State*, int, int) {}
static inline void lua_getfield(lua_State*, int, const char*) {}
static inline int lua_isnil(lua_State*, int) { return 1; }
static inline void lua_pop(lua_State*, int) {}
static inline void luaT_pushudata(lua_State*, void* p, const char* type) {
  const std::string t(type);
  if (t.find("Int") != std::string::npos) shim_tmp_int.push_back(reinterpret_cast<THIntTensor*>(p));
  else shim_tmp_float.push_back(reinterpret_cast<THFloatTensor*>(p));
}
static inline void shim_free_temporaries() {
  for (auto* t : shim_tmp_float) shim_free(t);
  for (auto* t : shim_tmp_int) shim_free(t);
  shim_tmp_float.clear(); shim_tmp_int.clear();
}

// ---- TH / THC on host memory -------------------------------------------------------------------
typedef ShimTensor<int> THCudaIntTensor;
static inline THIntTensor* THIntTensor_new() { return shim_new<int>(); }
static inline THCudaTensor* THCudaTensor_new(THCState*) { return shim_new<float>(); }
static inline THCudaIntTensor* THCudaIntTensor_new(THCState*) { return shim_new<int>(); }
static inline void THIntTensor_resize1d(THIntTensor* t, long a) { long s[1] = {a}; shim_resize(t, 1, s); }
static inline void THIntTensor_resize3d(THIntTensor* t, long a, long b, long c) { long s[3] = {a, b, c}; shim_resize(t, 3, s); }
static inline void THFloatTensor_resize5d(THFloatTensor* t, long a, long b, long c, long d, long e) {
  long s[5] = {a, b, c, d, e}; shim_resize(t, 5, s);
}
static inline void THCudaTensor_resize1d(THCState*, THCudaTensor* t, long a) { long s[1] = {a}; shim_resize(t, 1, s); }
static inline void THCudaIntTensor_resize1d(THCState*, THCudaIntTensor* t, long a) { long s[1] = {a}; shim_resize(t, 1, s); }
static inline void THCudaIntTensor_resize3d(THCState*, THCudaIntTensor* t, long a, long b, long c) {
  long s[3] = {a, b, c}; shim_resize(t, 3, s);
}
static inline void THIntTensor_set1d(THIntTensor* t, long i, int v) { t->data[i * t->stride[0]] = v; }
static inline void THFloatTensor_set1d(THFloatTensor* t, long i, float v) { t->data[i * t->stride[0]] = v; }
template <typename T> static inline ShimTensor<T>* shim_narrow0(ShimTensor<T>* t, long first, long size) {
  auto* r = new ShimTensor<T>();            // a borrowed view: shim_free deletes only the header
  r->nDimension = t->nDimension;
  for (int d = 0; d < t->nDimension; d++) { r->size[d] = t->size[d]; r->stride[d] = t->stride[d]; }
  r->size[0] = size;
  r->data = t->data + first * t->stride[0];
  return r;
}
static inline THIntTensor* THIntTensor_newNarrow(THIntTensor* t, int dim, long first, long size) {
  if (dim != 0) throw ShimError("shim: newNarrow only along dim 0");
  return shim_narrow0(t, first, size);
}
static inline THFloatTensor* THFloatTensor_newNarrow(THFloatTensor* t, int dim, long first, long size) {
  if (dim != 0) throw ShimError("shim: newNarrow only along dim 0");
  return shim_narrow0(t, first, size);
}
template <typename T> static inline void shim_copy(ShimTensor<T>* dst, ShimTensor<T>* src) {
  if (shim_numel(dst) != shim_numel(src)) throw ShimError("shim: copy size mismatch");
  if (!shim_contig(dst) || !shim_contig(src)) throw ShimError("shim: copy of a non-contiguous tensor");
  std::memcpy(dst->data, src->data, sizeof(T) * (size_t)shim_numel(src));
}
static inline void THFloatTensor_copyCuda(THCState*, THFloatTensor* d, THCudaTensor* s) { shim_copy(d, s); }
static inline void THCudaTensor_copyFloat(THCState*, THCudaTensor* d, THFloatTensor* s) { shim_copy(d, s); }
static inline void THCudaIntTensor_copyInt(THCState*, THCudaIntTensor* d, THIntTensor* s) { shim_copy(d, s); }
static inline float* THCudaTensor_data(THCState*, THCudaTensor* t) { return t->data; }
static inline int* THCudaIntTensor_data(THCState*, THCudaIntTensor* t) { return t->data; }
// THC's meanall: sum / numel. THC reduces in fp32 in an unspecified tree order; here fp64 accumulate, one rounding.
static inline float THCudaTensor_meanall(THCState*, THCudaTensor* t) {
  const long n = shim_numel(t);
  double s = 0.0;
  for (long i = 0; i < n; i++) s += t->data[i];
  return (float)(s / (double)n);
}
template <typename T, int Dim>
static inline THCDeviceTensor<T, Dim> toDeviceTensor(THCState*, THCudaIntTensor* t) {
  if (t->nDimension != Dim) throw ShimError("toDeviceTensor: dimension mismatch");
  THCDeviceTensor<T, Dim> d;
  d.data_ = t->data;
  for (int i = 0; i < Dim; i++) { d.size_[i] = t->size[i]; d.stride_[i] = t->stride[i]; }
  return d;
}

// ---- cuBLAS ------------------------------------------------------------------------------------
typedef int cublasStatus_t;
typedef int cublasHandle_t;
static const cublasStatus_t CUBLAS_STATUS_SUCCESS = 0;
static cublasHandle_t cublas_handle = 0;
static inline void init_cublas() {}
#define CHECK_CUBLAS(expr) do { if ((expr) != 0) throw ShimError("CUBLAS error"); } while (0)
static inline cublasStatus_t cublasSdot(cublasHandle_t, long n, const float* x, int, const float* y, int, float* r) {
  double acc = 0.0;
  for (long i = 0; i < n; i++) acc += (double)x[i] * (double)y[i];
  *r = (float)acc;
  return 0;
}
static inline cublasStatus_t cublasSscal(cublasHandle_t, long n, const float* a, float* x, int) {
  for (long i = 0; i < n; i++) x[i] = *a * x[i];
  return 0;
}
static inline cublasStatus_t cublasSaxpy(cublasHandle_t, long n, const float* a, const float* x, int, float* y, int) {
  for (long i = 0; i < n; i++) y[i] = *a * x[i] + y[i];
  return 0;
}

// ---- cuSPARSE (legacy API) ---------------------------------------------------------------------
typedef int cusparseStatus_t;
typedef int cusparseHandle_t;
enum cusparseOperation_t { CUSPARSE_OPERATION_NON_TRANSPOSE = 0, CUSPARSE_OPERATION_TRANSPOSE = 1 };
enum cusparseMatrixType_t { CUSPARSE_MATRIX_TYPE_GENERAL = 0, CUSPARSE_MATRIX_TYPE_SYMMETRIC = 1,
                            CUSPARSE_MATRIX_TYPE_HERMITIAN = 2, CUSPARSE_MATRIX_TYPE_TRIANGULAR = 3 };
enum cusparseFillMode_t { CUSPARSE_FILL_MODE_LOWER = 0, CUSPARSE_FILL_MODE_UPPER = 1 };
enum cusparseDiagType_t { CUSPARSE_DIAG_TYPE_NON_UNIT = 0, CUSPARSE_DIAG_TYPE_UNIT = 1 };
enum cusparseIndexBase_t { CUSPARSE_INDEX_BASE_ZERO = 0, CUSPARSE_INDEX_BASE_ONE = 1 };
struct cusparseMatDescr {          // cusparseCreateMatDescr's documented defaults
  cusparseMatrixType_t type = CUSPARSE_MATRIX_TYPE_GENERAL;
  cusparseFillMode_t fill = CUSPARSE_FILL_MODE_LOWER;
  cusparseDiagType_t diag = CUSPARSE_DIAG_TYPE_NON_UNIT;
  cusparseIndexBase_t base = CUSPARSE_INDEX_BASE_ZERO;
};
typedef cusparseMatDescr* cusparseMatDescr_t;
struct cusparseSolveAnalysisInfo {};
typedef cusparseSolveAnalysisInfo* cusparseSolveAnalysisInfo_t;
static cusparseHandle_t cusparse_handle = 0;
static inline void init_cusparse() {}
#define CHECK_CUSPARSE(expr) do { if ((expr) != 0) throw ShimError("CUSPARSE error"); } while (0)

static inline cusparseStatus_t cusparseCreateMatDescr(cusparseMatDescr_t* d) { *d = new cusparseMatDescr(); return 0; }
static inline cusparseStatus_t cusparseDestroyMatDescr(cusparseMatDescr_t d) { delete d; return 0; }   // deleting 0 is fine
static inline cusparseStatus_t cusparseSetMatType(cusparseMatDescr_t d, cusparseMatrixType_t t) { d->type = t; return 0; }
static inline cusparseStatus_t cusparseSetMatFillMode(cusparseMatDescr_t d, cusparseFillMode_t f) { d->fill = f; return 0; }
static inline cusparseStatus_t cusparseSetMatDiagType(cusparseMatDescr_t d, cusparseDiagType_t t) { d->diag = t; return 0; }
static inline cusparseStatus_t cusparseSetMatIndexBase(cusparseMatDescr_t d, cusparseIndexBase_t b) { d->base = b; return 0; }
static inline cusparseStatus_t cusparseCreateSolveAnalysisInfo(cusparseSolveAnalysisInfo_t* i) {
  *i = new cusparseSolveAnalysisInfo(); return 0;
}
static inline cusparseStatus_t cusparseDestroySolveAnalysisInfo(cusparseSolveAnalysisInfo_t i) { delete i; return 0; }
static inline cusparseStatus_t cusparseScsrsv_analysis(cusparseHandle_t, cusparseOperation_t, long, long, cusparseMatDescr_t,
                                                       const float*, const int*, const int*, cusparseSolveAnalysisInfo_t) {
  return 0;
}

static inline int shim_csr_at(const int* row, const int* col, long r, int c) {
  for (int q = row[r]; q < row[r + 1]; q++) if (col[q] == c) return q;
  return -1;
}

// ILU(0), in place, no pivoting: for every row i, for every k < i in its pattern: a_ik /= a_kk, then
// a_ij -= a_ik a_kj for the j > k of row i's pattern that row k holds too.
static inline cusparseStatus_t cusparseScsrilu0(cusparseHandle_t, cusparseOperation_t, long m, cusparseMatDescr_t d,
                                                float* val, const int* row, const int* col, cusparseSolveAnalysisInfo_t) {
  if (d->type != CUSPARSE_MATRIX_TYPE_GENERAL || d->base != CUSPARSE_INDEX_BASE_ZERO) return 1;
  for (long i = 0; i < m; i++)
    for (int q = row[i]; q < row[i + 1]; q++) {
      const int k = col[q];
      if (k >= i) continue;
      const int kk = shim_csr_at(row, col, k, k);
      if (kk < 0 || val[kk] == 0.0f) return 2;       // CUSPARSE_STATUS_ZERO_PIVOT
      val[q] = val[q] / val[kk];
      for (int q2 = row[i]; q2 < row[i + 1]; q2++) {
        if (col[q2] <= k) continue;
        const int kj = shim_csr_at(row, col, k, col[q2]);
        if (kj >= 0) val[q2] -= val[q] * val[kj];
      }
    }
  return 0;
}

// IC(0), in place, of a symmetric matrix whose UPPER triangle is stored: A ~ R^T R, R upper triangular on A's pattern.
// Row k of R: r_kk = sqrt(a_kk), r_kj = a_kj / r_kk; then the trailing update a_jl -= r_kj r_kl (j <= l, within pattern).
static inline cusparseStatus_t cusparseScsric0(cusparseHandle_t, cusparseOperation_t, long m, cusparseMatDescr_t d,
                                               float* val, const int* row, const int* col, cusparseSolveAnalysisInfo_t) {
  if (d->type != CUSPARSE_MATRIX_TYPE_SYMMETRIC || d->fill != CUSPARSE_FILL_MODE_UPPER ||
      d->base != CUSPARSE_INDEX_BASE_ZERO) return 1;   // the only form the reference uses
  for (long k = 0; k < m; k++) {
    const int dq = shim_csr_at(row, col, k, (int)k);
    if (dq < 0 || !(val[dq] > 0.0f)) return 2;
    val[dq] = sqrtf(val[dq]);
    for (int q = row[k]; q < row[k + 1]; q++) if (col[q] > k) val[q] = val[q] / val[dq];
    for (int q = row[k]; q < row[k + 1]; q++) {
      const int j = col[q];
      if (j <= k) continue;
      for (int q2 = row[k]; q2 < row[k + 1]; q2++) {
        const int l = col[q2];
        if (l < j) continue;
        const int jl = shim_csr_at(row, col, j, l);
        if (jl >= 0) val[jl] -= val[q] * val[q2];
      }
    }
  }
  return 0;
}

// op(T) x = alpha f, T = the descriptor's triangle of the stored matrix (entries on the other side are ignored).
static inline cusparseStatus_t cusparseScsrsv_solve(cusparseHandle_t, cusparseOperation_t op, long m, const float* alpha,
                                                    cusparseMatDescr_t d, const float* val, const int* row, const int* col,
                                                    cusparseSolveAnalysisInfo_t, const float* f, float* x) {
  if (d->base != CUSPARSE_INDEX_BASE_ZERO) return 1;
  if (d->type != CUSPARSE_MATRIX_TYPE_GENERAL && d->type != CUSPARSE_MATRIX_TYPE_TRIANGULAR) return 1;
  const bool lower = d->fill == CUSPARSE_FILL_MODE_LOWER, unit = d->diag == CUSPARSE_DIAG_TYPE_UNIT;
  if (op == CUSPARSE_OPERATION_NON_TRANSPOSE) {
    if (lower) {
      for (long i = 0; i < m; i++) {
        float v = *alpha * f[i], dg = 1.0f;
        for (int q = row[i]; q < row[i + 1]; q++) {
          if (col[q] < i) v -= val[q] * x[col[q]];
          else if (col[q] == i) dg = val[q];
        }
        x[i] = unit ? v : v / dg;
      }
    } else {
      for (long i = m - 1; i >= 0; i--) {
        float v = *alpha * f[i], dg = 1.0f;
        for (int q = row[i]; q < row[i + 1]; q++) {
          if (col[q] > i) v -= val[q] * x[col[q]];
          else if (col[q] == i) dg = val[q];
        }
        x[i] = unit ? v : v / dg;
      }
    }
    return 0;
  }
  // transposed: column-oriented substitution over the stored rows
  for (long i = 0; i < m; i++) x[i] = *alpha * f[i];
  if (!lower) {            // T upper => T^T lower: forward
    for (long i = 0; i < m; i++) {
      if (!unit) { const int dq = shim_csr_at(row, col, i, (int)i); if (dq < 0) return 2; x[i] = x[i] / val[dq]; }
      for (int q = row[i]; q < row[i + 1]; q++) if (col[q] > i) x[col[q]] -= val[q] * x[i];
    }
  } else {                 // T lower => T^T upper: backward
    for (long i = m - 1; i >= 0; i--) {
      if (!unit) { const int dq = shim_csr_at(row, col, i, (int)i); if (dq < 0) return 2; x[i] = x[i] / val[dq]; }
      for (int q = row[i]; q < row[i + 1]; q++) if (col[q] < i) x[col[q]] -= val[q] * x[i];
    }
  }
  return 0;
}

// y = alpha op(A) x + beta y. SYMMETRIC: the stored triangle stands for the full symmetric matrix.
static inline cusparseStatus_t cusparseScsrmv(cusparseHandle_t, cusparseOperation_t op, long m, long n, long, const float* alpha,
                                              cusparseMatDescr_t d, const float* val, const int* row, const int* col,
                                              const float* x, const float* beta, float* y) {
  if (d->base != CUSPARSE_INDEX_BASE_ZERO || op != CUSPARSE_OPERATION_NON_TRANSPOSE || m != n) return 1;
  const bool sym = d->type == CUSPARSE_MATRIX_TYPE_SYMMETRIC;
  if (!sym && d->type != CUSPARSE_MATRIX_TYPE_GENERAL) return 1;
  std::vector<float> acc((size_t)m, 0.0f);
  for (long r = 0; r < m; r++)
    for (int q = row[r]; q < row[r + 1]; q++) {
      acc[r] += val[q] * x[col[q]];
      if (sym && col[q] != r) acc[col[q]] += val[q] * x[r];
    }
  for (long r = 0; r < m; r++) y[r] = (*beta == 0.0f) ? *alpha * acc[r] : *alpha * acc[r] + *beta * y[r];
  return 0;
}
