// Minimal "CUDA on the host" + THC stand-in -- TEST INFRASTRUCTURE ONLY (see TH.h in this directory).
//
// The reference has no CPU implementation of solveLinearSystemJacobi (generic/tfluids.cc:836-839
// raises; init.lua:715 asserts a CudaTensor). To pin our Jacobi restatement to the REFERENCE's code
// rather than to our reading of it, ../ref_jacobi.cc compiles the reference's own kernel
// (generic/tfluids.cu:1765-1821) and host loop (:1823-1927) for the host through this header:
// __global__/__device__ become plain functions, threadIdx/blockIdx are thread-local variables that a
// serial LaunchKernel steps through the launch grid, and the few THC calls the loop makes are
// provided on host memory. Nothing here is product code.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iostream>
#include <string>

#include "luaT.h"   // lua_State stand-in + TH.h (ShimTensor, ShimError, THError)

#define __global__
#ifndef __forceinline__
#define __forceinline__ inline
#endif

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
static thread_local dim3 threadIdx, blockIdx, blockDim, gridDim;

static inline float __int_as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }

// ---- THC ---------------------------------------------------------------------------------------
struct THCState {};
typedef ShimTensor<float> THCudaTensor;
static inline THCState* cutorch_getstate(lua_State*) { static THCState s; return &s; }
static inline int THCudaTensor_isContiguous(THCState*, THCudaTensor* t) { return shim_contig(t); }

// THCDeviceTensor<T, Dim>: sizes/strides + chained operator[] down to a reference (THCDeviceTensor.cuh).
template <typename T, int Dim> struct THCDeviceSubTensor {
  T* p; const long* stride;
  __host__ __device__ THCDeviceSubTensor<T, Dim - 1> operator[](long i) const {
    return THCDeviceSubTensor<T, Dim - 1>{p + i * stride[0], stride + 1};
  }
};
template <typename T> struct THCDeviceSubTensor<T, 1> {
  T* p; const long* stride;
  __host__ __device__ T& operator[](long i) const { return p[i * stride[0]]; }
};
// a fully indexed 1-D tensor (the PCG copy kernels' pressure_pcg[ind]): reads as a T, assigns through
template <typename T> struct THCDeviceSubTensor<T, 0> {
  T* p; const long* stride;
  __host__ __device__ operator T&() const { return *p; }
  __host__ __device__ T& operator=(T v) const { *p = v; return *p; }
};
template <typename T, int Dim> struct THCDeviceTensor {
  T* data_; long size_[Dim]; long stride_[Dim];
  __host__ __device__ long getSize(int i) const { return size_[i]; }
  __host__ __device__ long getStride(int i) const { return stride_[i]; }
  __host__ __device__ THCDeviceSubTensor<T, Dim - 1> operator[](long i) const {
    return THCDeviceSubTensor<T, Dim - 1>{data_ + i * stride_[0], stride_ + 1};
  }
};
template <typename T, int Dim>
static inline THCDeviceTensor<T, Dim> toDeviceTensor(THCState*, THCudaTensor* t) {
  if (t->nDimension != Dim) throw ShimError("toDeviceTensor: dimension mismatch");
  THCDeviceTensor<T, Dim> d;
  d.data_ = t->data;
  for (int i = 0; i < Dim; i++) { d.size_[i] = t->size[i]; d.stride_[i] = t->stride[i]; }
  return d;
}

// the THC calls of the Jacobi host loop, generic/tfluids.cu:1869-1918, on host memory
static inline void THCudaTensor_zero(THCState*, THCudaTensor* t) {
  std::memset(t->data, 0, sizeof(float) * (size_t)shim_numel(t));
}
static inline void THCudaTensor_copy(THCState*, THCudaTensor* dst, THCudaTensor* src) {
  std::memcpy(dst->data, src->data, sizeof(float) * (size_t)shim_numel(src));
}
// r = a - alpha * b
static inline void THCudaTensor_csub(THCState*, THCudaTensor* r, THCudaTensor* a, float alpha, THCudaTensor* b) {
  const long n = shim_numel(a);
  for (long i = 0; i < n; i++) r->data[i] = a->data[i] - alpha * b->data[i];
}
static inline void THCudaTensor_resize2d(THCState*, THCudaTensor* t, long a, long b) {
  long sz[2] = {a, b}; shim_resize(t, 2, sz);
}
static inline void THCudaTensor_resize5d(THCState*, THCudaTensor* t, long a, long b, long c, long d, long e) {
  long sz[5] = {a, b, c, d, e}; shim_resize(t, 5, sz);
}
// L2 norm over dim 2 (1-based) of a [nbatch][numel] view. THC reduces in fp32 in an unspecified tree order;
// here fp64 accumulate + one rounding (the value only decides termination when pTol > 0).
static inline void THCudaTensor_norm(THCState*, THCudaTensor* r, THCudaTensor* a, float p, int dim, int keepdim) {
  (void)p; (void)dim; (void)keepdim;
  const long nb = a->size[0], n = a->size[1];
  for (long b = 0; b < nb; b++) {
    double s = 0.0;
    for (long i = 0; i < n; i++) { const double v = a->data[b * n + i]; s += v * v; }
    r->data[b] = (float)std::sqrt(s);
  }
}
static inline float THCudaTensor_maxall(THCState*, THCudaTensor* t) {
  const long n = shim_numel(t);
  float m = t->data[0];
  for (long i = 1; i < n; i++) if (t->data[i] > m) m = t->data[i];
  return m;
}
