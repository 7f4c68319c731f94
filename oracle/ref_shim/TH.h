// Minimal stand-in for Torch7's TH.h -- TEST INFRASTRUCTURE ONLY.
//
// Lets the reference's CPU tfluids translation unit (/root/reference/torch/tfluids/init.cu,
// compiled as C++ with -DBUILD_WITHOUT_CUDA_FUNCS) build without Torch7. Only the handful of
// TH entry points that translation unit touches are provided. Nothing here is product code;
// the product (fluidnet_amd/csrc) never includes this header.
#pragma once
#include <cfloat>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <unistd.h>
#include <execinfo.h>
#include <mutex>
#include <iostream>

#define TH_CONCAT_STRING_3(x, y, z) TH_CONCAT_STRING_3_EXPAND(x, y, z)
#define TH_CONCAT_STRING_3_EXPAND(x, y, z) #x #y #z
#define TH_CONCAT_3(x, y, z) TH_CONCAT_3_EXPAND(x, y, z)
#define TH_CONCAT_3_EXPAND(x, y, z) x##y##z
#define TH_CONCAT_4(x, y, z, w) TH_CONCAT_4_EXPAND(x, y, z, w)
#define TH_CONCAT_4_EXPAND(x, y, z, w) x##y##z##w

#define THTensor TH_CONCAT_3(TH, Real, Tensor)
#define THTensor_(NAME) TH_CONCAT_4(TH, Real, Tensor_, NAME)

#ifndef __host__
#define __host__
#endif
#ifndef __device__
#define __device__
#endif

struct ShimError : public std::runtime_error {
  explicit ShimError(const std::string& m) : std::runtime_error(m) {}
};

[[noreturn]] inline void THError(const char* msg, ...) { throw ShimError(msg); }

template <typename T>
struct ShimTensor {
  int nDimension = 0;
  long size[5] = {0, 0, 0, 0, 0};
  long stride[5] = {0, 0, 0, 0, 0};
  T* data = nullptr;
  bool owns = false;
  std::vector<T>* own_store = nullptr;
};

typedef ShimTensor<float> THFloatTensor;
typedef ShimTensor<double> THDoubleTensor;
typedef ShimTensor<int> THIntTensor;

template <typename T>
inline long shim_numel(const ShimTensor<T>* t) {
  if (t->nDimension == 0) return 0;
  long n = 1;
  for (int d = 0; d < t->nDimension; d++) n *= t->size[d];
  return n;
}
template <typename T>
inline void shim_set_contig(ShimTensor<T>* t, int nd, const long* sz) {
  t->nDimension = nd;
  long s = 1;
  for (int d = nd - 1; d >= 0; d--) { t->size[d] = sz[d]; t->stride[d] = s; s *= sz[d]; }
}
template <typename T>
inline bool shim_contig(const ShimTensor<T>* t) {
  long s = 1;
  for (int d = t->nDimension - 1; d >= 0; d--) {
    if (t->size[d] != 1 && t->stride[d] != s) return false;
    s *= t->size[d];
  }
  return true;
}
template <typename T>
inline void shim_resize(ShimTensor<T>* t, int nd, const long* sz) {
  long n = 1;
  for (int d = 0; d < nd; d++) n *= sz[d];
  if (t->owns) {
    t->own_store->resize(n);
    t->data = t->own_store->data();
  } else if (n > shim_numel(t)) {
    throw ShimError("shim: cannot grow a borrowed tensor");
  }
  shim_set_contig(t, nd, sz);
}
template <typename T>
inline ShimTensor<T>* shim_new() {
  auto* t = new ShimTensor<T>();
  t->owns = true;
  t->own_store = new std::vector<T>();
  return t;
}
template <typename T>
inline void shim_free(ShimTensor<T>* t) {
  if (t->owns) delete t->own_store;
  delete t;
}

#define SHIM_REAL_API(Real, T)                                                              \
  inline T* TH##Real##Tensor_data(TH##Real##Tensor* t) { return t->data; }                  \
  inline long TH##Real##Tensor_numel(TH##Real##Tensor* t) { return shim_numel(t); }         \
  inline int TH##Real##Tensor_isContiguous(TH##Real##Tensor* t) { return shim_contig(t); }  \
  inline TH##Real##Tensor* TH##Real##Tensor_new() { return shim_new<T>(); }                 \
  inline void TH##Real##Tensor_free(TH##Real##Tensor* t) { shim_free(t); }                  \
  inline void TH##Real##Tensor_resize1d(TH##Real##Tensor* t, long a) {                      \
    long sz[1] = {a}; shim_resize(t, 1, sz); }                                              \
  inline void TH##Real##Tensor_fill(TH##Real##Tensor* t, T v) {                             \
    long n = shim_numel(t); for (long i = 0; i < n; i++) t->data[i] = v; }

SHIM_REAL_API(Float, float)
SHIM_REAL_API(Double, double)

inline void THIntTensor_fill(THIntTensor* t, int v) {
  long n = shim_numel(t); for (long i = 0; i < n; i++) t->data[i] = v;
}
inline void THIntTensor_resize4d(THIntTensor* t, long a, long b, long c, long d) {
  long sz[4] = {a, b, c, d}; shim_resize(t, 4, sz);
}
inline THIntTensor* THIntTensor_newSelect(THIntTensor* t, int dim, long idx) {
  auto* r = new THIntTensor();
  r->nDimension = t->nDimension - 1;
  int o = 0;
  for (int d = 0; d < t->nDimension; d++) {
    if (d == dim) continue;
    r->size[o] = t->size[d]; r->stride[o] = t->stride[d]; o++;
  }
  r->data = t->data + idx * t->stride[dim];
  return r;
}
inline void THIntTensor_free(THIntTensor* t) { shim_free(t); }
inline int THIntTensor_get3d(const THIntTensor* t, long a, long b, long c) {
  return t->data[a * t->stride[0] + b * t->stride[1] + c * t->stride[2]];
}
inline void THIntTensor_set3d(THIntTensor* t, long a, long b, long c, int v) {
  t->data[a * t->stride[0] + b * t->stride[1] + c * t->stride[2]] = v;
}
inline int THIntTensor_get4d(const THIntTensor* t, long a, long b, long c, long d) {
  return t->data[a * t->stride[0] + b * t->stride[1] + c * t->stride[2] + d * t->stride[3]];
}
