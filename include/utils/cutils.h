// include/utils/cutils.h -- device helper templates the drivers call directly
// (reference: include/utils/cutils.h:18-28,193-202; net.cpp:344-349,414-415).
#pragma once
#include "gpu_context.h"

inline unsigned CudaTest(const char*) {  // device-wide sync + error check
  gpu_context::sync();
  return 0;
}

template <typename T>
void copy_async_device(int n, T* h_ptr, T* d_ptr) {
  GAIB_OR_DIE(gaib_memcpy_h2d(gpu_context::get(), d_ptr, h_ptr, sizeof(T) * (size_t)n));
}
template <typename T>
void malloc_device(int n, T*& ptr) {
  void* p = nullptr;
  GAIB_OR_DIE(gaib_malloc(gpu_context::get(), sizeof(T) * (size_t)(n > 0 ? n : 1), &p));
  ptr = static_cast<T*>(p);
}
template <typename T>
void free_device(T*& ptr) {
  GAIB_OR_DIE(gaib_free(gpu_context::get(), ptr));
  ptr = NULL;
}
