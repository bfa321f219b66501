// host_util.h -- internal helpers of the host C++ mirror.
#pragma once
#include "gaib.h"
#include "gpu_context.h"

namespace gaib_host {
// accumulates wall time into time_ops[op] when GAIB_SYNC_TIMERS=1 (the reference times every op
// around a device-wide sync: cutils.h:18-28; here the sync is opt-in so the default path stays
// asynchronous)
struct OpTimer {
  explicit OpTimer(char op);
  ~OpTimer();
  char op_;
  double t0_;
};

template <typename T>
inline T* dmalloc(size_t n) {
  void* p = nullptr;
  GAIB_OR_DIE(gaib_malloc(gpu_context::get(), (n > 0 ? n : 1) * sizeof(T), &p));
  return static_cast<T*>(p);
}
}  // namespace gaib_host
