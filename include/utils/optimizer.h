// include/utils/optimizer.h -- optimizer interface of the layer API.
// The reference's layers call `opt->update_gpu(n, dW, W)` with DEVICE pointers and keep one
// state block per weight buffer, keyed by its address (reference: include/utils/optimizer.h:23-59,
// 99-116; src/utilities/optimizer.cu:16-36) -- weight buffers therefore must not move.
// Only Adam is on the GNN path (graph_conv_layer.cpp:50, net.cpp:362); the tiny-dnn derived
// host-vector optimizers of the reference are not mirrored.
#pragma once
#include <unordered_map>
#include "global.h"

struct optimizer {
  virtual ~optimizer() = default;
  virtual void update(const vec_t& dW, vec_t& W) = 0;  // host vectors (kept for API parity)
  virtual void update_gpu(const size_t n, const float_t* dW, float_t* W) = 0;  // device pointers
  virtual void reset() {}
};

// Adam with eps INSIDE the square root; the beta powers advance once per update call, so a
// shared instance steps them once per layer per epoch (reference quirk Q6, optimizer.cpp:22-35).
struct adam : public optimizer {
  adam(float_t lr)
      : alpha(lr), b1(float_t(0.9)), b2(float_t(0.999)), b1_t(float_t(0.9)), b2_t(float_t(0.999)),
        eps(float_t(1e-8)) {}
  adam() : adam(0.01) {}
  ~adam() override;  // the per-weight moment buffers in HBM go back (the reference's instances live as long as the process)
  void update(const vec_t& dW, vec_t& W) override;
  void update_gpu(const size_t n, const float_t* dW, float_t* W) override;
  void reset() override;
  float_t alpha, b1, b2, b1_t, b2_t;
  // Beta powers in DEVICE memory (gaib_adam_step_dev), advanced there: what a recorded epoch (HIP graph) needs, since
  // by-value kernel arguments are frozen in the recording.  A one-way switch for all instances, made before the run's
  // first update; b1_t / b2_t above then stay at the value they had when an instance first stepped this way.
  static void keep_powers_on_device(bool on);
  static bool powers_on_device();

 private:
  float_t eps;
  float* d_pow = nullptr;  // {b1^t, b2^t} on the device
  struct state { float* m; float* v; size_t n; };
  std::unordered_map<const float_t*, state> dev_state;  // keyed by the weight's device address
  std::unordered_map<const vec_t*, std::pair<vec_t, vec_t>> host_state;
};
