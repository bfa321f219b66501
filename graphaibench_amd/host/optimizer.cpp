// optimizer.cpp -- Adam over device weights (gaib_adam_step) with per-buffer state.
#include "optimizer.h"
#include "host_util.h"

static bool g_powers_on_device = false;
void adam::keep_powers_on_device(bool on) { g_powers_on_device = on; }
bool adam::powers_on_device() { return g_powers_on_device; }

void adam::update_gpu(const size_t n, const float_t* dW, float_t* W) {
  gaib_ctx* c = gpu_context::get();
  auto it = dev_state.find(W);
  if (it == dev_state.end()) {
    state s;
    s.n = n;
    s.m = gaib_host::dmalloc<float>(n);
    s.v = gaib_host::dmalloc<float>(n);
    GAIB_OR_DIE(gaib_fill_f32(c, (int64_t)n, 0.f, s.m));
    GAIB_OR_DIE(gaib_fill_f32(c, (int64_t)n, 0.f, s.v));
    it = dev_state.emplace(W, s).first;
  }
  assert(it->second.n == n);
  // vertex-range partitions: the gradient of a replicated weight is the sum of the ranks' partial gradients
  if (gaib_comm* cm = gpu_context::comm()) GAIB_OR_DIE(gaib_allreduce_f32(cm, const_cast<float_t*>(dW), (int64_t)n));
  if (g_powers_on_device) {
    if (!d_pow) {  // first step this way: the powers move to the device at their current value
      d_pow = gaib_host::dmalloc<float>(2);
      const float h[2] = {b1_t, b2_t};
      GAIB_OR_DIE(gaib_memcpy_h2d(c, d_pow, h, sizeof(h)));
    }
    GAIB_OR_DIE(gaib_adam_step_dev(c, (int64_t)n, dW, W, it->second.m, it->second.v, alpha, b1, b2, eps, d_pow));
    return;  // advanced on the device
  }
  GAIB_OR_DIE(gaib_adam_step(c, (int64_t)n, dW, W, it->second.m, it->second.v, alpha, b1, b2, b1_t, b2_t, eps));
  b1_t *= b1;  // once per call: a shared optimizer advances per layer (Q6)
  b2_t *= b2;
}

void adam::update(const vec_t& dW, vec_t& W) {
  auto& st = host_state[&W];
  if (st.first.empty()) {
    st.first.assign(W.size(), 0.f);
    st.second.assign(W.size(), 0.f);
  }
  vec_t &mt = st.first, &vt = st.second;
  for (size_t i = 0; i < W.size(); i++) {
    mt[i] = b1 * mt[i] + (float_t(1) - b1) * dW[i];
    vt[i] = b2 * vt[i] + (float_t(1) - b2) * dW[i] * dW[i];
    W[i] -= alpha * (mt[i] / (float_t(1) - b1_t)) / std::sqrt((vt[i] / (float_t(1) - b2_t)) + eps);
  }
  b1_t *= b1;
  b2_t *= b2;
}

adam::~adam() { reset(); }

void adam::reset() {
  gaib_ctx* c = gpu_context::get();
  for (auto& kv : dev_state) {
    gaib_free(c, kv.second.m);
    gaib_free(c, kv.second.v);
  }
  dev_state.clear();
  host_state.clear();
  if (d_pow) {
    gaib_free(c, d_pow);
    d_pow = nullptr;
  }
  b1_t = b1;
  b2_t = b2;
}
