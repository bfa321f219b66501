// output_layers.cpp -- loss / l2norm / dense layers of the model head, device-resident.
#include "host_util.h"
#include "math_functions.hh"
#include "output_layers.h"

static inline gaib_ctx* C() { return gpu_context::get(); }

// ---- loss_layer -----------------------------------------------------------------------------------
loss_layer::loss_layer() : num_samples(0), num_cls(1), capacity_(0), phase_(net_phase::TRAIN), feat_in(NULL),
                           feat_out(NULL), d_losses(NULL), labels(NULL) {}
loss_layer::loss_layer(int nv, int ncls) : loss_layer(nv, ncls, NULL) {}
loss_layer::loss_layer(int nv, int ncls, label_t* ptr)
    : num_samples(nv), num_cls(ncls), capacity_(0), phase_(net_phase::TRAIN), feat_in(NULL), feat_out(NULL),
      d_losses(NULL), labels(ptr) {
  allocate(nv);
}
void loss_layer::allocate(int nv) {
  const size_t n = (size_t)nv * num_cls;
  if (feat_in) float_free_device(feat_in);
  if (feat_out) float_free_device(feat_out);
  if (d_losses) float_free_device(d_losses);
  float_malloc_device64(n, feat_in);
  float_malloc_device64(n, feat_out);
  float_malloc_device64((size_t)nv, d_losses);
  GAIB_OR_DIE(gaib_fill_f32(C(), n, 0.f, feat_in));
  GAIB_OR_DIE(gaib_fill_f32(C(), n, 0.f, feat_out));
  GAIB_OR_DIE(gaib_fill_f32(C(), nv, 0.f, d_losses));
  capacity_ = nv;
}
void loss_layer::update_dim_size(int x) {
  if (x > capacity_) allocate(x);
  num_samples = x;
}

void softmax_loss_layer::forward(size_t begin, size_t end, mask_t* masks) {
  softmax_cross_entropy_gpu(num_cls, begin, end, feat_in, masks, labels, d_losses, feat_out);
}
void softmax_loss_layer::backward(size_t begin, size_t end, mask_t* masks, float* grad_out) {
  d_softmax_cross_entropy_gpu(num_cls, begin, end, masks, labels, feat_out, grad_out);
}
acc_t softmax_loss_layer::get_prediction_loss(size_t begin, size_t end, size_t count, mask_t* masks) {
  return masked_avg_loss_gpu(begin, end, count, masks, d_losses);
}

// multi-label head (sigmoid_loss_layer.cpp:4-55): labels are [num_samples x num_cls] 0/1 bytes
void sigmoid_loss_layer::forward(size_t begin, size_t end, mask_t* masks) {
  sigmoid_cross_entropy_gpu(num_cls, begin, end, feat_in, masks, labels, d_losses, feat_out);
}
void sigmoid_loss_layer::backward(size_t begin, size_t end, mask_t* masks, float* grad_out) {
  d_sigmoid_cross_entropy_gpu(num_cls, begin, end, masks, labels, feat_out, grad_out);
}
acc_t sigmoid_loss_layer::get_prediction_loss(size_t begin, size_t end, size_t count, mask_t* masks) {
  return masked_avg_loss_gpu(begin, end, count, masks, d_losses);
}

// ---- row_block ------------------------------------------------------------------------------------
row_block::row_block(int r, int wi, int wo) : rows(r), cap(0), win(wi), wout(wo), acts(NULL), grads(NULL) {
  resize(r);
}
void row_block::resize(int r) {
  if (r > cap) {
    if (acts) float_free_device(acts);
    if (grads) float_free_device(grads);
    float_malloc_device64((size_t)r * win, acts);
    float_malloc_device64((size_t)r * wout, grads);
    GAIB_OR_DIE(gaib_fill_f32(C(), (int64_t)r * win, 0.f, acts));
    GAIB_OR_DIE(gaib_fill_f32(C(), (int64_t)r * wout, 0.f, grads));
    cap = r;
  }
  rows = r;
}

// ---- l2norm_layer (reference math: src/layers/l2norm_layer.cpp:19-64) -------------------------------
l2norm_layer::l2norm_layer(int nv, int len) : buf(nv, len, len) {}
void l2norm_layer::forward(float* feat_out) { l2norm(buf.rows, buf.win, buf.acts, feat_out); }
void l2norm_layer::backward(float* grad_out) { d_l2norm(buf.rows, buf.win, buf.acts, buf.grads, grad_out); }

// ---- dense_layer (reference math: src/layers/dense_layer.cpp:42-72) ---------------------------------
dense_layer::dense_layer(int nv, int in_len, int out_len, float lr)
    : buf(nv, in_len, out_len), d_weight(NULL), d_weight_grad(NULL), optm(new adam(lr)) {
  vec_t w;
  init_glorot(in_len, out_len, w, 1);  // the OpenMP path's init (dense_layer.cpp:30); its CUDA path draws from cuRAND
  float_malloc_device64(w.size(), d_weight);
  float_malloc_device64(w.size(), d_weight_grad);
  copy_float_device((int)w.size(), w.data(), d_weight);
  GAIB_OR_DIE(gaib_fill_f32(C(), (int64_t)w.size(), 0.f, d_weight_grad));
}
void dense_layer::forward(float* feat_out) { matmul(buf.rows, buf.wout, buf.win, buf.acts, d_weight, feat_out); }
void dense_layer::backward(float* grad_out) {
  matmul(buf.win, buf.wout, buf.rows, buf.acts, buf.grads, d_weight_grad, true);
  matmul(buf.rows, buf.win, buf.wout, buf.grads, d_weight, grad_out, false, true);
  optm->update_gpu((size_t)buf.win * buf.wout, d_weight_grad, d_weight);
}
