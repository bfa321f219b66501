// single-label output layer: softmax + cross entropy (reference: include/layers/softmax_loss_layer.h,
// src/layers/softmax_loss_layer.{cpp,cu}); gradient = (p - onehot) / (end - begin)  (Q8).
#pragma once
#include "loss_layer.h"

class softmax_loss_layer : public loss_layer {
 public:
  softmax_loss_layer() {}
  softmax_loss_layer(int nv, int n_cls) : loss_layer(nv, n_cls, NULL) {}
  softmax_loss_layer(int nv, int n_cls, label_t* ptr) : loss_layer(nv, n_cls, ptr) {}
  virtual void forward(size_t begin, size_t end, mask_t* masks);
  virtual void backward(size_t begin, size_t end, mask_t* masks, float* grad_out);
  virtual acc_t get_prediction_loss(size_t begin, size_t end, size_t count, mask_t* masks);
};
