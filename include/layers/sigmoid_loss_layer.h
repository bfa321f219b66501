// multi-label output layer (reference: include/layers/sigmoid_loss_layer.h).  None of the five
// BASELINE configs is multi-label: the class exists so drivers link; using it
// reports "not supported" and exits, following the reference's error convention.
#pragma once
#include "loss_layer.h"

class sigmoid_loss_layer : public loss_layer {
 public:
  sigmoid_loss_layer() {}
  sigmoid_loss_layer(int nv, int n_cls) : loss_layer(nv, n_cls, NULL) {}
  sigmoid_loss_layer(int nv, int n_cls, label_t* ptr) : loss_layer(nv, n_cls, ptr) {}
  virtual void forward(size_t begin, size_t end, mask_t* masks);
  virtual void backward(size_t begin, size_t end, mask_t* masks, float* grad_out);
  virtual acc_t get_prediction_loss(size_t begin, size_t end, size_t count, mask_t* masks);
};
