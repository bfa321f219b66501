// fully connected head hid -> num_classes (reference: include/layers/dense_layer.h,
// src/layers/dense_layer.cpp:42-72).  backward() also applies the layer's own Adam step, as the
// reference does.
#pragma once
#include "optimizer.h"

class dense_layer {
 private:
  bool is_bias;
  int num_samples;
  int dim_in;
  int dim_out;
  int capacity_;
  float* feat_in;
  float* grad_in;
  optimizer* optm;
  float* d_weight;
  float* d_weight_grad;

 public:
  dense_layer(int nv, int in_len, int out_len, float lr);
  void forward(float* feat_out);
  void backward(float* grad_out);
  float* get_feat_in() { return feat_in; }
  float* get_grad_in() { return grad_in; }
  float* weight_ptr() { return d_weight; }
  float* weight_grad_ptr() { return d_weight_grad; }
  void update_dim_size(int sz);
};
