// row-wise L2 normalisation layer, used in front of the dense head for GAT / sampling
// (reference: include/layers/l2norm_layer.h, src/layers/l2norm_layer.cpp:19-64).
#pragma once
#include "global.h"

class l2norm_layer {
 private:
  int num_samples;
  int dim;
  int capacity_;
  float* feat_in;
  float* grad_in;

 public:
  l2norm_layer(int nv, int len);
  void forward(float* feat_out);
  void backward(float* grad_out);
  float* get_feat_in() { return feat_in; }
  float* get_grad_in() { return grad_in; }
  void update_dim_size(int sz);
};
