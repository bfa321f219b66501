// GGNN is out of scope (CUDA-only in the reference, nv x nv gate matrices).  The
// class exists because the reference's net.cpp:620 instantiates Model<GGNN_layer>; every compute
// method reports that and exits.
#pragma once
#include "lgraph.h"
#include "optimizer.h"

class GGNN_layer {
 public:
  GGNN_layer(int, int, int, int, Graph*, bool, float, float, float) : feat_in(NULL), grad_in(NULL) {}
  void forward(float*) { unsupported(); }
  void backward(float*, float*) { unsupported(); }
  void update_weight(optimizer*) { unsupported(); }
  float* get_feat_in() { return feat_in; }
  float* get_grad_in() { return grad_in; }
  void set_feat_in(float* p) { feat_in = p; }
  void set_graph_ptr(Graph*) {}
  void set_netphase(net_phase) {}
  void update_dim_size(size_t) {}
  void print_layer_info() { std::cout << "GGNN layer (not supported on this backend)\n"; }

 private:
  static void unsupported() {
    fprintf(stderr, "GGNN_layer is not supported by the MI355X backend (out of scope)\n");
    exit(EXIT_FAILURE);
  }
  float* feat_in;
  float* grad_in;
};
