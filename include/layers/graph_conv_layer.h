// include/layers/graph_conv_layer.h -- graph convolution layers of the GNN path.
// Constructor, forward(feat_out), backward(feat_out, grad_out), update_weight(opt) and the
// accessors have the reference's signatures (include/layers/graph_conv_layer.h:6-106).  All
// buffers live in HBM; feat_out / grad_out are BORROWED pointers into the neighbouring layer
// (net.cpp:458-469, 591-614); layers are stored by value in std::vector, so copies are shallow
// and nothing is freed in a destructor (as in the reference).
#pragma once
#include "aggregator.h"
#include "lgraph.h"
#include "optimizer.h"

template <typename Aggregator>
class graph_conv_layer {
 public:
  graph_conv_layer(int id, int nv, int din, int dout, Graph* g, bool act, bool concat, float lr,
                   float feat_drop, float score_drop);
  float* get_feat_in() { return feat_in; }
  float* get_grad_in() { return grad_in; }
  void set_feat_in(float* ptr) { feat_in = ptr; }
  void set_graph_ptr(Graph* ptr) { graph = ptr; }
  void set_netphase(net_phase phase) { phase_ = phase; }
  void print_layer_info() {
    std::cout << "GraphConv Layer " << level_ << " with " << num_samples << " samples, dims: ["
              << dim_in << " x " << dim_out << "]\n";
  }
  void update_dim_size(size_t sz);
  // device buffers (tests, checkpoints, weight-gradient all-reduce)
  float* weight_neigh_ptr() { return d_W_neigh; }
  float* weight_neigh_grad_ptr() { return d_W_neigh_grad; }
  float* weight_self_ptr() { return d_W_self; }
  float* weight_self_grad_ptr() { return d_W_self_grad; }
  Aggregator& get_aggregator() { return aggr; }
  int get_dim_in() const { return dim_in; }
  int get_dim_out() const { return dim_out; }

 protected:
  int level_;
  int num_samples;
  int dim_in;
  int dim_out;
  Graph* graph;
  bool is_act;
  bool is_bias;  // hard-wired false in the reference (graph_conv_layer.cpp:8)
  bool use_concat;
  float feat_dropout_rate;
  float score_dropout_rate;
  float feat_scale;
  net_phase phase_;
  size_t capacity_;  // rows the device buffers were sized for
  uint64_t dropout_calls;

  float* feat_in;         // [nv*dim_in]   (level 0: the input features, set by the model)
  float* grad_in;         // [nv*dim_out]
  float* d_in_temp;       // [nv*dim_in]
  float* d_in_temp1;      // [nv*dim_in]   only if dim_in <= dim_out
  float* d_out_temp;      // [nv*dim_out]
  float* d_W_neigh;       // [dim_in*dim_out]
  float* d_W_neigh_grad;
  float* d_W_self;        // SAGE only
  float* d_W_self_grad;
  mask_t* dropout_mask;
  optimizer* optm;
  Aggregator aggr;
};

class GCN_layer : public graph_conv_layer<GCN_Aggregator> {
 public:
  GCN_layer(int id, int nv, int din, int dout, Graph* g, bool act, float lr, float feat_drop_rate,
            float score_drop_rate)
      : graph_conv_layer(id, nv, din, dout, g, act, false, lr, feat_drop_rate, score_drop_rate) {
    aggr.init(dim_in < dim_out ? dim_in : dim_out, nv);
  }
  void forward(float* feat_out);
  void backward(float* feat_out, float* grad_out);
  void update_weight(optimizer* opt);
};

class SAGE_layer : public graph_conv_layer<SAGE_Aggregator> {
 public:
  SAGE_layer(int id, int nv, int din, int dout, Graph* g, bool act, float lr, float feat_drop_rate,
             float score_drop_rate)
      : graph_conv_layer(id, nv, din, dout, g, act, true, lr, feat_drop_rate, score_drop_rate) {
    aggr.init(dim_in < dim_out ? dim_in : dim_out, nv);
  }
  void forward(float* feat_out);
  void backward(float* feat_out, float* grad_out);
  void update_weight(optimizer* opt);
};

class GAT_layer : public graph_conv_layer<GAT_Aggregator> {
 public:
  GAT_layer(int id, int nv, int din, int dout, Graph* g, bool act, float lr, float feat_drop_rate,
            float score_drop_rate)
      : graph_conv_layer(id, nv, din, dout, g, act, false, lr, feat_drop_rate, score_drop_rate) {
    aggr.init(dim_out, nv, g->sizeEdges(), lr, score_drop_rate);
  }
  void forward(float* feat_out);
  void backward(float* feat_out, float* grad_out);
  void update_weight(optimizer* opt);
};

#include "ggnn_layer_stub.h"
