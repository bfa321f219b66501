// include/layers/graph_conv_layer.h -- the graph convolution layers of the GNN path (device resident).
//
// What the drivers see is the reference's interface (include/layers/graph_conv_layer.h:6-106): a class template
// graph_conv_layer<Aggregator> and the concrete GCN_layer / SAGE_layer / GAT_layer with
//     L(id, nv, din, dout, Graph*, act, lr, feat_drop, score_drop)
//     forward(feat_out)   backward(feat_out, grad_out)   update_weight(optimizer*)
//     get_feat_in()  get_grad_in()  set_feat_in()  set_graph_ptr()  set_netphase()  update_dim_size()  print_layer_info()
// feat_out / grad_out are BORROWED pointers into the neighbouring layer (net.cpp:458-469, 591-614); layers are stored
// by value in std::vector (include/gnn/net.h:57), so copies are shallow and no destructor frees anything.
//
// Layout here: everything that does not depend on the aggregator type sits in gconv_state (shapes, flags, HBM
// buffers, growth on demand); the template adds the aggregator; the three layers are stamped out by one macro.
#pragma once
#include "aggregator.h"
#include "lgraph.h"
#include "optimizer.h"

class gconv_state {
 public:
  // ---- the reference's accessors -------------------------------------------------------------------------------
  float* get_feat_in() { return feat_in; }
  float* get_grad_in() { return grad_in; }
  void set_feat_in(float* ptr) { feat_in = ptr; agg_valid_ = false; }
  void set_graph_ptr(Graph* ptr) { graph = ptr; agg_valid_ = false; }
  void set_netphase(net_phase phase) { phase_ = phase; }
  void update_dim_size(size_t sz);  // number of vertices (subgraph sampling); buffers grow on demand
  void print_layer_info() {
    std::cout << "GraphConv Layer " << level_ << " with " << num_samples << " samples, dims: [" << dim_in << " x "
              << dim_out << "]\n";
  }
  // ---- extensions: device buffers for tests, checkpoints and the weight-gradient all-reduce ------------------
  float* weight_neigh_ptr() { return d_W_neigh; }
  float* weight_neigh_grad_ptr() { return d_W_neigh_grad; }
  float* weight_self_ptr() { return d_W_self; }
  float* weight_self_grad_ptr() { return d_W_self_grad; }
  int get_dim_in() const { return dim_in; }
  int get_dim_out() const { return dim_out; }
  // ---- extension: a constant input -----------------------------------------------------------------------------
  // The driver promises that the CONTENTS of feat_in and the graph do not change between forward calls: layer 0 of a
  // full-batch run without feature dropout (the reference re-aggregates the same features every epoch,
  // gcn_layer.cpp:22-25, sage_layer.cpp:17-21).  Where the layer aggregates first (dim_in <= dim_out) the aggregated
  // input A.X is then the same every epoch; the first forward computes it -- it is kept anyway, for the weight
  // gradient -- and later forwards run only the dense product(s) on it.  set_feat_in, set_graph_ptr, update_dim_size
  // and this call drop the kept aggregate.  Off unless asked for.
  void set_input_constant(bool on) { input_constant_ = on; agg_valid_ = false; }
  bool input_constant() const { return input_constant_; }
  // ---- extension: give the HBM back ------------------------------------------------------------------------------
  // Layers are copied by value (std::vector<layer_type>, include/gnn/net.h:57), so there is no destructor; a host that
  // builds and drops models in one process calls release() on the ONE copy it keeps (LearningGraph::dealloc's idiom):
  // weights, gradients, activations, masks and the layer's Adam state are freed, the object must not be used afterwards.
  // The concrete layers add their aggregator's buffers (release_all).
  void release();

 protected:
  gconv_state(int id, int nv, int din, int dout, Graph* g, bool act, bool concat, float lr, float feat_drop,
              float score_drop);
  // shape and mode
  int level_, num_samples, dim_in, dim_out;
  Graph* graph;
  bool is_act, is_bias /* hard-wired false in the reference, graph_conv_layer.cpp:8 */, use_concat;
  float feat_dropout_rate, score_dropout_rate, feat_scale;
  net_phase phase_;
  size_t capacity_;        // rows the HBM buffers were sized for
  uint64_t dropout_calls;  // counter feeding the dropout RNG
  bool input_constant_ = false, agg_valid_ = false;  // set_input_constant: d_in_temp1 holds A.feat_in of the current graph
  // HBM buffers
  float* feat_in;     // [nv x dim_in]   level 0: the input features, set by the model
  float* grad_in;     // [nv x dim_out]
  float* d_in_temp;   // [nv x dim_in]
  float* d_in_temp1;  // [nv x dim_in]   only where dim_in <= dim_out (the aggregated input, kept for the weight gradient)
  float* d_out_temp;  // [nv x dim_out]
  float *d_W_neigh, *d_W_neigh_grad;  // [dim_in x dim_out]
  float *d_W_self, *d_W_self_grad;    // SAGE only
  mask_t* dropout_mask;
  optimizer* optm;
};

template <typename Aggregator>
class graph_conv_layer : public gconv_state {
 public:
  graph_conv_layer(int id, int nv, int din, int dout, Graph* g, bool act, bool concat, float lr, float feat_drop,
                   float score_drop)
      : gconv_state(id, nv, din, dout, g, act, concat, lr, feat_drop, score_drop) {}
  Aggregator& get_aggregator() { return aggr; }
  void release_all() {  // the layer's buffers and the aggregator's
    aggr.release();
    release();
  }

 protected:
  Aggregator aggr;
};

// NAME(id, nv, din, dout, g, act, lr, feat_drop_rate, score_drop_rate): CONCAT = the layer has a self weight;
// AGGR_INIT runs in the constructor body (the aggregator works at the narrower of the two widths for GCN / SAGE,
// gcn_layer / sage_layer constructors; GAT attends over the output width and owns per-edge arrays)
#define GAIB_GCONV_LAYER(NAME, AGGREGATOR, CONCAT, AGGR_INIT)                                                    \
  class NAME : public graph_conv_layer<AGGREGATOR> {                                                             \
   public:                                                                                                       \
    NAME(int id, int nv, int din, int dout, Graph* g, bool act, float lr, float feat_drop_rate,                  \
         float score_drop_rate)                                                                                  \
        : graph_conv_layer(id, nv, din, dout, g, act, CONCAT, lr, feat_drop_rate, score_drop_rate) {             \
      AGGR_INIT;                                                                                                 \
    }                                                                                                            \
    void forward(float* feat_out);                                                                               \
    void backward(float* feat_out, float* grad_out);                                                             \
    void update_weight(optimizer* opt);                                                                          \
  }
GAIB_GCONV_LAYER(GCN_layer, GCN_Aggregator, false, aggr.init(dim_in < dim_out ? dim_in : dim_out, nv));
GAIB_GCONV_LAYER(SAGE_layer, SAGE_Aggregator, true, aggr.init(dim_in < dim_out ? dim_in : dim_out, nv));
GAIB_GCONV_LAYER(GAT_layer, GAT_Aggregator, false, aggr.init(dim_out, nv, g->sizeEdges(), lr, score_drop_rate));
#undef GAIB_GCONV_LAYER

#include "ggnn_layer_stub.h"
