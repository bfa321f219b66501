// layers.cpp -- graph convolution layers (GCN / SAGE / GAT) over the aggregators and matmul.
// Control flow follows the layer definitions of the reference (src/gnn/gconv/{gcn,sage,gat}_layer.cpp,
// src/gnn/graph_conv_layer.cpp): which of GEMM / aggregation runs first, what is cached for the
// weight gradient, the in-place d_relu on grad_in, and layer 0 skipping the input gradient.
#include "graph_conv_layer.h"
#include "host_util.h"
#include "math_functions.hh"

static inline gaib_ctx* C() { return gpu_context::get(); }

gconv_state::gconv_state(int id, int nv, int din, int dout, Graph* g, bool act, bool concat, float lr,
                         float feat_drop, float score_drop)
    : level_(id), num_samples(nv), dim_in(din), dim_out(dout), graph(g), is_act(act), is_bias(false),
      use_concat(concat), feat_dropout_rate(feat_drop), score_dropout_rate(score_drop),
      phase_(net_phase::TRAIN), capacity_((size_t)nv), dropout_calls(0), feat_in(NULL), grad_in(NULL),
      d_in_temp(NULL), d_in_temp1(NULL), d_out_temp(NULL), d_W_neigh(NULL), d_W_neigh_grad(NULL),
      d_W_self(NULL), d_W_self_grad(NULL), dropout_mask(NULL), optm(NULL) {
  assert(feat_dropout_rate >= 0. && feat_dropout_rate < 1.);
  assert(score_dropout_rate >= 0. && score_dropout_rate < 1.);
  feat_scale = 1. / (1. - feat_dropout_rate);
  const size_t nin = (size_t)nv * din, nout = (size_t)nv * dout, nw = (size_t)din * dout;
  // weights: Glorot with seed 1 for W_neigh of EVERY layer and seed 2 for W_self (Q7)
  vec_t w;
  init_glorot(din, dout, w, 1);
  d_W_neigh = gaib_host::dmalloc<float>(nw);
  d_W_neigh_grad = gaib_host::dmalloc<float>(nw);
  GAIB_OR_DIE(gaib_memcpy_h2d(C(), d_W_neigh, w.data(), sizeof(float) * nw));
  GAIB_OR_DIE(gaib_fill_f32(C(), nw, 0.f, d_W_neigh_grad));
  if (concat) {
    init_glorot(din, dout, w, 2);
    d_W_self = gaib_host::dmalloc<float>(nw);
    d_W_self_grad = gaib_host::dmalloc<float>(nw);
    GAIB_OR_DIE(gaib_memcpy_h2d(C(), d_W_self, w.data(), sizeof(float) * nw));
    GAIB_OR_DIE(gaib_fill_f32(C(), nw, 0.f, d_W_self_grad));
  }
  d_in_temp = gaib_host::dmalloc<float>(nin);
  d_out_temp = gaib_host::dmalloc<float>(nout);
  GAIB_OR_DIE(gaib_fill_f32(C(), nin, 0.f, d_in_temp));
  GAIB_OR_DIE(gaib_fill_f32(C(), nout, 0.f, d_out_temp));
  if (din <= dout) {
    d_in_temp1 = gaib_host::dmalloc<float>(nin);
    GAIB_OR_DIE(gaib_fill_f32(C(), nin, 0.f, d_in_temp1));
  }
  if (level_ > 0) {
    feat_in = gaib_host::dmalloc<float>(nin);
    GAIB_OR_DIE(gaib_fill_f32(C(), nin, 0.f, feat_in));
  }
  grad_in = gaib_host::dmalloc<float>(nout);
  GAIB_OR_DIE(gaib_fill_f32(C(), nout, 0.f, grad_in));
  if (feat_dropout_rate > 0.) dropout_mask = gaib_host::dmalloc<mask_t>(nin);
  optm = new adam(lr);
}

// see include/layers/graph_conv_layer.h: there is no destructor (layers are copied by value)
void gconv_state::release() {
  gaib_ctx* c = C();
  float** owned[] = {&grad_in, &d_in_temp, &d_in_temp1, &d_out_temp, &d_W_neigh, &d_W_neigh_grad, &d_W_self, &d_W_self_grad};
  for (float** p : owned) {
    if (*p) GAIB_OR_DIE(gaib_free(c, *p));
    *p = NULL;
  }
  if (level_ > 0 && feat_in) GAIB_OR_DIE(gaib_free(c, feat_in));  // level 0's input belongs to the model (set_feat_in)
  feat_in = NULL;
  if (dropout_mask) GAIB_OR_DIE(gaib_free(c, dropout_mask));
  dropout_mask = NULL;
  if (optm) {
    optm->reset();  // its per-weight moment buffers
    delete optm;
    optm = NULL;
  }
  capacity_ = 0;
  agg_valid_ = false;
}

// number of rows changes with subgraph sampling (training on subgraphs, evaluation on the full
// graph); buffers grow when needed (reference GPU build: src/gnn/graph_conv_layer.cu:57-83)
void gconv_state::update_dim_size(size_t x) {
  if (x > capacity_) {
    const size_t nin = x * dim_in, nout = x * dim_out;
    auto regrow = [&](float*& p, size_t n) {
      if (!p) return;
      GAIB_OR_DIE(gaib_free(C(), p));
      p = gaib_host::dmalloc<float>(n);
      GAIB_OR_DIE(gaib_fill_f32(C(), n, 0.f, p));
    };
    regrow(d_in_temp, nin);
    regrow(d_in_temp1, nin);
    regrow(d_out_temp, nout);
    if (level_ > 0) regrow(feat_in, nin);
    regrow(grad_in, nout);
    if (dropout_mask) {
      GAIB_OR_DIE(gaib_free(C(), dropout_mask));
      dropout_mask = gaib_host::dmalloc<mask_t>(nin);
    }
    capacity_ = x;
  }
  num_samples = (int)x;
  agg_valid_ = false;
}

template class graph_conv_layer<GCN_Aggregator>;
template class graph_conv_layer<SAGE_Aggregator>;
template class graph_conv_layer<GAT_Aggregator>;

// ---- GCN ---------------------------------------------------------------------------------------
void GCN_layer::forward(float* feat_out) {
  const size_t x = num_samples, y = dim_in, z = dim_out;
  float* in_data = feat_in;
  if (feat_dropout_rate > 0. && phase_ == net_phase::TRAIN) {
    dropout_gpu(x * y, feat_scale, feat_dropout_rate, in_data, dropout_mask, d_in_temp);
    in_data = d_in_temp;
  }
  // the activation (relu_gpu, gcn_layer.cpp:27) is fused into whichever op produces feat_out
  if (y > z) {  // shrink first, aggregate the narrow matrix
    matmul(x, z, y, in_data, d_W_neigh, d_out_temp);
    if (is_act) aggr.fuse_relu_once();
    aggr.aggregate(z, *graph, d_out_temp, feat_out);
  } else if (input_constant_ && agg_valid_ && in_data == feat_in) {
    // constant input (set_input_constant): A.X of an earlier forward is still in d_in_temp1, only the product is left
    if (is_act) matmul_relu(x, z, y, d_in_temp1, d_W_neigh, feat_out);
    else matmul(x, z, y, d_in_temp1, d_W_neigh, feat_out);
  } else {  // aggregate first; A.X is kept for the weight gradient.  One kernel: the product rides on the aggregation
    aggr.aggregate_matmul(y, *graph, in_data, d_in_temp1, true, d_W_neigh, false, z, feat_out, is_act);
    agg_valid_ = input_constant_ && in_data == feat_in;
  }
}

void GCN_layer::backward(float* feat_out, float* grad_out) {
  const size_t x = num_samples, y = dim_in, z = dim_out;
  // d_relu (gcn_layer.cpp:35) runs in place on grad_in with the post-activation output as mask (Q9)
  if (y > z) {
    if (is_act) d_relu_gpu(x * z, grad_in, feat_out, grad_in);
    if (level_ > 0) aggr.d_aggregate_matmul(z, *graph, grad_in, d_out_temp, true, d_W_neigh, true, y, grad_out);
    else aggr.d_aggregate(z, *graph, NULL, grad_in, d_out_temp);
    float* in_data = feat_dropout_rate > 0. ? d_in_temp : feat_in;
    matmul(y, z, x, in_data, d_out_temp, d_W_neigh_grad, true, false);
  } else {
    // the weight gradient goes first and applies the d_relu while it streams grad_in (one pass instead of two)
    if (is_act) matmul_drelu(y, z, x, d_in_temp1, grad_in, feat_out, d_W_neigh_grad);
    else matmul(y, z, x, d_in_temp1, grad_in, d_W_neigh_grad, true, false);
    if (level_ > 0) {
      if (y == z) {
        // A.(g.W^T) == (A.g).W^T: at equal widths the aggregation goes first and carries the product
        // (the reference order would cost a separate GEMM pass; same result up to summation order)
        aggr.d_aggregate_matmul(z, *graph, grad_in, d_in_temp, false, d_W_neigh, true, y, grad_out);
      } else {
        matmul(x, y, z, grad_in, d_W_neigh, d_in_temp, false, true);
        aggr.d_aggregate(y, *graph, NULL, d_in_temp, grad_out);
      }
    }
  }
  if (level_ != 0 && feat_dropout_rate > 0.)
    d_dropout_gpu(x * y, feat_scale, grad_out, dropout_mask, grad_out);
}

void GCN_layer::update_weight(optimizer* opt) {
  opt->update_gpu((size_t)dim_in * dim_out, d_W_neigh_grad, d_W_neigh);  // the model's shared optimizer (Q6)
}

// ---- SAGE --------------------------------------------------------------------------------------
void SAGE_layer::forward(float* feat_out) {
  const size_t x = num_samples, y = dim_in, z = dim_out;
  float* in_data = feat_in;
  if (feat_dropout_rate > 0. && phase_ == net_phase::TRAIN) {
    dropout_gpu(x * y, feat_scale, feat_dropout_rate, in_data, dropout_mask, d_in_temp);
    in_data = d_in_temp;
  }
  if (y > z) {
    matmul(x, z, y, in_data, d_W_neigh, d_out_temp);
    aggr.aggregate(z, *graph, d_out_temp, feat_out);
    // + X.W_self, with the activation fused into this last product
    if (is_act) matmul_relu(x, z, y, in_data, d_W_self, feat_out, false, false, true);
    else matmul(x, z, y, in_data, d_W_self, feat_out, false, false, true);
  } else if (input_constant_ && agg_valid_ && in_data == feat_in) {
    // constant input (set_input_constant): the mean-aggregated input of an earlier forward is still in d_in_temp1
    matmul(x, z, y, d_in_temp1, d_W_neigh, feat_out);
    if (is_act) matmul_relu(x, z, y, in_data, d_W_self, feat_out, false, false, true);
    else matmul(x, z, y, in_data, d_W_self, feat_out, false, false, true);
  } else {
    // one kernel: mean aggregation, then out = act(mean . W_neigh + X . W_self) on the matrix cores
    aggr.aggregate_matmul(y, *graph, in_data, d_in_temp1, true, d_W_neigh, false, z, feat_out, is_act, in_data,
                          d_W_self);
    agg_valid_ = input_constant_ && in_data == feat_in;
  }
}

void SAGE_layer::backward(float* feat_out, float* grad_out) {
  const size_t x = num_samples, y = dim_in, z = dim_out;
  float* in_data = feat_dropout_rate > 0. ? d_in_temp : feat_in;
  // the first product that streams grad_in applies the layer's d_relu on the way (in place, Q9)
  if (is_act) matmul_drelu(y, z, x, in_data, grad_in, feat_out, d_W_self_grad);
  else matmul(y, z, x, in_data, grad_in, d_W_self_grad, true, false);
  // grad_out = M^T-aggregated gradient . W_neigh^T + g . W_self^T: where the aggregation comes first, both
  // products ride on it (sage_layer.cpp:44-50 runs them as two GEMMs after / before the aggregation)
  if (y > z) {
    if (level_ > 0)
      aggr.d_aggregate_matmul(z, *graph, grad_in, d_out_temp, true, d_W_neigh, true, y, grad_out, grad_in, d_W_self);
    else aggr.d_aggregate(z, *graph, NULL, grad_in, d_out_temp);
    matmul(y, z, x, in_data, d_out_temp, d_W_neigh_grad, true, false);
  } else {
    matmul(y, z, x, d_in_temp1, grad_in, d_W_neigh_grad, true, false);
    if (level_ > 0) {
      if (y == z) {  // (M^T g).W^T instead of M^T (g.W^T): the products ride on the aggregation
        aggr.d_aggregate_matmul(z, *graph, grad_in, d_in_temp, false, d_W_neigh, true, y, grad_out, grad_in, d_W_self);
      } else {
        matmul(x, y, z, grad_in, d_W_neigh, d_in_temp, false, true);
        aggr.d_aggregate(y, *graph, NULL, d_in_temp, grad_out);
        matmul(x, y, z, grad_in, d_W_self, grad_out, false, true, true);  // += g.W_self^T
      }
    }
  }
  if (level_ != 0 && feat_dropout_rate > 0.)
    d_dropout_gpu(x * y, feat_scale, grad_out, dropout_mask, grad_out);
}

void SAGE_layer::update_weight(optimizer*) {
  optm->update_gpu((size_t)dim_in * dim_out, d_W_neigh_grad, d_W_neigh);  // the layer's own optimizer
  optm->update_gpu((size_t)dim_in * dim_out, d_W_self_grad, d_W_self);
}

// ---- GAT ---------------------------------------------------------------------------------------
void GAT_layer::forward(float* feat_out) {
  const size_t x = num_samples, y = dim_in, z = dim_out;
  float* in_data = feat_in;
  if (feat_dropout_rate > 0. && phase_ == net_phase::TRAIN) {
    dropout_gpu(x * y, feat_scale, feat_dropout_rate, in_data, dropout_mask, d_in_temp);
    in_data = d_in_temp;
  }
  matmul(x, z, y, in_data, d_W_neigh, d_out_temp);    // h = X.W
  aggr.set_training(phase_ == net_phase::TRAIN);      // attention dropout (score_drop), like feat_drop, only while training
  if (is_act) aggr.fuse_relu_once();
  aggr.aggregate(z, *graph, d_out_temp, feat_out);    // attention over h (+ relu)
}

void GAT_layer::backward(float* feat_out, float* grad_out) {
  const size_t x = num_samples, y = dim_in, z = dim_out;
  if (is_act) d_relu_gpu(x * z, grad_in, feat_out, grad_in);
  float* in_data = feat_dropout_rate > 0. ? d_in_temp : feat_in;
  // out_temp holds h on entry and the aggregated gradient on exit
  aggr.use_forward_output_once(feat_out);
  aggr.d_aggregate(z, *graph, d_out_temp, grad_in, d_out_temp);
  if (level_ != 0) {
    matmul(x, y, z, d_out_temp, d_W_neigh, grad_out, false, true);
    if (feat_dropout_rate > 0.) d_dropout_gpu(x * y, feat_scale, grad_out, dropout_mask, grad_out);
  }
  matmul(y, z, x, in_data, d_out_temp, d_W_neigh_grad, true);
}

void GAT_layer::update_weight(optimizer* opt) {
  opt->update_gpu((size_t)dim_in * dim_out, d_W_neigh_grad, d_W_neigh);
  aggr.update_weights(opt);
}
