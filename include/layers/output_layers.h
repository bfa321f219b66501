// include/layers/output_layers.h -- the model head of the GNN trainer, device resident:
//   row_block        shared storage of the two head layers below (input rows + incoming gradient)
//   l2norm_layer     row-wise L2 normalisation (GAT / sampling put it in front of the dense head)
//   dense_layer      fully connected hid -> classes; backward() also applies its own Adam step
//   loss_layer       logits / probabilities / per-vertex loss buffers + the virtual loss interface
//   softmax_loss_layer   single-label: softmax + cross entropy, gradient (p - onehot) / (end - begin)
//   sigmoid_loss_layer   multi-label: sigmoid + cross entropy on [n x num_cls] 0/1 labels
// Class names and public methods are the ones the reference's drivers call (net.cpp:421-615;
// reference declarations: include/gnn/loss_layer.h, include/layers/{l2norm,dense,softmax_loss,
// sigmoid_loss}_layer.h).  The per-class headers of the same names just include this file.
#pragma once
#include "global.h"
#include "optimizer.h"

// [rows x width_in] activations and [rows x width_out] gradients in HBM, regrown on demand
// (subgraph sampling trains on small graphs and evaluates on the full one)
struct row_block {
  row_block(int rows, int width_in, int width_out);
  void resize(int rows);
  int rows, cap, win, wout;
  float* acts;
  float* grads;
};

class l2norm_layer {
 public:
  l2norm_layer(int nv, int len);
  void forward(float* feat_out);   // feat_out[i,:] = feat_in[i,:] / max(|feat_in[i,:]|, 1e-6)
  void backward(float* grad_out);  // grad_out from get_grad_in() and get_feat_in()
  float* get_feat_in() { return buf.acts; }
  float* get_grad_in() { return buf.grads; }
  void update_dim_size(int sz) { buf.resize(sz); }

 private:
  row_block buf;
};

class dense_layer {
 public:
  dense_layer(int nv, int in_len, int out_len, float lr);
  void forward(float* feat_out);   // feat_out = feat_in . W
  void backward(float* grad_out);  // dW = feat_in^T . grad_in ; grad_out = grad_in . W^T ; W <- adam(dW)
  float* get_feat_in() { return buf.acts; }
  float* get_grad_in() { return buf.grads; }
  float* weight_ptr() { return d_weight; }
  float* weight_grad_ptr() { return d_weight_grad; }
  void update_dim_size(int sz) { buf.resize(sz); }

 private:
  row_block buf;
  float* d_weight;       // [in_len x out_len], Glorot seed 1 like the OpenMP path
  float* d_weight_grad;
  optimizer* optm;
};

class loss_layer {
 public:
  loss_layer();
  loss_layer(int nv, int n_cls);
  loss_layer(int nv, int n_cls, label_t* ptr);
  virtual ~loss_layer() {}
  virtual void forward(size_t begin, size_t end, mask_t* masks) {}
  virtual void backward(size_t begin, size_t end, mask_t* masks, float* grad_out) {}
  virtual acc_t get_prediction_loss(size_t begin, size_t end, size_t count, mask_t* masks) { return 0; }
  float* get_feat_in() { return feat_in; }    // logits
  float* get_feat_out() { return feat_out; }  // probabilities
  float* loss_buffer() { return d_losses; }   // per-vertex losses of the last forward (device)
  void set_labels_ptr(label_t* ptr) { labels = ptr; }
  void set_netphase(net_phase phase) { phase_ = phase; }
  void update_dim_size(int sz);
  void print_layer_info() {
    std::cout << "Output Layer with " << num_samples << " samples and " << num_cls << " classes\n";
  }

 protected:
  void allocate(int nv);
  int num_samples, num_cls, capacity_;
  net_phase phase_;
  float *feat_in, *feat_out;
  acc_t* d_losses;  // per-vertex loss
  label_t* labels;  // device
};

#define GAIB_LOSS_LAYER(NAME)                                                                        \
  class NAME : public loss_layer {                                                                   \
   public:                                                                                           \
    NAME() {}                                                                                        \
    NAME(int nv, int n_cls) : loss_layer(nv, n_cls, NULL) {}                                         \
    NAME(int nv, int n_cls, label_t* ptr) : loss_layer(nv, n_cls, ptr) {}                            \
    void forward(size_t begin, size_t end, mask_t* masks) override;                                  \
    void backward(size_t begin, size_t end, mask_t* masks, float* grad_out) override;                \
    acc_t get_prediction_loss(size_t begin, size_t end, size_t count, mask_t* masks) override;       \
  }
GAIB_LOSS_LAYER(softmax_loss_layer);
GAIB_LOSS_LAYER(sigmoid_loss_layer);
#undef GAIB_LOSS_LAYER
