// include/gnn/loss_layer.h -- output (loss) layer base of the GNN model.
// Same public interface as the reference class (include/gnn/loss_layer.h:5-33); feat_in / feat_out /
// per-vertex losses live in HBM.
#pragma once
#include "global.h"

class loss_layer {
 public:
  loss_layer();
  loss_layer(int nv, int n_cls);
  loss_layer(int nv, int n_cls, label_t* ptr);
  virtual ~loss_layer() {}
  float* get_feat_in() { return feat_in; }
  float* get_feat_out() { return feat_out; }
  virtual void forward(size_t begin, size_t end, mask_t* masks) {}
  virtual void backward(size_t begin, size_t end, mask_t* masks, float* grad_out) {}
  void set_labels_ptr(label_t* ptr) { labels = ptr; }
  virtual acc_t get_prediction_loss(size_t begin, size_t end, size_t count, mask_t* masks) { return 0; }
  void set_netphase(net_phase phase) { phase_ = phase; }
  void update_dim_size(int sz);
  void print_layer_info() {
    std::cout << "Output Layer with " << num_samples << " samples and " << num_cls << " classes\n";
  }

 protected:
  void allocate(int nv);
  int num_samples;
  int num_cls;
  int capacity_;
  net_phase phase_;
  float* feat_in;    // logits, device [nv*num_cls]
  float* feat_out;   // probabilities, device
  label_t* labels;   // device
  acc_t* d_losses;   // per-vertex loss, device [nv]
};
