// include/gnn/aggregator.h -- the operator classes of the GNN layer path.
// init / aggregate / d_aggregate (/ update_weights) with the reference's signatures
// (include/gnn/aggregator.h:21-88); `in`/`out` are DEVICE pointers, `out` is fully overwritten.
// Each call lowers to gaib_spmm / gaib_gat_* (include/gaib.h) on the process context.
#pragma once
#include "lgraph.h"
#include "math_functions.hh"
#include "optimizer.h"

class aggregator {
 public:
  aggregator() : n(0), length(0), fuse_relu(false) {}
  void set_vlen(int vlen) { length = vlen; }
  // extension: the next aggregate() call clamps its output at 0 (the layer's relu_gpu fused
  // into the aggregation's store); cleared by that call
  void fuse_relu_once() { fuse_relu = true; }
  // extension: free what the operator allocated (nothing for GCN / SAGE; GAT: attention vectors, per-edge arrays,
  // partition tables, its Adam state).  See gconv_state::release.
  void release() {}

 protected:
  // extension shared by the GCN / SAGE operators: agg = Op.in followed by out = act(agg . op(W)) -- the
  // layers' "aggregate first" branches -- as ONE kernel (gaib_spmm_gemm: the product runs on the matrix
  // cores inside the aggregating wave).  W is [len x len_out], or [len_out x len] with transW.  keep_agg = false
  // lets the kernel skip the store of agg (still needs the buffer).  On a partitioned graph (halo exchange)
  // it runs as aggregation + matmul.
  // rows2 / W2 (both or neither): + rows2 . op(W2) in the same store (the self term of a SAGE layer)
  void aggregate_then_matmul(int kind, int len, Graph& g, const float* in, float* agg, bool keep_agg,
                             const float* W, bool transW, int len_out, float* out, bool relu,
                             const float* rows2 = NULL, const float* W2 = NULL);

  int n;
  int length;  // feature vector length
  bool fuse_relu;
};

// out[i,:] = sum_e (vd[i]*vd[col_e]) * in[col_e,:]; backward is the same operator (symmetric)
class GCN_Aggregator : public aggregator {
 public:
  void init(int length, int nv, int ne = 0, float lr = 0.01, float drop_rate = 0.);
  void aggregate(int len, Graph& g, const float* in, float* out);
  void d_aggregate(int len, Graph& g, const float* feat_in, const float* grad_in, float* grad_out);
  // aggregate / d_aggregate fused with the following matmul (see aggregator::aggregate_then_matmul)
  void aggregate_matmul(int len, Graph& g, const float* in, float* agg, bool keep_agg, const float* W,
                        bool transW, int len_out, float* out, bool relu);
  void d_aggregate_matmul(int len, Graph& g, const float* grad_in, float* agg, bool keep_agg, const float* W,
                          bool transW, int len_out, float* out);
};

// forward: mean over neighbours (1/deg(i)); backward: its transpose (1/deg(col_e))
class SAGE_Aggregator : public aggregator {
 public:
  void init(int length, int nv, int ne = 0, float lr = 0.01, float drop_rate = 0.);
  void aggregate(int len, Graph& g, const float* in, float* out);
  void d_aggregate(int len, Graph& g, const float* feat_in, const float* grad_in, float* grad_out);
  // out = act(mean(in) . op(W) + rows_self . op(W_self)): neighbour and self product of the layer in one kernel
  void aggregate_matmul(int len, Graph& g, const float* in, float* agg, bool keep_agg, const float* W,
                        bool transW, int len_out, float* out, bool relu, const float* rows_self = NULL,
                        const float* W_self = NULL);
  void d_aggregate_matmul(int len, Graph& g, const float* grad_in, float* agg, bool keep_agg, const float* W,
                          bool transW, int len_out, float* out, const float* rows_self = NULL,
                          const float* W_self = NULL);
};

// single-head attention: p = softmax_row(leaky_relu_0.2(a_l.h_i + a_r.h_j)); out = P h.
// d_aggregate: alpha gradients + P^T g; no gradient through the scores into h (Q18).
class GAT_Aggregator : public aggregator {
 public:
  GAT_Aggregator();
  void init(int length, int nv, int ne = 0, float lr = 0.01, float drop_rate = 0.);
  void aggregate(int len, Graph& g, const float* in, float* out);
  void d_aggregate(int len, Graph& g, const float* feat_in, const float* grad_in, float* grad_out);
  void update_weights(optimizer* opt);
  void release();
  // extension (BASELINE config "GAT 8-head"; the reference is single-head): h independent attentions
  // on the column slices of width length/h.  Call right after init().
  void set_num_heads(int h);
  int num_heads() const { return heads; }
  // extension: the next d_aggregate() may read the layer's forward output rows `out` (post-activation is fine as
  // long as grad_in went through the matching d_relu): it replaces the per-row sum_e p_e dp_e by <grad_i, out_i>
  void use_forward_output_once(const float* out) { fwd_out = out; fwd_out_given = true; }
  // attention dropout (score_drop > 0) is applied while training only, like the layers' feature dropout; the layer
  // passes its phase on before every forward
  void set_training(bool on) { training = on; }
  // device state (tests / checkpoints)
  float* alpha_l_ptr() { return d_alpha_l; }
  float* alpha_r_ptr() { return d_alpha_r; }
  float* alpha_lgrad_ptr() { return d_alpha_lgrad; }
  float* alpha_rgrad_ptr() { return d_alpha_rgrad; }
  float* norm_scores_ptr();  // (materialised on demand after a one-sweep forward, which keeps only row statistics)
  float* temp_scores_ptr() { return d_temp_scores; }  // NULL for 4, 8, 16 heads: not materialised either
  float* scores_ptr() { return NULL; }  // leaky_relu(temp_scores): not materialised by this backend
  float* norm_scores_grad_ptr() { return d_norm_scores_grad; }
  float* norm_scores_dropped_ptr() { return d_norm_scores_drop; }  // NULL unless a training forward dropped attention
  mask_t* attn_masks_ptr() { return d_attn_masks; }

 private:
  // the temp_scores array is kept where re-forming the score is not a gain: 1 or 2 heads (measured on the
  // reddit-shaped graph: 4.0 vs 4.7 ms for scores + softmax backward at 1 head, 10.5 vs 10.1 ms at 8 heads)
  bool needs_temp() const { return !(heads == 4 || heads == 8 || heads == 16); }
  float epsilon;    // LeakyReLU negative slope (0.2)
  // attention dropout: the normalised scores of a training forward are masked and rescaled (the reference's CUDA path,
  // graph_operations.h:326-331; its OpenMP path has the call commented out, gat_aggregator.cpp:78-79), and backward goes
  // through the SAME mask: dp_e is masked and rescaled before the softmax backward (d_dropout, graph_operations.h:376-377
  // -- the `_naive` form; the `_warp` form the reference launches has it commented out at :419-422), and the transposed
  // aggregation uses the dropped attention.  Masks from the library's counter RNG (gaib_dropout), one per (edge, head).
  float attn_drop, attn_scale;
  bool training, dropped_last;  // dropped_last: the last forward applied a mask (backward must take the staged path)
  float* d_norm_scores_drop;    // [ne][heads] p . mask . scale of the last training forward
  mask_t* d_attn_masks;         // [ne][heads]
  size_t drop_cap;
  uint64_t drop_seed;
  bool dropping() const { return attn_drop > 0.f && training; }
  const float* apply_attn_dropout(size_t n_scores);  // d_norm_scores -> d_norm_scores_drop (+ masks); returns the latter
  size_t num_edges;
  int heads;
  float *d_alpha_l, *d_alpha_r, *d_alpha_lgrad, *d_alpha_rgrad;
  float *d_temp_scores, *d_norm_scores, *d_norm_scores_grad, *d_norm_scores_t;
  const float* fwd_out;  // see use_forward_output_once
  bool fwd_out_given = false;  // ... was called (a rank WITHOUT rows passes NULL and must still take the path the others take)
  float* d_tbuf;         // output of the fused backward sweep (the layer aliases feat_in and grad_out)
  size_t tbuf_floats;
  // vertex-range partitions: the [owned | halo] column table, the transposed aggregation's output, padded row sums and
  // the column sums of g
  float *d_ptab, *d_pout, *d_prs, *d_pcs;
  size_t ptab_floats, pvec_floats;
  // ... and of the one-sweep path on a partition: the [owned | halo] tables of gradient rows and (rowdot, max, 1/sum) records
  float *d_pgrad, *d_prec;
  size_t pgrad_floats, prec_floats;
  bool part_fused_last;  // the last partition forward was the one-sweep kernel (row statistics only)
  void ensure_partition_buffers(Graph& g, int len);
  void aggregate_partition(int len, Graph& g, const float* in, float* out);
  void d_aggregate_partition(int len, Graph& g, const float* grad_in, float* grad_out);
  // one-sweep forward (gaib_gat_forward_fused): per (row, head) the softmax's maximum and 1 / sum
  float* d_row_stats;
  size_t stats_floats;
  bool stats_valid;   // the last forward kept statistics instead of the attention array
  Graph* last_graph;  // ... of this graph and input, should the array be asked for
  const float* last_in;
  int last_len;
  void materialise_scores();
  optimizer* alpha_opt;
};
