/*
 * gaib_layers.h -- C handle API over the host C++ mirror (libgaib_gnn.so).
 *
 * The reference's layer/operator API is C++ (include/layers/graph_conv_layer.h,
 * include/gnn/aggregator.h, include/gnn/lgraph.h); a C++ caller uses the classes of this repo's
 * include/{gnn,layers,utils} directly.  This header exposes the SAME objects through opaque
 * handles so that non-C++ harnesses (tests/, bench.py via ctypes) drive exactly the code path
 * a linked GraphAIBench driver would: GCN_layer / SAGE_layer / GAT_layer ::forward / backward /
 * update_weight on a LearningGraph.
 */
#ifndef GAIB_LAYERS_H
#define GAIB_LAYERS_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { GAIBL_GCN = 0, GAIBL_SAGE = 1, GAIBL_GAT = 2 };
/* gaibl_layer_ptr selectors (device pointers) */
enum {
  GAIBL_FEAT_IN = 0, GAIBL_GRAD_IN = 1, GAIBL_W_NEIGH = 2, GAIBL_W_NEIGH_GRAD = 3, GAIBL_W_SELF = 4,
  GAIBL_W_SELF_GRAD = 5, GAIBL_ALPHA_L = 6, GAIBL_ALPHA_R = 7, GAIBL_ALPHA_LGRAD = 8,
  GAIBL_ALPHA_RGRAD = 9, GAIBL_NORM_SCORES = 10, GAIBL_TEMP_SCORES = 11, GAIBL_SCORES = 12,
  GAIBL_NORM_SCORES_GRAD = 13,
  GAIBL_NORM_SCORES_DROPPED = 14, /* GAT, score_drop > 0: attention . mask . scale of the last training forward */
  GAIBL_ATTN_MASKS = 15           /* ... and its masks (uint8 [ne][heads], returned through the float* type) */
};

void gaibl_init(int device, void* hip_stream); /* gpu_context::set */
void* gaibl_ctx(void);                         /* gaib_ctx* of the process */
void gaibl_sync(void);

/* LearningGraph* built the way Model::load_data does (net.cpp:88-203): host CSR -> optional
 * add_selfloop -> copy_to_gpu -> compute_vertex_data */
void* gaibl_graph_from_host(uint32_t nv, uint32_t ne, const uint32_t* rowptr, const uint32_t* colidx,
                            int add_selfloop);
void* gaibl_graph_adopt(void* gaib_graph_handle); /* wrap a gaib_graph that already lives in HBM */
void* gaibl_graph_device(void* graph);            /* gaib_graph* */
uint64_t gaibl_graph_num_edges(void* graph);
void gaibl_graph_free(void* graph);
/* vertex-range partitions (LearningGraph::set_halo): `halo_graph` is the gaib_graph of the
 * halo-column edges; begin() packs + starts the exchange, end() waits and returns the halo table */
typedef void (*gaibl_halo_begin_fn)(void* user, int len, const float* d_in);
typedef const float* (*gaibl_halo_end_fn)(void* user, int len);
void gaibl_graph_set_halo(void* graph, void* halo_graph, gaibl_halo_begin_fn begin, gaibl_halo_end_fn end,
                          void* user);

/* vertex-range partition built on the host (include/gnn/partition.h): every rank derives its share from the global
 * CSR.  gaibl_partition_array: which = 0 rowptr_own (int64) 1 colidx_own (uint32) 2 rowptr_halo (int64) 3 colidx_halo
 * (uint32) 4 degree 5 halo_gids 6 halo_degree 7 recv_counts 8 send_counts 9 send_idx (all int64); returns the element
 * count, *data points into the partition object.  gaibl_partition_make_graph uploads it and creates the halo plan on
 * `comm` (a gaib_comm*, NULL for world 1) -> LearningGraph*.  gaibl_set_comm: gpu_context::set_comm (every optimizer
 * step all-reduces its gradient first). */
void* gaibl_partition_build(uint32_t nv, const uint32_t* rowptr, const uint32_t* colidx, int rank, int world);
/* GAT: adds the [owned | halo] column space and its transpose (which = 10 rowptr_full 11 colidx_full 12 rowptr_t
 * 13 colidx_t 14 tperm); gaibl_partition_make_graph then builds a graph GAT layers can run on */
void gaibl_partition_build_gat(void* part, const uint32_t* rowptr, const uint32_t* colidx);
int64_t gaibl_partition_array(void* part, int which, const void** data);
void gaibl_partition_range(void* part, int64_t* lo, int64_t* hi);
void gaibl_partition_free(void* part);
void* gaibl_partition_make_graph(void* part, void* comm);
void gaibl_set_comm(void* comm);
void* gaibl_graph_halo_plan(void* graph); /* gaib_halo* or NULL */
/* LearningGraph::set_halo_plan: the halo-column gaib_graph + the gaib_halo plan every aggregation runs */
void gaibl_graph_set_halo_plan(void* graph, void* halo_graph, void* plan);

void* gaibl_layer_create(int kind, int level, int nv, int din, int dout, void* graph, int act, float lr,
                         float feat_drop, float score_drop);
void gaibl_layer_free(void* layer); /* gconv_state::release + the aggregator's: every device buffer of the layer */
void gaibl_layer_forward(void* layer, float* d_feat_out);
void gaibl_layer_backward(void* layer, float* d_feat_out, float* d_grad_out);
void gaibl_layer_update_weight(void* layer, void* optimizer);
void gaibl_layer_set_feat_in(void* layer, float* d_ptr);
void gaibl_layer_set_phase(void* layer, int phase); /* 0 TRAIN 1 TEST 2 VAL */
void gaibl_layer_set_heads(void* layer, int heads); /* GAT only: GAT_Aggregator::set_num_heads */
void gaibl_layer_set_input_constant(void* layer, int on); /* gconv_state::set_input_constant (layer 0, full batch) */
float* gaibl_layer_ptr(void* layer, int which);

/* host-only test hook for the GraphSAINT sampler (include/gnn/sampler.h): samples up to n vertices
 * with frontier size m from the training-masked graph and builds the induced, re-indexed subgraph.
 * Outputs are malloc'ed arrays the caller frees with gaibl_free_host. Returns the subgraph's nv. */
uint32_t gaibl_sample_subgraph(uint32_t nv, uint32_t ne, const uint32_t* rowptr, const uint32_t* colidx,
                               const uint8_t* train_masks, uint32_t n, uint32_t m, unsigned seed,
                               uint32_t** sub_rowptr, uint32_t** sub_colidx, uint32_t** kept_ids);
void gaibl_free_host(void* p);

/* row classes of a partitioned graph (LearningGraph::partition_mode, include/gnn/lgraph.h): mode 0 = column split over
 * all rows, 1 = interior rows in one pass + column split of the boundary rows, 2 = interior rows in one pass + the
 * boundary rows in one pass over [owned | halo]; -1 = by the rule (default; GAIB_PART_MODE overrides).
 * gaibl_graph_partition_mode decides (once per graph, for aggregations of `len` columns), builds the class graphs and
 * reports the mode, the boundary rows and their edges. */
void gaibl_graph_set_partition_mode(void* graph, int mode);
void gaibl_graph_set_halo_link_rows(void* graph, int64_t rows); /* callback transports: rows one peer pair moves per exchange */
int gaibl_graph_partition_mode(void* graph, int len, int64_t* n_boundary, int64_t* boundary_edges);

/* the halo-column half in pieces (LearningGraph::set_halo_pieces / halo_pieces, round 6).  A plan puts K slices on the wire
 * (gaib_halo_set_pieces); a callback transport names its slices here: range j = [begin[j], end[j]) of the halo table arrives in
 * slice piece[j], wait_piece(user, k) returns the table once slice k is there (stream-ordered; `user` is gaibl_graph_set_halo's).
 * A rank consumes the K slices in K' | K pieces: gaibl_graph_set_halo_consumption(K') forces it (GAIB_HALO_CONSUME too), -1 =
 * by the library's rule.  gaibl_graph_halo_pieces: K' for aggregations of `len` columns right now (1 = the whole half after the
 * exchange; decided after gaibl_graph_partition_mode). */
typedef const float* (*gaibl_halo_wait_piece_fn)(void* user, int piece);
void gaibl_graph_set_halo_pieces(void* graph, int n_pieces, int n_ranges, const int64_t* begin, const int64_t* end,
                                 const int* piece, gaibl_halo_wait_piece_fn wait_piece);
void gaibl_graph_set_halo_consumption(void* graph, int pieces);
int gaibl_graph_halo_pieces(void* graph, int len);

void* gaibl_adam_create(float lr);
void gaibl_adam_free(void* opt);

double gaibl_time_op(char op); /* time_ops[op] (seconds); needs GAIB_SYNC_TIMERS=1 */
void gaibl_reset_timers(void);

#ifdef __cplusplus
}
#endif
#endif
