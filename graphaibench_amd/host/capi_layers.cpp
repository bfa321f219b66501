// capi_layers.cpp -- opaque-handle C API over the host C++ classes (include/gaib_layers.h).
#include "gaib_layers.h"
#include "partition.h"
#include "graph_conv_layer.h"
#include "host_util.h"
#include "sampler.h"

namespace {
struct LayerBox {
  int kind;
  GCN_layer* gcn;
  SAGE_layer* sage;
  GAT_layer* gat;
};

template <typename L>
static float* common_ptr(L* l, int which) {
  switch (which) {
    case GAIBL_FEAT_IN: return l->get_feat_in();
    case GAIBL_GRAD_IN: return l->get_grad_in();
    case GAIBL_W_NEIGH: return l->weight_neigh_ptr();
    case GAIBL_W_NEIGH_GRAD: return l->weight_neigh_grad_ptr();
    case GAIBL_W_SELF: return l->weight_self_ptr();
    case GAIBL_W_SELF_GRAD: return l->weight_self_grad_ptr();
    default: return nullptr;
  }
}
}  // namespace

extern "C" {

void gaibl_init(int device, void* hip_stream) { gpu_context::set(device, hip_stream); }
void* gaibl_ctx(void) { return gpu_context::get(); }
void gaibl_sync(void) { gpu_context::sync(); }

void* gaibl_graph_from_host(uint32_t nv, uint32_t ne, const uint32_t* rowptr, const uint32_t* colidx,
                            int add_selfloop) {
  Graph* g = new Graph(true);
  g->allocateFrom(nv, ne);
  memcpy(g->row_host_ptr(), rowptr, sizeof(uint32_t) * ((size_t)nv + 1));
  if (ne) memcpy(g->edge_host_ptr(), colidx, sizeof(uint32_t) * ne);
  if (add_selfloop) g->add_selfloop();
  g->degree_counting();
  g->alloc_on_device();
  g->copy_to_gpu();
  g->compute_vertex_data();
  return g;
}
void* gaibl_graph_adopt(void* h) { return LearningGraph::adopt_device(static_cast<gaib_graph*>(h)); }
void* gaibl_graph_device(void* graph) { return static_cast<Graph*>(graph)->device_graph(); }
uint64_t gaibl_graph_num_edges(void* graph) { return static_cast<Graph*>(graph)->sizeEdges(); }
void gaibl_graph_set_halo(void* graph, void* halo_graph, gaibl_halo_begin_fn begin, gaibl_halo_end_fn end,
                          void* user) {
  static_cast<Graph*>(graph)->set_halo(static_cast<gaib_graph*>(halo_graph), begin, end, user);
}
void gaibl_graph_free(void* graph) {
  Graph* g = static_cast<Graph*>(graph);
  g->dealloc();
  delete g;
}

void* gaibl_layer_create(int kind, int level, int nv, int din, int dout, void* graph, int act, float lr,
                         float feat_drop, float score_drop) {
  Graph* g = static_cast<Graph*>(graph);
  LayerBox* b = new LayerBox{kind, nullptr, nullptr, nullptr};
  if (kind == GAIBL_GCN) b->gcn = new GCN_layer(level, nv, din, dout, g, act != 0, lr, feat_drop, score_drop);
  else if (kind == GAIBL_SAGE) b->sage = new SAGE_layer(level, nv, din, dout, g, act != 0, lr, feat_drop, score_drop);
  else if (kind == GAIBL_GAT) b->gat = new GAT_layer(level, nv, din, dout, g, act != 0, lr, feat_drop, score_drop);
  else {
    fprintf(stderr, "gaibl_layer_create: unknown kind %d\n", kind);
    exit(EXIT_FAILURE);
  }
  return b;
}

void gaibl_layer_free(void* layer) {
  LayerBox* b = static_cast<LayerBox*>(layer);
  if (!b) return;
  if (b->gcn) b->gcn->release_all(), delete b->gcn;
  if (b->sage) b->sage->release_all(), delete b->sage;
  if (b->gat) b->gat->release_all(), delete b->gat;
  delete b;
}

#define DISPATCH(b, call)                          \
  do {                                             \
    if ((b)->kind == GAIBL_GCN) (b)->gcn->call;    \
    else if ((b)->kind == GAIBL_SAGE) (b)->sage->call; \
    else (b)->gat->call;                           \
  } while (0)

void gaibl_layer_forward(void* layer, float* d_feat_out) {
  LayerBox* b = static_cast<LayerBox*>(layer);
  DISPATCH(b, forward(d_feat_out));
}
void gaibl_layer_backward(void* layer, float* d_feat_out, float* d_grad_out) {
  LayerBox* b = static_cast<LayerBox*>(layer);
  DISPATCH(b, backward(d_feat_out, d_grad_out));
}
void gaibl_layer_update_weight(void* layer, void* opt) {
  LayerBox* b = static_cast<LayerBox*>(layer);
  DISPATCH(b, update_weight(static_cast<optimizer*>(opt)));
}
void gaibl_layer_set_feat_in(void* layer, float* p) {
  LayerBox* b = static_cast<LayerBox*>(layer);
  DISPATCH(b, set_feat_in(p));
}
void gaibl_layer_set_input_constant(void* layer, int on) {
  LayerBox* b = static_cast<LayerBox*>(layer);
  DISPATCH(b, set_input_constant(on != 0));
}
void gaibl_layer_set_heads(void* layer, int heads) {
  LayerBox* b = static_cast<LayerBox*>(layer);
  if (b->kind != GAIBL_GAT) {
    fprintf(stderr, "gaibl_layer_set_heads: not a GAT layer\n");
    exit(EXIT_FAILURE);
  }
  b->gat->get_aggregator().set_num_heads(heads);
}
void gaibl_layer_set_phase(void* layer, int phase) {
  LayerBox* b = static_cast<LayerBox*>(layer);
  net_phase ph = phase == 0 ? net_phase::TRAIN : (phase == 1 ? net_phase::TEST : net_phase::VAL);
  DISPATCH(b, set_netphase(ph));
}

float* gaibl_layer_ptr(void* layer, int which) {
  LayerBox* b = static_cast<LayerBox*>(layer);
  if (which <= GAIBL_W_SELF_GRAD) {
    if (b->kind == GAIBL_GCN) return common_ptr(b->gcn, which);
    if (b->kind == GAIBL_SAGE) return common_ptr(b->sage, which);
    return common_ptr(b->gat, which);
  }
  if (b->kind != GAIBL_GAT) return nullptr;
  GAT_Aggregator& a = b->gat->get_aggregator();
  switch (which) {
    case GAIBL_ALPHA_L: return a.alpha_l_ptr();
    case GAIBL_ALPHA_R: return a.alpha_r_ptr();
    case GAIBL_ALPHA_LGRAD: return a.alpha_lgrad_ptr();
    case GAIBL_ALPHA_RGRAD: return a.alpha_rgrad_ptr();
    case GAIBL_NORM_SCORES: return a.norm_scores_ptr();
    case GAIBL_TEMP_SCORES: return a.temp_scores_ptr();
    case GAIBL_SCORES: return a.scores_ptr();
    case GAIBL_NORM_SCORES_GRAD: return a.norm_scores_grad_ptr();
    case GAIBL_NORM_SCORES_DROPPED: return a.norm_scores_dropped_ptr();
    case GAIBL_ATTN_MASKS: return reinterpret_cast<float*>(a.attn_masks_ptr());
    default: return nullptr;
  }
}

uint32_t gaibl_sample_subgraph(uint32_t nv, uint32_t ne, const uint32_t* rowptr, const uint32_t* colidx,
                               const uint8_t* train_masks, uint32_t n, uint32_t m, unsigned seed,
                               uint32_t** sub_rowptr, uint32_t** sub_colidx, uint32_t** kept_ids) {
  Graph full(false);
  full.allocateFrom(nv, ne);
  memcpy(full.row_host_ptr(), rowptr, sizeof(uint32_t) * ((size_t)nv + 1));
  if (ne) memcpy(full.edge_host_ptr(), colidx, sizeof(uint32_t) * ne);
  std::vector<mask_t> masks(train_masks, train_masks + nv);
  Graph* tg = full.generate_masked_graph(masks.data());
  size_t cnt = 0;
  for (auto b : masks) cnt += b;
  Sampler sampler(&full, tg, masks.data(), cnt);
  sampler.set_frontier_size(m);
  VertexSet st;
  sampler.select_vertices(n, st, seed);
  std::vector<mask_t> sm(nv);
  Graph sg(false);
  sampler.generateSubgraph(st, sm.data(), &sg);
  const uint32_t snv = (uint32_t)sg.size(), sne = (uint32_t)sg.sizeEdges();
  *sub_rowptr = (uint32_t*)malloc(sizeof(uint32_t) * ((size_t)snv + 1));
  *sub_colidx = (uint32_t*)malloc(sizeof(uint32_t) * (sne ? sne : 1));
  *kept_ids = (uint32_t*)malloc(sizeof(uint32_t) * (snv ? snv : 1));
  memcpy(*sub_rowptr, sg.row_host_ptr(), sizeof(uint32_t) * ((size_t)snv + 1));
  if (sne) memcpy(*sub_colidx, sg.edge_host_ptr(), sizeof(uint32_t) * sne);
  uint32_t k = 0;
  for (index_t v : st) (*kept_ids)[k++] = v;
  tg->dealloc();
  delete tg;
  full.dealloc();
  sg.dealloc();
  return snv;
}
void gaibl_free_host(void* p) { free(p); }

// ---- vertex-range partition (include/gnn/partition.h) ----
void* gaibl_partition_build(uint32_t nv, const uint32_t* rowptr, const uint32_t* colidx, int rank, int world) {
  return new VertexRangePartition(build_vertex_range_partition((int64_t)nv, rowptr, colidx, rank, world));
}
void gaibl_partition_build_gat(void* part, const uint32_t* rowptr, const uint32_t* colidx) {
  build_gat_structures(*static_cast<VertexRangePartition*>(part), rowptr, colidx);
}
int64_t gaibl_partition_array(void* part, int which, const void** data) {
  VertexRangePartition* P = static_cast<VertexRangePartition*>(part);
#define ARR(v) { *data = (v).data(); return (int64_t)(v).size(); }
  switch (which) {
    case 0: ARR(P->rowptr_own)
    case 1: ARR(P->colidx_own)
    case 2: ARR(P->rowptr_halo)
    case 3: ARR(P->colidx_halo)
    case 4: ARR(P->degree)
    case 5: ARR(P->halo_gids)
    case 6: ARR(P->halo_degree)
    case 7: ARR(P->recv_counts)
    case 8: ARR(P->send_counts)
    case 9: ARR(P->send_idx)
    case 10: ARR(P->rowptr_full)
    case 11: ARR(P->colidx_full)
    case 12: ARR(P->rowptr_t)
    case 13: ARR(P->colidx_t)
    case 14: ARR(P->tperm)
    default: *data = nullptr; return -1;
  }
#undef ARR
}
void gaibl_partition_range(void* part, int64_t* lo, int64_t* hi) {
  VertexRangePartition* P = static_cast<VertexRangePartition*>(part);
  *lo = P->lo;
  *hi = P->hi;
}
void gaibl_partition_free(void* part) { delete static_cast<VertexRangePartition*>(part); }
void* gaibl_partition_make_graph(void* part, void* comm) {
  return make_partitioned_graph(*static_cast<VertexRangePartition*>(part), static_cast<gaib_comm*>(comm));
}
void gaibl_set_comm(void* comm) { gpu_context::set_comm(static_cast<gaib_comm*>(comm)); }
void gaibl_graph_set_halo_plan(void* graph, void* halo_graph, void* plan) {
  static_cast<Graph*>(graph)->set_halo_plan(static_cast<gaib_graph*>(halo_graph), static_cast<gaib_halo*>(plan));
}
void* gaibl_graph_halo_plan(void* graph) { return static_cast<Graph*>(graph)->halo_plan(); }
void gaibl_graph_set_partition_mode(void* graph, int mode) { static_cast<Graph*>(graph)->set_partition_mode(mode); }
void gaibl_graph_set_halo_link_rows(void* graph, int64_t rows) { static_cast<Graph*>(graph)->set_halo_link_rows(rows); }
void gaibl_graph_set_halo_pieces(void* graph, int n_pieces, int n_ranges, const int64_t* begin, const int64_t* end,
                                 const int* piece, gaibl_halo_wait_piece_fn wait_piece) {
  static_cast<Graph*>(graph)->set_halo_pieces(n_pieces, n_ranges, begin, end, piece, wait_piece);
}
int gaibl_graph_halo_pieces(void* graph, int len) { return static_cast<Graph*>(graph)->halo_pieces(len); }
void gaibl_graph_set_halo_consumption(void* graph, int pieces) { static_cast<Graph*>(graph)->set_halo_consumption(pieces); }
int gaibl_graph_partition_mode(void* graph, int len, int64_t* n_boundary, int64_t* boundary_edges) {
  Graph* g = static_cast<Graph*>(graph);
  const int mode = g->partition_mode(len);
  if (n_boundary) *n_boundary = g->n_boundary();
  if (boundary_edges) *boundary_edges = g->boundary_edges();
  return mode;
}

void* gaibl_adam_create(float lr) { return static_cast<optimizer*>(new adam(lr)); }
void gaibl_adam_free(void* opt) { delete static_cast<optimizer*>(opt); }

double gaibl_time_op(char op) { return time_ops[op]; }
void gaibl_reset_timers(void) { time_ops.clear(); }
}
