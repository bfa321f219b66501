// aggregators.cpp -- GCN / SAGE / GAT aggregators: each operator call is one or a few entries of
// the C ABI (include/gaib.h) on the process context.
#include "aggregator.h"
#include "host_util.h"

using gaib_host::OpTimer;
// edges aggregated so far by this process: one count of the graph's edges per aggregation call (BASELINE's
// "aggregated edges"; bench.py counts the same way)
static inline void count_edges(Graph& g) { gpu_context::add_aggregated_edges((uint64_t)g.sizeEdges()); }
static inline gaib_ctx* C() { return gpu_context::get(); }

static gaib_graph* dev(Graph& g) {
  if (!g.device_graph()) g.copy_to_gpu();
  return g.device_graph();
}

// The halo-column half of a partitioned aggregation over `whole` (the mode's halo-column graph): in one pass after the whole
// exchange -- last(whole, table) --, or, where the exchange travels in K > 1 time slices (gaib_halo_set_pieces), piece by piece
// as the slices land: plain(piece k, table) in accumulate mode for every piece but the last non-empty one, which takes last()
// (the pass that carries the activation / the dense product).  Same terms per row, added piece by piece.  Ends the exchange.
template <class Plain, class Last>
static void halo_half(Graph& g, gaib_graph* whole, int len, Plain plain, Last last) {
  const int K = g.halo_pieces(len);
  if (K <= 1) {
    last(whole, g.halo_end(len));
    return;
  }
  int last_k = 0;
  for (int k = 0; k < K; k++)
    if (gaib_graph_ne(g.halo_piece_graph(k)) > 0) last_k = k;
  for (int k = 0; k < last_k; k++)
    if (gaib_graph_ne(g.halo_piece_graph(k)) > 0) plain(g.halo_piece_graph(k), g.halo_wait_piece(k));
  last(g.halo_piece_graph(last_k), g.halo_end(len));  // (pieces behind last_k are empty: waiting for all of them costs nothing)
}

// One aggregation.  On a vertex-range partition the work that needs no halo row runs while the halo rows are in flight
// (separate RCCL stream), the rest after they have arrived -- by row class (LearningGraph::partition_mode):
//   PART_SPLIT    owned-column edges of all rows meanwhile, halo-column edges added to the same rows after
//   PART_CLASSES  interior rows complete + the boundary rows' owned-column edges meanwhile, their halo-column edges after
//   PART_ONEPASS  interior rows complete meanwhile, the boundary rows in one pass over [owned | halo] after
static void aggregate_rows(Graph& g, int kind, int len, const float* in, float* out, bool relu = false, bool count = true) {
  if (count) count_edges(g);
  const int act = relu ? GAIB_RELU : 0;
  if (!g.has_halo()) {
    GAIB_OR_DIE(gaib_spmm_ex(C(), dev(g), kind, NULL, len, in, out, act));
    return;
  }
  const int mode = g.partition_mode(len);
  auto acc = [&](gaib_graph* gh, const float* halo) {
    GAIB_OR_DIE(gaib_spmm_ex(C(), gh, kind, NULL, len, halo, out, GAIB_ACCUMULATE));
  };
  auto acc_act = [&](gaib_graph* gh, const float* halo) {
    GAIB_OR_DIE(gaib_spmm_ex(C(), gh, kind, NULL, len, halo, out, GAIB_ACCUMULATE | act));
  };
  if (mode != Graph::PART_SPLIT) {
    g.halo_begin(len, in);
    GAIB_OR_DIE(gaib_spmm_ex(C(), g.class_interior(), kind, NULL, len, in, out, act));
    if (mode == Graph::PART_CLASSES) {
      const bool have_halo_edges = gaib_graph_ne(g.class_boundary_halo()) > 0;  // (else: no boundary row either)
      GAIB_OR_DIE(gaib_spmm_ex(C(), g.class_boundary_own(), kind, NULL, len, in, out, have_halo_edges ? 0 : act));
      if (have_halo_edges) halo_half(g, g.class_boundary_halo(), len, acc, acc_act);
      else g.halo_end(len);
    } else {
      const float* halo = g.halo_end(len);
      GAIB_OR_DIE(gaib_spmm_2t(C(), g.class_boundary_full(), kind, NULL, len, in, halo, (int64_t)g.size(), out, act));
    }
    return;
  }
  g.halo_begin(len, in);
  const bool have_halo_edges = gaib_graph_ne(g.halo_graph()) > 0;
  GAIB_OR_DIE(gaib_spmm_ex(C(), dev(g), kind, NULL, len, in, out, have_halo_edges ? 0 : act));
  if (!have_halo_edges) {
    g.halo_end(len);
    return;
  }
  halo_half(g, g.halo_graph(), len, acc, acc_act);
}

void aggregator::aggregate_then_matmul(int kind, int len, Graph& g, const float* in, float* agg, bool keep_agg,
                                       const float* W, bool transW, int len_out, float* out, bool relu,
                                       const float* rows2, const float* W2) {
  OpTimer t(OP_SPARSEMM);
  count_edges(g);
  const int flags = (relu ? GAIB_RELU : 0) | (keep_agg ? 0 : GAIB_AGG_SCRATCH);
  auto fused = [&](gaib_graph* dg, const float* src, int fl) {
    if (rows2)
      GAIB_OR_DIE(gaib_spmm_gemm2(C(), dg, kind, NULL, len, src, agg, W, transW ? 1 : 0, rows2, W2, len_out, out, fl));
    else
      GAIB_OR_DIE(gaib_spmm_gemm(C(), dg, kind, NULL, len, src, agg, W, transW ? 1 : 0, len_out, out, fl));
  };
  // a piece of the halo-column half that is not the last: the partial sums continue in agg, no product yet
  auto plain_acc = [&](gaib_graph* gh, const float* halo) {
    GAIB_OR_DIE(gaib_spmm_ex(C(), gh, kind, NULL, len, halo, agg, GAIB_ACCUMULATE));
  };
  const int mode = g.has_halo() ? g.partition_mode(len) : Graph::PART_SPLIT;
  if (mode != Graph::PART_SPLIT) {
    // the classes fill disjoint rows of ONE output: all of them take the fused kernel, or -- a shape it does not cover --
    // the aggregation runs class by class and the product(s) once over all rows
    if (!gaib_spmm_gemm_fusable(C(), kind, len, len_out, rows2 ? 1 : 0)) {
      aggregate_rows(g, kind, len, in, agg, false, false);
      const int act = relu ? GAIB_RELU : 0;
      GAIB_OR_DIE(gaib_sgemm_ex(C(), 0, transW ? 1 : 0, (int64_t)g.size(), len_out, len, agg, W, rows2 ? 0 : act, out));
      if (rows2)
        GAIB_OR_DIE(gaib_sgemm_ex(C(), 0, transW ? 1 : 0, (int64_t)g.size(), len_out, len, rows2, W2, GAIB_ACCUMULATE | act, out));
      return;
    }
    g.halo_begin(len, in);
    fused(g.class_interior(), in, flags | GAIB_OVERLAPS_TRANSFER);  // (the exchange is in flight: RCCL's kernels need CUs)
    if (mode == Graph::PART_CLASSES) {
      if (gaib_graph_ne(g.class_boundary_halo()) == 0) {  // no boundary row
        g.halo_end(len);
        return;
      }
      GAIB_OR_DIE(gaib_spmm_ex(C(), g.class_boundary_own(), kind, NULL, len, in, agg, 0));
      halo_half(g, g.class_boundary_halo(), len, plain_acc,
                [&](gaib_graph* gh, const float* halo) { fused(gh, halo, flags | GAIB_ACCUMULATE); });
    } else {
      const float* halo = g.halo_end(len);
      GAIB_OR_DIE(gaib_spmm_gemm_2t(C(), g.class_boundary_full(), kind, NULL, len, in, halo, (int64_t)g.size(), agg, W,
                                    transW ? 1 : 0, rows2, W2, len_out, out, flags));
    }
    return;
  }
  if (g.has_halo()) {
    // owned-column edges while the halo rows travel; the halo-column edges then continue the sums and
    // carry the dense product(s)
    g.halo_begin(len, in);  // every rank joins every exchange, also one without halo edges
    if (gaib_graph_ne(g.halo_graph()) == 0) {
      fused(dev(g), in, flags | GAIB_OVERLAPS_TRANSFER);
      g.halo_end(len);
      return;
    }
    GAIB_OR_DIE(gaib_spmm_ex(C(), dev(g), kind, NULL, len, in, agg, 0));
    halo_half(g, g.halo_graph(), len, plain_acc,
              [&](gaib_graph* gh, const float* halo) { fused(gh, halo, flags | GAIB_ACCUMULATE); });
    return;
  }
  fused(dev(g), in, flags);
}

// ---- GCN ---------------------------------------------------------------------------------------
void GCN_Aggregator::aggregate_matmul(int len, Graph& g, const float* in, float* agg, bool keep_agg,
                                      const float* W, bool transW, int len_out, float* out, bool relu) {
  aggregate_then_matmul(GAIB_W_GCN, len, g, in, agg, keep_agg, W, transW, len_out, out, relu);
}
void GCN_Aggregator::d_aggregate_matmul(int len, Graph& g, const float* grad_in, float* agg, bool keep_agg,
                                        const float* W, bool transW, int len_out, float* out) {
  aggregate_then_matmul(GAIB_W_GCN, len, g, grad_in, agg, keep_agg, W, transW, len_out, out, false);
}
void SAGE_Aggregator::aggregate_matmul(int len, Graph& g, const float* in, float* agg, bool keep_agg,
                                       const float* W, bool transW, int len_out, float* out, bool relu,
                                       const float* rows_self, const float* W_self) {
  aggregate_then_matmul(GAIB_W_MEAN, len, g, in, agg, keep_agg, W, transW, len_out, out, relu, rows_self, W_self);
}
void SAGE_Aggregator::d_aggregate_matmul(int len, Graph& g, const float* grad_in, float* agg, bool keep_agg,
                                         const float* W, bool transW, int len_out, float* out,
                                         const float* rows_self, const float* W_self) {
  aggregate_then_matmul(GAIB_W_MEAN_T, len, g, grad_in, agg, keep_agg, W, transW, len_out, out, false, rows_self,
                        W_self);
}
void GCN_Aggregator::init(int l, int nv, int, float, float) {
  length = l;
  n = nv;
}
void GCN_Aggregator::aggregate(int len, Graph& g, const float* in, float* out) {
  OpTimer t(OP_SPARSEMM);
  aggregate_rows(g, GAIB_W_GCN, len, in, out, fuse_relu);
  fuse_relu = false;
}
// the normalised adjacency is symmetric, so the derivative is the same operator
void GCN_Aggregator::d_aggregate(int len, Graph& g, const float*, const float* grad_in, float* grad_out) {
  OpTimer t(OP_SPARSEMM);
  aggregate_rows(g, GAIB_W_GCN, len, grad_in, grad_out);
}

// ---- SAGE --------------------------------------------------------------------------------------
void SAGE_Aggregator::init(int l, int nv, int, float, float) {
  length = l;
  n = nv;
}
void SAGE_Aggregator::aggregate(int len, Graph& g, const float* in, float* out) {
  OpTimer t(OP_SPARSEMM);
  aggregate_rows(g, GAIB_W_MEAN, len, in, out, fuse_relu);
  fuse_relu = false;
}
void SAGE_Aggregator::d_aggregate(int len, Graph& g, const float*, const float* grad_in, float* grad_out) {
  OpTimer t(OP_SPARSEMM);
  aggregate_rows(g, GAIB_W_MEAN_T, len, grad_in, grad_out);
}

// ---- GAT ---------------------------------------------------------------------------------------
GAT_Aggregator::GAT_Aggregator()
    : epsilon(0.2f), attn_drop(0.f), attn_scale(1.f), training(true), dropped_last(false), d_norm_scores_drop(NULL),
      d_attn_masks(NULL), drop_cap(0), drop_seed(0xA77E0000ull), num_edges(0), heads(1), d_alpha_l(NULL), d_alpha_r(NULL), d_alpha_lgrad(NULL),
      d_alpha_rgrad(NULL), d_temp_scores(NULL), d_norm_scores(NULL),
      d_norm_scores_grad(NULL), d_norm_scores_t(NULL), fwd_out(NULL), d_tbuf(NULL), tbuf_floats(0), d_ptab(NULL), d_pout(NULL),
      d_prs(NULL), d_pcs(NULL), ptab_floats(0), pvec_floats(0), d_pgrad(NULL), d_prec(NULL), pgrad_floats(0), prec_floats(0),
      part_fused_last(false), d_row_stats(NULL), stats_floats(0), stats_valid(false),
      last_graph(NULL), last_in(NULL), last_len(0), alpha_opt(NULL) {}

void GAT_Aggregator::init(int l, int nv, int ne, float lr, float drop_rate) {
  length = l;
  n = nv;
  attn_drop = drop_rate;
  assert(attn_drop >= 0. && attn_drop < 1.);
  attn_scale = 1.f / (1.f - attn_drop);  // gat_aggregator.cpp:7
  num_edges = (size_t)ne;
  // alpha_l / alpha_r: Glorot over (l, 1) with seeds 2 and 3, as the reference's CPU path
  // (gat_aggregator.cpp:11-12)
  vec_t al, ar;
  init_glorot(l, 1, al, 2);
  init_glorot(l, 1, ar, 3);
  d_alpha_l = gaib_host::dmalloc<float>(l);
  d_alpha_r = gaib_host::dmalloc<float>(l);
  d_alpha_lgrad = gaib_host::dmalloc<float>(l);
  d_alpha_rgrad = gaib_host::dmalloc<float>(l);
  copy_float_device(l, al.data(), d_alpha_l);
  copy_float_device(l, ar.data(), d_alpha_r);
  GAIB_OR_DIE(gaib_fill_f32(C(), l, 0.f, d_alpha_lgrad));
  GAIB_OR_DIE(gaib_fill_f32(C(), l, 0.f, d_alpha_rgrad));
  d_temp_scores = gaib_host::dmalloc<float>(num_edges);
  d_norm_scores = gaib_host::dmalloc<float>(num_edges);
  d_norm_scores_grad = gaib_host::dmalloc<float>(num_edges);
  d_norm_scores_t = gaib_host::dmalloc<float>(num_edges);
  epsilon = 0.2f;
  alpha_opt = new adam(lr);
}

// d_norm_scores (the softmax of this forward) -> p . mask . scale in a buffer of its own: backward needs both the
// undropped attention (softmax backward) and the dropped one (d_dropout of dp, transposed aggregation)
const float* GAT_Aggregator::apply_attn_dropout(size_t n_scores) {
  if (n_scores > drop_cap) {
    if (d_norm_scores_drop) float_free_device(d_norm_scores_drop);
    if (d_attn_masks) GAIB_OR_DIE(gaib_free(C(), d_attn_masks));
    float_malloc_device64(n_scores, d_norm_scores_drop);
    d_attn_masks = gaib_host::dmalloc<mask_t>(n_scores);
    drop_cap = n_scores;
  }
  OpTimer t(OP_DROPOUT);
  GAIB_OR_DIE(gaib_dropout(C(), (int64_t)n_scores, attn_scale, attn_drop, drop_seed++, d_norm_scores, d_attn_masks,
                           d_norm_scores_drop));
  dropped_last = true;
  return d_norm_scores_drop;
}

void GAT_Aggregator::release() {
  gaib_ctx* c = C();
  float** owned[] = {&d_alpha_l, &d_alpha_r, &d_alpha_lgrad, &d_alpha_rgrad, &d_temp_scores, &d_norm_scores, &d_norm_scores_grad,
                     &d_norm_scores_t, &d_norm_scores_drop, &d_tbuf, &d_ptab, &d_pout, &d_prs, &d_pcs, &d_pgrad, &d_prec,
                     &d_row_stats};
  for (float** p : owned) {
    if (*p) GAIB_OR_DIE(gaib_free(c, *p));
    *p = NULL;
  }
  if (d_attn_masks) GAIB_OR_DIE(gaib_free(c, d_attn_masks));
  d_attn_masks = NULL;
  drop_cap = tbuf_floats = ptab_floats = pvec_floats = pgrad_floats = prec_floats = stats_floats = 0;
  stats_valid = part_fused_last = dropped_last = false;
  last_graph = NULL;
  last_in = fwd_out = NULL;
  if (alpha_opt) {
    alpha_opt->reset();
    delete alpha_opt;
    alpha_opt = NULL;
  }
}

void GAT_Aggregator::set_num_heads(int h) {
  if (h < 1 || length % h != 0) {
    fprintf(stderr, "GAT_Aggregator::set_num_heads(%d): must divide the feature length %d\n", h, length);
    exit(EXIT_FAILURE);
  }
  if (h == heads) return;
  heads = h;
  float** arrays[] = {&d_norm_scores, &d_norm_scores_grad, &d_norm_scores_t};
  for (float** a : arrays) {
    float_free_device(*a);
    *a = gaib_host::dmalloc<float>(num_edges * heads);
  }
  // the temp_scores array only where the kernels do not form the pre-activation score again (see needs_temp)
  if (d_temp_scores) float_free_device(d_temp_scores);
  d_temp_scores = needs_temp() ? gaib_host::dmalloc<float>(num_edges * heads) : NULL;
}

// ---- GAT on a vertex-range partition ("h halo rows for the scores, g halo rows for the transposed
// aggregation") ----
// forward: the halo rows of h arrive while the owned rows are copied into one column table [owned | halo]; scores,
// edge softmax and aggregation then run unchanged on the rank's rectangular graph over that column space.
// backward: the reverse edge of (i -> c) belongs to the rank that owns c, so instead of the reverse-edge permutation the
// rank's TRANSPOSED local structure is used: column sums of g (alpha_r gradient) are row sums over the transpose, and
// the gradient aggregation out_c = sum_i p_(i->c) grad_i is an SpMM over the transpose whose halo rows -- partial sums
// for vertices of other ranks -- travel back to their owners and are added there (gaib_halo_reduce).
void GAT_Aggregator::ensure_partition_buffers(Graph& g, int len) {
  const size_t nc = g.size() + g.gat_n_halo();
  if (nc * len > ptab_floats) {
    if (d_ptab) float_free_device(d_ptab);
    if (d_pout) float_free_device(d_pout);
    float_malloc_device64(nc * len, d_ptab);
    float_malloc_device64(nc * len, d_pout);
    ptab_floats = nc * len;
  }
  if (nc * heads > pvec_floats) {
    if (d_prs) float_free_device(d_prs);
    if (d_pcs) float_free_device(d_pcs);
    float_malloc_device64(nc * heads, d_prs);
    float_malloc_device64(nc * heads, d_pcs);
    GAIB_OR_DIE(gaib_fill_f32(C(), (int64_t)(nc * heads), 0.f, d_prs));  // rows of halo vertices stay 0
    pvec_floats = nc * heads;
  }
  if (!d_temp_scores) d_temp_scores = gaib_host::dmalloc<float>(num_edges * heads);  // the row-side backward reads it
}

void GAT_Aggregator::aggregate_partition(int len, Graph& g, const float* in, float* out) {
  const size_t n_own = g.size(), n_halo = g.gat_n_halo();
  ensure_partition_buffers(g, len);
  dropped_last = false;
  part_fused_last = false;
  // One sweep (gaib_gat_forward_fused_rect) where the shape allows: the chunks over owned columns run while the halo rows
  // of h are on the wire (phase 0), the rest and the per-row combination after they have arrived (phase 1); only the row
  // statistics are kept.  Otherwise (other widths, attention dropout, gat_fused_fwd = 0) the staged pieces.
  bool fused = false;
  {
    OpTimer t(OP_SCORE);
    g.halo_begin(len, in);
    GAIB_OR_DIE(gaib_memcpy_d2d(C(), d_ptab, in, sizeof(float) * n_own * len));
    if (!dropping() && g.gat_symmetric()) {  // (an asymmetric graph: the staged path, which walks the transposed structure)
      const size_t need = n_own * heads * 2;
      if (need > stats_floats) {
        if (d_row_stats) float_free_device(d_row_stats);
        float_malloc_device64(need, d_row_stats);
        stats_floats = need;
      }
      const int rc = gaib_gat_forward_fused_rect(C(), g.gat_full_graph(), len, heads, d_ptab, d_alpha_l, d_alpha_r, epsilon,
                                                 fuse_relu ? 1 : 0, out, d_row_stats, 0);
      if (rc == GAIB_OK) fused = true;
      else if (rc != GAIB_ERR_UNSUPPORTED) GAIB_OR_DIE(rc);
    }
    const float* halo = g.halo_end(len);
    if (n_halo) GAIB_OR_DIE(gaib_memcpy_d2d(C(), d_ptab + n_own * len, halo, sizeof(float) * n_halo * len));
    if (fused) {
      GAIB_OR_DIE(gaib_gat_forward_fused_rect(C(), g.gat_full_graph(), len, heads, d_ptab, d_alpha_l, d_alpha_r, epsilon,
                                              fuse_relu ? 1 : 0, out, d_row_stats, 1));
      fuse_relu = false;
      part_fused_last = true;
      return;
    }
    GAIB_OR_DIE(gaib_gat_scores_mh(C(), g.gat_full_graph(), len, heads, d_ptab, d_alpha_l, d_alpha_r, epsilon,
                                   d_temp_scores, NULL, d_norm_scores));
  }
  const float* attn = dropping() ? apply_attn_dropout((size_t)g.sizeEdges() * heads) : d_norm_scores;
  OpTimer t(OP_SPARSEMM);
  GAIB_OR_DIE(gaib_spmm_mh(C(), g.gat_full_graph(), GAIB_W_EDGE, attn, heads, len, d_ptab, out,
                           fuse_relu ? GAIB_RELU : 0));
  fuse_relu = false;
}

void GAT_Aggregator::d_aggregate_partition(int len, Graph& g, const float* grad_in, float* grad_out) {
  const size_t n_own = g.size(), n_halo = g.gat_n_halo(), nc = n_own + n_halo;
  const int64_t ne = (int64_t)g.sizeEdges();
  gaib_graph *full = g.gat_full_graph(), *gt = g.gat_transposed_graph();
  const float* fwd = fwd_out;
  const bool fwd_given = fwd_out_given;
  fwd_out = NULL;
  fwd_out_given = false;
  ensure_partition_buffers(g, len);
  if (part_fused_last && fwd_given) {
    // the one-sweep backward on the rectangular graph: the owner of row i computes everything about i from i's own edge
    // list, given the halo vertices' h rows (d_ptab, from forward), grad rows and (rowdot, max, 1 / sum) records -- two
    // forward-direction exchanges, no transposed structure, no reverse exchange.  The chunks over owned columns run while
    // the grad rows are on the wire.
    OpTimer t(OP_ATTN);
    if (nc * len > pgrad_floats) {
      if (d_pgrad) float_free_device(d_pgrad);
      float_malloc_device64(nc * len, d_pgrad);
      pgrad_floats = nc * len;
    }
    if (nc * heads * 4 > prec_floats) {
      if (d_prec) float_free_device(d_prec);
      float_malloc_device64(nc * heads * 4, d_prec);
      prec_floats = nc * heads * 4;
    }
    GAIB_OR_DIE(gaib_gat_backward_rec(C(), (int64_t)n_own, len, heads, grad_in, fwd, d_row_stats, d_prec));
    GAIB_OR_DIE(gaib_memcpy_d2d(C(), d_pgrad, grad_in, sizeof(float) * n_own * len));
    g.halo_begin(len, grad_in);
    const int rc = gaib_gat_backward_fused_rect(C(), full, len, heads, d_ptab, d_pgrad, d_prec, d_alpha_l, d_alpha_r, epsilon,
                                                grad_out, d_alpha_lgrad, d_alpha_rgrad, 0);
    const float* halo = g.halo_end(len);  // (every rank ends every exchange it began, whatever rc says)
    if (rc == GAIB_OK) {
      if (n_halo) GAIB_OR_DIE(gaib_memcpy_d2d(C(), d_pgrad + n_own * len, halo, sizeof(float) * n_halo * len));
      g.halo_begin(4 * heads, d_prec);
      halo = g.halo_end(4 * heads);
      if (n_halo) GAIB_OR_DIE(gaib_memcpy_d2d(C(), d_prec + n_own * heads * 4, halo, sizeof(float) * n_halo * heads * 4));
      GAIB_OR_DIE(gaib_gat_backward_fused_rect(C(), full, len, heads, d_ptab, d_pgrad, d_prec, d_alpha_l, d_alpha_r, epsilon,
                                               grad_out, d_alpha_lgrad, d_alpha_rgrad, 1));
      return;
    }
    // forward took the one-sweep path and backward may not (option gat_fused_bwd = 0 on every rank -- options are set per
    // process, the same on all): the exchange above was for nothing, the staged pieces below form the attention again
    if (rc != GAIB_ERR_UNSUPPORTED) GAIB_OR_DIE(rc);
  }
  if (part_fused_last) {  // forward kept statistics only and backward has no forward output to use: the attention, staged
    GAIB_OR_DIE(gaib_gat_scores_mh(C(), full, len, heads, d_ptab, d_alpha_l, d_alpha_r, epsilon, d_temp_scores, NULL,
                                   d_norm_scores));
    part_fused_last = false;
  }
  {
    OpTimer t(OP_SCORE);
    GAIB_OR_DIE(gaib_sddmm_mh(C(), full, len, heads, grad_in, d_ptab, d_norm_scores_grad));
    if (dropped_last)  // d(out)/d(p_e) = mask_e . scale . <grad_i, h_c>
      GAIB_OR_DIE(gaib_d_dropout(C(), ne * heads, attn_scale, d_norm_scores_grad, d_attn_masks, d_norm_scores_grad));
  }
  {
    OpTimer t(OP_ATTN);
    GAIB_OR_DIE(gaib_gat_softmax_bwd_rows(C(), full, heads, d_norm_scores, d_norm_scores_grad, d_temp_scores, epsilon,
                                          d_norm_scores_t /* g_e */, d_prs));
    GAIB_OR_DIE(gaib_edge_gather_perm(C(), ne, heads, g.gat_tperm(), d_norm_scores_t, d_norm_scores_grad));
    GAIB_OR_DIE(gaib_edge_rowsum(C(), gt, heads, d_norm_scores_grad, d_pcs));
    GAIB_OR_DIE(gaib_gat_alpha_grads(C(), (int64_t)nc, len, heads, d_ptab, d_prs, d_pcs, d_alpha_lgrad, d_alpha_rgrad));
  }
  {
    OpTimer t(OP_TRANSPOSE);
    GAIB_OR_DIE(gaib_edge_gather_perm(C(), ne, heads, g.gat_tperm(), dropped_last ? d_norm_scores_drop : d_norm_scores,
                                      d_norm_scores_grad));
  }
  OpTimer t(OP_SPARSEMM);
  GAIB_OR_DIE(gaib_spmm_mh(C(), gt, GAIB_W_EDGE, d_norm_scores_grad, heads, len, grad_in, d_pout, 0));
  GAIB_OR_DIE(gaib_memcpy_d2d(C(), grad_out, d_pout, sizeof(float) * n_own * len));
  GAIB_OR_DIE(gaib_halo_reduce(g.halo_plan(), len, d_pout + n_own * len, grad_out));
}

void GAT_Aggregator::aggregate(int len, Graph& g, const float* in, float* out) {
  count_edges(g);
  if (g.gat_full_graph()) {
    aggregate_partition(len, g, in, out);
    return;
  }
  if (g.has_halo()) {
    fprintf(stderr, "GAT_Aggregator: this partitioned graph was built without build_gat_structures()\n");
    exit(EXIT_FAILURE);
  }
  // dense graphs at 64 columns: scores, edge softmax and aggregation in ONE sweep; only the row statistics (maximum,
  // 1 / sum) are kept and backward forms the attention again -- no [ne][heads] array is written or read.  norm_scores_ptr()
  // materialises the attention on demand (tests, checkpoints).
  dropped_last = false;
  if (!dropping()) {
    const size_t need = (size_t)g.size() * heads * 2;
    if (need > stats_floats) {
      if (d_row_stats) float_free_device(d_row_stats);
      float_malloc_device64(need, d_row_stats);
      stats_floats = need;
    }
    OpTimer t(OP_SPARSEMM);
    const int rc = gaib_gat_forward_fused(C(), dev(g), len, heads, in, d_alpha_l, d_alpha_r, epsilon, fuse_relu ? 1 : 0,
                                          out, d_row_stats);
    if (rc == GAIB_OK) {
      fuse_relu = false;
      stats_valid = true;
      last_graph = &g;
      last_in = in;
      last_len = len;
      return;
    }
    if (rc != GAIB_ERR_UNSUPPORTED) GAIB_OR_DIE(rc);
  }
  stats_valid = false;
  if (g.sizeEdges() > num_edges) {  // a larger graph than the one the layer was built on (sampling -> full graph)
    num_edges = g.sizeEdges();
    float** arrays[] = {&d_norm_scores, &d_norm_scores_grad, &d_norm_scores_t};
    for (float** a : arrays) {
      float_free_device(*a);
      *a = gaib_host::dmalloc<float>(num_edges * heads);
    }
    if (d_temp_scores) {
      float_free_device(d_temp_scores);
      d_temp_scores = gaib_host::dmalloc<float>(num_edges * heads);
    }
  }
  {
    OpTimer t(OP_SCORE);
    // the leaky-relu output itself is not materialised (NULL): nothing downstream reads it
    // nor is the pre-activation score where backward can form its sign again (d_temp_scores stays NULL then)
    GAIB_OR_DIE(gaib_gat_scores_mh(C(), dev(g), len, heads, in, d_alpha_l, d_alpha_r, epsilon, d_temp_scores,
                                   NULL, d_norm_scores));
  }
  const float* attn = dropping() ? apply_attn_dropout((size_t)g.sizeEdges() * heads) : d_norm_scores;
  OpTimer t(OP_SPARSEMM);
  GAIB_OR_DIE(gaib_spmm_mh(C(), dev(g), GAIB_W_EDGE, attn, heads, len, in, out, fuse_relu ? GAIB_RELU : 0));
  fuse_relu = false;
}

// feat_in and grad_out may be the same buffer (GAT_layer::backward passes out_temp for both):
// feat_in is last read by the alpha-gradient step, grad_out is first written by the final SpMM.
void GAT_Aggregator::d_aggregate(int len, Graph& g, const float* feat_in, const float* grad_in,
                                 float* grad_out) {
  count_edges(g);
  if (g.gat_full_graph()) {
    d_aggregate_partition(len, g, grad_in, grad_out);
    return;
  }
  if (fwd_out && !dropped_last) {
    // one sweep over the edges instead of four (gaib_gat_backward_fused): needs the layer's forward output and an
    // output that does not alias feat_in (GAT_layer::backward passes out_temp for both) -> a scratch of its own
    OpTimer t(OP_ATTN);
    const size_t need = (size_t)g.size() * len;
    if (need > tbuf_floats) {
      if (d_tbuf) float_free_device(d_tbuf);
      float_malloc_device64(need, d_tbuf);
      tbuf_floats = need;
    }
    const int rc = gaib_gat_backward_fused(C(), dev(g), len, heads, feat_in, grad_in, fwd_out, d_alpha_l, d_alpha_r,
                                           stats_valid ? NULL : d_norm_scores, stats_valid ? d_row_stats : NULL, epsilon,
                                           d_tbuf, d_alpha_lgrad, d_alpha_rgrad);
    if (rc == GAIB_OK) {
      fwd_out = NULL;
      fwd_out_given = false;
      GAIB_OR_DIE(gaib_memcpy_d2d(C(), grad_out, d_tbuf, sizeof(float) * need));
      return;
    }
    if (rc != GAIB_ERR_UNSUPPORTED) GAIB_OR_DIE(rc);  // a real failure; UNSUPPORTED = this shape / graph takes the staged path
  }
  if (stats_valid) materialise_scores();  // (the staged kernels read the attention array)
  {
    OpTimer t(OP_SCORE);
    GAIB_OR_DIE(gaib_sddmm_mh(C(), dev(g), len, heads, grad_in, feat_in, d_norm_scores_grad));
    if (dropped_last)  // d(out)/d(p_e) = mask_e . scale . <grad_i, h_c>; sum_e p_e dp_e is still <grad_i, out_i>
      GAIB_OR_DIE(gaib_d_dropout(C(), (int64_t)(g.sizeEdges() * heads), attn_scale, d_norm_scores_grad, d_attn_masks,
                                 d_norm_scores_grad));
  }
  {
    OpTimer t(OP_ATTN);
    // with the layer's forward output at hand the softmax backward is one pass over the edge arrays:
    // sum_e p_e dp_e == <grad_i, out_i> (out_i = sum_e p_e h_col; where relu cut out_i the gradient is 0 too)
    const float* fwd = fwd_out;
    fwd_out = NULL;
    fwd_out_given = false;
    // the pass that walks rev anyway also leaves the transposed attention p[rev(e)] in d_norm_scores_t, so the
    // gradient aggregation reads its weights linearly
    if (d_temp_scores)
      GAIB_OR_DIE(gaib_gat_softmax_bwd_alpha_ex(C(), dev(g), len, heads, feat_in, d_norm_scores, d_norm_scores_grad,
                                                d_temp_scores, epsilon, NULL, d_alpha_lgrad, d_alpha_rgrad,
                                                fwd ? grad_in : NULL, fwd, d_norm_scores_t));
    else
      GAIB_OR_DIE(gaib_gat_softmax_bwd_alpha_re(C(), dev(g), len, heads, feat_in, d_alpha_l, d_alpha_r, d_norm_scores,
                                                d_norm_scores_grad, epsilon, NULL, d_alpha_lgrad, d_alpha_rgrad,
                                                fwd ? grad_in : NULL, fwd, d_norm_scores_t));
  }
  if (dropped_last) {  // the gradient flows back along the DROPPED attention: its transpose replaces the undropped one
    OpTimer t2(OP_TRANSPOSE);
    GAIB_OR_DIE(gaib_edge_transpose_mh(C(), dev(g), heads, d_norm_scores_drop, d_norm_scores_t));
  }
  OpTimer t(OP_SPARSEMM);
  GAIB_OR_DIE(gaib_spmm_mh(C(), dev(g), GAIB_W_EDGE, d_norm_scores_t, heads, len, grad_in, grad_out, 0));
}

// the attention [ne][heads] of the last forward, for callers that want the array the one-sweep forward did not write
void GAT_Aggregator::materialise_scores() {
  if (!stats_valid || !last_graph) return;
  Graph& g = *last_graph;
  if (g.sizeEdges() > num_edges) {
    num_edges = g.sizeEdges();
    float** arrays[] = {&d_norm_scores, &d_norm_scores_grad, &d_norm_scores_t};
    for (float** a : arrays) {
      float_free_device(*a);
      *a = gaib_host::dmalloc<float>(num_edges * heads);
    }
    if (d_temp_scores) {
      float_free_device(d_temp_scores);
      d_temp_scores = gaib_host::dmalloc<float>(num_edges * heads);
    }
  }
  GAIB_OR_DIE(gaib_gat_scores_mh(C(), dev(g), last_len, heads, last_in, d_alpha_l, d_alpha_r, epsilon, d_temp_scores, NULL,
                                 d_norm_scores));
  stats_valid = false;
}
float* GAT_Aggregator::norm_scores_ptr() {
  materialise_scores();
  return d_norm_scores;
}

void GAT_Aggregator::update_weights(optimizer*) {
  alpha_opt->update_gpu(length, d_alpha_lgrad, d_alpha_l);
  alpha_opt->update_gpu(length, d_alpha_rgrad, d_alpha_r);
}
