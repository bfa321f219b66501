// lgraph.cpp -- LearningGraph: host CSR construction + HBM residency.
// API parity with the reference class (include/gnn/lgraph.h, src/gnn/lgraph.{cpp,cu}); the
// normalisers are computed by the device kernels behind gaib_graph_compute_*.
#include <stdlib.h>
#include <string.h>
#include "lgraph.h"
#include "host_util.h"

LearningGraph* LearningGraph::adopt_device(gaib_graph* g) {
  LearningGraph* lg = new LearningGraph(true);
  lg->num_vertices_ = (index_t)gaib_graph_nv(g);
  lg->num_edges_ = (index_t)gaib_graph_ne(g);
  lg->dev_ = g;
  return lg;
}

void LearningGraph::halo_begin(int len, const float* d_in) {
  if (halo_plan_) GAIB_OR_DIE(gaib_halo_exchange_begin(halo_plan_, len, d_in));
  else halo_begin_(halo_user_, len, d_in);
}
const float* LearningGraph::halo_end(int len) {
  if (!halo_plan_) return halo_end_(halo_user_, len);
  const float* table = NULL;
  GAIB_OR_DIE(gaib_halo_exchange_end(halo_plan_, &table));
  return table;
}

const float* LearningGraph::halo_wait_piece(int j) {
  const int k = (j + 1) * (pieces_slices_ / pieces_built_) - 1;  // the last slice of piece j
  if (!halo_plan_) return halo_wait_piece_(halo_user_, k);
  const float* table = NULL;
  GAIB_OR_DIE(gaib_halo_exchange_wait_piece(halo_plan_, k, &table));
  return table;
}

void LearningGraph::drop_pieces() {
  for (gaib_graph*& p : pieces_) {
    if (p) gaib_graph_destroy(p);
    p = NULL;
  }
  pieces_built_ = pieces_slices_ = 0;
}

// ---- the model the two rules share (partition_mode: which form; consumption_rule: how many pieces) ----------------------
// One aggregation of a rank whose rows keep a halo-column half, on rows of `len` columns, with the exchange priced at the
// most rows one peer pair moves x row bytes / GAIB_LINK_GBS (default 100 GB/s per pair; bench.py --gpus N puts the link rate
// it measured there).  Rates are the ones measured on the shard benchmarks (scripts/papers_shard.py; profiles/r04/shard_*,
// profiles/r06/shard/): the row kernels gather 512-B rows at 7.5 TB/s; a pass that continues partial sums reaches
// 4.8 TB/s on row segments of <= 3 edges, 7.7 TB/s from 12 edges on, and 8 % less where it is one of several pieces.
namespace {
struct SplitShape {
  double row_bytes, t_wire;
  int64_t ne_own, ne_halo, rows_half;  // owned-column edges of ALL rows; halo-column edges; rows of the halo-column half
};
double wire_seconds(int64_t link_rows, int len) {
  const double link_gbs = getenv("GAIB_LINK_GBS") ? atof(getenv("GAIB_LINK_GBS")) : 100.0;
  return link_rows * 4.0 * len / ((link_gbs > 0 ? link_gbs : 100.0) * 1e9);
}
// the column split with the K slices consumed in kc pieces: owned-column work first, piece j once slice (j + 1) K / kc - 1 has
// landed -- at (j + 1) / kc of the exchange --, every piece a read + write of the half's partial sums on top of its gathers
double model_split(const SplitShape& s, int kc) {
  const double e = s.rows_half > 0 ? (double)s.ne_halo / ((double)s.rows_half * kc) : 0.0;  // edges per row segment
  double rate = 4.8e12 + (e - 3.0) / 9.0 * 2.9e12;
  rate = rate < 4.8e12 ? 4.8e12 : (rate > 7.7e12 ? 7.7e12 : rate);
  if (kc > 1) rate *= 0.92;
  const double t_piece = (s.ne_halo * (s.row_bytes + 8) + kc * 2.0 * s.rows_half * s.row_bytes) / rate / kc;
  double t = s.ne_own * (s.row_bytes + 8) / 7.5e12;
  for (int j = 0; j < kc; j++) {
    const double arrive = s.t_wire * (j + 1) / kc;
    t = (t > arrive ? t : arrive) + t_piece;
  }
  return t;
}
// the K' | K with the shortest modelled aggregation; a further piece must buy 2 % (the model is not better than that)
int best_consumption(const SplitShape& s, int K, double* t_best) {
  int best = 1;
  double best_t = 0.0;
  for (int kc = 1; kc <= K; kc++) {
    if (K % kc) continue;
    const double t = model_split(s, kc);
    if (kc == 1 || t < best_t * 0.98) {
      best = kc;
      best_t = t;
    }
  }
  if (t_best) *t_best = best_t;
  return best;
}
}  // namespace

// How many pieces this rank consumes an exchange of K slices in, for rows of `len` columns (see lgraph.h)
int LearningGraph::consumption_rule(int K, int len) {
  gaib_graph* half = part_mode_ == PART_SPLIT ? halo_dev_ : cls_bhalo_;
  int64_t link_rows = link_rows_;
  if (link_rows < 0 && halo_plan_) link_rows = gaib_halo_link_rows(halo_plan_);
  if (link_rows < 0) link_rows = gaib_graph_nc(half) / 7 + 1;
  // (classes: interior rows + the boundary rows' owned columns run before the first wait = all of dev_'s edges)
  const SplitShape s{4.0 * len, wire_seconds(link_rows, len), gaib_graph_ne(dev_), gaib_graph_ne(half), gaib_graph_nv(half)};
  double t = 0.0;
  const int best = best_consumption(s, K, &t);
  if (getenv("GAIB_PART_VERBOSE"))
    fprintf(stderr, "[gaib] halo consumption: %d piece(s) of %d slice(s): modelled aggregation %.3f ms (in one piece %.3f ms; "
            "exchange %.3f ms per link)\n", best, K, t * 1e3, model_split(s, 1) * 1e3, s.t_wire * 1e3);
  return best;
}

int LearningGraph::halo_pieces(int len) {
  const int K = halo_slices();
  if (K <= 1 || part_mode_ < 0) return 1;
  gaib_graph* half = part_mode_ == PART_SPLIT ? halo_dev_ : (part_mode_ == PART_CLASSES ? cls_bhalo_ : NULL);
  if (!half || gaib_graph_ne(half) == 0) return 1;  // (the one-pass forms wait for the whole exchange)
  int want = pieces_want_;
  if (want < 0) {
    const char* e = getenv("GAIB_HALO_CONSUME");
    if (e && *e) want = atoi(e);
  }
  if (want < 0) {
    if (pieces_rule_k_ != K || pieces_rule_len_ != len) {
      pieces_rule_ = consumption_rule(K, len);
      pieces_rule_k_ = K;
      pieces_rule_len_ = len;
    }
    want = pieces_rule_;
  }
  if (want > K) want = K;
  while (want > 1 && K % want) want--;  // K' | K
  if (want <= 1) return 1;
  if (pieces_built_ == want && pieces_slices_ == K) return want;
  drop_pieces();
  std::vector<int64_t> rb, re;
  std::vector<int> rp;
  const int m = K / want;  // slices per piece
  if (halo_plan_) {
    int64_t b[64], e[64];
    for (int k = 0; k < K; k++) {
      int n = 0;
      GAIB_OR_DIE(gaib_halo_piece_ranges(halo_plan_, k, 64, b, e, &n));
      for (int j = 0; j < n; j++) {
        rb.push_back(b[j]);
        re.push_back(e[j]);
        rp.push_back(k / m);
      }
    }
  } else {
    rb = cb_range_begin_;
    re = cb_range_end_;
    for (int s : cb_range_piece_) rp.push_back(s / m);
  }
  GAIB_OR_DIE(gaib_graph_split_pieces(gpu_context::get(), half, want, (int)rb.size(), rb.data(), re.data(), rp.data(), pieces_));
  pieces_built_ = want;
  pieces_slices_ = K;
  return want;
}

// The mode of a partitioned graph's aggregations.  GAIB_PART_MODE = split | classes | onepass | onepass_all | auto (default), or
// set_partition_mode.  The rule (auto) models one aggregation in each form and takes the shorter (model_split above):
//   one pass    : max(exchange, the interior rows' work) + the boundary rows over [owned | halo] after the last row has landed
//   column split: the owned-column work of all rows, then the halo-column half -- one more read + write of the boundary rows'
//                 partial sums, below the gather rate on short row segments -- piece by piece where the exchange travels in
//                 slices, each piece as soon as its slices have landed (round 6; in as many pieces as the model likes best)
// with the exchange priced at GAIB_LINK_GBS per peer pair (default 100; unmeasured on this pool's one-GPU boxes: bench.py
// --gpus N measures the link first and puts the figure there) and the kernels at the rates measured on the shard benchmarks
// (scripts/papers_shard.py).  Where next to no row is interior (a random vertex order, or cut edges spread over
// every vertex: under 10 % of the edges in interior rows) the classes are not worth their second launch: the column split then
// runs over all rows as in round 3, the one pass over all rows of one [owned | halo] graph.  A rank decides for itself:
// every form runs the same exchange, so ranks need not agree.
int LearningGraph::partition_mode(int len) {
  if (part_mode_ >= 0) return part_mode_;
  if (!has_halo()) return part_mode_ = PART_SPLIT;
  int want = part_mode_wanted_;
  if (want < 0) {
    const char* e = getenv("GAIB_PART_MODE");
    if (e && *e) {
      if (!strcmp(e, "split") || !strcmp(e, "0")) want = PART_SPLIT;
      else if (!strcmp(e, "classes") || !strcmp(e, "1")) want = PART_CLASSES;
      else if (!strcmp(e, "onepass") || !strcmp(e, "2")) want = PART_ONEPASS;
      else if (!strcmp(e, "onepass_all") || !strcmp(e, "3")) want = PART_ONEPASS_ALL;
      else if (strcmp(e, "auto")) {
        fprintf(stderr, "GAIB_PART_MODE=%s: want split | classes | onepass | onepass_all | auto\n", e);
        exit(EXIT_FAILURE);
      }
    }
  }
  gaib_ctx* ctx = gpu_context::get();
  if (want == PART_SPLIT) return part_mode_ = PART_SPLIT;
  // the interior class first: its size decides (auto) and every class form needs it
  gaib_graph* gi = NULL;
  {
    const int rc = gaib_graph_split_classes(ctx, dev_, halo_dev_, &gi, NULL, NULL, NULL, &n_boundary_, &boundary_edges_, 0);
    // (a caller's halo graph without gaib_graph_set_vertex_norm -- per-edge weights only -- cannot be cut into classes: the
    // column split of round 3 needs nothing more than the two graphs; an explicit wish for classes fails loudly instead)
    if (rc != GAIB_OK && want < 0) return part_mode_ = PART_SPLIT;
    GAIB_OR_DIE(rc);
  }
  const int64_t ne_all = gaib_graph_ne(dev_) + gaib_graph_ne(halo_dev_);
  const int64_t ne_int = gaib_graph_ne(gi), ne_bhalo = gaib_graph_ne(halo_dev_);
  const bool few_interior = 10 * ne_int < ne_all;
  bool all_boundary = want == PART_ONEPASS_ALL;
  int mode = all_boundary ? (int)PART_ONEPASS : want;
  if (mode < 0) {
    int64_t link_rows = link_rows_;
    if (link_rows < 0 && halo_plan_) link_rows = gaib_halo_link_rows(halo_plan_);
    if (link_rows < 0) link_rows = (ne_bhalo ? gaib_graph_nc(halo_dev_) : 0) / 7 + 1;  // (a callback transport that gave no figure: 8 ranks)
    const double row_bytes = 4.0 * len;
    const double t_exchange = wire_seconds(link_rows, len);
    // one pass: the interior rows' work hides the exchange (where a fair share of the edges is interior), the boundary rows --
    // all rows where next to none is interior -- run over [owned | halo] after the last row has landed
    const double t_interior = few_interior ? 0.0 : ne_int * (row_bytes + 8) / 7.5e12;
    const double t_onepass = (t_exchange > t_interior ? t_exchange : t_interior) + (ne_all - (few_interior ? 0 : ne_int)) * (row_bytes + 8) / 7.5e12;
    // the column split (of the boundary rows; of all rows where next to none is interior): the owned-column work of ALL rows
    // hides the exchange, the halo-column half follows -- piece by piece where the exchange travels in slices (round 6), in as
    // many pieces as serve it best
    const SplitShape shape{row_bytes, t_exchange, gaib_graph_ne(dev_), ne_bhalo, few_interior ? (int64_t)size() : n_boundary_};
    const int K = halo_slices();
    double t_split = 0.0;
    const int kc = best_consumption(shape, K > 1 ? K : 1, &t_split);
    const bool onepass = t_onepass <= t_split;
    mode = onepass ? PART_ONEPASS : (few_interior ? PART_SPLIT : PART_CLASSES);
    all_boundary = onepass && few_interior;
    if (getenv("GAIB_PART_VERBOSE"))
      fprintf(stderr, "[gaib] partition mode %s%s: %lld of %lld rows on the boundary (%.1f %% of the edges in interior rows), "
              "exchange %.3f ms per link in %d slice(s); modelled aggregation: one pass %.3f ms, column split %.3f ms in %d piece(s)\n",
              mode == PART_ONEPASS ? "onepass" : (mode == PART_CLASSES ? "classes" : "split"), all_boundary ? " (all rows)" : "",
              (long long)n_boundary_, (long long)size(), 100.0 * ne_int / (ne_all > 0 ? ne_all : 1), t_exchange * 1e3, K,
              t_onepass * 1e3, t_split * 1e3, kc);
  }
  if (mode == PART_SPLIT) {
    gaib_graph_destroy(gi);
    return part_mode_ = PART_SPLIT;
  }
  if (all_boundary) {  // one [owned | halo] graph over all rows (the interior class is empty)
    gaib_graph_destroy(gi);
    gi = NULL;
    GAIB_OR_DIE(gaib_graph_split_classes(ctx, dev_, halo_dev_, &gi, NULL, NULL, &cls_bfull_, NULL, NULL, GAIB_SPLIT_ALL_BOUNDARY));
  } else if (mode == PART_ONEPASS) {
    GAIB_OR_DIE(gaib_graph_split_classes(ctx, dev_, halo_dev_, NULL, NULL, NULL, &cls_bfull_, NULL, NULL, 0));
  } else {
    GAIB_OR_DIE(gaib_graph_split_classes(ctx, dev_, halo_dev_, NULL, &cls_bown_, &cls_bhalo_, NULL, NULL, NULL, 0));
  }
  cls_int_ = gi;
  return part_mode_ = mode;
}

void LearningGraph::allocateFrom(index_t nv, index_t ne) {
  num_vertices_ = nv;
  num_edges_ = ne;
  rowptr_ = new index_t[(size_t)nv + 1];
  colidx_ = new index_t[ne > 0 ? ne : 1];
  rowptr_[0] = 0;
}

void LearningGraph::degree_counting() {
  index_t m = 0;
#pragma omp parallel for reduction(max : m)
  for (int64_t v = 0; v < (int64_t)num_vertices_; v++) {
    index_t d = rowptr_[v + 1] - rowptr_[v];
    if (d > m) m = d;
  }
  max_degree = m;
}

// Insert v into row v, keeping the row sorted: entries <= v stay, v goes after them, the rest
// shift by one; row v as a whole shifts by v.  Same result as the reference's single pass
// (lgraph.h:185-218) for sorted rows without self loops (its precondition, Q16).
void LearningGraph::add_selfloop() {
  assert(rowptr_ && colidx_);
  const size_t nv = num_vertices_;
  index_t* nc = new index_t[(size_t)num_edges_ + nv];
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t v = 0; v < (int64_t)nv; v++) {
    const index_t b = rowptr_[v], e = rowptr_[v + 1];
    index_t* dst = nc + b + v;
    index_t k = b;
    while (k < e && colidx_[k] <= (index_t)v) *dst++ = colidx_[k++];
    *dst++ = (index_t)v;
    while (k < e) *dst++ = colidx_[k++];
  }
  for (size_t v = 0; v <= nv; v++) rowptr_[v] += (index_t)v;
  delete[] colidx_;
  colidx_ = nc;
  num_edges_ += num_vertices_;
  if (dev_) {  // the device copy is stale
    gaib_graph_destroy(dev_);
    dev_ = NULL;
  }
}

LearningGraph* LearningGraph::generate_masked_graph(mask_t* masks) {
  const size_t n = size();
  LearningGraph* mg = new LearningGraph(is_device);
  std::vector<index_t> off(n + 1, 0);
  for (size_t s = 0; s < n; s++) {
    index_t d = 0;
    if (masks[s] == 1)
      for (index_t e = rowptr_[s]; e < rowptr_[s + 1]; e++) d += masks[colidx_[e]] == 1;
    off[s + 1] = off[s] + d;
  }
  mg->allocateFrom((index_t)n, off[n]);
#pragma omp parallel for
  for (int64_t s = 0; s < (int64_t)n; s++) {
    mg->fixEndEdge((index_t)s, off[s + 1]);
    if (masks[s] != 1) continue;
    index_t idx = off[s];
    for (index_t e = rowptr_[s]; e < rowptr_[s + 1]; e++)
      if (masks[colidx_[e]] == 1) mg->constructEdge(idx++, colidx_[e]);
  }
  std::cout << "masked graph: num_vertices = " << mg->size() << ", num_edges = " << mg->sizeEdges() << "\n";
  return mg;
}

void LearningGraph::alloc_on_device() { /* storage is created by copy_to_gpu (one hipMalloc set) */ }
void LearningGraph::alloc_on_device(index_t) {}

void LearningGraph::copy_to_gpu() {
  assert(rowptr_ && colidx_);
  if (dev_) gaib_graph_destroy(dev_);
  dev_ = NULL;
  GAIB_OR_DIE(gaib_graph_create(gpu_context::get(), num_vertices_, num_edges_, rowptr_, 32, colidx_, 0, &dev_));
}

void LearningGraph::copy_to_cpu() {
  assert(dev_);
  const int64_t nv = gaib_graph_nv(dev_), ne = gaib_graph_ne(dev_);
  std::vector<int64_t> rp(nv + 1);
  delete[] rowptr_;
  delete[] colidx_;
  allocateFrom((index_t)nv, (index_t)ne);
  GAIB_OR_DIE(gaib_memcpy_d2h(gpu_context::get(), rp.data(), gaib_graph_rowptr(dev_), sizeof(int64_t) * (nv + 1)));
  for (int64_t i = 0; i <= nv; i++) rowptr_[i] = (index_t)rp[i];
  GAIB_OR_DIE(gaib_memcpy_d2h(gpu_context::get(), colidx_, gaib_graph_colidx(dev_), sizeof(index_t) * ne));
}

void LearningGraph::compute_vertex_data() {
  if (!dev_) copy_to_gpu();
  GAIB_OR_DIE(gaib_graph_compute_vertex_data(gpu_context::get(), dev_));
  delete[] vertex_data_;
  vertex_data_ = NULL;
}

void LearningGraph::compute_edge_data() {
  if (!dev_) copy_to_gpu();
  GAIB_OR_DIE(gaib_graph_compute_edge_data(gpu_context::get(), dev_));
  delete[] edge_data_;
  edge_data_ = NULL;
}

vdata_t LearningGraph::get_vertex_data(index_t vid) {
  if (!vertex_data_) {
    assert(dev_ && gaib_graph_vertex_data(dev_));
    vertex_data_ = new vdata_t[num_vertices_];
    GAIB_OR_DIE(gaib_memcpy_d2h(gpu_context::get(), vertex_data_, gaib_graph_vertex_data(dev_),
                                sizeof(vdata_t) * num_vertices_));
  }
  return vertex_data_[vid];
}

edata_t LearningGraph::get_edge_data(index_t eid) {
  if (!edge_data_) {
    assert(dev_ && gaib_graph_edge_data(dev_));
    edge_data_ = new edata_t[num_edges_];
    GAIB_OR_DIE(gaib_memcpy_d2h(gpu_context::get(), edge_data_, gaib_graph_edge_data(dev_),
                                sizeof(edata_t) * num_edges_));
  }
  return edge_data_[eid];
}

void LearningGraph::dealloc() {
  delete[] rowptr_;
  delete[] colidx_;
  delete[] vertex_data_;
  delete[] edge_data_;
  rowptr_ = colidx_ = NULL;
  vertex_data_ = edge_data_ = NULL;
  if (dev_) gaib_graph_destroy(dev_);
  dev_ = NULL;
  gaib_graph* cls[] = {cls_int_, cls_bown_, cls_bhalo_, cls_bfull_};  // built by this object, whoever owns the rest
  for (gaib_graph* c : cls)
    if (c) gaib_graph_destroy(c);
  cls_int_ = cls_bown_ = cls_bhalo_ = cls_bfull_ = NULL;
  drop_pieces();
  part_mode_ = -1;
  if (owns_partition_) {
    if (halo_plan_) gaib_halo_destroy(halo_plan_);
    if (halo_dev_) gaib_graph_destroy(halo_dev_);
    if (gat_full_) gaib_graph_destroy(gat_full_);
    if (gat_t_) gaib_graph_destroy(gat_t_);
    if (gat_tperm_) gaib_free(gpu_context::get(), gat_tperm_);
  }
  halo_plan_ = NULL;
  halo_dev_ = gat_full_ = gat_t_ = NULL;
  gat_tperm_ = NULL;
  owns_partition_ = false;
}

void LearningGraph::print_graph() {
  std::cout << "Printing the graph: \n";
  for (index_t n = 0; n < num_vertices_; n++) {
    std::cout << "vertex " << n << ": degree = " << get_degree(n) << " edgelist = [ ";
    for (index_t e = rowptr_[n]; e != rowptr_[n + 1]; e++) std::cout << colidx_[e] << " ";
    std::cout << "]\n";
  }
}
