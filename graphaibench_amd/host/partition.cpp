// partition.cpp -- vertex-range partition builder of the multi-rank trainer (include/gnn/partition.h).
#include <algorithm>
#include <math.h>
#include "partition.h"
#include <stdlib.h>
#include "host_util.h"

std::vector<int64_t> vertex_range_bounds(int64_t n, int world) {
  const int64_t per = (n + world - 1) / world;
  std::vector<int64_t> b(world + 1);
  for (int p = 0; p <= world; p++) b[p] = std::min<int64_t>((int64_t)p * per, n);
  return b;
}

VertexRangePartition build_vertex_range_partition(int64_t n, const index_t* rowptr, const index_t* colidx, int rank,
                                                  int world) {
  VertexRangePartition P;
  P.rank = rank;
  P.world = world;
  P.n_global = n;
  const std::vector<int64_t> bounds = vertex_range_bounds(n, world);
  P.lo = bounds[rank];
  P.hi = bounds[rank + 1];
  const int64_t n_own = P.hi - P.lo;
  // ---- halo set: columns of the owned rows outside [lo, hi), ascending ----
  std::vector<uint8_t> mark((size_t)n, 0);
  for (int64_t v = P.lo; v < P.hi; v++)
    for (index_t e = rowptr[v]; e < rowptr[v + 1]; e++) {
      const index_t c = colidx[e];
      if ((int64_t)c < P.lo || (int64_t)c >= P.hi) mark[c] = 1;
    }
  // Round 5: a peer range of which this rank needs at least a share `complete` of the rows is taken WHOLE -- its
  // send list then is its full row range, one run of consecutive rows, which the RCCL transport sends straight from the
  // matrix without packing (comm.hip; the same rule as dist.py's split_by_owner, so both builders give the same partition)
  // The share from which that pays (dist.py: complete_halo_threshold): the rows nobody reads cross world - 1 links, the pack
  // they save is a read and a write at ~5 TB/s: share > 1 - 2 (world - 1) link / 5000 GB/s, link = GAIB_LINK_GBS (100), >= 0.5
  const char* ch = getenv("GAIB_COMPLETE_HALO");
  const char* lk = getenv("GAIB_LINK_GBS");
  const double link_gbs = lk && *lk ? atof(lk) : 100.0;
  const double complete = ch && *ch ? atof(ch) : (world < 2 ? 0.0 : std::max(0.5, 1.0 - 2.0 * (world - 1) * link_gbs / 5000.0));
  if (complete > 0.0)
    for (int q = 0; q < world; q++) {
      if (q == rank) continue;
      const int64_t size_q = bounds[q + 1] - bounds[q];
      int64_t cnt = 0;
      for (int64_t v = bounds[q]; v < bounds[q + 1]; v++) cnt += mark[v];
      if (size_q > 0 && (double)cnt >= complete * (double)size_q && cnt < size_q)
        for (int64_t v = bounds[q]; v < bounds[q + 1]; v++) mark[v] = 1;
    }
  std::vector<int64_t> halo_slot((size_t)n, -1);  // global id -> position in halo_gids
  for (int64_t v = 0; v < n; v++)
    if (mark[v]) {
      halo_slot[v] = (int64_t)P.halo_gids.size();
      P.halo_gids.push_back(v);
      P.halo_degree.push_back((int64_t)rowptr[v + 1] - (int64_t)rowptr[v]);
    }
  P.recv_counts.assign(world, 0);
  for (int q = 0, k = 0; q < world; q++) {
    while (k < (int)P.halo_gids.size() && P.halo_gids[k] < bounds[q + 1]) {
      P.recv_counts[q]++;
      k++;
    }
  }
  // ---- the rows' edges split by column owner; the order of the edges inside a row is kept ----
  P.rowptr_own.assign(n_own + 1, 0);
  P.rowptr_halo.assign(n_own + 1, 0);
  P.degree.resize(n_own);
  for (int64_t i = 0; i < n_own; i++) {
    const int64_t v = P.lo + i;
    int64_t a = 0, b = 0;
    for (index_t e = rowptr[v]; e < rowptr[v + 1]; e++) {
      const index_t c = colidx[e];
      ((int64_t)c >= P.lo && (int64_t)c < P.hi) ? a++ : b++;
    }
    P.rowptr_own[i + 1] = P.rowptr_own[i] + a;
    P.rowptr_halo[i + 1] = P.rowptr_halo[i] + b;
    P.degree[i] = a + b;
  }
  P.colidx_own.resize(std::max<int64_t>(P.rowptr_own[n_own], 1));
  P.colidx_halo.resize(std::max<int64_t>(P.rowptr_halo[n_own], 1));
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t i = 0; i < n_own; i++) {
    const int64_t v = P.lo + i;
    int64_t a = P.rowptr_own[i], b = P.rowptr_halo[i];
    for (index_t e = rowptr[v]; e < rowptr[v + 1]; e++) {
      const index_t c = colidx[e];
      if ((int64_t)c >= P.lo && (int64_t)c < P.hi) P.colidx_own[a++] = (index_t)(c - P.lo);
      else P.colidx_halo[b++] = (index_t)halo_slot[c];
    }
  }
  P.colidx_own.resize(P.rowptr_own[n_own]);
  P.colidx_halo.resize(P.rowptr_halo[n_own]);
  // ---- what every peer q will ask of this rank: the columns inside [lo, hi) that q's rows touch, ascending
  //      (== the segment of q's halo_gids that this rank owns) ----
  P.send_counts.assign(world, 0);
  std::vector<uint8_t> want((size_t)std::max<int64_t>(n_own, 1));
  for (int q = 0; q < world; q++) {
    if (q == rank) continue;
    std::fill(want.begin(), want.end(), 0);
    for (int64_t v = bounds[q]; v < bounds[q + 1]; v++)
      for (index_t e = rowptr[v]; e < rowptr[v + 1]; e++) {
        const index_t c = colidx[e];
        if ((int64_t)c >= P.lo && (int64_t)c < P.hi) want[c - P.lo] = 1;
      }
    if (complete > 0.0) {  // (the peer applies the rule above to THIS rank's range: same count, same decision)
      int64_t cnt = 0;
      for (int64_t i = 0; i < n_own; i++) cnt += want[i];
      if (n_own > 0 && (double)cnt >= complete * (double)n_own && cnt < n_own) std::fill(want.begin(), want.end(), 1);
    }
    for (int64_t i = 0; i < n_own; i++)
      if (want[i]) {
        P.send_idx.push_back(i);
        P.send_counts[q]++;
      }
  }
  return P;
}

void build_gat_structures(VertexRangePartition& P, const index_t* rowptr, const index_t* colidx) {
  const int64_t n_own = P.n_own(), n_halo = P.n_halo(), nc = n_own + n_halo;
  // global id -> local column id: owned vertices first, then the halo vertices in ascending global order
  auto local_of = [&](index_t c) -> index_t {
    if ((int64_t)c >= P.lo && (int64_t)c < P.hi) return (index_t)(c - P.lo);
    const auto it = std::lower_bound(P.halo_gids.begin(), P.halo_gids.end(), (int64_t)c);
    return (index_t)(n_own + (it - P.halo_gids.begin()));
  };
  P.rowptr_full.assign(n_own + 1, 0);
  for (int64_t i = 0; i < n_own; i++) P.rowptr_full[i + 1] = P.rowptr_full[i] + P.degree[i];
  const int64_t ne = P.rowptr_full[n_own];
  P.colidx_full.resize(std::max<int64_t>(ne, 1));
#pragma omp parallel for schedule(dynamic, 1024)
  for (int64_t i = 0; i < n_own; i++) {
    int64_t k = P.rowptr_full[i];
    for (index_t e = rowptr[P.lo + i]; e < rowptr[P.lo + i + 1]; e++) P.colidx_full[k++] = local_of(colidx[e]);
  }
  P.colidx_full.resize(ne);
  // structural symmetry of this rank's rows, checked against the global CSR (sorted rows): the reverse of every edge exists
  int asym = 0;
#pragma omp parallel for schedule(dynamic, 1024) reduction(| : asym)
  for (int64_t i = 0; i < n_own; i++) {
    const index_t v = (index_t)(P.lo + i);
    for (index_t e = rowptr[v]; e < rowptr[v + 1]; e++) {
      const index_t c = colidx[e];
      if (!std::binary_search(colidx + rowptr[c], colidx + rowptr[c + 1], v)) asym = 1;
    }
  }
  P.rows_symmetric = asym == 0;
  // transpose by counting sort on the local column id; stable, so the rows of a column come out ascending
  P.rowptr_t.assign(nc + 1, 0);
  for (int64_t e = 0; e < ne; e++) P.rowptr_t[P.colidx_full[e] + 1]++;
  for (int64_t v = 0; v < nc; v++) P.rowptr_t[v + 1] += P.rowptr_t[v];
  P.colidx_t.resize(std::max<int64_t>(ne, 1));
  P.tperm.resize(std::max<int64_t>(ne, 1));
  std::vector<int64_t> cur(P.rowptr_t.begin(), P.rowptr_t.end() - 1);
  for (int64_t i = 0; i < n_own; i++)
    for (int64_t e = P.rowptr_full[i]; e < P.rowptr_full[i + 1]; e++) {
      const int64_t k = cur[P.colidx_full[e]]++;
      P.colidx_t[k] = (index_t)i;
      P.tperm[k] = (index_t)e;
    }
  P.colidx_t.resize(ne);
  P.tperm.resize(ne);
}

// deg^-1/2 (0 for isolated vertices) and (float)(1.0 / float(deg)) with the roundings of compute_vertex_data
// (src/gnn/lgraph.cpp:22-34) and sage_aggregator.cpp:18,44 -- the same expressions as csrc/graph.hip
static void normalisers(const std::vector<int64_t>& deg, std::vector<float>& vd, std::vector<float>& inv) {
  vd.resize(std::max<size_t>(deg.size(), 1));
  inv.resize(std::max<size_t>(deg.size(), 1));
  for (size_t i = 0; i < deg.size(); i++) {
    const float t = sqrtf((float)deg[i]);
    vd[i] = (t == 0.0f) ? 0.0f : (float)(1.0 / (double)t);
    inv[i] = (float)(1.0 / (double)(float)deg[i]);
  }
}

static float* upload(const std::vector<float>& h) {
  float* d = gaib_host::dmalloc<float>(h.size());
  GAIB_OR_DIE(gaib_memcpy_h2d(gpu_context::get(), d, h.data(), sizeof(float) * h.size()));
  return d;
}

LearningGraph* make_partitioned_graph(const VertexRangePartition& P, gaib_comm* comm) {
  gaib_ctx* ctx = gpu_context::get();
  const int64_t n_own = P.n_own(), n_halo = P.n_halo();
  std::vector<float> vd, inv, vd_h, inv_h;
  normalisers(P.degree, vd, inv);
  normalisers(P.halo_degree, vd_h, inv_h);
  float *d_vd = upload(vd), *d_inv = upload(inv), *d_vd_h = upload(vd_h), *d_inv_h = upload(inv_h);
  gaib_graph* g_own = nullptr;
  const index_t dummy = 0;
  GAIB_OR_DIE(gaib_graph_create(ctx, n_own, (int64_t)P.colidx_own.size(), P.rowptr_own.data(), 64,
                                P.colidx_own.empty() ? &dummy : P.colidx_own.data(), 0, &g_own));
  GAIB_OR_DIE(gaib_graph_set_vertex_norm(ctx, g_own, d_vd, d_inv, d_vd, d_inv));
  LearningGraph* lg = LearningGraph::adopt_device(g_own);
  if (P.world > 1) {
    if (!comm) {
      fprintf(stderr, "make_partitioned_graph: world %d needs a communicator\n", P.world);
      exit(EXIT_FAILURE);
    }
    // every rank takes part in every exchange, also one without halo rows of its own
    gaib_graph* g_halo = nullptr;
    GAIB_OR_DIE(gaib_graph_create_rect(ctx, n_own, std::max<int64_t>(n_halo, 1), (int64_t)P.colidx_halo.size(),
                                       P.rowptr_halo.data(), 64, P.colidx_halo.empty() ? &dummy : P.colidx_halo.data(), 0,
                                       &g_halo));
    GAIB_OR_DIE(gaib_graph_set_vertex_norm(ctx, g_halo, d_vd, d_inv, d_vd_h, d_inv_h));
    gaib_halo* plan = nullptr;
    GAIB_OR_DIE(gaib_halo_create(comm, P.send_counts.data(), P.send_idx.data(), 0, P.recv_counts.data(), &plan));
    // the exchange in time slices where the ranges are large enough to pay for it (the same figure on every rank)
    GAIB_OR_DIE(gaib_halo_set_pieces(plan, gaib_halo_default_pieces(P.n_global, P.world)));
    lg->set_halo_plan(g_halo, plan);
  }
  if (!P.rowptr_full.empty()) {
    // GAT: one rectangular graph over [owned | halo] columns + its transpose + the edge permutation between them
    const int64_t nc = n_own + n_halo, ne = (int64_t)P.colidx_full.size();
    gaib_graph *g_full = nullptr, *g_t = nullptr;
    GAIB_OR_DIE(gaib_graph_create_rect(ctx, n_own, nc, ne, P.rowptr_full.data(), 64,
                                       P.colidx_full.empty() ? &dummy : P.colidx_full.data(), 0, &g_full));
    GAIB_OR_DIE(gaib_graph_create_rect(ctx, nc, std::max<int64_t>(n_own, 1), ne, P.rowptr_t.data(), 64,
                                       P.colidx_t.empty() ? &dummy : P.colidx_t.data(), 0, &g_t));
    index_t* d_tperm = gaib_host::dmalloc<index_t>((size_t)ne);
    if (ne) GAIB_OR_DIE(gaib_memcpy_h2d(ctx, d_tperm, P.tperm.data(), sizeof(index_t) * (size_t)ne));
    lg->set_gat_partition(g_full, g_t, d_tperm, n_halo, ne);
    // the one-sweep kernels need a structurally symmetric GLOBAL graph: every rank checked its rows; all ranks take the
    // same path (they run different collectives in backward), so the verdict is the sum over the ranks
    double asym = P.rows_symmetric ? 0.0 : 1.0;
    if (P.world > 1) GAIB_OR_DIE(gaib_allreduce_host_f64(comm, &asym, 1));
    lg->set_gat_symmetric(asym == 0.0);
    if (asym != 0.0 && P.rank == 0)
      fprintf(stderr, "make_partitioned_graph: the graph is not structurally symmetric (%d rank(s) hold an edge without its "
              "reverse): GAT on this partition takes the staged path (transposed structure + reverse exchange)\n", (int)asym);
  }
  lg->own_partition_objects();
  float* tmp[] = {d_vd, d_inv, d_vd_h, d_inv_h};  // set_vertex_norm copied them
  for (float* p : tmp) GAIB_OR_DIE(gaib_free(ctx, p));
  return lg;
}
