// sampler.cpp -- GraphSAINT-style frontier sampler over host CSR (see include/gnn/sampler.h).
#include <random>
#include "sampler.h"

Sampler::Sampler(Graph* g, Graph* tg, mask_t* masks, size_t count)
    : m(DEFAULT_SIZE_FRONTIER), count_(count), full_graph(g), masked_graph(tg) {
  for (size_t i = 0; i < full_graph->size(); i++)
    if (masks[i] == 1) trainingNodes.push_back((index_t)i);
  avg_deg = masked_graph->size() ? (int)(masked_graph->sizeEdges() / masked_graph->size()) : 0;
  subg_deg = avg_deg > SAMPLE_CLIP ? SAMPLE_CLIP : avg_deg;
}

namespace {
// Fenwick tree over non-negative integer weights: point update, prefix-sum search
struct Fenwick {
  std::vector<int64_t> t;
  std::vector<int64_t> w;
  int n, top;
  explicit Fenwick(int n_) : t(n_ + 1, 0), w(n_, 0), n(n_) {
    top = 1;
    while (top * 2 <= n) top *= 2;
  }
  void set(int i, int64_t v) {
    int64_t d = v - w[i];
    w[i] = v;
    for (int k = i + 1; k <= n; k += k & -k) t[k] += d;
  }
  int64_t total() const {
    int64_t s = 0;
    for (int k = n; k > 0; k -= k & -k) s += t[k];
    return s;
  }
  // smallest index i with prefix(i) > r, 0 <= r < total()
  int find(int64_t r) const {
    int pos = 0;
    for (int step = top; step > 0; step >>= 1)
      if (pos + step <= n && t[pos + step] <= r) {
        pos += step;
        r -= t[pos];
      }
    return pos;
  }
};
inline int64_t clipped_degree(Graph* g, index_t v) {
  int64_t d = (int64_t)g->edge_end_host(v) - (int64_t)g->edge_begin_host(v);
  return d > SAMPLE_CLIP ? SAMPLE_CLIP : d;
}
}  // namespace

size_t Sampler::select_vertices(index_t n, VertexSet& st, unsigned seed) {
  if (trainingNodes.empty() || n == 0) return st.size();
  const index_t fm = n < m ? n : m;
  std::mt19937 rng(seed);
  std::vector<index_t> frontier(fm);
  Fenwick fw((int)fm);
  for (index_t i = 0; i < fm; i++) {
    const index_t v = trainingNodes[rng() % trainingNodes.size()];
    frontier[i] = v;
    st.insert(v);
    fw.set((int)i, clipped_degree(masked_graph, v));
  }
  for (index_t itr = 0; itr < n - fm; itr++) {
    const int64_t tot = fw.total();
    if (tot == 0) break;  // every frontier vertex is isolated in the training graph
    const int slot = fw.find((int64_t)(rng() % (uint64_t)tot));
    const index_t v = frontier[slot];
    const index_t deg = masked_graph->edge_end_host(v) - masked_graph->edge_begin_host(v);
    const index_t u = masked_graph->getEdgeDstHost(masked_graph->edge_begin_host(v) + (index_t)(rng() % deg));
    st.insert(u);
    frontier[slot] = u;
    fw.set(slot, clipped_degree(masked_graph, u));
  }
  return st.size();
}

void Sampler::generateSubgraph(VertexSet& vertex_set, mask_t* masks, Graph* sg) {
  const size_t nfull = full_graph->size();
  std::fill(masks, masks + nfull, 0);
  std::vector<index_t> new_id(nfull, 0);
  index_t k = 0;
  for (index_t v : vertex_set) {  // ascending order -> monotone relabelling keeps rows sorted
    masks[v] = 1;
    new_id[v] = k++;
  }
  const index_t nv = k;
  std::vector<index_t> off(nv + 1, 0);
  k = 0;
  for (index_t v : vertex_set) {
    index_t d = 0;
    for (index_t e = full_graph->edge_begin_host(v); e < full_graph->edge_end_host(v); e++)
      d += masks[full_graph->getEdgeDstHost(e)];
    off[k + 1] = off[k] + d;
    k++;
  }
  sg->dealloc();
  sg->allocateFrom(nv, off[nv]);
  k = 0;
  for (index_t v : vertex_set) {
    index_t idx = off[k];
    for (index_t e = full_graph->edge_begin_host(v); e < full_graph->edge_end_host(v); e++) {
      const index_t dst = full_graph->getEdgeDstHost(e);
      if (masks[dst]) sg->constructEdge(idx++, new_id[dst]);
    }
    sg->fixEndEdge(k, off[k + 1]);
    k++;
  }
}
