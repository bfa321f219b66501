// sampler.cpp -- GraphSAINT frontier sampler over host CSR (see include/gnn/sampler.h): returns the reference's
// vertex sets and subgraphs for the same seeds (pinned against the reference's own sampler.cpp, tests/golden).
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <utility>
#include "sampler.h"

Sampler::Sampler(Graph* g, Graph* tg, mask_t* masks, size_t count)
    : m(DEFAULT_SIZE_FRONTIER), count_(count), full_graph(g), masked_graph(tg) {
  for (size_t i = 0; i < full_graph->size(); i++)
    if (masks[i] == 1) trainingNodes.push_back((index_t)i);
  avg_deg = masked_graph->size() ? (int)(masked_graph->sizeEdges() / masked_graph->size()) : 0;
  subg_deg = avg_deg > SAMPLE_CLIP ? SAMPLE_CLIP : avg_deg;
}

namespace {

inline int degree_of(Graph* g, index_t v) { return (int)(g->edge_end_host(v) - g->edge_begin_host(v)); }
inline int clipped(int d) { return d > SAMPLE_CLIP ? SAMPLE_CLIP : d; }

// The GraphSAINT "dashboard" the reference samples from (src/gnn/sampler.cpp:146-294, after GraphSAINT's
// ipdps19_cpp/sample.cpp): every frontier vertex owns as many consecutive slots as its clipped degree, a
// uniform draw over ALL slots (dead ones are rejected and redrawn) therefore picks a frontier vertex with
// probability proportional to its degree.  A popped vertex's slots die in place; the dashboard is rebuilt
// without them only when appending the replacement would outgrow its storage.  Which draws get rejected --
// and with that the whole rand_r stream -- depends on this layout and on WHEN the rebuilds happen, and the
// rebuild trigger in the reference is its std::vector capacity.  To return the same vertex set for the same
// seed this class keeps the same layout and tracks that capacity number (libstdc++ growth rules) explicitly.
class Dashboard {
 public:
  struct Slot {
    int vertex;  // -1: dead
    int link;    // head slot of a run: -(run length); other slots: distance back to the head
    int owner;   // 1-based index of the owning frontier entry
  };
  struct Entry {
    int weight;  // slots owned (0 once popped)
    int alive;
    int end;     // running sum of weights: one past this entry's last slot at the time it was appended
    int vertex;
  };

  explicit Dashboard(size_t reserved) : cap_(reserved) {}
  size_t slots() const { return slot_.size(); }
  const Slot& at(size_t j) const { return slot_[j]; }
  size_t entries() const { return entry_.size(); }

  // append a frontier entry; its slots go to [previous end, previous end + weight)
  void push(int vertex, int weight) {
    const int start = entry_.empty() ? 0 : entry_.back().end;
    entry_.push_back(Entry{weight, 1, start + weight, vertex});
    grow_to((size_t)(start + weight));
    write_run(slot_, start, start + weight, vertex, (int)entry_.size());
  }
  // the whole initial frontier at once: ONE growth step for all runs (sampler.cpp:181-195)
  void push_all(const std::vector<std::pair<int, int>>& vw) {
    int end = 0;
    for (auto& p : vw) {
      end += p.second;
      entry_.push_back(Entry{p.second, 1, end, p.first});
    }
    grow_to((size_t)end);
    int start = 0;
    for (size_t i = 0; i < vw.size(); i++) {
      write_run(slot_, start, entry_[i].end, vw[i].first, (int)i + 1);
      start = entry_[i].end;
    }
  }
  // head slot of the run that slot j belongs to
  size_t head(size_t j) const { return slot_[j].link < 0 ? j : j - (size_t)slot_[j].link; }
  // kill the run starting at head slot h
  void pop(size_t h) {
    Entry& e = entry_[(size_t)slot_[h].owner - 1];
    e.alive = 0;
    e.weight = 0;
    const size_t len = (size_t)(-slot_[h].link);
    for (size_t j = h; j < h + len; j++) slot_[j].vertex = -1;
  }
  // would appending `extra` slots outgrow the storage?  (sampler.cpp:226)
  bool must_rebuild(int extra) const { return slot_.size() + (size_t)extra > cap_; }
  // drop dead runs and zero-weight entries, renumber owners (sampler.cpp:227-270)
  void rebuild() {
    if (getenv("GAIB_SAMPLER_TRACE")) fprintf(stderr, "dashboard rebuild at %zu slots, cap %zu\n", slot_.size(), cap_);
    std::vector<int> run_end(entry_.size());
    int total = 0;
    for (size_t i = 0; i < entry_.size(); i++) run_end[i] = (total += entry_[i].weight);
    std::vector<Slot> fresh((size_t)total, Slot{0, 0, 0});
    for (size_t i = 0; i < entry_.size(); i++) {
      entry_[i].end = run_end[i];
      if (entry_[i].alive) write_run(fresh, i ? run_end[i - 1] : 0, run_end[i], entry_[i].vertex, (int)i + 1);
    }
    std::vector<int> new_index(entry_.size());
    int live = 0;
    for (size_t i = 0; i < entry_.size(); i++) new_index[i] = (live += entry_[i].alive);
    if (fresh.size() > cap_) cap_ = fresh.size();  // vector::assign reallocates to the exact size only when it must
    slot_.swap(fresh);
    for (auto& sl : slot_) sl.owner = new_index[(size_t)sl.owner - 1];
    size_t k = 0;
    for (size_t i = 0; i < entry_.size(); i++)
      if (entry_[i].weight != 0) entry_[k++] = entry_[i];
    entry_.resize(k);
  }

 private:
  static void write_run(std::vector<Slot>& v, int start, int end, int vertex, int owner) {
    for (int j = start; j < end; j++) v[(size_t)j] = Slot{vertex, j == start ? start - end : j - start, owner};
  }
  // checkGSDB (sampler.cpp:149-159): double the reservation once if it is too small, then resize -- and a resize
  // beyond the reservation grows it to size + max(size, missing) like libstdc++'s vector does
  void grow_to(size_t n) {
    if (cap_ < n) cap_ *= 2;
    if (n > cap_) cap_ = slot_.size() + std::max(slot_.size(), n - slot_.size());
    slot_.resize(n, Slot{0, 0, 0});
  }
  std::vector<Slot> slot_;
  std::vector<Entry> entry_;
  size_t cap_;  // what std::vector::capacity() of the reference's dashboard arrays would be
};

}  // namespace

// Same vertex set as the reference for the same (graph, training set, n, seed): same rand_r stream, same
// dashboard (see above).  Where the reference would divide by zero or spin forever (every frontier vertex
// isolated in the training graph) this returns what has been collected so far.
size_t Sampler::select_vertices(index_t n, VertexSet& st, unsigned seed) {
  if (trainingNodes.empty() || n == 0) return st.size();
  if (n < m) m = n;  // (sticks for later calls, as in the reference)
  unsigned state = seed;
  Dashboard db((size_t)(subg_deg * m * ETA));
  std::vector<std::pair<int, int>> first(m);
  for (index_t i = 0; i < m; i++) {
    const index_t v = trainingNodes[(size_t)rand_r(&state) % trainingNodes.size()];
    st.insert(v);
    first[i] = std::make_pair((int)v, clipped(degree_of(masked_graph, v)));
  }
  db.push_all(first);
  for (index_t itr = 0; itr < n - m; itr++) {
    if (db.slots() == 0) break;
    size_t pick = 0;
    bool found = false;
    for (size_t tries = 0; tries < 64 * db.slots() + 1024; tries++) {  // rejection sampling over dead slots
      pick = (size_t)rand_r(&state) % db.slots();
      if (db.at(pick).vertex != -1) {
        found = true;
        break;
      }
    }
    if (!found) break;  // (practically: no live slot left)
    const size_t h = db.head(pick);
    const index_t v = (index_t)db.at(h).vertex;
    const int deg = degree_of(masked_graph, v);
    int fresh_vertex = -1, fresh_weight = 0;
    if (deg != 0) {
      const index_t e = masked_graph->edge_begin_host(v) + (index_t)(rand_r(&state) % deg);
      fresh_vertex = (int)masked_graph->getEdgeDstHost(e);
      st.insert((index_t)fresh_vertex);
      db.pop(h);
      fresh_weight = clipped(degree_of(masked_graph, (index_t)fresh_vertex));
    }
    if (db.must_rebuild(fresh_weight)) db.rebuild();
    if (db.entries() == 0 && fresh_weight == 0) break;
    db.push(fresh_vertex, fresh_weight);
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
