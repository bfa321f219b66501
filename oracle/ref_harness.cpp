// ref_harness.cpp -- thin extern "C" shim over the REFERENCE's own LearningGraph / Reader.
//
// TEST INFRASTRUCTURE ONLY.  This file contains no reference code: it #includes the
// reference headers from where they lie (/root/reference/include/gnn) and is linked with
// the reference's src/gnn/lgraph.cpp and src/gnn/reader.cpp compiled unmodified (see
// oracle/Makefile, target `ref`).  Output goes to oracle/_ref/ (git-ignored).  It exists to
// pin oracle/gnn_oracle.c's a1 functions and the binary reader against the real reference.
//
// include/gnn/global.h:61 hard-defines ENABLE_GPU, so LearningGraph's accessors
// (edge_begin/getEdgeDst, lgraph.h:161-166) read the d_* "device" pointers; the subclass
// below points them at the host arrays, which is what a CPU build would see.
#include "lgraph.h"
#include "reader.h"
#include "sampler.h"
#include <cstring>

std::map<char, double> time_ops;  // extern in include/gnn/global.h:77

class RefGraph : public LearningGraph {
 public:
  RefGraph() : LearningGraph(false) {}
  void alias() {
    d_rowptr_ = rowptr_;
    d_colidx_ = colidx_;
    d_vertex_data_ = vertex_data_;
    d_edge_data_ = edge_data_;
  }
  void load(uint32_t nv, uint32_t ne, const uint32_t* rp, const uint32_t* ci) {
    allocateFrom(nv, ne);
    memcpy(rowptr_, rp, sizeof(uint32_t) * (nv + 1));
    memcpy(colidx_, ci, sizeof(uint32_t) * ne);
    alias();
  }
  const uint32_t* rp() const { return rowptr_; }
  const uint32_t* ci() const { return colidx_; }
  const float* vd() const { return vertex_data_; }
  const float* ed() const { return edge_data_; }
};

extern "C" {

// LearningGraph::add_selfloop (include/gnn/lgraph.h:185-218)
void ref_add_selfloop(uint32_t nv, uint32_t ne, const uint32_t* rp, const uint32_t* ci,
                      uint32_t* rp_out, uint32_t* ci_out) {
  RefGraph g;
  g.load(nv, ne, rp, ci);
  g.add_selfloop();
  memcpy(rp_out, g.rp(), sizeof(uint32_t) * (nv + 1));
  memcpy(ci_out, g.ci(), sizeof(uint32_t) * (size_t)(ne + nv));
}

// LearningGraph::compute_vertex_data (src/gnn/lgraph.cpp:22-34)
void ref_vertex_data(uint32_t nv, uint32_t ne, const uint32_t* rp, const uint32_t* ci, float* vd) {
  RefGraph g;
  g.load(nv, ne, rp, ci);
  g.compute_vertex_data();
  memcpy(vd, g.vd(), sizeof(float) * nv);
}

// LearningGraph::compute_edge_data (src/gnn/lgraph.cpp:6-20)
void ref_edge_data(uint32_t nv, uint32_t ne, const uint32_t* rp, const uint32_t* ci, float* ed) {
  RefGraph g;
  g.load(nv, ne, rp, ci);
  g.compute_edge_data();
  g.alias();
  memcpy(ed, g.ed(), sizeof(float) * ne);
}

// Reader::bin_read_graph / bin_read_vlabels / bin_read_features (src/gnn/reader.cpp:414-457,
// 347-412, 248-268).  DATASET_PATH must be set (and end with '/') before this library is
// loaded: include/gnn/configs.h:5 reads it at static-init time.
// Two-call protocol: first call with NULL buffers returns sizes.
int ref_read_dataset(const char* name, uint32_t* nv, uint32_t* ne, uint32_t* rp, uint32_t* ci,
                     int* num_cls, uint8_t* labels, int* feat_len, float* feats, int want_feats) {
  static RefGraph* g = nullptr;
  static Reader* r = nullptr;
  static std::vector<label_t> lab;
  static std::vector<float> ft;
  static int ncls = 0, flen = 0;
  if (!rp) {
    delete r;
    g = new RefGraph();
    r = new Reader(name);
    r->bin_read_graph(g);
    ncls = r->bin_read_vlabels(lab, true);
    flen = want_feats ? (int)r->bin_read_features(ft) : 0;
    *nv = (uint32_t)g->size();
    *ne = (uint32_t)g->sizeEdges();
    *num_cls = ncls;
    *feat_len = flen;
    return 0;
  }
  memcpy(rp, g->rp(), sizeof(uint32_t) * (g->size() + 1));
  memcpy(ci, g->ci(), sizeof(uint32_t) * g->sizeEdges());
  if (labels) memcpy(labels, lab.data(), lab.size());
  if (feats && flen) memcpy(feats, ft.data(), sizeof(float) * ft.size());
  return 0;
}

// LearningGraph::generate_masked_graph (include/gnn/lgraph.h:231-270): two-call protocol (NULL ci_out -> *ne_out)
void ref_masked_graph(uint32_t nv, uint32_t ne, const uint32_t* rp, const uint32_t* ci, uint8_t* masks,
                      uint32_t* ne_out, uint32_t* rp_out, uint32_t* ci_out) {
  RefGraph g;
  g.load(nv, ne, rp, ci);
  LearningGraph* mg = g.generate_masked_graph(masks);
  *ne_out = (uint32_t)mg->sizeEdges();
  if (ci_out) {
    for (uint32_t v = 0; v <= nv; v++) rp_out[v] = v == 0 ? 0 : (uint32_t)mg->edge_end_host(v - 1);
    for (uint32_t e = 0; e < *ne_out; e++) ci_out[e] = mg->getEdgeDstHost(e);
  }
}

// Sampler::select_vertices + Sampler::generateSubgraph (src/gnn/sampler.cpp:146-294, 128-144), driven the way
// net.cpp:288-300 does: Sampler(full graph, training-masked graph, training masks, count).
// kept[] receives the sorted vertex set (capacity n); the subgraph comes back through the two-call protocol:
// first call (sub_ci == NULL) samples and returns sizes, second call copies.
static RefGraph* s_sub = nullptr;
uint32_t ref_sample_subgraph(uint32_t nv, uint32_t ne, const uint32_t* rp, const uint32_t* ci,
                             uint8_t* train_masks, uint32_t n, unsigned seed, uint32_t* kept,
                             uint32_t* sub_ne, uint32_t* sub_rp, uint32_t* sub_ci) {
  if (sub_ci || sub_rp) {
    uint32_t snv = (uint32_t)s_sub->size();
    memcpy(sub_rp, s_sub->rp(), sizeof(uint32_t) * (snv + 1));
    if (s_sub->sizeEdges()) memcpy(sub_ci, s_sub->ci(), sizeof(uint32_t) * s_sub->sizeEdges());
    return snv;
  }
  RefGraph g;
  g.load(nv, ne, rp, ci);
  size_t count = 0;
  for (uint32_t v = 0; v < nv; v++) count += train_masks[v] == 1;
  LearningGraph* tg = g.generate_masked_graph(train_masks);
  Sampler sampler(&g, tg, train_masks, count);
  VertexSet st;
  sampler.select_vertices(n, st, seed);
  std::vector<mask_t> m(nv);
  delete s_sub;
  s_sub = new RefGraph();
  sampler.generateSubgraph(st, m.data(), s_sub);
  uint32_t k = 0;
  for (auto v : st) kept[k++] = v;
  *sub_ne = (uint32_t)s_sub->sizeEdges();
  return (uint32_t)st.size();
}
}
