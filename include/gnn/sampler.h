// include/gnn/sampler.h -- GraphSAINT-style frontier sampler (host side).
// Interface of the reference class (include/gnn/sampler.h:9-67; algorithm src/gnn/sampler.cpp:146-294,
// after GraphSAINT's ipdps19 sample.cpp): keep a frontier of m training vertices; n - m times pick
// a frontier slot with probability proportional to its (clipped) degree in the masked training
// graph, replace it by a uniformly chosen neighbour and add that neighbour to the vertex set; the
// subgraph is the one the FULL graph induces on the set, re-indexed in ascending vertex order.
// Our implementation keeps the slot weights in a Fenwick tree (O(log m) per draw) instead of the
// reference's "dashboard" arrays, and draws from std::mt19937: the distribution is the same, the
// random stream (rand_r in the reference) is not, so sampled sets are not comparable draw by draw.
#pragma once
#include <set>
#include "lgraph.h"
#define ETA 1.5           // (reference constants, kept for source compatibility)
#define SAMPLE_CLIP 3000  // degree clip in sampling

typedef std::set<index_t> VertexSet;
typedef std::vector<index_t> VertexList;

class Sampler {
 public:
  Sampler(Graph* g, Graph* tg, mask_t* masks, size_t count);
  ~Sampler() {}
  // build the subgraph induced by `vertex_set` on the full graph; masks[v] = 1 for kept vertices
  void generateSubgraph(VertexSet& vertex_set, mask_t* masks, Graph* sg);
  // sample up to n vertices into vertex_set (may return fewer: repeated picks); returns its size
  size_t select_vertices(index_t n, VertexSet& vertex_set, unsigned seed);
  void set_frontier_size(index_t f) { m = f; }  // extension (tests); default DEFAULT_SIZE_FRONTIER

 protected:
  index_t m;  // frontier size
  size_t count_;
  int avg_deg;
  int subg_deg;
  Graph* full_graph;
  Graph* masked_graph;
  std::vector<index_t> trainingNodes;
};
