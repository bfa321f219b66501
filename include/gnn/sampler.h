// include/gnn/sampler.h -- GraphSAINT frontier sampler (host side).
// Interface of the reference class (include/gnn/sampler.h:9-67; algorithm src/gnn/sampler.cpp:146-294,
// after GraphSAINT's ipdps19 sample.cpp): keep a frontier of m training vertices; n - m times pick
// a frontier vertex with probability proportional to its (clipped) degree in the masked training
// graph, replace it by a uniformly chosen neighbour and add that neighbour to the vertex set; the
// subgraph is the one the FULL graph induces on the set, re-indexed in ascending vertex order.
// select_vertices consumes the same rand_r stream over the same "dashboard" layout as the reference, so
// the same (graph, training set, n, seed) gives the SAME vertex set and subgraph, bit for bit
// (tests/test_sampler_cpu.py against the reference's own sampler.cpp in oracle/_ref and against
// tests/golden/sampler_*.npz).
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
