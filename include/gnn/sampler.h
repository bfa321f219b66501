// include/gnn/sampler.h -- GraphSAINT-style subgraph sampler interface (reference:
// include/gnn/sampler.h, src/gnn/sampler.cpp:146-294).  Sampling is off by default (subg_size = 0,
// net.cpp:38) and is the last of SURVEY 8f's "next" rows: the class is declared so drivers link;
// selecting subg_size > 0 reports that it is not implemented yet and exits.
#pragma once
#include <set>
#include "lgraph.h"

typedef std::set<index_t> VertexSet;
typedef std::vector<index_t> VertexList;

class Sampler {
 public:
  Sampler(Graph* g, Graph* tg, mask_t* masks, size_t count);
  ~Sampler() {}
  void generateSubgraph(VertexSet& vertex_set, mask_t* masks, Graph* sg);
  size_t select_vertices(index_t n, VertexSet& vertex_set, unsigned seed);

 protected:
  size_t count_;
  Graph* full_graph;
  Graph* masked_graph;
};
