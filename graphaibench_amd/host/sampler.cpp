// sampler.cpp -- placeholder for the GraphSAINT-style sampler (not built yet).
#include "sampler.h"

Sampler::Sampler(Graph* g, Graph* tg, mask_t*, size_t count) : count_(count), full_graph(g), masked_graph(tg) {}
static void not_yet() {
  fprintf(stderr, "subgraph sampling (subg_size > 0) is not implemented by the MI355X backend yet\n");
  exit(EXIT_FAILURE);
}
void Sampler::generateSubgraph(VertexSet&, mask_t*, Graph*) { not_yet(); }
size_t Sampler::select_vertices(index_t, VertexSet&, unsigned) { not_yet(); return 0; }
