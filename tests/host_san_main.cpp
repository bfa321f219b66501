// Sanitizer driver for the HOST-ONLY C++ of the product (tests/test_sanitizers.py builds it with
// -fsanitize=address,undefined): binary dataset reader, LearningGraph host methods (add_selfloop, degree_counting,
// generate_masked_graph), GraphSAINT sampler, vertex-range partition builder + GAT structures.  No device call is made
// (the GPU libraries are linked only to resolve symbols), so it runs in the CPU-only container.
#include <cstdio>
#include <fstream>
#include <random>
#include <set>
#include <string>
#include <vector>
#include "lgraph.h"
#include "partition.h"
#include "reader.h"
#include "sampler.h"

static void write_dataset(const std::string& dir, int nv, std::vector<int64_t>& rp, std::vector<uint32_t>& ci) {
  std::mt19937 rng(7);
  std::vector<std::set<uint32_t>> adj(nv);
  for (int i = 0; i < nv * 4; i++) {
    uint32_t u = rng() % nv, v = rng() % nv;
    if (u == v) continue;
    adj[u].insert(v);
    adj[v].insert(u);
  }
  for (int v = 1; v < nv; v++) { adj[0].insert(v); adj[v].insert(0); }  // a hub row
  rp.assign(nv + 1, 0);
  ci.clear();
  for (int v = 0; v < nv; v++) {
    for (uint32_t c : adj[v]) ci.push_back(c);
    rp[v + 1] = (int64_t)ci.size();
  }
  int maxdeg = 0;
  for (int v = 0; v < nv; v++) maxdeg = std::max<int>(maxdeg, (int)adj[v].size());
  std::ofstream(dir + "graph.meta.txt") << nv << "\n" << ci.size() << "\n4 8 1 2\n" << maxdeg << "\n5\n3\n0\n0 "
                                       << nv / 2 << " " << nv / 2 << "\n" << nv / 2 << " " << nv * 3 / 4 << " " << nv / 4
                                       << "\n" << nv * 3 / 4 << " " << nv << " " << nv - nv * 3 / 4 << "\n";
  std::ofstream(dir + "graph.vertex.bin", std::ios::binary).write((const char*)rp.data(), sizeof(int64_t) * rp.size());
  std::ofstream(dir + "graph.edge.bin", std::ios::binary).write((const char*)ci.data(), sizeof(uint32_t) * ci.size());
  std::vector<uint8_t> lab(nv);
  for (int v = 0; v < nv; v++) lab[v] = v % 3;
  std::ofstream(dir + "graph.vlabel.bin", std::ios::binary).write((const char*)lab.data(), lab.size());
  std::vector<float> f((size_t)nv * 5, 0.5f);
  std::ofstream(dir + "graph.feats.bin", std::ios::binary).write((const char*)f.data(), sizeof(float) * f.size());
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string root = argv[1];  // DATASET_PATH must point here (with a trailing slash); dataset name "cora"
  const int nv = 600;
  std::vector<int64_t> rp;
  std::vector<uint32_t> ci;
  write_dataset(root + "cora/", nv, rp, ci);
  Graph* g = new Graph(false);
  Reader reader("cora");
  reader.bin_read_graph(g);
  std::vector<float> feats;
  std::vector<label_t> labels, hot;
  reader.bin_read_features(feats);
  reader.bin_read_vlabels(labels, true);
  reader.bin_read_vlabels(hot, false);
  std::vector<mask_t> mtrain(nv);
  size_t b = 0, e = 0;
  const size_t cnt = reader.bin_read_masks("train", nv, b, e, mtrain.data());
  g->degree_counting();
  // sampler on the masked graph, then the induced subgraph
  Graph* tg = g->generate_masked_graph(mtrain.data());
  Sampler sampler(g, tg, mtrain.data(), cnt);
  VertexSet st;
  sampler.select_vertices(120, st, 1u);
  std::vector<mask_t> sm(nv);
  Graph sg(false);
  sampler.generateSubgraph(st, sm.data(), &sg);
  // self loops + vertex-range partitions of every rank, with the GAT structures
  g->add_selfloop();
  size_t total = 0;
  for (int world : {1, 2, 5}) {
    for (int r = 0; r < world; r++) {
      VertexRangePartition P = build_vertex_range_partition(nv, g->row_start_host_ptr(), g->edge_dst_host_ptr(), r, world);
      build_gat_structures(P, g->row_start_host_ptr(), g->edge_dst_host_ptr());
      total += P.colidx_full.size() + P.tperm.size() + P.send_idx.size();
    }
  }
  printf("host_san_main done: %zu vertices in the sample, %zu partition entries\n", (size_t)sg.size(), total);
  tg->dealloc();
  delete tg;
  sg.dealloc();
  g->dealloc();
  delete g;
  return 0;
}
