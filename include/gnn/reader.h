// include/gnn/reader.h -- dataset reader for the GNN path.
// Binary format as the reference parses it (src/gnn/reader.cpp:414-457, Appendix B of SURVEY.md):
//   $DATASET_PATH/<name>/graph.meta.txt   nv ne vid_size eid_size vlabel_size elabel_size max_degree
//                                         feat_len num_vertex_classes num_edge_classes
//                                         train_begin train_end train_count  val_* test_*
//   graph.vertex.bin int64[nv+1] | graph.edge.bin uint32[ne] | graph.vlabel.bin uint8[nv] |
//   graph.feats.bin fp32[nv*feat_len]
// Method names and argument lists are the reference's (include/gnn/reader.h:23-40): the drivers call
// bin_read_graph / bin_read_features / bin_read_vlabels / bin_read_masks.  The legacy .csgr readers are unreachable in
// the reference (dataset_type = 1, net.cpp:80); they are declared for API parity and report "not supported".
#pragma once
#include "lgraph.h"

class Reader {
 public:
  explicit Reader(std::string dataset = "") : dataset_str(dataset) {}
  void init(std::string dataset) { dataset_str = dataset; }

  // ---- binary format (what net.cpp uses) ---------------------------------------------------------------------
  void bin_read_graph(LearningGraph* g);                     // meta file + CSR arrays; must run first
  size_t bin_read_features(std::vector<float>& feats);       // -> feat_len
  int bin_read_vlabels(std::vector<label_t>& labels, bool is_single_class = true);  // -> number of classes
  // masks are the contiguous meta ranges: fills masks[begin, end) with 1, returns the sample count
  size_t bin_read_masks(std::string mask_type, size_t n, size_t& begin, size_t& end, mask_t* masks);

  // ---- legacy Galois .csgr format: report and exit -------------------------------------------------------------
  void csgr_read_graph(LearningGraph* g);
  size_t csgr_read_labels(std::vector<label_t>& labels, bool is_single_class = true);
  size_t csgr_read_features(std::vector<float>& feats, std::string filetype = "bin");
  size_t csgr_read_masks(std::string mask_type, size_t n, size_t& begin, size_t& end, mask_t* masks);

 private:
  std::string dataset_str, inputfile_path;
  // graph.meta.txt
  index_t num_vertices_, num_edges_, feat_len;
  int num_vertex_classes, num_edge_classes;
  int train_begin, train_end, train_count, val_begin, val_end, val_count, test_begin, test_end, test_count;
};
