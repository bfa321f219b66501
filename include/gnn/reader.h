// include/gnn/reader.h -- dataset reader for the GNN path.
// Binary format as the reference parses it (src/gnn/reader.cpp:414-457, Appendix B of SURVEY.md):
//   $DATASET_PATH/<name>/graph.meta.txt   nv ne vid_size eid_size vlabel_size elabel_size max_degree
//                                         feat_len num_vertex_classes num_edge_classes
//                                         train_begin train_end train_count  val_* test_*
//   graph.vertex.bin int64[nv+1] | graph.edge.bin uint32[ne] | graph.vlabel.bin uint8[nv] |
//   graph.feats.bin fp32[nv*feat_len]
// The legacy .csgr readers are unreachable in the reference (dataset_type = 1, net.cpp:80); they are
// declared for API parity and report "not supported".
#pragma once
#include "lgraph.h"

class Reader {
 private:
  std::string dataset_str;
  std::string inputfile_path;
  index_t feat_len;
  int num_vertex_classes;
  int num_edge_classes;
  index_t num_vertices_;
  index_t num_edges_;
  int train_begin, train_end, train_count;
  int val_begin, val_end, val_count;
  int test_begin, test_end, test_count;

 public:
  Reader() : dataset_str("") {}
  Reader(std::string dataset) : dataset_str(dataset) {}
  void init(std::string dataset) { dataset_str = dataset; }

  size_t csgr_read_labels(std::vector<label_t>& labels, bool is_single_class = true);
  size_t csgr_read_features(std::vector<float>& feats, std::string filetype = "bin");
  size_t csgr_read_masks(std::string mask_type, size_t n, size_t& begin, size_t& end, mask_t* masks);
  void csgr_read_graph(LearningGraph* g);

  size_t bin_read_features(std::vector<float>& feats);
  size_t bin_read_masks(std::string mask_type, size_t n, size_t& begin, size_t& end, mask_t* masks);
  void bin_read_graph(LearningGraph* g);
  int bin_read_vlabels(std::vector<label_t>& labels, bool is_single_class = true);
};
