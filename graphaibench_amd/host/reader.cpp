// reader.cpp -- binary dataset reader (format: include/gnn/reader.h; reference parser
// src/gnn/reader.cpp:248-268,347-457).
#include <fstream>
#include "configs.h"
#include "reader.h"

namespace {
template <typename T>
void read_exact(const std::string& fname, T* dst, size_t count) {
  std::ifstream f(fname.c_str(), std::ios::binary);
  if (!f.good()) {
    std::cerr << "Failed to open file: " << fname << "\n";
    exit(1);
  }
  f.read(reinterpret_cast<char*>(dst), sizeof(T) * count);
  if ((size_t)f.gcount() != sizeof(T) * count) {
    std::cerr << "Short read on " << fname << ": wanted " << sizeof(T) * count << " bytes, got " << f.gcount() << "\n";
    exit(1);
  }
}
void csgr_unsupported() {
  std::cerr << "the legacy .csgr dataset format is not supported (the reference never selects it: net.cpp:80)\n";
  exit(1);
}
}  // namespace

void Reader::bin_read_graph(LearningGraph* g) {
  inputfile_path = dataset_root() + dataset_str + "/";
  std::cout << "input file path: " << inputfile_path << ", graph name: " << dataset_str << "\n";
  std::ifstream meta((inputfile_path + "graph.meta.txt").c_str());
  if (!meta.good()) {
    std::cerr << "Failed to open file: " << inputfile_path << "graph.meta.txt\n";
    exit(1);
  }
  int vid_size = 0, eid_size = 0, vlabel_size = 0, elabel_size = 0, max_degree = 0;
  train_begin = train_end = train_count = val_begin = val_end = val_count = 0;
  test_begin = test_end = test_count = 0;
  feat_len = 0;
  num_vertex_classes = num_edge_classes = 0;
  meta >> num_vertices_ >> num_edges_ >> vid_size >> eid_size >> vlabel_size >> elabel_size >> max_degree >>
      feat_len >> num_vertex_classes >> num_edge_classes;
  meta >> train_begin >> train_end >> train_count >> val_begin >> val_end >> val_count >> test_begin >>
      test_end >> test_count;
  // the element sizes the reference asserts (reader.cpp:433-436) decide how the files are parsed: refuse others
  if (vid_size != 4 || eid_size != 8 || vlabel_size != 1) {
    std::cerr << "graph.meta.txt: unsupported sizes (vid " << vid_size << ", eid " << eid_size << ", vlabel "
              << vlabel_size << "; need 4 / 8 / 1)\n";
    exit(1);
  }
  // max_degree is only an assert in the reference (reader.cpp:437, compiled out with NDEBUG) and is recomputed by
  // degree_counting() anyway: a dataset the reference binaries accept is not rejected here
  if (!(max_degree > 0 && (index_t)max_degree < num_vertices_))
    std::cerr << "graph.meta.txt: max_degree " << max_degree << " is outside (0, num_vertices): ignored, recomputed\n";
  g->allocateFrom(num_vertices_, num_edges_);
  std::vector<int64_t> rows((size_t)num_vertices_ + 1);
  read_exact<int64_t>(inputfile_path + "graph.vertex.bin", rows.data(), rows.size());
  read_exact<index_t>(inputfile_path + "graph.edge.bin", g->edge_host_ptr(), num_edges_);
  index_t* row = g->row_host_ptr();
  for (size_t i = 0; i <= num_vertices_; i++) row[i] = (index_t)rows[i];  // int64 on disk, uint32 in memory (Q13)
}

size_t Reader::bin_read_features(std::vector<float>& feats) {
  std::cout << "Reading features ... N x D: " << num_vertices_ << " x " << feat_len << "\n";
  feats.resize((size_t)num_vertices_ * feat_len);
  if (feat_len) read_exact<float>(inputfile_path + "graph.feats.bin", feats.data(), feats.size());
  return feat_len;
}

int Reader::bin_read_vlabels(std::vector<label_t>& labels, bool is_single_class) {
  assert(num_vertex_classes > 0 && num_vertex_classes < 255);
  std::vector<label_t> raw(num_vertices_);
  read_exact<label_t>(inputfile_path + "graph.vlabel.bin", raw.data(), raw.size());
  if (is_single_class) {
    std::cout << "Using single-class (one-hot) labels\n";
    labels = raw;
  } else {
    std::cout << "Using multi-class (multi-hot) labels\n";
    labels.assign((size_t)num_vertices_ * num_vertex_classes, 0);
    for (size_t v = 0; v < num_vertices_; v++)
      if (raw[v] < num_vertex_classes) labels[v * num_vertex_classes + raw[v]] = 1;
  }
  return num_vertex_classes;
}

// masks are the contiguous ranges of the meta file; the *.masks.bin files are never read (Q5)
size_t Reader::bin_read_masks(std::string mask_type, size_t n, size_t& begin, size_t& end, mask_t* masks) {
  size_t count;
  if (mask_type == "train") { begin = train_begin; end = train_end; count = train_count; }
  else if (mask_type == "val") { begin = val_begin; end = val_end; count = val_count; }
  else { begin = test_begin; end = test_end; count = test_count; }
  if (masks) {
    std::fill(masks, masks + n, 0);
    for (size_t i = begin; i < end && i < n; i++) masks[i] = 1;
  }
  std::cout << mask_type << "_mask range: [" << begin << ", " << end << ") Number of valid samples: " << count
            << " (" << (float)count / (float)n * 100.f << "%)\n";
  return count;
}

size_t Reader::csgr_read_labels(std::vector<label_t>&, bool) { csgr_unsupported(); return 0; }
size_t Reader::csgr_read_features(std::vector<float>&, std::string) { csgr_unsupported(); return 0; }
size_t Reader::csgr_read_masks(std::string, size_t, size_t&, size_t&, mask_t*) { csgr_unsupported(); return 0; }
void Reader::csgr_read_graph(LearningGraph*) { csgr_unsupported(); }
