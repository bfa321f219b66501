// include/gnn/lgraph.h -- LearningGraph: CSR container of the GNN path, host arrays plus the
// HBM-resident copy the aggregation kernels read.
// API mirrors the reference class (include/gnn/lgraph.h:20-277): the reader writes through
// row_host_ptr()/edge_host_ptr(), the model calls add_selfloop / alloc_on_device / copy_to_gpu /
// compute_vertex_data / compute_edge_data, layers only hold a Graph*.
// Not mirrored (dead or out of scope in the reference itself): CSR segmenting (lgraph.cpp:55-157
// is commented out there), print_test.
#pragma once
#include <algorithm>
#include <vector>
#include "global.h"
#include "gaib.h"

class LearningGraph {
 protected:
  bool is_device;
  index_t num_vertices_;
  index_t num_edges_;
  index_t max_degree;
  index_t* rowptr_;  // host, [nv+1]
  index_t* colidx_;  // host, [ne]
  vdata_t* vertex_data_;  // host copies, filled on demand
  edata_t* edge_data_;
  gaib_graph* dev_;  // CSR + normalisers + kernel schedules in HBM
  // vertex-range partitions (no reference counterpart): the rows' edges are split into
  // an owned-column graph (dev_: column ids index the layer's own feature rows) and a halo-column
  // graph (halo_dev_: column ids index the halo table).  halo_begin_ packs the rows other ranks need
  // and starts the all-to-all; halo_end_ waits and returns the halo table [n_halo x len].
  gaib_graph* halo_dev_;
  void (*halo_begin_)(void* user, int len, const float* d_in);
  const float* (*halo_end_)(void* user, int len);
  void* halo_user_;
  gaib_halo* halo_plan_;  // the exchange behind the C ABI (gaib_halo_exchange_begin/end); set_halo_plan
  // GAT on a partition (include/gnn/partition.h build_gat_structures): the rows over one [owned | halo] column space,
  // the transposed structure and the edge permutation between them
  gaib_graph* gat_full_;
  gaib_graph* gat_t_;
  index_t* gat_tperm_;
  index_t gat_n_halo_;
  bool owns_partition_;  // make_partitioned_graph built halo_dev_ / halo_plan_ / gat_*: dealloc() releases them too
  bool gat_symmetric_ = true;
  // Row classes of the partition (gaib_graph_split_classes; the reference partitioner's owned rows / halo vertices,
  // src/partitioner/graph_partition.cc:70-80): INTERIOR rows -- no halo-column edge -- are aggregated in one pass (with
  // the dense product riding on it) while the halo rows travel; BOUNDARY rows either keep the column split (owned-column
  // edges meanwhile, halo-column edges after arrival: everything overlaps, one more pass over their partial sums) or are
  // aggregated in ONE pass over [owned | halo] after arrival (no partial sums; only the interior work hides the
  // exchange).  Decided once per graph at its first aggregation (partition_mode) and built then.
  gaib_graph *cls_int_, *cls_bown_, *cls_bhalo_, *cls_bfull_;
  int part_mode_;         // PART_*, -1 = not decided yet
  int part_mode_wanted_;  // set_partition_mode / GAIB_PART_MODE; -1 = by the rule
  int64_t n_boundary_, boundary_edges_, link_rows_;
  // Round 6: the exchange in time slices (gaib_halo_set_pieces).  The plan puts K slices on the wire (the same K on every
  // rank); each rank CONSUMES them in K' pieces of K / K' consecutive slices, K' | K, its own choice: where the mode keeps a
  // halo-column half (PART_SPLIT: halo_dev_, PART_CLASSES: cls_bhalo_) that half is cut into K' piece graphs
  // (gaib_graph_split_pieces over the plan's column ranges) and aggregated piece by piece in accumulate mode as the slices land
  // (host/aggregators.cpp halo_half), so the wire hides under the halo-column work too, not only under the owned-column pass.
  // Every further piece costs one more read + write of the rows' partial sums and shorter row segments per pass (measured:
  // + 6 % of a step at K' = 2, + 25 % at 4, profiles/r06/shard/), so K' comes from a rule that prices both (halo_pieces):
  // a rank whose owned-column work already covers the exchange keeps K' = 1 and pays nothing.
  gaib_graph* pieces_[GAIB_GRAPH_MAX_PIECES];
  int pieces_built_;   // K' the piece graphs were cut for (0 = none) ...
  int pieces_slices_;  // ... out of this many slices on the wire
  int pieces_want_;    // set_halo_consumption / GAIB_HALO_CONSUME; -1 = by the rule
  int pieces_rule_;    // what the rule chose (cached with the K and row length it chose for)
  int pieces_rule_k_, pieces_rule_len_;
  // callback transports (set_halo) bring the slice structure themselves: set_halo_pieces
  int cb_pieces_;
  std::vector<int64_t> cb_range_begin_, cb_range_end_;
  std::vector<int> cb_range_piece_;
  const float* (*halo_wait_piece_)(void* user, int piece);
  void drop_pieces();
  int halo_slices() const { return halo_plan_ ? gaib_halo_pieces(halo_plan_) : (halo_wait_piece_ ? cb_pieces_ : 1); }
  int consumption_rule(int K, int len);

 public:
  typedef size_t iterator;
  LearningGraph(bool use_gpu)
      : is_device(use_gpu), num_vertices_(0), num_edges_(0), max_degree(0), rowptr_(NULL),
        colidx_(NULL), vertex_data_(NULL), edge_data_(NULL), dev_(NULL), halo_dev_(NULL),
        halo_begin_(NULL), halo_end_(NULL), halo_user_(NULL), halo_plan_(NULL), gat_full_(NULL), gat_t_(NULL),
        gat_tperm_(NULL), gat_n_halo_(0), owns_partition_(false), cls_int_(NULL), cls_bown_(NULL), cls_bhalo_(NULL),
        cls_bfull_(NULL), part_mode_(-1), part_mode_wanted_(-1), n_boundary_(0), boundary_edges_(0), link_rows_(-1),
        pieces_built_(0), pieces_slices_(0), pieces_want_(-1), pieces_rule_(0), pieces_rule_k_(0), pieces_rule_len_(0), cb_pieces_(1),
        halo_wait_piece_(NULL) {
    for (gaib_graph*& p : pieces_) p = NULL;
  }
  LearningGraph() : LearningGraph(true) {}
  // wrap a graph that already lives in HBM (synthetic / partitioned graphs built on device)
  static LearningGraph* adopt_device(gaib_graph* g);

  size_t size() { return (size_t)num_vertices_; }
  size_t sizeEdges() { return (size_t)num_edges_; }
  bool on_device() { return is_device; }
  index_t get_max_degree() { return max_degree; }
  index_t get_degree(index_t v) { return rowptr_[v + 1] - rowptr_[v]; }
  iterator begin() const { return iterator(0); }
  iterator end() const { return iterator(num_vertices_); }

  // construction on the host (what Reader::bin_read_graph drives, reader.cpp:414-457)
  void allocateFrom(index_t nv, index_t ne);
  void fixEndEdge(index_t vid, index_t row_end) { rowptr_[vid + 1] = row_end; }
  void constructEdge(index_t eid, index_t dst) {
    assert(dst < num_vertices_ && eid < num_edges_);
    colidx_[eid] = dst;
  }
  index_t* row_start_host_ptr() { return rowptr_; }
  index_t*& row_host_ptr() { return rowptr_; }
  index_t* edge_dst_host_ptr() { return colidx_; }
  index_t*& edge_host_ptr() { return colidx_; }
  index_t getEdgeDstHost(index_t eid) { return colidx_[eid]; }
  index_t edge_begin_host(index_t vid) { return rowptr_[vid]; }
  index_t edge_end_host(index_t vid) { return rowptr_[vid + 1]; }
  void degree_counting();
  void add_selfloop();  // host CSR, sorted insert (rows must be sorted, no self loops)
  LearningGraph* generate_masked_graph(mask_t* masks);

  // device side
  void alloc_on_device();
  void alloc_on_device(index_t n);
  void copy_to_gpu();
  void copy_to_cpu();
  void compute_vertex_data();
  void compute_edge_data();
  void dealloc();
  gaib_graph* device_graph() { return dev_; }
  void set_halo(gaib_graph* halo_graph, void (*begin)(void*, int, const float*),
                const float* (*end)(void*, int), void* user) {
    halo_dev_ = halo_graph;
    halo_begin_ = begin;
    halo_end_ = end;
    halo_user_ = user;
  }
  // the same with the exchange running behind the C ABI (gaib_comm / gaib_halo: RCCL or peer-to-peer pull);
  // the callback form above stays for launchers that bring their own transport (torch.distributed in dist.py)
  void set_halo_plan(gaib_graph* halo_graph, gaib_halo* plan) {
    halo_dev_ = halo_graph;
    halo_plan_ = plan;
  }
  gaib_halo* halo_plan() { return halo_plan_; }
  // the halo graph, the exchange plan and the GAT structures handed over so far belong to this object from here on:
  // dealloc() destroys them (the plan before its communicator -- the caller's order, as with gaib_halo_destroy)
  void own_partition_objects() { owns_partition_ = true; }
  // ne_full = all edges of the owned rows (owned- and halo-column): what sizeEdges() reports from here on, so that
  // per-edge buffers of the aggregators are sized for the combined graph
  void set_gat_partition(gaib_graph* full, gaib_graph* transposed, index_t* d_tperm, size_t n_halo, size_t ne_full) {
    gat_full_ = full;
    gat_t_ = transposed;
    gat_tperm_ = d_tperm;
    gat_n_halo_ = (index_t)n_halo;
    num_edges_ = (index_t)ne_full;
  }
  // false: an edge of the GLOBAL graph lacks its reverse (make_partitioned_graph checks, summed over the ranks): the
  // one-sweep GAT kernels on the partition are then wrong (they read a row's in-edges off its out-edges); staged path
  void set_gat_symmetric(bool yes) { gat_symmetric_ = yes; }
  bool gat_symmetric() const { return gat_symmetric_; }
  gaib_graph* gat_full_graph() { return gat_full_; }
  gaib_graph* gat_transposed_graph() { return gat_t_; }
  const index_t* gat_tperm() { return gat_tperm_; }
  size_t gat_n_halo() { return gat_n_halo_; }
  // ---- row classes (see above) ----
  enum { PART_SPLIT = 0,    // round 3: owned-column pass over ALL rows, halo-column pass over all rows
         PART_CLASSES = 1,  // interior rows in one pass; boundary rows by the column split
         PART_ONEPASS = 2,  // interior rows in one pass; boundary rows in one pass over [owned | halo] after arrival
         PART_ONEPASS_ALL = 3  // (a wish only; partition_mode reports PART_ONEPASS) every row counts as a boundary row:
                               // one pass over all rows of one [owned | halo] graph after arrival
  };
  void set_partition_mode(int mode) {  // -1: by the rule
    part_mode_wanted_ = mode;
    part_mode_ = -1;
    drop_pieces();  // (cut from the former mode's halo-column half)
  }
  // callback transports (set_halo): the most rows one peer pair moves per exchange (the plan form knows: gaib_halo_link_rows)
  void set_halo_link_rows(int64_t rows) {
    link_rows_ = rows;
    pieces_rule_k_ = 0;  // (the consumption rule prices the exchange by it: decide again)
  }
  // the mode of this graph's aggregations of `len` columns; the first call decides and builds the class graphs
  int partition_mode(int len);
  gaib_graph* class_interior() { return cls_int_; }
  gaib_graph* class_boundary_own() { return cls_bown_; }
  gaib_graph* class_boundary_halo() { return cls_bhalo_; }
  gaib_graph* class_boundary_full() { return cls_bfull_; }
  int64_t n_boundary() const { return n_boundary_; }
  int64_t boundary_edges() const { return boundary_edges_; }
  bool has_halo() const { return halo_dev_ != NULL; }
  gaib_graph* halo_graph() { return halo_dev_; }
  void halo_begin(int len, const float* d_in);
  const float* halo_end(int len);
  // ---- the halo-column half piece by piece (see pieces_ above) ----
  // callback transports: the exchange lands in n_pieces slices; range j = [begin[j], end[j]) of the halo table belongs to slice
  // piece[j]; wait_piece(user, k) returns the table once slice k is there (stream-ordered).  Plans: gaib_halo_set_pieces.
  void set_halo_pieces(int n_pieces, int n_ranges, const int64_t* begin, const int64_t* end, const int* piece,
                       const float* (*wait_piece)(void*, int)) {
    drop_pieces();
    cb_pieces_ = n_pieces;
    cb_range_begin_.assign(begin, begin + n_ranges);
    cb_range_end_.assign(end, end + n_ranges);
    cb_range_piece_.assign(piece, piece + n_ranges);
    halo_wait_piece_ = wait_piece;
  }
  // pieces the halo-column half of an aggregation of `len` columns is consumed in RIGHT NOW: K' | K (the plan's / the callbacks'
  // slices), forced (set_halo_consumption, GAIB_HALO_CONSUME) or by the rule; 1 = the whole half after gaib_halo_exchange_end
  // (also wherever the mode has no such half).  The piece graphs are (re)built here if need be.
  int halo_pieces(int len);
  void set_halo_consumption(int pieces) { pieces_want_ = pieces; }  // -1: by the rule
  gaib_graph* halo_piece_graph(int j) { return pieces_[j]; }
  // the compute stream continues once piece j -- slices j K / K' ... (j + 1) K / K' - 1 -- has landed
  const float* halo_wait_piece(int j);
  // device pointers, as the reference's ENABLE_GPU accessors return them.  Row pointers are
  // int64 in HBM (the reference's uint32 offsets overflow past 2^32 edges*features).
  const int64_t* row_start_ptr() const { return gaib_graph_rowptr(dev_); }
  const index_t* edge_dst_ptr() const { return gaib_graph_colidx(dev_); }
  const edata_t* edge_data_ptr() const { return gaib_graph_edge_data(dev_); }
  const vdata_t* vertex_data_ptr() const { return gaib_graph_vertex_data(dev_); }
  // host-side values (copied back on first use; tests)
  vdata_t get_vertex_data(index_t vid);
  edata_t get_edge_data(index_t eid);
  void print_graph();
};

typedef LearningGraph Graph;
typedef LearningGraph GraphCPU;
typedef LearningGraph GraphGPU;
