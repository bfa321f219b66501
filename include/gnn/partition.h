// include/gnn/partition.h -- vertex-range partition of a LearningGraph for one-process-per-GPU training.
// No counterpart in the reference's GNN code; the scheme is the reference partitioner's
// PartitionedGraph::edgecut_induced_partition1D (src/partitioner/graph_partition.cc:128-178): rank p owns the vertex
// range [p*ceil(N/P), (p+1)*ceil(N/P)), its subgraph's vertex set is owned + halo (the columns its rows touch outside
// the range), with a local -> global id map.  Here the rows' edges are additionally split by column owner (an
// owned-column CSR and a halo-column CSR), so the owned half of an aggregation runs while the halo rows travel
// (include/gnn/lgraph.h set_halo_plan, host/aggregators.cpp).  The same split as graphaibench_amd/dist.py, which
// tests/test_dist_cpu.py pins against the reference partitioner itself (oracle/_ref/libref_partition.so).
#pragma once
#include <stdint.h>
#include <vector>
#include "gaib.h"
#include "lgraph.h"

struct VertexRangePartition {
  int rank, world;
  int64_t n_global, lo, hi;                    // this rank owns global rows [lo, hi)
  std::vector<int64_t> rowptr_own, rowptr_halo;  // [n_own + 1]
  std::vector<index_t> colidx_own;             // local ids in [0, n_own)
  std::vector<index_t> colidx_halo;            // ids in [0, n_halo): index into halo_gids / the halo table
  std::vector<int64_t> degree;                 // [n_own] full (global) degree of every owned row
  std::vector<int64_t> halo_gids;              // [n_halo] global ids, ascending (hence grouped by owner rank)
  std::vector<int64_t> halo_degree;            // [n_halo] global degree of every halo vertex
  std::vector<int64_t> recv_counts;            // [world] halo rows owned by rank q
  std::vector<int64_t> send_counts;            // [world] rows rank q needs from this rank ...
  std::vector<int64_t> send_idx;               // ... and their local row ids, grouped by destination, ascending
  // GAT on a partition (build_gat_structures): the rows over ONE column space [owned | halo] (local id of a halo
  // vertex = n_own + its position in halo_gids), edges in the global CSR order of the row, and the transposed structure
  // (rows = the n_own + n_halo local column ids, columns = owned rows) with the edge permutation CSC position -> edge id
  std::vector<int64_t> rowptr_full;   // [n_own + 1]
  std::vector<index_t> colidx_full;   // ids in [0, n_own + n_halo)
  std::vector<int64_t> rowptr_t;      // [n_own + n_halo + 1]
  std::vector<index_t> colidx_t;      // owned row ids
  std::vector<index_t> tperm;         // tperm[k] = edge id (in rowptr_full / colidx_full order) of transposed entry k
  // build_gat_structures: every edge (i -> c) of this rank's rows has its reverse (c -> i) in the GLOBAL graph.  The
  // one-sweep GAT kernels on a partition read the in-edges of a row off its out-edges and are only correct on a
  // structurally symmetric graph; make_partitioned_graph sums the flag over the ranks (all of them must take the same path)
  bool rows_symmetric = true;
  int64_t n_own() const { return hi - lo; }
  int64_t n_halo() const { return (int64_t)halo_gids.size(); }
};

// bounds[p] = min(p * ceil(n / world), n), p = 0..world  (graph_partition.cc:131-133,151-153)
std::vector<int64_t> vertex_range_bounds(int64_t n, int world);

// Every rank holds the GLOBAL host CSR (the dataset files are global) and derives its share without communication:
// its own halo set from its rows, and what each peer will ask of it from the peer's rows.
VertexRangePartition build_vertex_range_partition(int64_t n_global, const index_t* rowptr, const index_t* colidx,
                                                  int rank, int world);

// fills rowptr_full / colidx_full / rowptr_t / colidx_t / tperm (needs the global CSR again)
void build_gat_structures(VertexRangePartition& part, const index_t* rowptr, const index_t* colidx);

// The rank's LearningGraph: owned-column CSR + halo-column CSR in HBM with the GLOBAL normalisers (a halo vertex's
// local degree is truncated), and the halo plan on `comm` that every aggregation of the layers will run.
// comm may be NULL for world == 1.  The returned graph borrows comm.
LearningGraph* make_partitioned_graph(const VertexRangePartition& part, gaib_comm* comm);
// If build_gat_structures was run on `part`, the graph also carries the combined and the transposed graph GAT needs
// (LearningGraph::set_gat_partition).
