// spmm_part.hip -- the aggregation kernels of spmm_kernels.h instantiated for the ROW CLASSES of a vertex-range partition
// (PART = true): a class graph holds a compact subset of a rank's rows (gaib_graph_split_classes, graph.hip) -- the interior
// rows, or the boundary rows' owned-column / halo-column / all edges -- and
//   * row r of the class stands for row row_map[r] of the caller's matrices (out, the continued partial sums, rows2, y);
//   * column ids >= n_first index a SECOND table: the halo table behind the rank's own feature rows, two allocations
//     that one pass over [owned | halo] reads side by side (gaib_spmm_2t / gaib_spmm_gemm_2t).
// No reference counterpart (the reference has no multi-GPU GNN); the partition structure is the reference partitioner's
// owned (master) rows + halo vertices, src/partitioner/graph_partition.cc:70-80,128-178, include/graph_partition.h:21-22,36-37.
// Sums and their order are those of the whole-graph kernels (spmm.hip): a class graph keeps the edge order of its rows.
// A translation unit of its own so the two sets of instantiations compile side by side.
#include "spmm_kernels.h"

namespace {

template <int VEC, int CT, int WMODE>
int part_w64(gaib_ctx* ctx, const gaib_graph* g, const SpmmArgs& a) {
  constexpr int U = (VEC * CT >= 8) ? 4 : (VEC * CT >= 4 ? 8 : 16);  // as launch_w64
  const bool buf = a.in_bytes != 0 && ctx->spmm_addr_mode != 2 && (!a.in2 || a.in2_bytes != 0);
  return buf ? launch_w64_u<VEC, CT, WMODE, U, 1, true>(ctx, g, a) : launch_w64_u<VEC, CT, WMODE, U, 0, true>(ctx, g, a);
}

template <int VEC, int WMODE>
int part_ct(gaib_ctx* ctx, const gaib_graph* g, const SpmmArgs& a, int lanes) {
  if (lanes <= 64) return part_w64<VEC, 1, WMODE>(ctx, g, a);
  if (lanes <= 128) return part_w64<VEC, 2, WMODE>(ctx, g, a);
  return part_w64<VEC, 4, WMODE>(ctx, g, a);
}

// lane vector by width, the rule of dispatch_vec (spmm_kernels.h): one row per wave, the narrowest vector that covers
// the row in at most two passes
template <int WMODE>
int part_vec(gaib_ctx* ctx, const gaib_graph* g, const SpmmArgs& a0, int len) {
  const uintptr_t al = (uintptr_t)a0.in | (uintptr_t)a0.out | (uintptr_t)a0.in2;
  int vmax = 1;
  if (len % 4 == 0 && (al & 15) == 0) vmax = 4;
  else if (len % 2 == 0 && (al & 7) == 0) vmax = 2;
  int vec = vmax;
  if (len <= 64) vec = 1;
  else if (len <= 256 && vmax >= 2) vec = 2;
  else if (len <= 128) vec = 1;
  const int slab = 256 * vec;
  for (int c0 = 0; c0 < len; c0 += slab) {
    SpmmArgs a = a0;
    a.in = a0.in + c0;
    if (a0.in2) a.in2 = a0.in2 + c0;
    a.out = a0.out + c0;
    a.ncols = (len - c0 < slab) ? (len - c0) : slab;
    if (a.in_bytes) a.in_bytes -= (uint32_t)(c0 * 4);
    if (a.in2_bytes) a.in2_bytes -= (uint32_t)(c0 * 4);
    const int lanes = (a.ncols + vec - 1) / vec;
    int rc;
    if (vec == 4) rc = part_ct<4, WMODE>(ctx, g, a, lanes);
    else if (vec == 2) rc = part_ct<2, WMODE>(ctx, g, a, lanes);
    else rc = part_ct<1, WMODE>(ctx, g, a, lanes);
    if (rc != GAIB_OK) return rc;
  }
  return GAIB_OK;
}

}  // namespace

int gaib_spmm_part_plain(gaib_ctx* ctx, const gaib_graph* g, const void* spmm_args, int wmode, int len) {
  const SpmmArgs& a = *static_cast<const SpmmArgs*>(spmm_args);
  return wmode == 0 ? part_vec<0>(ctx, g, a, len) : part_vec<1>(ctx, g, a, len);
}

int gaib_spmm_part_fused(gaib_ctx* ctx, const gaib_graph* g, const void* spmm_args, const void* fuse_args, float* heavy_scratch,
                         int vec, int wmode) {
  const SpmmArgs& a = *static_cast<const SpmmArgs*>(spmm_args);
  const FuseArgs& f = *static_cast<const FuseArgs*>(fuse_args);
  if (vec == 1) return wmode == 0 ? launch_fused<1, 0, true>(ctx, g, a, f, heavy_scratch) : launch_fused<1, 1, true>(ctx, g, a, f, heavy_scratch);
  return wmode == 0 ? launch_fused<2, 0, true>(ctx, g, a, f, heavy_scratch) : launch_fused<2, 1, true>(ctx, g, a, f, heavy_scratch);
}
