// spmm.hip -- CSR SpMM neighbour aggregation for gfx950 (the hot loop).
//
//   out[i,:] = sum_{e in row i} w_e * in[col_e,:]
//
// replaces GCN_Aggregator::update_all (src/gnn/gconv/gcn_aggregator.cpp:48-77), the two
// SAGE_Aggregator loops (sage_aggregator.cpp:7-54), the file-static update_all of the GAT
// aggregator (gat_aggregator.cpp:26-45) and, on the reference's CUDA side, update_all_gcn /
// update_all_sage / reduce_warp / reduce_cta (include/gnn/graph_operations.h:8-178).
//
// Design (MI355X):
//   * HBM-bound gather: per aggregated edge one 4*len-byte feature row + 4 B colidx
//     (+ 4 B weight).  No LDS staging of feature rows: a 64-lane wave already covers a
//     512-B row with one dwordx2 load per lane, and up to U such loads are kept in flight
//     per wave (<= 64 VGPRs -> 8 waves/SIMD), which is what covers the ~2 us loaded HBM
//     latency.  colidx / weights are read coalesced, 64 edges per wave instruction, and
//     broadcast with v_readlane (SGPR) so the row base address is scalar.
//   * one wave per row ("w64" kernels) when a row needs >= 32 lanes; for narrow rows
//     (len/VEC < 32) a wave is cut into 64/G groups, one row per group ("sub" kernels).
//   * rows are summed in CSR order with separate multiply and add (no FMA), i.e. exactly
//     the OpenMP loop's rounding: results are bit-identical for rows up to the heavy
//     threshold.
//   * power-law tail: rows with more than `heavy_thr` edges are skipped by the light kernel
//     and handled by a workgroup-per-row kernel (16 waves split the edge list, partial sums
//     meet in LDS and are added in wave order -> deterministic).
//   * blockIdx -> row-block mapping is XCD-aware: consecutive row blocks land on the same
//     XCD (blocks b and b+8 share one), so neighbouring rows share that XCD's 4 MB L2.
#include <algorithm>
#include "spmm_kernels.h"

namespace {

// ---- dense graphs: aggregation by ordered 64-edge chunks -------------------------------------------------------------
// Where rows have hundreds of edges and the feature table sits in the Infinity Cache but not in the 4 MB L2 (reddit:
// 490 edges per row over 60 MB), the row-per-wave kernels gather from all over the table at any moment.  The chunk
// list of the graph (SDDMM's, ordered by column block) turns that into a sweep: a wave sums the 64 edges of one chunk
// -- G lanes x 16 B per edge, 64/G edges per instruction -- into a partial row, a second kernel adds a row's partials in
// row order.  Fixed order, so deterministic, but not the CSR-order sum of the one-row kernels (the oracle's order).
// HT > 0 (WMODE 3, heads == HT in {4, 8, 16}): the chunk's [64][HT] weight block is read once, one edge per lane
// (HT/4 16-byte loads), parked in the wave's LDS slice with a row stride of HT + 1 floats and picked up per (edge, head)
// from there: 2 + 2 wave instructions per chunk instead of one 4-byte global load per lane and edge.
template <int G, int WMODE, int HT = 0>
__global__ __launch_bounds__(256) void spmm_chunk_kernel(int64_t n_chunks, const uint32_t* chunk_row,
                                                         const uint32_t* chunk_ebase, const uint32_t* chunk_start,
                                                         SpmmArgs a, float* partial) {
  constexpr int U = G < 8 ? G : 8;
  constexpr bool MH = WMODE >= 3;
  constexpr int WS = HT + 1;  // LDS row stride (floats)
  __shared__ float wlds[HT > 0 ? 4 * 64 * WS : 1];
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const int sl = lane & (G - 1), gbase = lane & ~(G - 1);
  const int64_t row = chunk_row[c];
  const int64_t eb = chunk_ebase[c];
  const int64_t rb = a.rowptr[row];
  const int64_t rem = a.rowptr[row + 1] - eb;
  const int n = rem < 64 ? (int)rem : 64;
  const uint32_t cl = a.col[eb + (lane < n ? lane : 0)];
  float wl = 0.f;  // weight of edge `lane` of the chunk (0 past the end)
  if (lane < n) {
    if constexpr (WMODE == 0) wl = a.rw[row];
    else if constexpr (WMODE == 1 || WMODE == 2) wl = load_edge_w<WMODE>(a, eb + lane);
  }
  const bool colok = sl * 4 < a.ncols;
  const int coff = colok ? sl * 4 : 0;
  const int head = MH ? coff / a.dh : 0;
  float* wd = wlds + (HT > 0 ? (threadIdx.x >> 6) * 64 * WS : 0);
  if constexpr (HT > 0) {
    f32x4_t wv[HT / 4];
    const float* src = a.ew + (eb + (lane < n ? lane : 0)) * HT;
#pragma unroll
    for (int q = 0; q < HT / 4; ++q) wv[q] = reinterpret_cast<const f32x4_t*>(src)[q];
#pragma unroll
    for (int q = 0; q < HT / 4; ++q)
#pragma unroll
      for (int k = 0; k < 4; ++k) wd[lane * WS + 4 * q + k] = lane < n ? wv[q][k] : 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // LDS operations of one wave complete in order
  }
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < G; j += U) {
    f32x4_t x[U];
    float w[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ei = gbase + j + u;  // edge of the chunk this group handles now
      const uint32_t cj = (uint32_t)__shfl((int)cl, ei, 64);
      x[u] = *reinterpret_cast<const f32x4_t*>(a.in + (int64_t)cj * a.ld + coff);
      if constexpr (HT > 0) w[u] = wd[ei * WS + head];
      else if constexpr (MH) w[u] = ei < n ? load_edge_w<WMODE>(a, eb + ei, head) : 0.f;
      else w[u] = __shfl(wl, ei, 64);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      // lanes past the end of a short chunk gathered the row of the chunk's FIRST edge: select the product, do not rely
      // on a zero weight (0 * Inf = NaN would enter the partial sum; the row kernels never touch non-existent edges)
      const bool live = gbase + j + u < n;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float t = live ? w[u] * x[u][k] : 0.f;
        acc[k] = acc[k] + t;
      }
    }
  }
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
  }
  const int64_t slot = (int64_t)chunk_start[row] + (eb - rb) / 64;
  if (gbase == 0 && colok) *reinterpret_cast<f32x4_t*>(partial + slot * a.ncols + coff) = acc;
}

// out[row] (+)= sum of the row's chunk partials, in row order.  One wave per row, 16 B per lane (ncols <= 256).
__global__ __launch_bounds__(256) void spmm_chunk_reduce_kernel(SpmmArgs a, const uint32_t* chunk_start,
                                                                const float* partial) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= a.n_rows) return;
  const int lane = threadIdx.x & 63;
  if (lane * 4 >= a.ncols) return;
  const int64_t c0 = chunk_start[row], c1 = chunk_start[row + 1];
  float* o = a.out + row * a.ldo + lane * 4;
  f32x4_t s = {0.f, 0.f, 0.f, 0.f};
  if (a.accumulate) s = *reinterpret_cast<const f32x4_t*>(o);
  const float* p = partial + lane * 4;
  int64_t k = c0;
  for (; k + 4 <= c1; k += 4) {
    f32x4_t t[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) t[u] = *reinterpret_cast<const f32x4_t*>(p + (k + u) * a.ncols);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int q = 0; q < 4; ++q) s[q] += t[u][q];
  }
  for (; k < c1; ++k) {
    const f32x4_t t = *reinterpret_cast<const f32x4_t*>(p + k * a.ncols);
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] += t[q];
  }
  if (a.relu) {
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] = s[q] > 0.f ? s[q] : 0.f;
  }
  *reinterpret_cast<f32x4_t*>(o) = s;
}

template <int WMODE>
int launch_chunked(gaib_ctx* ctx, gaib_graph* g, const SpmmArgs& a) {
  GAIB_TRY(gaib_graph_ensure_chunks(ctx, g));
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (size_t)g->n_chunks * a.ncols + 256));
  float* partial = (float*)ctx->ws;
  const unsigned grid = (unsigned)cdiv64(g->n_chunks > 0 ? g->n_chunks : 1, 4);
  const int lanes = (a.ncols + 3) / 4;
  {
    // col + gathered row + the edge's weights; a partial row per chunk written (and read again by the reduction)
    const double wb = WMODE == 0 ? 0.0 : (WMODE >= 3 ? 4.0 * a.heads : 4.0);
    ProfScope ps(ctx, "spmm_chunk", (double)g->ne * (4.0 * a.ncols + 4.0 + wb) + (double)g->n_chunks * 4.0 * a.ncols,
                 2.0 * (double)g->ne * a.ncols);
#define GAIB_CHUNK(GG) \
  spmm_chunk_kernel<GG, WMODE><<<grid, 256, 0, ctx->stream>>>(g->n_chunks, g->chunk_row, g->chunk_ebase, g->chunk_start, a, partial)
#define GAIB_CHUNK_H(GG, HH) \
  spmm_chunk_kernel<GG, 3, HH><<<grid, 256, 0, ctx->stream>>>(g->n_chunks, g->chunk_row, g->chunk_ebase, g->chunk_start, a, partial)
    const bool w16 = WMODE == 3 && (((uintptr_t)a.ew) & 15) == 0;
    if (w16 && lanes > 8 && lanes <= 16 && a.heads == 8) GAIB_CHUNK_H(16, 8);
    else if (w16 && lanes > 8 && lanes <= 16 && a.heads == 4) GAIB_CHUNK_H(16, 4);
    else if (w16 && lanes > 8 && lanes <= 16 && a.heads == 16) GAIB_CHUNK_H(16, 16);
    else if (w16 && lanes > 16 && lanes <= 32 && a.heads == 8) GAIB_CHUNK_H(32, 8);
    else if (w16 && lanes > 32 && a.heads == 8) GAIB_CHUNK_H(64, 8);
    else if (lanes <= 1) GAIB_CHUNK(1);
    else if (lanes <= 2) GAIB_CHUNK(2);
    else if (lanes <= 4) GAIB_CHUNK(4);
    else if (lanes <= 8) GAIB_CHUNK(8);
    else if (lanes <= 16) GAIB_CHUNK(16);
    else if (lanes <= 32) GAIB_CHUNK(32);
    else GAIB_CHUNK(64);
#undef GAIB_CHUNK
#undef GAIB_CHUNK_H
    GAIB_LAUNCH_CHECK();
  }
  ProfScope ps(ctx, "spmm_chunk_reduce", ((double)g->n_chunks + (double)a.n_rows * (a.accumulate ? 2 : 1)) * 4.0 * a.ncols);
  spmm_chunk_reduce_kernel<<<(unsigned)cdiv64(a.n_rows, 4), 256, 0, ctx->stream>>>(a, g->chunk_start, partial);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

// out[r][0..lp) = in[r][0..len) followed by zeros; lp % 4 == 0, out 16-B aligned.  One float4 of `out` per thread.
__global__ __launch_bounds__(256) void pad_rows_kernel(int64_t n4, int len, int lp4, const float* in, f32x4_t* out) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / lp4;
    const int c = (int)(i - r * lp4) * 4;
    const float* src = in + r * len + c;
    f32x4_t v;
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = c + k < len ? src[k] : 0.f;
    out[i] = v;
  }
}

// 128-B lines a gathered row of `bytes` bytes touches on average when rows are `bytes` apart (any 4-B alignment) ...
inline double lines_packed(int bytes) { return 1.0 + (bytes - 4) / 128.0; }
// ... and when rows start on 64-B boundaries `stride` bytes apart (stride % 64 == 0)
inline double lines_strided(int bytes, int stride) {
  const int at0 = (bytes + 127) / 128, at64 = (bytes + 64 + 127) / 128;
  return stride % 128 == 0 ? at0 : 0.5 * (at0 + at64);
}

}  // namespace

// fills the launch arguments shared by every aggregation path; *wmode = kernel weight mode
// d_in2 / n_first: column ids >= n_first index the second table d_in2 (row id - n_first) -- NULL: one table
static int spmm_setup(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len,
                      const float* d_in, float* d_out, int flags, int heads, SpmmArgs* pa, int* wmode,
                      const float* d_in2 = nullptr, int64_t n_first = 0) {
  SpmmArgs& a = *pa;
  const bool part = g->row_map != nullptr || d_in2 != nullptr;  // a row class of a partition (spmm_part.hip)
  GAIB_CHECK(!d_in2 || (n_first >= 0 && n_first <= g->nc), "gaib_spmm: n_first (%lld) outside the %lld columns",
             (long long)n_first, (long long)g->nc);
  GAIB_HIP(hipSetDevice(ctx->device));
  GAIB_TRY(gaib_graph_ensure_heavy(ctx, g, ctx->spmm_heavy_threshold));
  a.rowptr = g->rowptr;
  a.col = g->colidx;
  a.rw = nullptr;
  a.ew = nullptr;
  a.rev = nullptr;
  a.in = d_in;
  a.out = d_out;
  a.ld = len;
  a.ncols = len;
  a.n_rows = (int)g->nv;
  a.heavy_thr = g->heavy_thr;
  a.row_list = nullptr;
  a.row_order = nullptr;
  a.nblocks = 0;
  a.per_xcd = 0;
  a.xcd_chunk = 0;
  a.accumulate = (flags & GAIB_ACCUMULATE) ? 1 : 0;
  a.relu = (flags & GAIB_RELU) ? 1 : 0;
  a.heads = heads;
  a.dh = heads > 0 ? len / heads : len;
  a.col_flagged = nullptr;
  a.compact = 0;
  a.row_map = g->row_map;
  a.in2 = d_in2;
  a.n_first = d_in2 ? (uint32_t)n_first : 0xffffffffu;
  a.in2_bytes = 0;
  if (!part && ctx->spmm_gather_mode == 3 && g->nc == g->nv && !g->col_vdata) {
    GAIB_TRY(gaib_graph_ensure_hot_flags(ctx, g, len));
    a.col_flagged = g->colidx_flagged;
  }
  a.ldo = len;
  // Rows whose byte length is not a multiple of 64 straddle more 128-B lines than they fill (a 188-B row of the
  // 47-class output layer touches 2.44 lines on average, 2 when rows start on 64-B boundaries), and the gather time
  // follows the lines touched: products, D = 47: 5.28 ms, D = 48: 4.04 ms.  Where re-striding saves more than 10 %
  // of the lines, the input table is copied once into rows of whole 64-B pieces (0.9 GB of streaming traffic against
  // 40 GB of gathers at products / 47) and gathered from there; output rows keep the caller's stride.
  {
    const int bytes = len * 4, stride = (bytes + 63) & ~63;
    // (not inside a side-stream section: the copy buffer belongs to the main stream's calls)
    if (ctx->spmm_pad && !part && !ctx->forked && len > 16 && stride != bytes && g->ne > 4 * g->nc &&
        lines_strided(bytes, stride) < 0.9 * lines_packed(bytes)) {
      const int lp = stride / 4;
      GAIB_TRY(gaib_pad_reserve(ctx, (size_t)g->nc * (size_t)stride));
      const int64_t n4 = g->nc * (int64_t)(lp / 4);
      const unsigned grid = (unsigned)std::min<int64_t>(cdiv64(n4, 256), (int64_t)ctx->num_cus * 16);
      pad_rows_kernel<<<grid, 256, 0, ctx->stream>>>(n4, len, lp / 4, d_in, reinterpret_cast<f32x4_t*>(ctx->pad));
      GAIB_LAUNCH_CHECK();
      a.in = reinterpret_cast<const float*>(ctx->pad);
      a.ld = lp;
    }
  }
  // feature table = nc rows of a.ld floats; the 32-bit buffer path needs it below 4 GB
  const int64_t table_bytes = (d_in2 ? n_first : g->nc) * a.ld * 4;
  a.in_bytes = table_bytes < ((int64_t)1 << 32) ? (uint32_t)table_bytes : 0u;
  if (d_in2) {
    const int64_t t2 = (g->nc - n_first) * a.ld * 4;
    a.in2_bytes = t2 < ((int64_t)1 << 32) ? (uint32_t)t2 : 0u;
  }
  switch (weight_kind) {
    case GAIB_W_GCN:
      GAIB_TRY(gaib_graph_ensure_w_gcn(ctx, g));
      a.ew = g->w_gcn;
      *wmode = 1;
      return GAIB_OK;
    case GAIB_W_MEAN:
      GAIB_TRY(gaib_graph_ensure_inv_deg(ctx, g));
      a.rw = g->inv_deg;
      *wmode = 0;
      return GAIB_OK;
    case GAIB_W_MEAN_T:
      GAIB_TRY(gaib_graph_ensure_w_mean_t(ctx, g));
      a.ew = g->w_mean_t;
      *wmode = 1;
      return GAIB_OK;
    case GAIB_W_EDGE:
      GAIB_CHECK(d_edge_w || g->ne == 0, "gaib_spmm: GAIB_W_EDGE needs d_edge_w");
      a.ew = d_edge_w;
      *wmode = heads > 1 ? 3 : 1;
      return GAIB_OK;
    case GAIB_W_EDGE_T:
      GAIB_CHECK(d_edge_w || g->ne == 0, "gaib_spmm: GAIB_W_EDGE_T needs d_edge_w");
      GAIB_TRY(gaib_graph_ensure_rev(ctx, g));
      a.ew = d_edge_w;
      a.rev = g->rev;
      *wmode = heads > 1 ? 4 : 2;
      return GAIB_OK;
    default:
      gaib_set_error("gaib_spmm: unknown weight_kind %d", weight_kind);
      return GAIB_ERR_INVALID;
  }
}

// Where the ordered-chunk aggregation pays (scripts/ab_chunked.py, microbench_layers.py): a chunk of 64 edges of a row
// with d edges spans 64/d of the columns, so it is L2-sized only on long rows -- on the reddit shape the rows above the
// heavy threshold hold half of the edges and the two aggregations of a layer drop from 6.65 to 4.1 ms, on uniform random
// graphs of the same density nothing is gained (-12 % at 256 edges per row, +5-8 % at 32-128).  Rule: a table that can
// live in the Infinity Cache and at least a quarter of the edges in heavy rows.
static bool chunk_rule(gaib_ctx* ctx, gaib_graph* g, int64_t ld) {
  if (gaib_graph_ensure_heavy(ctx, g, ctx->spmm_heavy_threshold) != GAIB_OK) return false;
  return g->nc * ld * 4 <= ((int64_t)512 << 20) && 4 * g->heavy_edges >= g->ne && g->ne > 0;
}

// the row-class kernels (PART = true) live in a translation unit of their own: spmm_part.hip
int gaib_spmm_part_plain(gaib_ctx* ctx, const gaib_graph* g, const void* spmm_args, int wmode, int len);
int gaib_spmm_part_fused(gaib_ctx* ctx, const gaib_graph* g, const void* spmm_args, const void* fuse_args, float* heavy_scratch,
                         int vec, int wmode);

static int spmm_impl(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len,
                     const float* d_in, float* d_out, int flags, int heads = 1, const float* d_in2 = nullptr,
                     int64_t n_first = 0) {
  GAIB_CHECK(ctx && g, "gaib_spmm: NULL ctx/graph");
  GAIB_CHECK(len >= 0, "gaib_spmm: len < 0");
  GAIB_CHECK(ctx->device == g->device, "gaib_spmm: graph lives on device %d, ctx on %d", g->device,
             ctx->device);
  if (len == 0 || g->nv == 0) return GAIB_OK;
  GAIB_CHECK(d_in && d_out, "gaib_spmm: NULL feature pointer");
  GAIB_CHECK(d_in != d_out, "gaib_spmm: in and out must not alias");
  SpmmArgs a;
  int wmode = 0;
  GAIB_TRY(spmm_setup(ctx, g, weight_kind, d_edge_w, len, d_in, d_out, flags, heads, &a, &wmode, d_in2, n_first));
  if (g->row_map || d_in2) {
    GAIB_CHECK(d_in2 != d_out, "gaib_spmm: in2 and out must not alias");
    GAIB_CHECK(heads == 1 && wmode <= 1, "gaib_spmm: a row class of a partition aggregates with GAIB_W_GCN / _MEAN / _MEAN_T / "
                                          "single-head _EDGE weights (got kind %d, %d heads)", weight_kind, heads);
    return gaib_spmm_part_plain(ctx, g, &a, wmode, len);
  }
  // dense graphs over a table that fits the Infinity Cache: ordered chunks (see spmm_chunk_kernel)
  const bool chunk_shape = len % 4 == 0 && len <= 256 && a.ld % 4 == 0 && a.ldo % 4 == 0 && g->ne > 0 &&
                           ((((uintptr_t)a.in | (uintptr_t)a.out) & 15) == 0) && (wmode < 3 || a.dh % 4 == 0);
  const bool chunk_auto = ctx->spmm_chunked < 0 && chunk_rule(ctx, g, a.ld);
  if (chunk_shape && (ctx->spmm_chunked == 1 || (ctx->spmm_chunked < 0 && chunk_auto))) {
    switch (wmode) {
      case 0: return launch_chunked<0>(ctx, g, a);
      case 1: return launch_chunked<1>(ctx, g, a);
      case 2: return launch_chunked<2>(ctx, g, a);
      case 3: return launch_chunked<3>(ctx, g, a);
      default: return launch_chunked<4>(ctx, g, a);
    }
  }
  switch (wmode) {
    case 0: return dispatch_vec<0>(ctx, g, a, len);
    case 1: return dispatch_vec<1>(ctx, g, a, len);
    case 2: return dispatch_vec<2>(ctx, g, a, len);
    case 3: return dispatch_vec<3>(ctx, g, a, len);
    default: return dispatch_vec<4>(ctx, g, a, len);
  }
}

extern "C" int gaib_spmm(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w,
                         int len, const float* d_in, float* d_out) {
  return spmm_impl(ctx, g, weight_kind, d_edge_w, len, d_in, d_out, 0);
}

extern "C" int gaib_spmm_acc(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w,
                             int len, const float* d_in, float* d_out) {
  return spmm_impl(ctx, g, weight_kind, d_edge_w, len, d_in, d_out, GAIB_ACCUMULATE);
}

extern "C" int gaib_spmm_ex(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w,
                            int len, const float* d_in, float* d_out, int flags) {
  return spmm_impl(ctx, g, weight_kind, d_edge_w, len, d_in, d_out, flags);
}

extern "C" int gaib_spmm_2t(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len,
                            const float* d_in, const float* d_in2, int64_t n_first, float* d_out, int flags) {
  GAIB_CHECK(ctx && g, "gaib_spmm_2t: NULL ctx/graph");
  GAIB_CHECK(d_in2 || n_first >= g->nc || g->nv == 0 || g->ne == 0, "gaib_spmm_2t: NULL second table with columns beyond n_first");
  return spmm_impl(ctx, g, weight_kind, d_edge_w, len, d_in, d_out, flags, 1, d_in2, n_first);
}

// agg = A.in ; out = act(agg . op(W) [+ rows2 . op(W2)]).  Fused on the matrix cores when the shape allows,
// otherwise gaib_spmm followed by gaib_sgemm (same results up to summation order).
// shape-only part of the "can the product ride on the aggregation" decision (what the row classes of a partition must
// agree on: they share one output matrix, so either every class runs the fused kernel or none does)
static bool fuse_shape_ok(gaib_ctx* ctx, int weight_kind, int len_in, int len_out, bool dual) {
  if (!ctx->spmm_fuse || len_in < 1 || len_in > 128 || len_out < 1) return false;
  if (len_in > 64 && len_in % 2 != 0) return false;
  if (!(weight_kind == GAIB_W_GCN || weight_kind == GAIB_W_MEAN || weight_kind == GAIB_W_MEAN_T || weight_kind == GAIB_W_EDGE))
    return false;
  return fuse_strip_rows(len_in <= 64 ? 64 : 128, len_out, dual) != 0;
}

extern "C" int gaib_spmm_gemm_fusable(gaib_ctx* ctx, int weight_kind, int len_in, int len_out, int dual) {
  return ctx && fuse_shape_ok(ctx, weight_kind, len_in, len_out, dual != 0) ? 1 : 0;
}

static int spmm_gemm_impl(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len_in,
                          const float* d_in, float* d_agg, const float* d_W, int transW, const float* d_rows2,
                          const float* d_W2, int len_out, float* d_out, int flags, const float* d_in2 = nullptr,
                          int64_t n_first = 0) {
  GAIB_CHECK(ctx && g, "gaib_spmm_gemm: NULL ctx/graph");
  GAIB_CHECK(len_in >= 0 && len_out >= 0, "gaib_spmm_gemm: negative length");
  GAIB_CHECK(ctx->device == g->device, "gaib_spmm_gemm: graph lives on device %d, ctx on %d", g->device,
             ctx->device);
  GAIB_CHECK((flags & ~(GAIB_RELU | GAIB_AGG_SCRATCH | GAIB_ACCUMULATE | GAIB_OVERLAPS_TRANSFER)) == 0,
             "gaib_spmm_gemm: unsupported flags %d", flags);
  GAIB_CHECK((d_rows2 == nullptr) == (d_W2 == nullptr), "gaib_spmm_gemm2: rows2 and W2 go together");
  if (g->nv == 0 || len_out == 0) return GAIB_OK;
  GAIB_CHECK(d_in && d_agg && d_W && d_out, "gaib_spmm_gemm: NULL pointer");
  GAIB_CHECK(d_in != d_agg && d_agg != d_out && d_in != d_out && d_rows2 != d_out && d_rows2 != d_agg,
             "gaib_spmm_gemm: buffers must not alias");
  const bool dual = d_rows2 != nullptr;
  const bool part = g->row_map != nullptr || d_in2 != nullptr;
  const uintptr_t al = (uintptr_t)d_in | (uintptr_t)d_agg | (uintptr_t)d_in2;
  // the weight matrices [len_out x (len_in+4)] and 16 row strips must fit the CU's 160 KB of LDS;
  // 65..128 columns need 8-byte lanes
  const int kpad = len_in <= 64 ? 64 : 128;
  const bool lanes_ok = len_in <= 64 ? true : (len_in % 2 == 0 && (al & 7) == 0);
  // dense graphs over a cache-sized table aggregate faster by ordered chunks (spmm_chunk_kernel) than row by row
  // inside the fused kernel: two kernels there
  const bool dense = !part && ctx->spmm_chunked != 0 && len_in % 4 == 0 && (al & 15) == 0 &&
                     (ctx->spmm_chunked == 1 || chunk_rule(ctx, g, len_in));
  // (a row class may be without edges -- isolated interior vertices -- and still owes its rows of y)
  const bool shape_ok = ctx->spmm_fuse != 0 && !dense && len_in >= 1 && len_in <= 128 && lanes_ok && (g->ne > 0 || part) &&
                        g->nv >= 1 &&
                        (weight_kind == GAIB_W_GCN || weight_kind == GAIB_W_MEAN ||
                         weight_kind == GAIB_W_MEAN_T || weight_kind == GAIB_W_EDGE);
  // two products whose matrices do not fit LDS together (SAGE's 100 -> 256 input layer): the neighbour product still
  // rides on the aggregation, the self term follows as an accumulating GEMM that also applies the activation
  // (whole graphs only: a row class of a partition -- row map, second feature table -- cannot take this route, the
  // accumulating GEMM below runs over rows [0, nv) of rows2 / out; it gets GAIB_ERR_UNSUPPORTED further down)
  if (!part && shape_ok && dual && fuse_strip_rows(kpad, len_out, true) == 0 && fuse_strip_rows(kpad, len_out, false) != 0) {
    GAIB_TRY(spmm_gemm_impl(ctx, g, weight_kind, d_edge_w, len_in, d_in, d_agg, d_W, transW, nullptr, nullptr, len_out, d_out,
                            flags & ~GAIB_RELU));
    return gaib_sgemm_ex(ctx, 0, transW, g->nv, len_out, len_in, d_rows2, d_W2,
                         GAIB_ACCUMULATE | ((flags & GAIB_RELU) ? GAIB_RELU : 0), d_out);
  }
  const bool fusable = shape_ok && fuse_strip_rows(kpad, len_out, dual) != 0;
  if (part) {
    // the classes of a partition write disjoint rows of ONE output: the caller (host/aggregators.cpp) asks
    // gaib_spmm_gemm_fusable first and runs aggregation + one dense product over all rows where the answer is no
    if (!fusable) {
      gaib_set_error("gaib_spmm_gemm: a row class of a partition needs a fusable shape (len_in %d, len_out %d%s): aggregate "
                     "with gaib_spmm_ex / gaib_spmm_2t and multiply all rows at once", len_in, len_out, dual ? ", two products" : "");
      return GAIB_ERR_UNSUPPORTED;
    }
  }
  // 129..256 columns (the hidden width 256 of scripts/run-sage-products.sh): op(W) does not fit LDS next to the strips,
  // so the aggregation runs as two 128-column K-slabs through the same kernel, each with its [len_out x 128] slab of
  // op(W) in LDS: slab 0 writes y = agg[:, :128] . op(W)[:128, :], slab 1 adds agg[:, 128:] . op(W)[128:, :] (and applies
  // the activation).  The gathered bytes are the same as one 1-KB row gather per edge; colidx / weights are streamed
  // twice and y is read-modify-written once -- against a separate pass over agg and a 2 x 2.45 M x 256 x 256 GEMM.
  // A second product (SAGE's self term) follows as an accumulating GEMM.
  const bool kslab = !part && ctx->spmm_fuse != 0 && !dense && len_in > 128 && len_in <= 256 && len_in % 2 == 0 && (al & 7) == 0 &&
                     len_out >= 1 && fuse_strip_rows(128, len_out, false) != 0 && g->ne > 0 &&
                     (weight_kind == GAIB_W_GCN || weight_kind == GAIB_W_MEAN || weight_kind == GAIB_W_MEAN_T ||
                      weight_kind == GAIB_W_EDGE);
  if (kslab) {
    SpmmArgs a0;
    int wmode = 0;
    GAIB_TRY(spmm_setup(ctx, g, weight_kind, d_edge_w, len_in, d_in, d_agg, 0, 1, &a0, &wmode));
    const size_t wt_bytes = (sizeof(float) * (size_t)len_out * len_in + 255) & ~(size_t)255;
    const size_t hv_bytes = (sizeof(float) * (size_t)g->n_heavy * len_in + 255) & ~(size_t)255;
    GAIB_TRY(gaib_ws_reserve(ctx, wt_bytes + hv_bytes + 256));
    float* wt = (float*)ctx->ws;
    float* hv = (float*)((char*)ctx->ws + wt_bytes);
    int* counter = (int*)((char*)hv + hv_bytes);
    const float* wk = d_W;  // [len_out][len_in], k-contiguous
    if (!transW) {
      const int n = len_in * len_out;
      transpose_small_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(len_in, len_out, d_W, wt);
      GAIB_LAUNCH_CHECK();
      wk = wt;
    }
    for (int k0 = 0; k0 < len_in; k0 += 128) {
      SpmmArgs a = a0;
      a.in = a0.in + k0;
      a.ncols = len_in - k0 < 128 ? len_in - k0 : 128;
      a.accumulate = 0;
      a.out = (flags & GAIB_AGG_SCRATCH) ? nullptr : d_agg + k0;
      if (a.in_bytes) a.in_bytes -= (uint32_t)(k0 * 4);
      FuseArgs f;
      f.wt = wk + k0;
      f.wt2 = nullptr;
      f.rows2 = nullptr;
      f.ldw = len_in;
      f.y = d_out;
      f.ldy = len_out;
      f.n_out = len_out;
      f.tile_counter = counter;
      f.agg_in = (flags & GAIB_ACCUMULATE) ? d_agg + k0 : nullptr;
      f.y_accum = k0 > 0;
      f.overlaps_transfer = (flags & GAIB_OVERLAPS_TRANSFER) ? 1 : 0;
      f.tile_xcd = tile_xcd_arg(ctx, g);
      f.relu = ((flags & GAIB_RELU) && !dual && k0 + 128 >= len_in) ? 1 : 0;
      f.heavy_agg = hv + k0;
      f.heavy_rows = g->heavy_rows;
      f.n_heavy = (int)g->n_heavy;
      GAIB_TRY(wmode == 0 ? (launch_fused<2, 0>(ctx, g, a, f, hv + k0)) : (launch_fused<2, 1>(ctx, g, a, f, hv + k0)));
    }
    if (dual)
      return gaib_sgemm_ex(ctx, 0, transW, g->nv, len_out, len_in, d_rows2, d_W2,
                           GAIB_ACCUMULATE | ((flags & GAIB_RELU) ? GAIB_RELU : 0), d_out);
    return GAIB_OK;
  }
  if (!fusable) {
    const int act = (flags & GAIB_RELU) ? GAIB_RELU : 0;
    GAIB_TRY(spmm_impl(ctx, g, weight_kind, d_edge_w, len_in, d_in, d_agg, flags & GAIB_ACCUMULATE));
    GAIB_TRY(gaib_sgemm_ex(ctx, 0, transW, g->nv, len_out, len_in, d_agg, d_W, dual ? 0 : act, d_out));
    if (dual) return gaib_sgemm_ex(ctx, 0, transW, g->nv, len_out, len_in, d_rows2, d_W2, GAIB_ACCUMULATE | act, d_out);
    return GAIB_OK;
  }
  SpmmArgs a;
  int wmode = 0;
  GAIB_TRY(spmm_setup(ctx, g, weight_kind, d_edge_w, len_in, d_in, d_agg, 0, 1, &a, &wmode, d_in2, n_first));
  a.accumulate = 0;  // (the fused kernel takes the partial sums through f.agg_in)
  if (flags & GAIB_AGG_SCRATCH) a.out = nullptr;  // the caller does not read agg: skip its store
  // scratch: op(W) (and op(W2)) k-contiguous + the heavy rows' aggregates + the tile counter
  const size_t wt_bytes = (sizeof(float) * (size_t)len_out * len_in + 255) & ~(size_t)255;
  const size_t hv_bytes = (sizeof(float) * (size_t)g->n_heavy * len_in + 255) & ~(size_t)255;
  GAIB_TRY(gaib_ws_reserve(ctx, 2 * wt_bytes + hv_bytes + 256));
  float* wt = (float*)ctx->ws;
  float* wt2 = (float*)((char*)ctx->ws + wt_bytes);
  float* hv = (float*)((char*)ctx->ws + 2 * wt_bytes);
  int* counter = (int*)((char*)hv + hv_bytes);
  FuseArgs f;
  f.wt = d_W;  // with transW, W is [len_out x len_in]: already k-contiguous
  f.wt2 = d_W2;
  if (!transW) {
    const int n = len_in * len_out;
    transpose_small_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(len_in, len_out, d_W, wt);
    GAIB_LAUNCH_CHECK();
    f.wt = wt;
    if (dual) {
      transpose_small_kernel<<<(n + 255) / 256, 256, 0, ctx->stream>>>(len_in, len_out, d_W2, wt2);
      GAIB_LAUNCH_CHECK();
      f.wt2 = wt2;
    }
  }
  f.rows2 = d_rows2;
  f.ldw = len_in;
  f.y_accum = 0;
  f.overlaps_transfer = (flags & GAIB_OVERLAPS_TRANSFER) ? 1 : 0;
  f.tile_xcd = tile_xcd_arg(ctx, g);
  f.y = d_out;
  f.ldy = len_out;
  f.n_out = len_out;
  f.tile_counter = counter;
  f.agg_in = (flags & GAIB_ACCUMULATE) ? d_agg : nullptr;
  f.relu = (flags & GAIB_RELU) ? 1 : 0;
  f.heavy_agg = hv;
  f.heavy_rows = g->heavy_rows;
  f.n_heavy = (int)g->n_heavy;
  if (part) return gaib_spmm_part_fused(ctx, g, &a, &f, hv, len_in <= 64 ? 1 : 2, wmode);
  if (len_in <= 64) {
    return wmode == 0 ? launch_fused<1, 0>(ctx, g, a, f, hv) : launch_fused<1, 1>(ctx, g, a, f, hv);
  }
  return wmode == 0 ? launch_fused<2, 0>(ctx, g, a, f, hv) : launch_fused<2, 1>(ctx, g, a, f, hv);
}

extern "C" int gaib_spmm_gemm(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w,
                              int len_in, const float* d_in, float* d_agg, const float* d_W, int transW,
                              int len_out, float* d_out, int flags) {
  return spmm_gemm_impl(ctx, g, weight_kind, d_edge_w, len_in, d_in, d_agg, d_W, transW, nullptr, nullptr, len_out,
                        d_out, flags);
}

extern "C" int gaib_spmm_gemm2(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w,
                               int len_in, const float* d_in, float* d_agg, const float* d_W, int transW,
                               const float* d_rows2, const float* d_W2, int len_out, float* d_out, int flags) {
  GAIB_CHECK(d_rows2 && d_W2, "gaib_spmm_gemm2: NULL second operand");
  return spmm_gemm_impl(ctx, g, weight_kind, d_edge_w, len_in, d_in, d_agg, d_W, transW, d_rows2, d_W2, len_out,
                        d_out, flags);
}

extern "C" int gaib_spmm_gemm_2t(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len_in,
                                 const float* d_in, const float* d_in2, int64_t n_first, float* d_agg, const float* d_W,
                                 int transW, const float* d_rows2, const float* d_W2, int len_out, float* d_out, int flags) {
  GAIB_CHECK(ctx && g, "gaib_spmm_gemm_2t: NULL ctx/graph");
  GAIB_CHECK(d_in2 || n_first >= g->nc || g->nv == 0 || g->ne == 0, "gaib_spmm_gemm_2t: NULL second table with columns beyond n_first");
  GAIB_CHECK(!d_in2 || (d_in2 != d_agg && d_in2 != d_out), "gaib_spmm_gemm_2t: buffers must not alias");
  return spmm_gemm_impl(ctx, g, weight_kind, d_edge_w, len_in, d_in, d_agg, d_W, transW, d_rows2, d_W2, len_out, d_out, flags,
                        d_in2, n_first);
}

extern "C" int gaib_spmm_mh(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w,
                            int heads, int len, const float* d_in, float* d_out, int flags) {
  GAIB_CHECK(heads >= 1 && len % heads == 0, "gaib_spmm_mh: heads (%d) must divide len (%d)", heads, len);
  GAIB_CHECK(heads == 1 || weight_kind == GAIB_W_EDGE || weight_kind == GAIB_W_EDGE_T,
             "gaib_spmm_mh: per-head weights need GAIB_W_EDGE or GAIB_W_EDGE_T");
  return spmm_impl(ctx, g, weight_kind, d_edge_w, len, d_in, d_out, flags, heads);
}
