// graph.hip -- device half of LearningGraph (include/gnn/lgraph.h:20-277 in the reference):
// CSR upload, add_selfloop, normalisers, and the once-per-graph schedules the aggregation
// kernels use (heavy-row list, per-edge weights, reverse-edge permutation).
//
// HBM layout: rowptr int64[nv+1] | colidx uint32[ne] | vdata/inv_deg fp32[nv] |
// w_gcn / w_mean_t / edata fp32[ne] | rev uint32[ne].  Everything is streamed linearly by the
// kernels that use it; only the feature rows are gathered.
#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <functional>
#include <vector>
#include "common.h"

namespace {

__global__ void widen_rowptr_kernel(int64_t n, const uint32_t* in, int64_t* out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (int64_t)in[i];
}

// compute_vertex_data: src/gnn/lgraph.cpp:22-34 (kernel: lgraph.cu:6-12)
__global__ void vertex_data_kernel(int64_t nv, const int64_t* rowptr, float* vd) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  uint32_t deg = (uint32_t)(rowptr[v + 1] - rowptr[v]);
  float temp = sqrtf((float)deg);
  vd[v] = (temp == 0.0f) ? 0.0f : (float)(1.0 / (double)temp);
}

// (float)(1.0 / float(deg)) as sage_aggregator.cpp:18,44 evaluate it
__global__ void inv_deg_kernel(int64_t nv, const int64_t* rowptr, float* inv) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  uint32_t deg = (uint32_t)(rowptr[v + 1] - rowptr[v]);
  inv[v] = (float)(1.0 / (double)(float)deg);
}

// one wave per row; lanes stride the row's edges
template <int KIND>  // 0: edge_data (lgraph.cpp:6-20)  1: w_gcn = vd[i]*vd[j] (gcn_aggregator.cpp:61-66)
__global__ __launch_bounds__(256) void edge_weight_kernel(int64_t nv, const int64_t* rowptr,
                                                          const uint32_t* col, const float* vd,
                                                          const float* cvd, float* ew) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  if (KIND == 0) {
    float c_i = sqrtf((float)(uint32_t)(e1 - e0));
    for (int64_t e = e0 + lane; e < e1; e += 64) {
      uint32_t j = col[e];
      float c_j = sqrtf((float)(uint32_t)(rowptr[j + 1] - rowptr[j]));
      ew[e] = (c_i == 0.0f || c_j == 0.0f) ? 0.0f : (float)(1.0 / (double)(c_i * c_j));
    }
  } else {
    float a = vd[row];
    for (int64_t e = e0 + lane; e < e1; e += 64) ew[e] = a * cvd[col[e]];
  }
}

__global__ void gather_by_col_kernel(int64_t ne, const uint32_t* col, const float* pv, float* ew) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < ne) ew[e] = pv[col[e]];
}

// reverse-edge permutation: rev[e] = position of (dst -> src) in row dst
// (binary search as math_functions.cpp:32-44,60-73; graph_operations.h:340-361)
__global__ __launch_bounds__(256) void rev_kernel(int64_t nv, const int64_t* rowptr,
                                                  const uint32_t* col, uint32_t* rev, int* bad) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  for (int64_t e = e0 + lane; e < e1; e += 64) {
    uint32_t dst = col[e];
    int64_t l = rowptr[dst], r = rowptr[dst + 1] - 1, idx = -1;
    while (r >= l) {
      int64_t mid = l + (r - l) / 2;
      uint32_t v = col[mid];
      if (v == (uint32_t)row) { idx = mid; break; }
      if (v < (uint32_t)row) l = mid + 1;
      else r = mid - 1;
    }
    if (idx < 0) { *bad = 1; rev[e] = (uint32_t)e; }
    else rev[e] = (uint32_t)idx;
  }
}

__global__ void heavy_rows_kernel(int64_t nv, const int64_t* rowptr, int thr, uint32_t* list,
                                  unsigned long long* count, unsigned long long* maxdeg) {
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  int64_t deg = rowptr[v + 1] - rowptr[v];
  if (deg > thr) {
    unsigned long long p = atomicAdd(count, 1ull);
    if (list) list[p] = (uint32_t)v;
    else atomicAdd(count + 2, (unsigned long long)deg);  // counting pass: edges held by heavy rows
  }
  if (maxdeg) atomicMax(maxdeg, (unsigned long long)deg);
}

// add_selfloop (include/gnn/lgraph.h:185-218): row i keeps its sorted order with i inserted
// after the last entry <= i; new rowptr[i] = rowptr[i] + i.
__global__ __launch_bounds__(256) void selfloop_kernel(int64_t nv, const int64_t* rowptr,
                                                       const uint32_t* col, int64_t* rowptr_out,
                                                       uint32_t* col_out) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row > nv) return;
  const int lane = threadIdx.x & 63;
  if (row == nv) {
    if (lane == 0) rowptr_out[nv] = rowptr[nv] + nv;
    return;
  }
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  int cnt = 0;
  for (int64_t e = e0 + lane; e < e1; e += 64) {
    uint32_t dst = col[e];
    bool after = dst > (uint32_t)row;
    col_out[e + row + (after ? 1 : 0)] = dst;
    cnt += after ? 0 : 1;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) {
    col_out[e0 + row + cnt] = (uint32_t)row;
    rowptr_out[row] = e0 + row;
  }
}

inline dim3 grid1d(int64_t n, int block) { return dim3((unsigned)cdiv64(n > 0 ? n : 1, block)); }

int new_graph(int64_t nv, int64_t ne, int device, gaib_graph** out) {
  gaib_graph* g = new gaib_graph();
  memset(g, 0, sizeof(*g));
  g->device = device;
  g->nv = nv;
  g->nc = nv;
  g->ne = ne;
  g->heavy_thr = -1;
  g->hot_threshold = -1;
  g->near_frac = -1.f;
  g->max_degree = -1;
  GAIB_HIP(hipMalloc(&g->rowptr, sizeof(int64_t) * (size_t)(nv + 1)));
  GAIB_HIP(hipMalloc(&g->colidx, sizeof(uint32_t) * (size_t)(ne > 0 ? ne : 1)));
  g->dev_bytes = sizeof(int64_t) * (nv + 1) + sizeof(uint32_t) * ne;
  *out = g;
  return GAIB_OK;
}

}  // namespace

extern "C" int gaib_graph_create(gaib_ctx* ctx, int64_t nv, int64_t ne, const void* rowptr,
                                 int rowptr_bits, const uint32_t* colidx, int src_on_device,
                                 gaib_graph** out) {
  return gaib_graph_create_rect(ctx, nv, nv, ne, rowptr, rowptr_bits, colidx, src_on_device, out);
}

extern "C" int gaib_graph_set_vertex_norm(gaib_ctx* ctx, gaib_graph* g, const float* d_row_vdata,
                                          const float* d_row_inv_deg, const float* d_col_vdata,
                                          const float* d_col_inv_deg) {
  GAIB_CHECK(ctx && g && d_col_vdata && d_col_inv_deg, "gaib_graph_set_vertex_norm: NULL argument");
  GAIB_HIP(hipSetDevice(ctx->device));
  const size_t nc = (size_t)(g->nc > 0 ? g->nc : 1), nv = (size_t)(g->nv > 0 ? g->nv : 1);
  if (!g->col_vdata) {
    GAIB_HIP(hipMalloc(&g->col_vdata, sizeof(float) * nc));
    GAIB_HIP(hipMalloc(&g->col_inv_deg, sizeof(float) * nc));
    g->dev_bytes += 2 * sizeof(float) * g->nc;
  }
  GAIB_HIP(hipMemcpyAsync(g->col_vdata, d_col_vdata, sizeof(float) * g->nc, hipMemcpyDeviceToDevice, ctx->stream));
  GAIB_HIP(hipMemcpyAsync(g->col_inv_deg, d_col_inv_deg, sizeof(float) * g->nc, hipMemcpyDeviceToDevice, ctx->stream));
  if (!g->vdata) {
    GAIB_HIP(hipMalloc(&g->vdata, sizeof(float) * nv));
    g->dev_bytes += sizeof(float) * g->nv;
  }
  if (d_row_vdata)
    GAIB_HIP(hipMemcpyAsync(g->vdata, d_row_vdata, sizeof(float) * g->nv, hipMemcpyDeviceToDevice, ctx->stream));
  else
    vertex_data_kernel<<<grid1d(g->nv, 256), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->vdata);
  GAIB_LAUNCH_CHECK();
  if (d_row_inv_deg) {
    if (!g->inv_deg) {
      GAIB_HIP(hipMalloc(&g->inv_deg, sizeof(float) * nv));
      g->dev_bytes += sizeof(float) * g->nv;
    }
    GAIB_HIP(hipMemcpyAsync(g->inv_deg, d_row_inv_deg, sizeof(float) * g->nv, hipMemcpyDeviceToDevice, ctx->stream));
  }
  // cached per-edge weights derive from these
  if (g->w_gcn) {
    edge_weight_kernel<1><<<grid1d(g->nv, 4), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->colidx,
                                                                     g->vdata, g->col_vdata, g->w_gcn);
    GAIB_LAUNCH_CHECK();
  }
  if (g->w_mean_t) {
    gather_by_col_kernel<<<grid1d(g->ne, 256), 256, 0, ctx->stream>>>(g->ne, g->colidx, g->col_inv_deg,
                                                                      g->w_mean_t);
    GAIB_LAUNCH_CHECK();
  }
  return GAIB_OK;
}

extern "C" int gaib_graph_create_rect(gaib_ctx* ctx, int64_t nv, int64_t nc, int64_t ne,
                                      const void* rowptr, int rowptr_bits, const uint32_t* colidx,
                                      int src_on_device, gaib_graph** out) {
  GAIB_CHECK(ctx && out && rowptr, "gaib_graph_create: NULL argument");
  GAIB_CHECK(nc >= 0 && nc < (int64_t)1 << 32, "gaib_graph_create_rect: need 0 <= nc < 2^32");
  GAIB_CHECK(nv >= 0 && ne >= 0, "gaib_graph_create: negative size");
  GAIB_CHECK(nv < (int64_t)1 << 31, "gaib_graph_create: nv must be < 2^31");
  GAIB_CHECK(ne < (int64_t)1 << 32, "gaib_graph_create: ne must be < 2^32 (edge ids are uint32)");
  GAIB_CHECK(rowptr_bits == 32 || rowptr_bits == 64, "gaib_graph_create: rowptr_bits must be 32 or 64");
  GAIB_CHECK(ne == 0 || colidx, "gaib_graph_create: colidx is NULL");
  GAIB_HIP(hipSetDevice(ctx->device));
  gaib_graph* g = nullptr;
  GAIB_TRY(new_graph(nv, ne, ctx->device, &g));
  g->nc = nc;
  hipMemcpyKind kind = src_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
  if (rowptr_bits == 64) {
    GAIB_HIP(hipMemcpyAsync(g->rowptr, rowptr, sizeof(int64_t) * (nv + 1), kind, ctx->stream));
  } else {
    uint32_t* tmp = nullptr;
    GAIB_HIP(hipMalloc(&tmp, sizeof(uint32_t) * (nv + 1)));
    GAIB_HIP(hipMemcpyAsync(tmp, rowptr, sizeof(uint32_t) * (nv + 1), kind, ctx->stream));
    widen_rowptr_kernel<<<grid1d(nv + 1, 256), 256, 0, ctx->stream>>>(nv + 1, tmp, g->rowptr);
    GAIB_LAUNCH_CHECK();
    GAIB_HIP(hipStreamSynchronize(ctx->stream));
    GAIB_HIP(hipFree(tmp));
  }
  if (ne > 0)
    GAIB_HIP(hipMemcpyAsync(g->colidx, colidx, sizeof(uint32_t) * ne, kind, ctx->stream));
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  // validate the two ends of rowptr (cheap; the kernels trust the rest)
  int64_t ends[2] = {0, 0};
  GAIB_HIP(hipMemcpy(&ends[0], g->rowptr, sizeof(int64_t), hipMemcpyDeviceToHost));
  GAIB_HIP(hipMemcpy(&ends[1], g->rowptr + nv, sizeof(int64_t), hipMemcpyDeviceToHost));
  if (ends[0] != 0 || ends[1] != ne) {
    gaib_set_error("gaib_graph_create: rowptr[0]=%lld rowptr[nv]=%lld but ne=%lld",
                   (long long)ends[0], (long long)ends[1], (long long)ne);
    gaib_graph_destroy(g);
    return GAIB_ERR_INVALID;
  }
  *out = g;
  return GAIB_OK;
}

extern "C" int gaib_graph_destroy(gaib_graph* g) {
  if (!g) return GAIB_OK;
  (void)hipSetDevice(g->device);
  void* ptrs[] = {g->rowptr, g->colidx,   g->vdata, g->edata,      g->inv_deg,  g->col_vdata,
                  g->col_inv_deg, g->w_gcn, g->w_mean_t, g->rev,   g->heavy_rows,
                  g->chunk_row, g->chunk_ebase, g->chunk_start, g->colidx_flagged, g->row_map};
  for (void* p : ptrs)
    if (p) (void)hipFree(p);
  delete g;
  return GAIB_OK;
}

extern "C" int64_t gaib_graph_nv(const gaib_graph* g) { return g ? g->nv : -1; }
extern "C" int64_t gaib_graph_ne(const gaib_graph* g) { return g ? g->ne : -1; }
extern "C" int64_t gaib_graph_nc(const gaib_graph* g) { return g ? g->nc : -1; }
extern "C" const int64_t* gaib_graph_rowptr(const gaib_graph* g) { return g ? g->rowptr : nullptr; }
extern "C" const uint32_t* gaib_graph_colidx(const gaib_graph* g) { return g ? g->colidx : nullptr; }
extern "C" const float* gaib_graph_vertex_data(const gaib_graph* g) { return g ? g->vdata : nullptr; }
extern "C" const float* gaib_graph_edge_data(const gaib_graph* g) { return g ? g->edata : nullptr; }
extern "C" int64_t gaib_graph_device_bytes(const gaib_graph* g) { return g ? g->dev_bytes : -1; }

extern "C" int gaib_graph_add_selfloop(gaib_ctx* ctx, const gaib_graph* g, gaib_graph** out) {
  GAIB_CHECK(ctx && g && out, "gaib_graph_add_selfloop: NULL argument");
  GAIB_CHECK(g->nc == g->nv, "gaib_graph_add_selfloop: square graphs only");
  GAIB_CHECK(g->ne + g->nv < (int64_t)1 << 32, "gaib_graph_add_selfloop: ne+nv must be < 2^32");
  GAIB_HIP(hipSetDevice(ctx->device));
  gaib_graph* n = nullptr;
  GAIB_TRY(new_graph(g->nv, g->ne + g->nv, ctx->device, &n));
  selfloop_kernel<<<grid1d(g->nv + 1, 4), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->colidx,
                                                                 n->rowptr, n->colidx);
  GAIB_LAUNCH_CHECK();
  *out = n;
  return GAIB_OK;
}

extern "C" int gaib_graph_compute_vertex_data(gaib_ctx* ctx, gaib_graph* g) {
  GAIB_CHECK(ctx && g, "gaib_graph_compute_vertex_data: NULL argument");
  GAIB_HIP(hipSetDevice(ctx->device));
  if (!g->vdata) {
    GAIB_HIP(hipMalloc(&g->vdata, sizeof(float) * (size_t)(g->nv > 0 ? g->nv : 1)));
    g->dev_bytes += sizeof(float) * g->nv;
  }
  vertex_data_kernel<<<grid1d(g->nv, 256), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->vdata);
  GAIB_LAUNCH_CHECK();
  // the cached per-edge GCN weights derive from vdata
  if (g->w_gcn) {
    edge_weight_kernel<1><<<grid1d(g->nv, 4), 256, 0, ctx->stream>>>(
        g->nv, g->rowptr, g->colidx, g->vdata, g->col_vdata ? g->col_vdata : g->vdata, g->w_gcn);
    GAIB_LAUNCH_CHECK();
  }
  return GAIB_OK;
}

extern "C" int gaib_graph_compute_edge_data(gaib_ctx* ctx, gaib_graph* g) {
  GAIB_CHECK(ctx && g, "gaib_graph_compute_edge_data: NULL argument");
  GAIB_HIP(hipSetDevice(ctx->device));
  if (!g->edata) {
    GAIB_HIP(hipMalloc(&g->edata, sizeof(float) * (size_t)(g->ne > 0 ? g->ne : 1)));
    g->dev_bytes += sizeof(float) * g->ne;
  }
  GAIB_CHECK(g->nc == g->nv, "gaib_graph_compute_edge_data: square graphs only");
  edge_weight_kernel<0><<<grid1d(g->nv, 4), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->colidx,
                                                                   nullptr, nullptr, g->edata);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

int gaib_graph_ensure_inv_deg(gaib_ctx* ctx, gaib_graph* g) {
  if (g->inv_deg) return GAIB_OK;
  GAIB_NOT_WHILE_CAPTURING(ctx, "building the graph's 1/degree table");
  GAIB_HIP(hipMalloc(&g->inv_deg, sizeof(float) * (size_t)(g->nv > 0 ? g->nv : 1)));
  g->dev_bytes += sizeof(float) * g->nv;
  inv_deg_kernel<<<grid1d(g->nv, 256), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->inv_deg);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

int gaib_graph_ensure_w_gcn(gaib_ctx* ctx, gaib_graph* g) {
  if (g->w_gcn) return GAIB_OK;
  GAIB_NOT_WHILE_CAPTURING(ctx, "building the graph's GCN edge weights");
  if (!g->vdata) GAIB_TRY(gaib_graph_compute_vertex_data(ctx, g));
  GAIB_HIP(hipMalloc(&g->w_gcn, sizeof(float) * (size_t)(g->ne > 0 ? g->ne : 1)));
  g->dev_bytes += sizeof(float) * g->ne;
  GAIB_CHECK(g->nc == g->nv || g->col_vdata,
             "rectangular graph: call gaib_graph_set_vertex_norm before GAIB_W_GCN");
  edge_weight_kernel<1><<<grid1d(g->nv, 4), 256, 0, ctx->stream>>>(
      g->nv, g->rowptr, g->colidx, g->vdata, g->col_vdata ? g->col_vdata : g->vdata, g->w_gcn);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

int gaib_graph_ensure_w_mean_t(gaib_ctx* ctx, gaib_graph* g) {
  if (g->w_mean_t) return GAIB_OK;
  GAIB_NOT_WHILE_CAPTURING(ctx, "building the graph's transpose-mean edge weights");
  GAIB_TRY(gaib_graph_ensure_inv_deg(ctx, g));
  GAIB_HIP(hipMalloc(&g->w_mean_t, sizeof(float) * (size_t)(g->ne > 0 ? g->ne : 1)));
  g->dev_bytes += sizeof(float) * g->ne;
  GAIB_CHECK(g->nc == g->nv || g->col_inv_deg,
             "rectangular graph: call gaib_graph_set_vertex_norm before GAIB_W_MEAN_T");
  gather_by_col_kernel<<<grid1d(g->ne, 256), 256, 0, ctx->stream>>>(
      g->ne, g->colidx, g->col_inv_deg ? g->col_inv_deg : g->inv_deg, g->w_mean_t);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

// The reverse-edge permutation as a sort-merge instead of one binary search per edge (the reference's way,
// math_functions.cpp:32-44,60-73, kept above as rev_kernel: 13.4 ms at the reddit shape, 112 M dependent probes).
// A stable sort of the edge ids by column id lists the edges in (column, row) order -- the CSC order.  For a
// structurally symmetric graph with sorted rows that is the CSR order of the reverse edges: position j of the sorted
// list holds the edge (row = r_j, col = c_j) whose reverse (c_j, r_j) is edge j itself.  So rev = the inverse of the
// sort's permutation; a second pass verifies col[rev[e]] == row(e) inside row col[e] for every edge (the symmetry
// check the binary search gave for free).  Radix sort over the significant bits of the column ids only.
__global__ void iota_u32_kernel(int64_t n, uint32_t* x) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (uint32_t)i;
}
__global__ void invert_perm_kernel(int64_t n, const uint32_t* sorted_ids, uint32_t* rev) {
  int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (j < n) rev[sorted_ids[j]] = (uint32_t)j;
}
__global__ __launch_bounds__(256) void rev_check_kernel(int64_t nv, const int64_t* rowptr, const uint32_t* col,
                                                        const uint32_t* rev, int* bad) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  for (int64_t e = e0 + lane; e < e1; e += 64) {
    const uint32_t c = col[e];
    const int64_t r = (int64_t)rev[e];
    if (r < rowptr[c] || r >= rowptr[c + 1] || col[r] != (uint32_t)row) *bad = 1;
  }
}

int gaib_graph_ensure_rev(gaib_ctx* ctx, gaib_graph* g) {
  if (g->rev) return GAIB_OK;
  GAIB_NOT_WHILE_CAPTURING(ctx, "building the graph's reverse-edge permutation");
  GAIB_CHECK(g->nc == g->nv, "reverse-edge permutation: square graphs only");
  if (g->rows_unsorted) {
    gaib_set_error("reverse-edge permutation: this graph's rows are not sorted by column id (gaib_graph_reorder keeps every "
                   "row's edge order): call gaib_graph_sort_rows first -- GAT backward / gaib_edge_transpose need sorted rows");
    return GAIB_ERR_UNSUPPORTED;
  }
  uint32_t* rev = nullptr;
  int* bad = nullptr;
  const int64_t ne = g->ne;
  GAIB_HIP(hipMalloc(&rev, sizeof(uint32_t) * (size_t)(ne > 0 ? ne : 1)));
  GAIB_HIP(hipMalloc(&bad, sizeof(int)));
  GAIB_HIP(hipMemsetAsync(bad, 0, sizeof(int), ctx->stream));
  // small graphs (and the knob graph_rev_search = 1): the per-edge binary search
  const bool by_search = ne < (1 << 16) || ctx->graph_rev_search == 1;
  if (by_search) {
    rev_kernel<<<grid1d(g->nv, 4), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->colidx, rev, bad);
    GAIB_LAUNCH_CHECK();
  } else {
    uint32_t *keys_out = nullptr, *ids = nullptr, *ids_out = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    int end_bit = 1;
    while (end_bit < 32 && ((int64_t)1 << end_bit) < g->nv) end_bit++;
    hipError_t e = hipMalloc(&keys_out, sizeof(uint32_t) * ne);
    if (e == hipSuccess) e = hipMalloc(&ids, sizeof(uint32_t) * ne);
    if (e == hipSuccess) e = hipMalloc(&ids_out, sizeof(uint32_t) * ne);
    if (e == hipSuccess)
      e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, g->colidx, keys_out, ids, ids_out, ne, 0, end_bit,
                                             ctx->stream);  // (64-bit item count: graphs of 2^31 .. 2^32 - 1 edges, round 6)
    if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes);
    if (e == hipSuccess) {
      iota_u32_kernel<<<grid1d(ne, 256), 256, 0, ctx->stream>>>(ne, ids);
      e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, g->colidx, keys_out, ids, ids_out, ne, 0, end_bit,
                                             ctx->stream);  // radix sort is stable: ties keep the row order
    }
    if (e == hipSuccess) {
      invert_perm_kernel<<<grid1d(ne, 256), 256, 0, ctx->stream>>>(ne, ids_out, rev);
      rev_check_kernel<<<grid1d(g->nv, 4), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->colidx, rev, bad);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (keys_out) (void)hipFree(keys_out);
    if (ids) (void)hipFree(ids);
    if (ids_out) (void)hipFree(ids_out);
    if (tmp) (void)hipFree(tmp);
    if (e != hipSuccess) {
      (void)hipFree(rev);
      (void)hipFree(bad);
      gaib_set_error("reverse-edge permutation (sort): %s", hipGetErrorString(e));
      return e == hipErrorOutOfMemory ? GAIB_ERR_NOMEM : GAIB_ERR_HIP;
    }
  }
  int hbad = 0;
  GAIB_HIP(hipMemcpyAsync(&hbad, bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  GAIB_HIP(hipFree(bad));
  if (hbad) {
    (void)hipFree(rev);
    gaib_set_error("graph is not structurally symmetric: a reverse edge is missing "
                   "(reference asserts, math_functions.cpp:70)");
    return GAIB_ERR_ASYMMETRIC;
  }
  g->rev = rev;
  g->dev_bytes += sizeof(uint32_t) * g->ne;
  return GAIB_OK;
}

// Does the vertex numbering carry locality?  Share of the edges of every 16th row whose column id lies within 32 768
// ids of the row id (8 XCD L2s of 4 MB hold 8 192 rows of 128 floats each).  A random numbering of N vertices gives
// 65 536 / N (2.7 % at the products size), a numbering with communities in consecutive ids most of the edges.  Square
// graphs only (a rectangular graph's columns index another table).  Measured once per graph (lazily), used by the
// fused kernel's tile supply (spmm.hip).
__global__ void locality_kernel(int64_t nv, const int64_t* rowptr, const uint32_t* col, unsigned long long* cnt) {
  const int64_t r = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 16;
  unsigned long long near = 0, all = 0;
  if (r < nv) {
    const int64_t e0 = rowptr[r], e1 = rowptr[r + 1];
    for (int64_t e = e0; e < e1; ++e) {
      const int64_t d = (int64_t)col[e] - r;
      near += (d < 0 ? -d : d) <= 32768 && d != 0;
    }
    all = (unsigned long long)(e1 - e0);
  }
  near = wave_sum_u32((unsigned)near);
  all = wave_sum_u32((unsigned)(all > 0xffffffu ? 0xffffffu : all));
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(cnt, near);
    atomicAdd(cnt + 1, all);
  }
}

int gaib_graph_ensure_locality(gaib_ctx* ctx, gaib_graph* g) {
  if (g->near_frac >= 0.f) return GAIB_OK;
  if (g->nc != g->nv || g->nv < 65536 * 4) {  // rectangular, or small enough for any numbering to be "near"
    g->near_frac = 0.f;
    return GAIB_OK;
  }
  GAIB_NOT_WHILE_CAPTURING(ctx, "measuring the locality of the graph's numbering");
  unsigned long long* cnt = nullptr;
  GAIB_HIP(hipMalloc(&cnt, 2 * sizeof(unsigned long long)));
  struct Release {
    void* p;
    ~Release() { (void)hipFree(p); }
  } release{cnt};
  GAIB_HIP(hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned long long), ctx->stream));
  locality_kernel<<<grid1d(cdiv64(g->nv, 16), 256), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->colidx, cnt);
  GAIB_LAUNCH_CHECK();
  unsigned long long h[2] = {0, 0};
  GAIB_HIP(hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  g->near_frac = h[1] ? (float)((double)h[0] / (double)h[1]) : 0.f;
  return GAIB_OK;
}

int gaib_graph_ensure_heavy(gaib_ctx* ctx, gaib_graph* g, int thr) {
  if (g->heavy_thr == thr) return GAIB_OK;
  GAIB_NOT_WHILE_CAPTURING(ctx, "building the graph's heavy-row list");
  unsigned long long* cnt = nullptr;
  GAIB_HIP(hipMalloc(&cnt, 3 * sizeof(unsigned long long)));
  struct Release {  // (every early return below goes through GAIB_HIP / GAIB_LAUNCH_CHECK)
    void* p;
    ~Release() { (void)hipFree(p); }
  } release{cnt};
  GAIB_HIP(hipMemsetAsync(cnt, 0, 3 * sizeof(unsigned long long), ctx->stream));
  heavy_rows_kernel<<<grid1d(g->nv, 256), 256, 0, ctx->stream>>>(g->nv, g->rowptr, thr, nullptr,
                                                                 cnt, cnt + 1);
  GAIB_LAUNCH_CHECK();
  unsigned long long h[3] = {0, 0, 0};
  GAIB_HIP(hipMemcpyAsync(h, cnt, sizeof(h), hipMemcpyDeviceToHost, ctx->stream));
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  if (g->heavy_rows) {
    GAIB_HIP(hipFree(g->heavy_rows));
    g->heavy_rows = nullptr;
  }
  g->n_heavy = (int64_t)h[0];
  g->max_degree = (int64_t)h[1];
  g->heavy_edges = (int64_t)h[2];
  if (g->n_heavy > 0) {
    // [0, n): row ids ascending; [n, 2n): launch order = slots by descending degree
    GAIB_HIP(hipMalloc(&g->heavy_rows, sizeof(uint32_t) * 2 * (size_t)g->n_heavy));
    GAIB_HIP(hipMemsetAsync(cnt, 0, sizeof(unsigned long long), ctx->stream));
    heavy_rows_kernel<<<grid1d(g->nv, 256), 256, 0, ctx->stream>>>(g->nv, g->rowptr, thr,
                                                                   g->heavy_rows, cnt, nullptr);
    GAIB_LAUNCH_CHECK();
    // the atomic append leaves the list in arrival order: sort it (ascending row id) so that it can
    // be searched (fused aggregation) and the launch order is the same on every run
    std::vector<uint32_t> rows((size_t)g->n_heavy);
    GAIB_HIP(hipMemcpyAsync(rows.data(), g->heavy_rows, sizeof(uint32_t) * rows.size(),
                            hipMemcpyDeviceToHost, ctx->stream));
    GAIB_HIP(hipStreamSynchronize(ctx->stream));
    std::sort(rows.begin(), rows.end());
    // longest rows first: the workgroup-per-row kernel then ends on its shortest rows
    std::vector<int64_t> rp((size_t)g->nv + 1);
    GAIB_HIP(hipMemcpyAsync(rp.data(), g->rowptr, sizeof(int64_t) * rp.size(), hipMemcpyDeviceToHost,
                            ctx->stream));
    GAIB_HIP(hipStreamSynchronize(ctx->stream));
    std::vector<uint32_t> order(rows.size());
    for (size_t k = 0; k < order.size(); ++k) order[k] = (uint32_t)k;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
      return rp[rows[x] + 1] - rp[rows[x]] > rp[rows[y] + 1] - rp[rows[y]];
    });
    rows.insert(rows.end(), order.begin(), order.end());
    GAIB_HIP(hipMemcpyAsync(g->heavy_rows, rows.data(), sizeof(uint32_t) * rows.size(),
                            hipMemcpyHostToDevice, ctx->stream));
    GAIB_HIP(hipStreamSynchronize(ctx->stream));
  }
  g->heavy_thr = thr;
  return GAIB_OK;
}

// gather mode 3: flag cold columns.  The hot set is the highest-degree vertices whose feature rows
// fit `spmm_hot_bytes` (default 3 MB of the 4 MB per-XCD L2); everything else is gathered with the
// streaming (nt) policy so it does not push the hub rows out.
__global__ void flag_cold_kernel(int64_t ne, const uint32_t* col, const int64_t* rowptr, int64_t thr,
                                 uint32_t* out) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= ne) return;
  const uint32_t c = col[e];
  const int64_t deg = rowptr[c + 1] - rowptr[c];
  out[e] = deg >= thr ? c : (c | 0x80000000u);
}

int gaib_graph_ensure_hot_flags(gaib_ctx* ctx, gaib_graph* g, int len) {
  int64_t hot_rows = (int64_t)ctx->spmm_hot_bytes / ((int64_t)len * 4);
  if (hot_rows < 1) hot_rows = 1;
  if (hot_rows > g->nv) hot_rows = g->nv;
  // threshold = degree of the hot_rows-th highest-degree vertex
  if (g->colidx_flagged && g->hot_rows == hot_rows) return GAIB_OK;
  GAIB_NOT_WHILE_CAPTURING(ctx, "building the graph's hot-column flags");
  std::vector<int64_t> rp((size_t)g->nv + 1);
  GAIB_HIP(hipMemcpyAsync(rp.data(), g->rowptr, sizeof(int64_t) * (g->nv + 1), hipMemcpyDeviceToHost, ctx->stream));
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  std::vector<int64_t> deg((size_t)g->nv);
  for (int64_t v = 0; v < g->nv; ++v) deg[v] = rp[v + 1] - rp[v];
  std::nth_element(deg.begin(), deg.begin() + (hot_rows - 1), deg.end(), std::greater<int64_t>());
  const int64_t thr = deg[hot_rows - 1];
  if (!g->colidx_flagged) {
    GAIB_HIP(hipMalloc(&g->colidx_flagged, sizeof(uint32_t) * (size_t)(g->ne > 0 ? g->ne : 1)));
    g->dev_bytes += sizeof(uint32_t) * g->ne;
  }
  if (g->ne > 0) {
    flag_cold_kernel<<<grid1d(g->ne, 256), 256, 0, ctx->stream>>>(g->ne, g->colidx, g->rowptr, thr, g->colidx_flagged);
    GAIB_LAUNCH_CHECK();
  }
  g->hot_threshold = thr;
  g->hot_rows = hot_rows;
  return GAIB_OK;
}

// 64-edge chunk list: chunk c covers edges [chunk_ebase[c], min(+64, row end)) of row chunk_row[c]; chunk_start[v] =
// number of chunks in the rows before v.  Built once per graph, on the device (a host build with a comparison sort
// took 4 s at 115 M edges): scan of ceil(deg/64), one wave per row fills its chunks, and -- gat_chunk_sort -- a stable
// radix sort by the column block of each chunk's first edge.  Chunks are independent in the kernels that walk the
// list, so their order is free; sorted, the chunks in flight gather feature rows from one window of columns (rows are
// sorted by column, so a chunk of a long row covers a compact column interval).
namespace {
__global__ void chunk_count_kernel(int64_t nv, const int64_t* rowptr, uint32_t* cnt) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < nv) cnt[v] = (uint32_t)((rowptr[v + 1] - rowptr[v] + 63) / 64);
  else if (v == nv) cnt[v] = 0u;
}
__global__ __launch_bounds__(256) void chunk_fill_kernel(int64_t nv, const int64_t* rowptr, const uint32_t* colidx,
                                                         const uint32_t* chunk_start, uint32_t* crow, uint32_t* cbase,
                                                         uint32_t* key, uint32_t* ident) {
  const int64_t v = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (v >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[v];
  const uint32_t c0 = chunk_start[v], c1 = chunk_start[v + 1];
  for (uint32_t c = c0 + lane; c < c1; c += 64) {
    const int64_t e = e0 + (int64_t)(c - c0) * 64;
    crow[c] = (uint32_t)v;
    cbase[c] = (uint32_t)e;
    if (key) {
      key[c] = colidx[e] >> 10;
      ident[c] = c;
    }
  }
}
__global__ void chunk_permute_kernel(int64_t n, const uint32_t* order, const uint32_t* crow, const uint32_t* cbase,
                                     uint32_t* crow2, uint32_t* cbase2) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    crow2[i] = crow[order[i]];
    cbase2[i] = cbase[order[i]];
  }
}
}  // namespace

int gaib_graph_ensure_chunks(gaib_ctx* ctx, gaib_graph* g) {
  if (g->chunk_row) return GAIB_OK;
  GAIB_NOT_WHILE_CAPTURING(ctx, "building the graph's edge-chunk list");
  GAIB_HIP(hipSetDevice(ctx->device));
  hipStream_t st = ctx->stream;
  const int64_t nv = g->nv;
  uint32_t* cnt = nullptr;
  GAIB_HIP(hipMalloc(&cnt, sizeof(uint32_t) * (size_t)(nv + 1)));
  GAIB_HIP(hipMalloc(&g->chunk_start, sizeof(uint32_t) * (size_t)(nv + 1)));
  chunk_count_kernel<<<(unsigned)cdiv64(nv + 1, 256), 256, 0, st>>>(nv, g->rowptr, cnt);
  GAIB_LAUNCH_CHECK();
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  GAIB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt, g->chunk_start, (int)(nv + 1), st));
  GAIB_HIP(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 1));
  GAIB_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, cnt, g->chunk_start, (int)(nv + 1), st));
  uint32_t nch32 = 0;
  GAIB_HIP(hipMemcpyAsync(&nch32, g->chunk_start + nv, sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  GAIB_HIP(hipStreamSynchronize(st));
  GAIB_HIP(hipFree(tmp));
  GAIB_HIP(hipFree(cnt));
  const int64_t nch = nch32;
  const size_t nalloc = (size_t)(nch > 0 ? nch : 1);
  GAIB_HIP(hipMalloc(&g->chunk_row, sizeof(uint32_t) * nalloc));
  GAIB_HIP(hipMalloc(&g->chunk_ebase, sizeof(uint32_t) * nalloc));
  const bool sorted = ctx->gat_chunk_sort && nch > 1;
  uint32_t *key = nullptr, *ident = nullptr, *key2 = nullptr, *order = nullptr;
  if (sorted) {
    GAIB_HIP(hipMalloc(&key, sizeof(uint32_t) * nalloc));
    GAIB_HIP(hipMalloc(&ident, sizeof(uint32_t) * nalloc));
    GAIB_HIP(hipMalloc(&key2, sizeof(uint32_t) * nalloc));
    GAIB_HIP(hipMalloc(&order, sizeof(uint32_t) * nalloc));
  }
  if (nv > 0) {
    chunk_fill_kernel<<<(unsigned)cdiv64(nv, 4), 256, 0, st>>>(nv, g->rowptr, g->colidx, g->chunk_start, g->chunk_row,
                                                             g->chunk_ebase, key, ident);
    GAIB_LAUNCH_CHECK();
  }
  if (sorted) {
    tmp = nullptr;
    tmp_bytes = 0;
    GAIB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key, key2, ident, order, (int)nch, 0, 32, st));
    GAIB_HIP(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 1));
    GAIB_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key, key2, ident, order, (int)nch, 0, 32, st));  // stable
    uint32_t *crow2 = nullptr, *cbase2 = nullptr;
    GAIB_HIP(hipMalloc(&crow2, sizeof(uint32_t) * nalloc));
    GAIB_HIP(hipMalloc(&cbase2, sizeof(uint32_t) * nalloc));
    chunk_permute_kernel<<<(unsigned)cdiv64(nch, 256), 256, 0, st>>>(nch, order, g->chunk_row, g->chunk_ebase, crow2,
                                                                    cbase2);
    GAIB_LAUNCH_CHECK();
    GAIB_HIP(hipStreamSynchronize(st));
    void* old[] = {g->chunk_row, g->chunk_ebase, key, ident, key2, order, tmp};
    for (void* q : old) GAIB_HIP(hipFree(q));
    g->chunk_row = crow2;
    g->chunk_ebase = cbase2;
  }
  g->n_chunks = nch;
  g->dev_bytes += 2 * sizeof(uint32_t) * nch + sizeof(uint32_t) * (nv + 1);
  return GAIB_OK;
}

// ---- opt-in relabelling of a graph -----------------------------------------------------
// A numbering with locality is worth 1.2-1.6x to the aggregation (the gathered rows of concurrently processed rows meet
// in an XCD's L2).  gaib_graph_reorder builds the SAME graph under a new numbering computed on the device from the graph
// alone; every row keeps the ORDER of its edges, so a row's fp32 sum is the same sequence of additions and the
// aggregation's outputs are bit-identical once un-permuted (scripts/locality_study.py asserts it).
//   GAIB_ORDER_DEGREE     hubs first: vertices by descending degree (stable in the old id)
//   GAIB_ORDER_BFS        breadth-first levels from the highest-degree vertex, inside a level by old id; vertices the
//                         search does not reach keep their relative order at the end
//   GAIB_ORDER_CM         the same levels, inside a level by the position of the first parent (Cuthill-McKee): children of
//                         one vertex, and of neighbouring vertices, become neighbours -- what meshes / road-like graphs want
// The caller permutes feature rows with gaib_gather_rows(old_of_new) and un-permutes outputs with
// gaib_gather_rows(new_of_old).
__global__ void degree_key_kernel(int64_t nv, const int64_t* rowptr, uint32_t maxdeg, uint32_t* key) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < nv) key[v] = maxdeg - (uint32_t)(rowptr[v + 1] - rowptr[v]);
}
__global__ void bfs_level_kernel(int64_t nv, const int64_t* rowptr, const uint32_t* col, uint32_t cur, uint32_t* level,
                                 unsigned* changed) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv || level[v] != cur) return;
  bool any = false;
  for (int64_t e = rowptr[v]; e < rowptr[v + 1]; ++e) {
    const uint32_t c = col[e];
    if (level[c] == 0xffffffffu) {  // every writer writes the same value: the level of a vertex is its distance
      level[c] = cur + 1;
      any = true;
    }
  }
  if (any) *changed = 1u;
}
__global__ void cap_level_kernel(int64_t nv, uint32_t cap, uint32_t* level) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < nv && level[v] == 0xffffffffu) level[v] = cap;
}
__global__ void argmax_degree_kernel(int64_t nv, const int64_t* rowptr, unsigned long long* best) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= nv) return;
  // (degree, smallest id) as one 64-bit key: max over degree, ties to the smaller id
  const unsigned long long k = ((unsigned long long)(rowptr[v + 1] - rowptr[v]) << 32) | (0xffffffffu - (uint32_t)v);
  atomicMax(best, k);
}
// Cuthill-McKee inside the breadth-first levels (GAIB_ORDER_CM): a vertex of level L sorts by the POSITION of its first parent
// (the smallest new position among its neighbours of level L - 1), so the children of one vertex -- and of neighbouring
// vertices -- become neighbours in the numbering.  One wave per vertex of the level's segment.
__global__ __launch_bounds__(256) void cm_key_kernel(int64_t seg_begin, int64_t seg_count, const uint32_t* order,
                                                     const int64_t* rowptr, const uint32_t* col, const uint32_t* level,
                                                     uint32_t L, const uint32_t* pos, uint32_t* key) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= seg_count) return;
  const int lane = threadIdx.x & 63;
  const uint32_t v = order[seg_begin + i];
  uint32_t best = 0xffffffffu;
  for (int64_t e = rowptr[v] + lane; e < rowptr[v + 1]; e += 64) {
    const uint32_t c = col[e];
    if (level[c] + 1 == L) best = min(best, pos[c]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) best = min(best, (uint32_t)__shfl_xor((int)best, o, 64));
  if (lane == 0) key[i] = best;
}
__global__ void set_pos_kernel(int64_t seg_begin, int64_t seg_count, const uint32_t* order, uint32_t* pos) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < seg_count) pos[order[seg_begin + i]] = (uint32_t)(seg_begin + i);
}
__global__ void level_hist_kernel(int64_t nv, const uint32_t* level, uint32_t nlevels, unsigned* hist) {
  const int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (v < nv) atomicAdd(&hist[min(level[v], nlevels - 1)], 1u);
}
__global__ void invert_order_kernel(int64_t nv, const uint32_t* old_of_new, int64_t* new_of_old, int64_t* old_of_new64,
                                    const int64_t* rowptr, int64_t* deg_new) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nv) return;
  const uint32_t o = old_of_new[k];
  new_of_old[o] = k;
  if (old_of_new64) old_of_new64[k] = o;
  deg_new[k] = rowptr[o + 1] - rowptr[o];
}
__global__ __launch_bounds__(256) void relabel_rows_kernel(int64_t nv, const int64_t* rowptr_old, const uint32_t* col_old,
                                                           const uint32_t* old_of_new, const int64_t* new_of_old,
                                                           const int64_t* rowptr_new, uint32_t* col_new) {
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);  // one wave per new row
  if (k >= nv) return;
  const int lane = threadIdx.x & 63;
  const uint32_t o = old_of_new[k];
  const int64_t s = rowptr_old[o], n = rowptr_old[o + 1] - s, d = rowptr_new[k];
  for (int64_t j = lane; j < n; j += 64) col_new[d + j] = (uint32_t)new_of_old[col_old[s + j]];  // edge j stays edge j
}

extern "C" int gaib_graph_reorder(gaib_ctx* ctx, gaib_graph* g, int method, gaib_graph** out, int64_t* d_new_of_old,
                                  int64_t* d_old_of_new) {
  GAIB_CHECK(ctx && g && out && d_new_of_old, "gaib_graph_reorder: NULL argument");
  GAIB_CHECK(g->nc == g->nv, "gaib_graph_reorder: square graphs only");
  GAIB_CHECK(method == GAIB_ORDER_DEGREE || method == GAIB_ORDER_BFS || method == GAIB_ORDER_CM,
             "gaib_graph_reorder: unknown method %d", method);
  GAIB_CHECK(g->nv < ((int64_t)1 << 31), "gaib_graph_reorder: more than 2^31 vertices");
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_graph_reorder");
  GAIB_HIP(hipSetDevice(ctx->device));
  *out = nullptr;
  const int64_t nv = g->nv, ne = g->ne;
  hipStream_t st = ctx->stream;
  struct Bufs {  // freed on every return path
    std::vector<void*> p;
    ~Bufs() {
      for (void* q : p) (void)hipFree(q);
    }
    int get(void** q, size_t bytes) {
      hipError_t e = hipMalloc(q, bytes ? bytes : 16);
      if (e != hipSuccess) {
        gaib_set_error("gaib_graph_reorder: hipMalloc(%zu): %s", bytes, hipGetErrorString(e));
        return GAIB_ERR_NOMEM;
      }
      p.push_back(*q);
      return GAIB_OK;
    }
  } bufs;
  uint32_t *key = nullptr, *key2 = nullptr, *ids = nullptr, *order = nullptr;
  int64_t* deg_new = nullptr;
  unsigned long long* scalar = nullptr;
  GAIB_TRY(bufs.get((void**)&key, sizeof(uint32_t) * nv));
  GAIB_TRY(bufs.get((void**)&key2, sizeof(uint32_t) * nv));
  GAIB_TRY(bufs.get((void**)&ids, sizeof(uint32_t) * nv));
  GAIB_TRY(bufs.get((void**)&order, sizeof(uint32_t) * nv));
  GAIB_TRY(bufs.get((void**)&deg_new, sizeof(int64_t) * (nv + 1)));
  GAIB_TRY(bufs.get((void**)&scalar, sizeof(unsigned long long) * 2));
  GAIB_HIP(hipMemsetAsync(scalar, 0, sizeof(unsigned long long) * 2, st));
  argmax_degree_kernel<<<grid1d(nv, 256), 256, 0, st>>>(nv, g->rowptr, scalar);
  GAIB_LAUNCH_CHECK();
  unsigned long long best = 0;
  GAIB_HIP(hipMemcpyAsync(&best, scalar, sizeof(best), hipMemcpyDeviceToHost, st));
  GAIB_HIP(hipStreamSynchronize(st));
  const uint32_t maxdeg = (uint32_t)(best >> 32), hub = 0xffffffffu - (uint32_t)(best & 0xffffffffu);
  int key_bits = 32;
  uint32_t cm_levels = 0;
  if (method == GAIB_ORDER_DEGREE) {
    degree_key_kernel<<<grid1d(nv, 256), 256, 0, st>>>(nv, g->rowptr, maxdeg, key);
    GAIB_LAUNCH_CHECK();
    key_bits = 1;
    while (key_bits < 32 && (maxdeg >> key_bits)) ++key_bits;
  } else {
    GAIB_HIP(hipMemsetAsync(key, 0xff, sizeof(uint32_t) * nv, st));
    const uint32_t zero = 0;
    if (nv > 0) GAIB_HIP(hipMemcpyAsync(key + hub, &zero, sizeof(zero), hipMemcpyHostToDevice, st));
    uint32_t cur = 0;
    for (;; ++cur) {  // level-synchronous: one launch and one 4-byte read-back per level (~10 on a power-law graph)
      GAIB_HIP(hipMemsetAsync(scalar + 1, 0, sizeof(unsigned), st));
      bfs_level_kernel<<<grid1d(nv, 256), 256, 0, st>>>(nv, g->rowptr, g->colidx, cur, key, (unsigned*)(scalar + 1));
      GAIB_LAUNCH_CHECK();
      unsigned changed = 0;
      GAIB_HIP(hipMemcpyAsync(&changed, scalar + 1, sizeof(changed), hipMemcpyDeviceToHost, st));
      GAIB_HIP(hipStreamSynchronize(st));
      if (!changed || cur > (1u << 20)) break;
    }
    cap_level_kernel<<<grid1d(nv, 256), 256, 0, st>>>(nv, cur + 1, key);  // unreached vertices: after the last level
    GAIB_LAUNCH_CHECK();
    cm_levels = cur;
    key_bits = 1;
    while (key_bits < 32 && ((cur + 1) >> key_bits)) ++key_bits;
  }
  iota_u32_kernel<<<grid1d(nv, 256), 256, 0, st>>>(nv, ids);
  GAIB_LAUNCH_CHECK();
  size_t tmp_bytes = 0;
  GAIB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, key, key2, ids, order, (int)nv, 0, key_bits, st));
  void* tmp = nullptr;
  GAIB_TRY(bufs.get(&tmp, tmp_bytes));
  GAIB_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, key, key2, ids, order, (int)nv, 0, key_bits, st));  // stable
  if (method == GAIB_ORDER_CM && nv > 1) {
    // level by level, in order: the positions of level L - 1 are final when level L is keyed.  `order` holds the vertices by
    // (level, old id): a level is a contiguous segment of it; a segment is re-sorted by its keys (stable: ties keep the old id
    // order) and its positions are published before the next level reads them.  The unreached vertices (last segment) stay.
    const uint32_t nlevels = cm_levels + 2;  // levels 0 .. cm_levels, + the capped "unreached" level
    unsigned* hist = nullptr;
    uint32_t *pos = nullptr, *seg_key = nullptr, *seg_key2 = nullptr, *seg_ord2 = nullptr;
    GAIB_TRY(bufs.get((void**)&hist, sizeof(unsigned) * nlevels));
    GAIB_TRY(bufs.get((void**)&pos, sizeof(uint32_t) * nv));
    GAIB_TRY(bufs.get((void**)&seg_key, sizeof(uint32_t) * nv));
    GAIB_TRY(bufs.get((void**)&seg_key2, sizeof(uint32_t) * nv));
    GAIB_TRY(bufs.get((void**)&seg_ord2, sizeof(uint32_t) * nv));
    GAIB_HIP(hipMemsetAsync(hist, 0, sizeof(unsigned) * nlevels, st));
    level_hist_kernel<<<grid1d(nv, 256), 256, 0, st>>>(nv, key, nlevels, hist);
    GAIB_LAUNCH_CHECK();
    std::vector<unsigned> h_hist(nlevels);
    GAIB_HIP(hipMemcpyAsync(h_hist.data(), hist, sizeof(unsigned) * nlevels, hipMemcpyDeviceToHost, st));
    GAIB_HIP(hipStreamSynchronize(st));
    set_pos_kernel<<<grid1d(nv, 256), 256, 0, st>>>(0, nv, order, pos);  // the (level, id) order to start from
    GAIB_LAUNCH_CHECK();
    int64_t seg_begin = h_hist[0];
    for (uint32_t L = 1; L <= cm_levels; ++L) {
      const int64_t cnt = h_hist[L];
      if (cnt > 1) {
        cm_key_kernel<<<grid1d(cnt, 4), 256, 0, st>>>(seg_begin, cnt, order, g->rowptr, g->colidx, key, L, pos, seg_key);
        GAIB_LAUNCH_CHECK();
        size_t need = 0;
        GAIB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, need, seg_key, seg_key2, order + seg_begin, seg_ord2, (int)cnt, 0, 32, st));
        GAIB_CHECK(need <= tmp_bytes, "gaib_graph_reorder: sort workspace");
        GAIB_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, need, seg_key, seg_key2, order + seg_begin, seg_ord2, (int)cnt, 0, 32, st));
        GAIB_HIP(hipMemcpyAsync(order + seg_begin, seg_ord2, sizeof(uint32_t) * cnt, hipMemcpyDeviceToDevice, st));
        set_pos_kernel<<<grid1d(cnt, 256), 256, 0, st>>>(seg_begin, cnt, order, pos);
        GAIB_LAUNCH_CHECK();
      }
      seg_begin += cnt;
    }
  }
  invert_order_kernel<<<grid1d(nv, 256), 256, 0, st>>>(nv, order, d_new_of_old, d_old_of_new, g->rowptr, deg_new);
  GAIB_LAUNCH_CHECK();
  gaib_graph* r = nullptr;
  GAIB_TRY(new_graph(nv, ne, ctx->device, &r));
  size_t scan_bytes = 0;
  hipError_t e = hipcub::DeviceScan::ExclusiveSum(nullptr, scan_bytes, deg_new, r->rowptr, (int)(nv + 1), st);
  void* stmp = nullptr;
  int rc = e == hipSuccess ? bufs.get(&stmp, scan_bytes) : GAIB_ERR_HIP;
  if (rc == GAIB_OK) {
    e = hipcub::DeviceScan::ExclusiveSum(stmp, scan_bytes, deg_new, r->rowptr, (int)(nv + 1), st);
    if (e == hipSuccess) {
      relabel_rows_kernel<<<grid1d(nv, 4), 256, 0, st>>>(nv, g->rowptr, g->colidx, order, d_new_of_old, r->rowptr, r->colidx);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) rc = GAIB_ERR_HIP;
  }
  if (rc != GAIB_OK) {
    if (e != hipSuccess) gaib_set_error("gaib_graph_reorder: %s", hipGetErrorString(e));
    (void)gaib_graph_destroy(r);
    return rc;
  }
  rc = gaib_graph_compute_vertex_data(ctx, r);  // degrees travel with the vertices: the same normalisers, renamed
  if (rc != GAIB_OK) {
    (void)gaib_graph_destroy(r);
    return rc;
  }
  r->rows_unsorted = 1;  // (each row kept its edge order under new names)
  *out = r;
  return GAIB_OK;
}

namespace {
__global__ __launch_bounds__(256) void row_col_key_kernel(int64_t nv, const int64_t* rowptr, const uint32_t* col, uint64_t* key) {
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nv) return;
  for (int64_t e = rowptr[r] + (threadIdx.x & 63); e < rowptr[r + 1]; e += 64) key[e] = ((uint64_t)r << 32) | col[e];
}
__global__ void key_low_kernel(int64_t ne, const uint64_t* key, uint32_t* col) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < ne) col[e] = (uint32_t)(key[e] & 0xffffffffull);
}
}  // namespace

// Sort every row's column ids ascending (one radix sort of (row, column) keys).  For graphs that gaib_graph_reorder
// relabelled and that GAT backward / gaib_edge_transpose are to run on: those derive the reverse-edge permutation from
// sorted rows (math_functions.cpp:32-44 searches them).  Aggregations over the sorted graph sum a row's terms in another
// order than the original numbering did: equal up to fp32 rounding, no longer bit for bit.  Per-edge caches are dropped.
extern "C" int gaib_graph_sort_rows(gaib_ctx* ctx, gaib_graph* g) {
  GAIB_CHECK(ctx && g, "gaib_graph_sort_rows: NULL argument");
  GAIB_CHECK(!g->row_map, "gaib_graph_sort_rows: not on a class graph");
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_graph_sort_rows");
  GAIB_HIP(hipSetDevice(ctx->device));
  const int64_t ne = g->ne, nv = g->nv;
  if (ne > 0) {
    uint64_t *key = nullptr, *key2 = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    int end_bit = 33;
    while (end_bit < 64 && ((int64_t)1 << (end_bit - 32)) < nv) end_bit++;
    hipError_t e = hipMalloc(&key, sizeof(uint64_t) * ne);
    if (e == hipSuccess) e = hipMalloc(&key2, sizeof(uint64_t) * ne);
    if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortKeys(nullptr, tmp_bytes, key, key2, ne, 0, end_bit, ctx->stream);  // (64-bit count)
    if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes);
    if (e == hipSuccess) {
      row_col_key_kernel<<<grid1d(nv, 4), 256, 0, ctx->stream>>>(nv, g->rowptr, g->colidx, key);
      e = hipcub::DeviceRadixSort::SortKeys(tmp, tmp_bytes, key, key2, ne, 0, end_bit, ctx->stream);
    }
    if (e == hipSuccess) {
      key_low_kernel<<<grid1d(ne, 256), 256, 0, ctx->stream>>>(ne, key2, g->colidx);
      e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    void* owned[] = {key, key2, tmp};
    for (void* p : owned)
      if (p) (void)hipFree(p);
    if (e != hipSuccess) {
      gaib_set_error("gaib_graph_sort_rows: %s", hipGetErrorString(e));
      return e == hipErrorOutOfMemory ? GAIB_ERR_NOMEM : GAIB_ERR_HIP;
    }
  }
  // everything laid out per edge follows the old order
  void** per_edge[] = {(void**)&g->edata, (void**)&g->w_gcn, (void**)&g->w_mean_t, (void**)&g->rev, (void**)&g->chunk_row,
                       (void**)&g->chunk_ebase, (void**)&g->chunk_start, (void**)&g->colidx_flagged};
  for (void** p : per_edge) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  g->n_chunks = 0;
  g->hot_threshold = -1;
  g->near_frac = -1.f;
  g->rows_unsorted = 0;
  return GAIB_OK;
}

extern "C" int gaib_graph_locality(gaib_ctx* ctx, gaib_graph* g, float* h_near_frac) {
  GAIB_CHECK(ctx && g && h_near_frac, "gaib_graph_locality: NULL argument");
  GAIB_HIP(hipSetDevice(ctx->device));
  GAIB_TRY(gaib_graph_ensure_locality(ctx, g));
  *h_near_frac = g->near_frac;
  return GAIB_OK;
}

extern "C" int gaib_graph_stats(gaib_ctx* ctx, gaib_graph* g, int64_t* h_n_heavy, int64_t* h_heavy_edges,
                                int64_t* h_max_degree) {
  GAIB_CHECK(ctx && g, "gaib_graph_stats: NULL argument");
  GAIB_HIP(hipSetDevice(ctx->device));
  GAIB_TRY(gaib_graph_ensure_heavy(ctx, g, ctx->spmm_heavy_threshold));
  if (h_n_heavy) *h_n_heavy = g->n_heavy;
  if (h_heavy_edges) *h_heavy_edges = g->heavy_edges;
  if (h_max_degree) *h_max_degree = g->max_degree;
  return GAIB_OK;
}


// ---- row classes of a vertex-range partition --------------------------------------------------------------------------
// The reference partitioner's local graph marks the owned (master) rows and appends the halo vertices behind them
// (src/partitioner/graph_partition.cc:70-80, include/graph_partition.h:21-22,36-37).  An owned row whose edges all stay
// inside the range -- an INTERIOR row -- needs nothing from the exchange: its aggregation (and the dense product riding on
// it) is complete in one pass while the halo rows travel.  Only BOUNDARY rows wait for them.  From the rank's owned-column
// graph and halo-column graph (same rows) this builds the class graphs, each COMPACT (row k = the k-th row of the class, in
// ascending order) with the map back to the rank's rows and the normalisers of its rows and columns:
//   interior : the rows without halo-column edges                         columns: owned
//   bnd_own  : the boundary rows' owned-column edges                      columns: owned
//   bnd_halo : the boundary rows' halo-column edges                       columns: halo
//   bnd_full : both, per row [owned-column edges..., halo-column edges...]  columns: [owned | halo] (halo id + n_own)
// Edge order inside a row is kept, so the sums of a class graph are the sums of the graphs it was cut from.
namespace {

__global__ void class_flag_kernel(int64_t nv, const int64_t* rp_halo, int all_boundary, uint32_t* is_bnd) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nv) is_bnd[i] = (all_boundary || rp_halo[i + 1] > rp_halo[i]) ? 1u : 0u;
}
// pos_b = exclusive scan of is_bnd: boundary row i is row pos_b[i] of its class, interior row i is row i - pos_b[i]
__global__ void class_map_kernel(int64_t nv, const uint32_t* is_bnd, const uint32_t* pos_b, uint32_t* map_b, uint32_t* map_i) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nv) return;
  if (is_bnd[i]) map_b[pos_b[i]] = (uint32_t)i;
  else map_i[i - pos_b[i]] = (uint32_t)i;
}
// deg[k] = edges of class row k in source a (+ source b); deg[n] = 0 closes the exclusive scan into a rowptr
__global__ void class_deg_kernel(int64_t n, const uint32_t* map, const int64_t* rp_a, const int64_t* rp_b, int64_t* deg) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k > n) return;
  if (k == n) {
    deg[k] = 0;
    return;
  }
  const int64_t r = map[k];
  int64_t d = rp_a[r + 1] - rp_a[r];
  if (rp_b) d += rp_b[r + 1] - rp_b[r];
  deg[k] = d;
}
// one wave per class row: its edges from source a, then from source b with the column ids shifted by `shift_b`
__global__ __launch_bounds__(256) void class_copy_kernel(int64_t n, const uint32_t* map, const int64_t* rp_a, const uint32_t* col_a,
                                                         const int64_t* rp_b, const uint32_t* col_b, uint32_t shift_b,
                                                         const int64_t* rp_out, uint32_t* col_out) {
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n) return;
  const int lane = threadIdx.x & 63;
  const int64_t r = map[k];
  int64_t o = rp_out[k];
  const int64_t a0 = rp_a[r], a1 = rp_a[r + 1];
  for (int64_t e = a0 + lane; e < a1; e += 64) col_out[o + (e - a0)] = col_a[e];
  if (rp_b) {
    o += a1 - a0;
    const int64_t b0 = rp_b[r], b1 = rp_b[r + 1];
    for (int64_t e = b0 + lane; e < b1; e += 64) col_out[o + (e - b0)] = col_b[e] + shift_b;
  }
}
__global__ void gather_f32_kernel(int64_t n, const uint32_t* idx, const float* in, float* out) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) out[k] = in[idx[k]];
}

// one class graph: rows map[0..n) of source a (+ b), nc columns
int build_class(gaib_ctx* ctx, int64_t n, const uint32_t* map, int64_t n_rows_all, const gaib_graph* a, const gaib_graph* b,
                uint32_t shift_b, int64_t nc, gaib_graph** out) {
  gaib_graph* g = nullptr;
  // rowptr first (the edge count comes out of the scan)
  int64_t* deg = nullptr;
  GAIB_HIP(hipMalloc(&deg, sizeof(int64_t) * (size_t)(n + 1)));
  struct Release {
    void* p;
    ~Release() { (void)hipFree(p); }
  } rel_deg{deg};
  class_deg_kernel<<<grid1d(n + 1, 256), 256, 0, ctx->stream>>>(n, map, a->rowptr, b ? b->rowptr : nullptr, deg);
  GAIB_LAUNCH_CHECK();
  int64_t* rp = nullptr;
  GAIB_HIP(hipMalloc(&rp, sizeof(int64_t) * (size_t)(n + 1)));
  Release rel_rp{rp};
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  GAIB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, deg, rp, (int)(n + 1), ctx->stream));
  GAIB_HIP(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 8));
  Release rel_tmp{tmp};
  GAIB_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, deg, rp, (int)(n + 1), ctx->stream));
  int64_t ne = 0;
  GAIB_HIP(hipMemcpyAsync(&ne, rp + n, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  GAIB_CHECK(ne < (int64_t)1 << 32, "gaib_graph_split_classes: a class with %lld edges (edge ids are uint32)", (long long)ne);
  GAIB_TRY(new_graph(n, ne, ctx->device, &g));
  struct Guard {  // (every early return below goes through GAIB_HIP / GAIB_LAUNCH_CHECK: the half-built graph goes with it)
    gaib_graph* g;
    ~Guard() {
      if (g) (void)gaib_graph_destroy(g);
    }
  } guard{g};
  g->nc = nc;
  GAIB_HIP(hipMemcpyAsync(g->rowptr, rp, sizeof(int64_t) * (size_t)(n + 1), hipMemcpyDeviceToDevice, ctx->stream));
  if (n > 0) {
    class_copy_kernel<<<grid1d(n, 4), 256, 0, ctx->stream>>>(n, map, a->rowptr, a->colidx, b ? b->rowptr : nullptr,
                                                             b ? b->colidx : nullptr, shift_b, g->rowptr, g->colidx);
    GAIB_LAUNCH_CHECK();
  }
  GAIB_HIP(hipMalloc(&g->row_map, sizeof(uint32_t) * (size_t)(n > 0 ? n : 1)));
  if (n > 0) GAIB_HIP(hipMemcpyAsync(g->row_map, map, sizeof(uint32_t) * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
  g->n_out_rows = n_rows_all;
  g->dev_bytes += sizeof(uint32_t) * n;
  // normalisers: the rows' through the map; the columns' from the source graphs (a: owned space, b: halo space)
  const size_t n1 = (size_t)(n > 0 ? n : 1), nc1 = (size_t)(nc > 0 ? nc : 1);
  GAIB_HIP(hipMalloc(&g->vdata, sizeof(float) * n1));
  GAIB_HIP(hipMalloc(&g->inv_deg, sizeof(float) * n1));
  GAIB_HIP(hipMalloc(&g->col_vdata, sizeof(float) * nc1));
  GAIB_HIP(hipMalloc(&g->col_inv_deg, sizeof(float) * nc1));
  g->dev_bytes += sizeof(float) * (2 * n + 2 * nc);
  if (n > 0) {
    gather_f32_kernel<<<grid1d(n, 256), 256, 0, ctx->stream>>>(n, map, a->vdata, g->vdata);
    gather_f32_kernel<<<grid1d(n, 256), 256, 0, ctx->stream>>>(n, map, a->inv_deg, g->inv_deg);
    GAIB_LAUNCH_CHECK();
  }
  GAIB_HIP(hipMemcpyAsync(g->col_vdata, a->col_vdata, sizeof(float) * a->nc, hipMemcpyDeviceToDevice, ctx->stream));
  GAIB_HIP(hipMemcpyAsync(g->col_inv_deg, a->col_inv_deg, sizeof(float) * a->nc, hipMemcpyDeviceToDevice, ctx->stream));
  const int64_t off = a->nc;
  if (b) {
    GAIB_HIP(hipMemcpyAsync(g->col_vdata + off, b->col_vdata, sizeof(float) * b->nc, hipMemcpyDeviceToDevice, ctx->stream));
    GAIB_HIP(hipMemcpyAsync(g->col_inv_deg + off, b->col_inv_deg, sizeof(float) * b->nc, hipMemcpyDeviceToDevice, ctx->stream));
  }
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  guard.g = nullptr;
  *out = g;
  return GAIB_OK;
}

}  // namespace

extern "C" int gaib_graph_set_row_map(gaib_ctx* ctx, gaib_graph* g, const uint32_t* d_row_map, int64_t n_out_rows) {
  GAIB_CHECK(ctx && g, "gaib_graph_set_row_map: NULL argument");
  GAIB_HIP(hipSetDevice(ctx->device));
  if (!d_row_map) {
    if (g->row_map) GAIB_HIP(hipFree(g->row_map));
    g->row_map = nullptr;
    g->n_out_rows = 0;
    return GAIB_OK;
  }
  GAIB_CHECK(n_out_rows >= g->nv, "gaib_graph_set_row_map: %lld output rows for a graph of %lld rows", (long long)n_out_rows,
             (long long)g->nv);
  if (!g->row_map) {
    GAIB_HIP(hipMalloc(&g->row_map, sizeof(uint32_t) * (size_t)(g->nv > 0 ? g->nv : 1)));
    g->dev_bytes += sizeof(uint32_t) * g->nv;
  }
  GAIB_HIP(hipMemcpyAsync(g->row_map, d_row_map, sizeof(uint32_t) * (size_t)g->nv, hipMemcpyDeviceToDevice, ctx->stream));
  g->n_out_rows = n_out_rows;
  return GAIB_OK;
}

extern "C" const uint32_t* gaib_graph_row_map(const gaib_graph* g) { return g ? g->row_map : nullptr; }

extern "C" int gaib_graph_split_classes(gaib_ctx* ctx, const gaib_graph* g_own, const gaib_graph* g_halo, gaib_graph** interior,
                                        gaib_graph** bnd_own, gaib_graph** bnd_halo, gaib_graph** bnd_full,
                                        int64_t* h_n_boundary, int64_t* h_boundary_edges, int flags) {
  GAIB_CHECK(ctx && g_own && g_halo, "gaib_graph_split_classes: NULL argument");
  GAIB_CHECK((flags & ~GAIB_SPLIT_ALL_BOUNDARY) == 0, "gaib_graph_split_classes: unknown flags %d", flags);
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_graph_split_classes");
  GAIB_CHECK(g_own->nv == g_halo->nv, "gaib_graph_split_classes: the two graphs must hold the same rows (%lld vs %lld)",
             (long long)g_own->nv, (long long)g_halo->nv);
  GAIB_CHECK(g_own->nc == g_own->nv, "gaib_graph_split_classes: the owned-column graph must be square");
  GAIB_CHECK(!g_own->row_map && !g_halo->row_map, "gaib_graph_split_classes: class graphs cannot be split again");
  GAIB_CHECK(g_own->col_vdata && g_own->vdata && g_own->inv_deg && g_halo->col_vdata && g_halo->vdata && g_halo->inv_deg,
             "gaib_graph_split_classes: call gaib_graph_set_vertex_norm (rows and columns) on both graphs first");
  GAIB_CHECK(g_own->nc + g_halo->nc < (int64_t)1 << 32, "gaib_graph_split_classes: owned + halo columns must be < 2^32");
  GAIB_HIP(hipSetDevice(ctx->device));
  const int64_t nv = g_own->nv;
  uint32_t *is_bnd = nullptr, *pos = nullptr, *map_b = nullptr, *map_i = nullptr;
  void* tmp = nullptr;
  const size_t n1 = (size_t)(nv + 1);
  hipError_t e = hipMalloc(&is_bnd, sizeof(uint32_t) * n1);
  if (e == hipSuccess) e = hipMalloc(&pos, sizeof(uint32_t) * n1);
  if (e == hipSuccess) e = hipMalloc(&map_b, sizeof(uint32_t) * n1);
  if (e == hipSuccess) e = hipMalloc(&map_i, sizeof(uint32_t) * n1);
  struct Release {
    void** p;
    int n;
    ~Release() {
      for (int i = 0; i < n; ++i)
        if (p[i]) (void)hipFree(p[i]);
    }
  };
  void* owned[] = {is_bnd, pos, map_b, map_i, nullptr};
  Release rel{owned, 5};
  if (e != hipSuccess) {
    gaib_set_error("gaib_graph_split_classes: %s", hipGetErrorString(e));
    return GAIB_ERR_NOMEM;
  }
  GAIB_HIP(hipMemsetAsync(is_bnd, 0, sizeof(uint32_t) * n1, ctx->stream));
  if (nv > 0) {
    class_flag_kernel<<<grid1d(nv, 256), 256, 0, ctx->stream>>>(nv, g_halo->rowptr, (flags & GAIB_SPLIT_ALL_BOUNDARY) ? 1 : 0,
                                                                is_bnd);
    GAIB_LAUNCH_CHECK();
  }
  size_t tmp_bytes = 0;
  GAIB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, is_bnd, pos, (int)(nv + 1), ctx->stream));
  GAIB_HIP(hipMalloc(&tmp, tmp_bytes > 0 ? tmp_bytes : 8));
  owned[4] = tmp;
  GAIB_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, is_bnd, pos, (int)(nv + 1), ctx->stream));
  uint32_t n_b32 = 0;
  GAIB_HIP(hipMemcpyAsync(&n_b32, pos + nv, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  const int64_t n_b = n_b32, n_i = nv - n_b;
  if (nv > 0) {
    class_map_kernel<<<grid1d(nv, 256), 256, 0, ctx->stream>>>(nv, is_bnd, pos, map_b, map_i);
    GAIB_LAUNCH_CHECK();
  }
  gaib_graph* made[4] = {nullptr, nullptr, nullptr, nullptr};
  int rc = GAIB_OK;
  if (interior) rc = build_class(ctx, n_i, map_i, nv, g_own, nullptr, 0, g_own->nc, &made[0]);
  if (rc == GAIB_OK && bnd_own) rc = build_class(ctx, n_b, map_b, nv, g_own, nullptr, 0, g_own->nc, &made[1]);
  if (rc == GAIB_OK && bnd_halo) rc = build_class(ctx, n_b, map_b, nv, g_halo, nullptr, 0, g_halo->nc, &made[2]);
  if (rc == GAIB_OK && bnd_full)
    rc = build_class(ctx, n_b, map_b, nv, g_own, g_halo, (uint32_t)g_own->nc, g_own->nc + g_halo->nc, &made[3]);
  if (rc != GAIB_OK) {
    for (gaib_graph* m : made) gaib_graph_destroy(m);
    return rc;
  }
  if (interior) *interior = made[0];
  if (bnd_own) *bnd_own = made[1];
  if (bnd_halo) *bnd_halo = made[2];
  if (bnd_full) *bnd_full = made[3];
  if (h_n_boundary) *h_n_boundary = n_b;
  if (h_boundary_edges) {
    // edges (owned- and halo-column) of the boundary rows: what waits for the exchange in the one-pass form
    int64_t be = 0;
    if (made[3]) be = made[3]->ne;
    else if (made[1] && made[2]) be = made[1]->ne + made[2]->ne;
    else if (made[0]) be = g_own->ne + g_halo->ne - made[0]->ne;
    else be = -1;
    *h_boundary_edges = be;
  }
  return GAIB_OK;
}

// ---- pieces of a halo-column graph (round 6: the exchange in time slices, gaib_halo_set_pieces) ----
// Piece k of a rank's halo-column graph = the edges whose column (a row of the halo table) travels in slice k of the exchange:
// the same rows, the same column space, every row's edges in the row's order.  Aggregating piece 0, 1, ... in accumulate mode
// adds a row's terms piece by piece and inside a piece in column order -- for ONE peer (the slices are then consecutive column
// ranges) exactly the order of the uncut graph, for several peers the piece-major order of the same terms.
namespace {

struct PiecePtrs {
  int64_t* rp[GAIB_GRAPH_MAX_PIECES];
  uint32_t* col[GAIB_GRAPH_MAX_PIECES];
};

// piece of column c: the range [rb[j], re[j]) it lies in (ranges ascending, disjoint), -1 if none
__device__ __forceinline__ int piece_of(uint32_t c, int n_ranges, const uint32_t* rb, const uint32_t* re, const int* rpiece) {
  int lo = 0, hi = n_ranges;  // last range whose begin is <= c
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (rb[mid] <= c) lo = mid;
    else hi = mid;
  }
  return (n_ranges > 0 && c >= rb[lo] && c < re[lo]) ? rpiece[lo] : -1;
}

// one wave per row: cnt[k * (nv + 1) + row] = the row's edges in piece k; *n_lost counts edges in no piece
__global__ __launch_bounds__(256) void piece_count_kernel(int64_t nv, const int64_t* rowptr, const uint32_t* col, int n_pieces,
                                                          int n_ranges, const uint32_t* rb, const uint32_t* re, const int* rpiece,
                                                          int64_t* cnt, unsigned long long* n_lost) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row > nv) return;
  const int lane = threadIdx.x & 63;
  if (row == nv) {  // closes every piece's exclusive scan
    if (lane < n_pieces) cnt[(int64_t)lane * (nv + 1) + nv] = 0;
    return;
  }
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  int64_t mine = 0;  // lane k < n_pieces carries piece k's count
  unsigned lost = 0;
  for (int64_t base = e0; base < e1; base += 64) {
    const int64_t e = base + lane;
    const int pc = e < e1 ? piece_of(col[e], n_ranges, rb, re, rpiece) : -2;
    lost += pc == -1 ? 1u : 0u;
    for (int k = 0; k < n_pieces; ++k) {
      const int c = __popcll(__ballot(pc == k));
      if (lane == k) mine += c;
    }
  }
  if (lane < n_pieces) cnt[(int64_t)lane * (nv + 1) + row] = mine;
  lost = wave_sum_u32(lost);
  if (lane == 0 && lost) atomicAdd(n_lost, (unsigned long long)lost);
}

__global__ __launch_bounds__(256) void piece_fill_kernel(int64_t nv, const int64_t* rowptr, const uint32_t* col, int n_pieces,
                                                         int n_ranges, const uint32_t* rb, const uint32_t* re, const int* rpiece,
                                                         PiecePtrs out) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  int64_t off = lane < n_pieces ? out.rp[lane][row] : 0;  // lane k: where piece k's next edge of this row goes
  const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
  for (int64_t base = e0; base < e1; base += 64) {
    const int64_t e = base + lane;
    const uint32_t c = e < e1 ? col[e] : 0u;
    const int pc = e < e1 ? piece_of(c, n_ranges, rb, re, rpiece) : -2;
    for (int k = 0; k < n_pieces; ++k) {
      const unsigned long long m = __ballot(pc == k);
      const int64_t o = __shfl(off, k, 64);  // (int64 shuffle: two 32-bit halves)
      if (pc == k) out.col[k][o + __popcll(m & below)] = c;
      if (lane == k) off += __popcll(m);
    }
  }
}

}  // namespace

extern "C" int gaib_graph_split_pieces(gaib_ctx* ctx, const gaib_graph* g, int n_pieces, int n_ranges, const int64_t* h_range_begin,
                                       const int64_t* h_range_end, const int* h_range_piece, gaib_graph** out) {
  GAIB_CHECK(ctx && g && out && (n_ranges == 0 || (h_range_begin && h_range_end && h_range_piece)),
             "gaib_graph_split_pieces: NULL argument");
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_graph_split_pieces");
  GAIB_CHECK(n_pieces >= 1 && n_pieces <= GAIB_GRAPH_MAX_PIECES, "gaib_graph_split_pieces: %d pieces (1 .. %d)", n_pieces,
             GAIB_GRAPH_MAX_PIECES);
  GAIB_CHECK(n_ranges >= 0 && n_ranges <= 4096, "gaib_graph_split_pieces: %d column ranges (at most 4096)", n_ranges);
  GAIB_CHECK(g->col_vdata && g->vdata && g->inv_deg,
             "gaib_graph_split_pieces: call gaib_graph_set_vertex_norm (rows and columns) on the graph first");
  // ranges sorted by their first column, disjoint, inside the column space
  std::vector<int> order((size_t)n_ranges);
  for (int j = 0; j < n_ranges; ++j) order[(size_t)j] = j;
  std::sort(order.begin(), order.end(), [&](int a, int b) { return h_range_begin[a] < h_range_begin[b]; });
  std::vector<uint32_t> rb, re;
  std::vector<int> rpc;
  int64_t last_end = 0;
  for (int j : order) {
    const int64_t b = h_range_begin[j], e = h_range_end[j];
    if (e <= b) continue;
    GAIB_CHECK(b >= last_end && e <= g->nc, "gaib_graph_split_pieces: column range [%lld, %lld) overlaps another or leaves the %lld "
               "columns", (long long)b, (long long)e, (long long)g->nc);
    GAIB_CHECK(h_range_piece[j] >= 0 && h_range_piece[j] < n_pieces, "gaib_graph_split_pieces: range of piece %d (of %d)",
               h_range_piece[j], n_pieces);
    rb.push_back((uint32_t)b);
    re.push_back((uint32_t)e);
    rpc.push_back(h_range_piece[j]);
    last_end = e;
  }
  const int nr = (int)rb.size();
  GAIB_HIP(hipSetDevice(ctx->device));
  const int64_t nv = g->nv;
  struct Release {
    std::vector<void*> p;
    ~Release() {
      for (void* q : p)
        if (q) (void)hipFree(q);
    }
  } rel;
  auto dmalloc = [&](size_t bytes) -> void* {
    void* q = nullptr;
    if (hipMalloc(&q, bytes > 0 ? bytes : 16) != hipSuccess) {
      (void)hipGetLastError();
      return nullptr;
    }
    rel.p.push_back(q);
    return q;
  };
  uint32_t* d_rb = (uint32_t*)dmalloc(sizeof(uint32_t) * (size_t)nr);
  uint32_t* d_re = (uint32_t*)dmalloc(sizeof(uint32_t) * (size_t)nr);
  int* d_rpc = (int*)dmalloc(sizeof(int) * (size_t)nr);
  int64_t* cnt = (int64_t*)dmalloc(sizeof(int64_t) * (size_t)n_pieces * (size_t)(nv + 1));
  int64_t* rp = (int64_t*)dmalloc(sizeof(int64_t) * (size_t)n_pieces * (size_t)(nv + 1));
  unsigned long long* d_lost = (unsigned long long*)dmalloc(sizeof(unsigned long long));
  if (!d_rb || !d_re || !d_rpc || !cnt || !rp || !d_lost) {
    gaib_set_error("gaib_graph_split_pieces: out of device memory");
    return GAIB_ERR_NOMEM;
  }
  if (nr) {
    GAIB_HIP(hipMemcpyAsync(d_rb, rb.data(), sizeof(uint32_t) * (size_t)nr, hipMemcpyHostToDevice, ctx->stream));
    GAIB_HIP(hipMemcpyAsync(d_re, re.data(), sizeof(uint32_t) * (size_t)nr, hipMemcpyHostToDevice, ctx->stream));
    GAIB_HIP(hipMemcpyAsync(d_rpc, rpc.data(), sizeof(int) * (size_t)nr, hipMemcpyHostToDevice, ctx->stream));
  }
  GAIB_HIP(hipMemsetAsync(d_lost, 0, sizeof(unsigned long long), ctx->stream));
  piece_count_kernel<<<grid1d(nv + 1, 4), 256, 0, ctx->stream>>>(nv, g->rowptr, g->colidx, n_pieces, nr, d_rb, d_re, d_rpc, cnt, d_lost);
  GAIB_LAUNCH_CHECK();
  size_t tmp_bytes = 0;
  GAIB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, cnt, rp, (int)(nv + 1), ctx->stream));
  void* tmp = dmalloc(tmp_bytes);
  GAIB_CHECK(tmp, "gaib_graph_split_pieces: out of device memory");
  for (int k = 0; k < n_pieces; ++k)
    GAIB_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tmp_bytes, cnt + (int64_t)k * (nv + 1), rp + (int64_t)k * (nv + 1), (int)(nv + 1),
                                              ctx->stream));
  unsigned long long lost = 0;
  int64_t ne_k[GAIB_GRAPH_MAX_PIECES];
  GAIB_HIP(hipMemcpyAsync(&lost, d_lost, sizeof(lost), hipMemcpyDeviceToHost, ctx->stream));
  for (int k = 0; k < n_pieces; ++k)
    GAIB_HIP(hipMemcpyAsync(&ne_k[k], rp + (int64_t)k * (nv + 1) + nv, sizeof(int64_t), hipMemcpyDeviceToHost, ctx->stream));
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  GAIB_CHECK(lost == 0, "gaib_graph_split_pieces: %llu edges point at columns outside every piece's ranges", lost);
  gaib_graph* made[GAIB_GRAPH_MAX_PIECES] = {};
  struct Guard {
    gaib_graph** m;
    int n;
    bool armed;
    ~Guard() {
      for (int k = 0; armed && k < n; ++k) gaib_graph_destroy(m[k]);
    }
  } guard{made, n_pieces, true};
  PiecePtrs ptrs;
  memset(&ptrs, 0, sizeof(ptrs));
  const size_t n1 = (size_t)(nv > 0 ? nv : 1), nc1 = (size_t)(g->nc > 0 ? g->nc : 1);
  for (int k = 0; k < n_pieces; ++k) {
    GAIB_TRY(new_graph(nv, ne_k[k], ctx->device, &made[k]));
    gaib_graph* q = made[k];
    q->nc = g->nc;
    GAIB_HIP(hipMemcpyAsync(q->rowptr, rp + (int64_t)k * (nv + 1), sizeof(int64_t) * (size_t)(nv + 1), hipMemcpyDeviceToDevice,
                            ctx->stream));
    ptrs.rp[k] = q->rowptr;
    ptrs.col[k] = q->colidx;
    // the rows' and columns' normalisers: the source graph's
    GAIB_HIP(hipMalloc(&q->vdata, sizeof(float) * n1));
    GAIB_HIP(hipMalloc(&q->inv_deg, sizeof(float) * n1));
    GAIB_HIP(hipMalloc(&q->col_vdata, sizeof(float) * nc1));
    GAIB_HIP(hipMalloc(&q->col_inv_deg, sizeof(float) * nc1));
    q->dev_bytes += sizeof(float) * (2 * nv + 2 * g->nc);
    GAIB_HIP(hipMemcpyAsync(q->vdata, g->vdata, sizeof(float) * (size_t)nv, hipMemcpyDeviceToDevice, ctx->stream));
    GAIB_HIP(hipMemcpyAsync(q->inv_deg, g->inv_deg, sizeof(float) * (size_t)nv, hipMemcpyDeviceToDevice, ctx->stream));
    GAIB_HIP(hipMemcpyAsync(q->col_vdata, g->col_vdata, sizeof(float) * (size_t)g->nc, hipMemcpyDeviceToDevice, ctx->stream));
    GAIB_HIP(hipMemcpyAsync(q->col_inv_deg, g->col_inv_deg ? g->col_inv_deg : g->col_vdata, sizeof(float) * (size_t)g->nc,
                            hipMemcpyDeviceToDevice, ctx->stream));
    if (g->row_map) {  // a piece of a row class keeps the class's map
      GAIB_HIP(hipMalloc(&q->row_map, sizeof(uint32_t) * n1));
      GAIB_HIP(hipMemcpyAsync(q->row_map, g->row_map, sizeof(uint32_t) * (size_t)nv, hipMemcpyDeviceToDevice, ctx->stream));
      q->n_out_rows = g->n_out_rows;
      q->dev_bytes += sizeof(uint32_t) * nv;
    }
    q->rows_unsorted = g->rows_unsorted;
  }
  if (nv > 0 && g->ne > 0) {
    piece_fill_kernel<<<grid1d(nv, 4), 256, 0, ctx->stream>>>(nv, g->rowptr, g->colidx, n_pieces, nr, d_rb, d_re, d_rpc, ptrs);
    GAIB_LAUNCH_CHECK();
  }
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  guard.armed = false;
  for (int k = 0; k < n_pieces; ++k) out[k] = made[k];
  return GAIB_OK;
}

// ---- csr2csc (math_functions.hh:45; cusparseCsr2cscEx2, math_functions.cu:345-358): the transpose of a general CSR matrix
// with 32-bit offsets, everything in device memory.  A stable radix sort of the edges by column id: inside a column the
// rows come out ascending (the order of cuSPARSE's ALG1).
namespace {
__global__ __launch_bounds__(256) void expand_rows_kernel(int nrows, const int* rowptr, uint32_t* rows) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= nrows) return;
  for (int e = rowptr[r] + (threadIdx.x & 63); e < rowptr[r + 1]; e += 64) rows[e] = (uint32_t)r;
}
__global__ void csc_fill_kernel(int64_t nnz, const uint32_t* perm, const uint32_t* rows, const float* values, float* valuesT,
                                int* colidxT) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const uint32_t e = perm[k];
  colidxT[k] = (int)rows[e];
  if (values) valuesT[k] = values[e];
}
// rowptrT[c] = first position of a key >= c in the sorted column ids
__global__ void csc_offsets_kernel(int ncols, int64_t nnz, const uint32_t* sorted_cols, int* rowptrT) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > ncols) return;
  int64_t lo = 0, hi = nnz;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (sorted_cols[mid] < (uint32_t)c) lo = mid + 1;
    else hi = mid;
  }
  rowptrT[c] = (int)lo;
}
}  // namespace

extern "C" int gaib_csr2csc(gaib_ctx* ctx, int nrows, int ncols, int nnz, const float* d_values, const int* d_rowptr,
                            const int* d_colidx, float* d_valuesT, int* d_rowptrT, int* d_colidxT) {
  GAIB_CHECK(ctx && nrows >= 0 && ncols >= 0 && nnz >= 0, "gaib_csr2csc: bad size");
  GAIB_CHECK(d_rowptr && d_rowptrT && (nnz == 0 || (d_colidx && d_colidxT)), "gaib_csr2csc: NULL pointer");
  GAIB_CHECK((d_values == nullptr) == (d_valuesT == nullptr) || nnz == 0, "gaib_csr2csc: values and valuesT go together");
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_csr2csc");
  GAIB_HIP(hipSetDevice(ctx->device));
  if (nnz == 0) {
    GAIB_HIP(hipMemsetAsync(d_rowptrT, 0, sizeof(int) * (size_t)(ncols + 1), ctx->stream));
    return GAIB_OK;
  }
  uint32_t *keys_out = nullptr, *ids = nullptr, *perm = nullptr, *rows = nullptr;
  void* tmp = nullptr;
  size_t tmp_bytes = 0;
  int end_bit = 1;
  while (end_bit < 32 && ((int64_t)1 << end_bit) < ncols) end_bit++;
  const uint32_t* keys_in = reinterpret_cast<const uint32_t*>(d_colidx);
  hipError_t e = hipMalloc(&keys_out, sizeof(uint32_t) * (size_t)nnz);
  if (e == hipSuccess) e = hipMalloc(&ids, sizeof(uint32_t) * (size_t)nnz);
  if (e == hipSuccess) e = hipMalloc(&perm, sizeof(uint32_t) * (size_t)nnz);
  if (e == hipSuccess) e = hipMalloc(&rows, sizeof(uint32_t) * (size_t)nnz);
  if (e == hipSuccess) e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, keys_in, keys_out, ids, perm, nnz, 0, end_bit, ctx->stream);
  if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes);
  if (e == hipSuccess) {
    iota_u32_kernel<<<grid1d(nnz, 256), 256, 0, ctx->stream>>>(nnz, ids);
    expand_rows_kernel<<<grid1d(nrows, 4), 256, 0, ctx->stream>>>(nrows, d_rowptr, rows);
    e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, keys_in, keys_out, ids, perm, nnz, 0, end_bit, ctx->stream);
  }
  if (e == hipSuccess) {
    csc_fill_kernel<<<grid1d(nnz, 256), 256, 0, ctx->stream>>>(nnz, perm, rows, d_values, d_valuesT, d_colidxT);
    csc_offsets_kernel<<<grid1d(ncols + 1, 256), 256, 0, ctx->stream>>>(ncols, nnz, keys_out, d_rowptrT);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  void* owned[] = {keys_out, ids, perm, rows, tmp};
  for (void* p : owned)
    if (p) (void)hipFree(p);
  if (e != hipSuccess) {
    gaib_set_error("gaib_csr2csc: %s", hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? GAIB_ERR_NOMEM : GAIB_ERR_HIP;
  }
  return GAIB_OK;
}
