// gat.hip -- GAT attention kernels for gfx950: edge scores + edge-softmax, SDDMM,
// softmax-backward + alpha gradients, symmetric edge transpose.
//
// replaces GAT_Aggregator::aggregate / d_aggregate (src/gnn/gconv/gat_aggregator.cpp:57-200)
// and the CUDA kernels compute_attn_score_warp, compute_scores_grad_warp,
// compute_alpha_grad_warp, symmetric_csr_transpose_kernel
// (include/gnn/graph_operations.h:191-467).
//
// Differences in HOW (results agree to fp32 rounding):
//   * a_l.h[i] and a_r.h[j] are computed once per VERTEX (O(N*D)) and gathered per edge
//     (4 B) instead of recomputing a_r.h[col_e] per EDGE (O(E*D), gat_aggregator.cpp:71).
//   * the alpha gradients  sum_e g_e*h[col_e]  and  sum_i (sum_e g_e)*h[i]  are regrouped by
//     vertex using the reverse-edge permutation:  alpha_r' = sum_v cs[v]*h[v],
//     alpha_l' = sum_v rs[v]*h[v]  with rs = row sums and cs = column sums of g
//     (O(E + N*D) instead of O(E*D)); reduction is a fixed two-level tree, no float atomics.
//   * the transposed-score SpMM reads p[rev(e)] directly (gaib_spmm GAIB_W_EDGE_T); the
//     reverse-edge permutation is built once per graph instead of a binary search per call.
#include "common.h"

namespace {

// s[v] = <alpha, h[v,:]> for two alpha vectors at once.  One wave per row.
__global__ __launch_bounds__(256) void vertex_dots_kernel(int64_t nv, int len, const float* h,
                                                          const float* al, const float* ar,
                                                          float* sl, float* sr) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const float* hr = h + row * (int64_t)len;
  float pl = 0.f, pr = 0.f;
  for (int c = lane; c < len; c += 64) {
    float x = hr[c];
    pl += al[c] * x;
    pr += ar[c] * x;
  }
  pl = wave_sum(pl);
  pr = wave_sum(pr);
  if (lane == 0) {
    sl[row] = pl;
    sr[row] = pr;
  }
}

// per row: temp = sl[i] + sr[col]; scores = leaky_relu; norm = softmax over the row.
// (gat_aggregator.cpp:64-77; softmax math_functions.cpp:485-494: max-subtracted, expf, divide)
__global__ __launch_bounds__(256) void edge_softmax_kernel(int64_t nv, const int64_t* rowptr,
                                                           const uint32_t* col, const float* sl,
                                                           const float* sr, float eps, float* temp,
                                                           float* scores, float* norm) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  if (e0 == e1) return;
  const float s_src = sl[row];
  if (e1 - e0 <= 64) {
    // whole row in registers
    const int64_t e = e0 + lane;
    const bool ok = e < e1;
    float s = -INFINITY;
    if (ok) {
      float t = s_src + sr[col[e]];
      temp[e] = t;
      s = t > 0.0f ? t : eps * t;
      scores[e] = s;
    }
    const float mx = wave_max(s);
    const float ex = ok ? expf(s - mx) : 0.f;
    const float den = wave_sum(ex);
    if (ok) norm[e] = ex / den;
    return;
  }
  float mx = -INFINITY;
  for (int64_t e = e0 + lane; e < e1; e += 64) {
    float t = s_src + sr[col[e]];
    temp[e] = t;
    float s = t > 0.0f ? t : eps * t;
    scores[e] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float den = 0.f;
  for (int64_t e = e0 + lane; e < e1; e += 64) {  // each lane re-reads only its own writes
    float ex = expf(scores[e] - mx);
    norm[e] = ex;
    den += ex;
  }
  den = wave_sum(den);
  for (int64_t e = e0 + lane; e < e1; e += 64) norm[e] = norm[e] / den;
}

// SDDMM: out[e] = <grad[i,:], feat[col_e,:]>.  One wave per row; the row's grad vector
// stays in registers (up to 4 floats per lane = len 256), wider rows loop.
template <int VEC>
__global__ __launch_bounds__(256) void sddmm_kernel(int64_t nv, const int64_t* rowptr,
                                                    const uint32_t* col, int len,
                                                    const float* grad, const float* feat,
                                                    float* out_e) {
  int row32 = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (row32 >= nv) return;
  const int64_t row = __builtin_amdgcn_readfirstlane(row32);  // wave-uniform -> scalar loads
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  const float* gr = grad + row * (int64_t)len;
  if (len <= 64 * VEC) {
    float g[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) {
      int c = lane * VEC + v;
      g[v] = c < len ? gr[c] : 0.f;
    }
    for (int64_t base = e0; base < e1; base += 64) {
      const int64_t rem = e1 - base;
      const int n = rem < 64 ? (int)rem : 64;
      uint32_t c = lane < n ? col[base + lane] : 0u;
      float res = 0.f;
      for (int j = 0; j < n; ++j) {
        const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, j);
        const float* fr = feat + (int64_t)cj * len;
        float p = 0.f;
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
          int cc = lane * VEC + v;
          float x = cc < len ? fr[cc] : 0.f;
          p += g[v] * x;
        }
        p = wave_sum(p);
        if (lane == j) res = p;
      }
      if (lane < n) out_e[base + lane] = res;
    }
  } else {
    for (int64_t base = e0; base < e1; base += 64) {
      const int64_t rem = e1 - base;
      const int n = rem < 64 ? (int)rem : 64;
      uint32_t c = lane < n ? col[base + lane] : 0u;
      float res = 0.f;
      for (int j = 0; j < n; ++j) {
        const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, j);
        const float* fr = feat + (int64_t)cj * len;
        float p = 0.f;
        for (int cc = lane; cc < len; cc += 64) p += gr[cc] * fr[cc];
        p = wave_sum(p);
        if (lane == j) res = p;
      }
      if (lane < n) out_e[base + lane] = res;
    }
  }
}


// SDDMM, edge-parallel: one wave per 64-edge chunk (gaib_graph_ensure_chunks), G lanes x float4 per
// edge, so one wave instruction gathers 64/G feature rows (1 KB) and a dot product costs
// 4 FMAs + log2(G) shuffles.  Group k of the wave owns edges k*G .. k*G+G-1 of the chunk; in step
// j it reduces its j-th edge and lane k*G+j (which sits in group k) keeps the result, so after G
// steps lane l holds edge l and the store is coalesced.  Perfectly balanced for power-law rows.
template <int G, int U>
__global__ __launch_bounds__(256) void sddmm_chunk_kernel(int64_t n_chunks, const uint32_t* chunk_row,
                                                          const uint32_t* chunk_ebase,
                                                          const int64_t* rowptr, const uint32_t* col,
                                                          int len, const float* grad, const float* feat,
                                                          float* out_e) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const int sl = lane & (G - 1), gbase = lane & ~(G - 1);
  const int64_t row = chunk_row[c];
  const int64_t eb = chunk_ebase[c];
  const int64_t rem = rowptr[row + 1] - eb;
  const int n = rem < 64 ? (int)rem : 64;
  const uint32_t cl = col[eb + (lane < n ? lane : 0)];
  const bool colok = sl * 4 < len;
  const int coff = colok ? sl * 4 : 0;
  f4 g4 = {0.f, 0.f, 0.f, 0.f};
  if (colok) g4 = *reinterpret_cast<const f4*>(grad + row * (int64_t)len + coff);
  float res = 0.f;
#pragma unroll
  for (int j = 0; j < G; j += U) {
    f4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t cj = (uint32_t)__shfl((int)cl, gbase + j + u, 64);
      x[u] = *reinterpret_cast<const f4*>(feat + (int64_t)cj * len + coff);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float p = g4[0] * x[u][0] + g4[1] * x[u][1] + g4[2] * x[u][2] + g4[3] * x[u][3];
#pragma unroll
      for (int o = G / 2; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
      if (sl == j + u) res = p;
    }
  }
  if (lane < n) out_e[eb + lane] = res;
}

// softmax backward per row (math_functions.cpp:496-514, closed form of the :497-504 branch)
// + leaky-relu' (gat_aggregator.cpp:145).  Writes ds into scores[], g into gbuf[], and the
// row sum of g into rs[].
__global__ __launch_bounds__(256) void softmax_bwd_kernel(int64_t nv, const int64_t* rowptr,
                                                          const float* p, const float* dp,
                                                          const float* temp, float eps,
                                                          float* scores, float* gbuf, float* rs) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  float dot = 0.f;
  for (int64_t e = e0 + lane; e < e1; e += 64) dot += p[e] * dp[e];
  dot = wave_sum(dot);
  float gs = 0.f;
  for (int64_t e = e0 + lane; e < e1; e += 64) {
    const float pe = p[e], dpe = dp[e];
    const float x = pe * (1.0f - pe) * dpe;
    const float ds = x - (dot - pe * dpe) * pe;
    scores[e] = ds;
    const float ge = ds * (temp[e] > 0.0f ? 1.0f : eps);
    gbuf[e] = ge;
    gs += ge;
  }
  gs = wave_sum(gs);
  if (lane == 0) rs[row] = gs;
}

// cs[v] = sum_{e in row v} g[rev[e]]  == column sum of g (structurally symmetric graph)
__global__ __launch_bounds__(256) void colsum_kernel(int64_t nv, const int64_t* rowptr,
                                                     const uint32_t* rev, const float* gbuf,
                                                     float* cs) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  float s = 0.f;
  for (int64_t e = e0 + lane; e < e1; e += 64) s += gbuf[rev[e]];
  s = wave_sum(s);
  if (lane == 0) cs[row] = s;
}

// partial[b][0][c] = sum_{v in strip b} rs[v]*h[v][c]; partial[b][1][c] likewise with cs.
// 256 threads: thread t owns column (t % cw) of every (256/cw)-th row of the strip, where
// cw = min(len,256) rounded to a divisor layout; generic: loop columns.
__global__ __launch_bounds__(256) void alpha_partial_kernel(int64_t nv, int len, const float* h,
                                                            const float* rs, const float* cs,
                                                            int64_t rows_per_block, float* partial) {
  extern __shared__ float sm[];  // [2][256]
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < nv) ? r0 + rows_per_block : nv;
  float* out = partial + (int64_t)blockIdx.x * 2 * len;
  for (int c0 = 0; c0 < len; c0 += 256) {
    const int cw = (len - c0 < 256) ? (len - c0) : 256;  // columns in this pass
    const int rpp = 256 / cw;                            // rows per pass (>=1)
    const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
    float al = 0.f, ar = 0.f;
    if (tr < rpp) {
      for (int64_t v = r0 + tr; v < r1; v += rpp) {
        const float x = h[v * (int64_t)len + c0 + tc];
        al += rs[v] * x;
        ar += cs[v] * x;
      }
    }
    sm[threadIdx.x] = (tr < rpp) ? al : 0.f;
    sm[256 + threadIdx.x] = (tr < rpp) ? ar : 0.f;
    __syncthreads();
    if (threadIdx.x < cw) {
      float sl_ = 0.f, sr_ = 0.f;
      for (int k = 0; k < rpp; ++k) {
        sl_ += sm[k * cw + threadIdx.x];
        sr_ += sm[256 + k * cw + threadIdx.x];
      }
      out[c0 + threadIdx.x] = sl_;
      out[len + c0 + threadIdx.x] = sr_;
    }
    __syncthreads();
  }
}

__global__ void alpha_final_kernel(int nblocks, int len, const float* partial, float* lgrad,
                                   float* rgrad) {
  int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= len) return;
  float sl_ = 0.f, sr_ = 0.f;
  for (int b = 0; b < nblocks; ++b) {
    sl_ += partial[(int64_t)b * 2 * len + c];
    sr_ += partial[(int64_t)b * 2 * len + len + c];
  }
  lgrad[c] = sl_;
  rgrad[c] = sr_;
}

__global__ void edge_gather_kernel(int64_t ne, const uint32_t* rev, const float* in, float* out) {
  int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < ne) out[e] = in[rev[e]];
}

inline unsigned rowgrid(int64_t nv) { return (unsigned)cdiv64(nv > 0 ? nv : 1, 4); }

}  // namespace

extern "C" int gaib_gat_scores(gaib_ctx* ctx, gaib_graph* g, int len, const float* d_h,
                               const float* d_alpha_l, const float* d_alpha_r, float epsilon,
                               float* d_temp_scores, float* d_scores, float* d_norm_scores) {
  GAIB_CHECK(ctx && g, "gaib_gat_scores: NULL ctx/graph");
  GAIB_CHECK(len > 0, "gaib_gat_scores: len must be > 0");
  if (g->nv == 0) return GAIB_OK;
  GAIB_CHECK(d_h && d_alpha_l && d_alpha_r && d_temp_scores && d_scores && d_norm_scores,
             "gaib_gat_scores: NULL pointer");
  GAIB_HIP(hipSetDevice(ctx->device));
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * 2 * (size_t)g->nv));
  float* sl = (float*)ctx->ws;
  float* sr = sl + g->nv;
  { ProfScope ps(ctx, "gat_vertex_dots");
  vertex_dots_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, len, d_h, d_alpha_l,
                                                              d_alpha_r, sl, sr);
  }
  GAIB_LAUNCH_CHECK();
  { ProfScope ps(ctx, "gat_edge_softmax");
  edge_softmax_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(
      g->nv, g->rowptr, g->colidx, sl, sr, epsilon, d_temp_scores, d_scores, d_norm_scores);
  }
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_sddmm(gaib_ctx* ctx, gaib_graph* g, int len, const float* d_grad,
                          const float* d_feat, float* d_out_e) {
  GAIB_CHECK(ctx && g, "gaib_sddmm: NULL ctx/graph");
  GAIB_CHECK(len > 0, "gaib_sddmm: len must be > 0");
  if (g->nv == 0 || g->ne == 0) return GAIB_OK;
  GAIB_CHECK(d_grad && d_feat && d_out_e, "gaib_sddmm: NULL pointer");
  GAIB_HIP(hipSetDevice(ctx->device));
  const bool vec_ok = (len % 4 == 0) && len <= 256 &&
                      ((((uintptr_t)d_grad | (uintptr_t)d_feat) & 15) == 0) && ctx->gat_fast;
  if (vec_ok) {
    GAIB_TRY(gaib_graph_ensure_chunks(ctx, g));
    ProfScope ps(ctx, "gat_sddmm");
    const unsigned grid = (unsigned)cdiv64(g->n_chunks > 0 ? g->n_chunks : 1, 4);
#define GAIB_SDDMM(G, U)                                                                             \
  sddmm_chunk_kernel<G, U><<<grid, 256, 0, ctx->stream>>>(g->n_chunks, g->chunk_row, g->chunk_ebase, \
                                                          g->rowptr, g->colidx, len, d_grad, d_feat, d_out_e)
    if (len <= 4) GAIB_SDDMM(1, 1);
    else if (len <= 8) GAIB_SDDMM(2, 2);
    else if (len <= 16) GAIB_SDDMM(4, 4);
    else if (len <= 32) GAIB_SDDMM(8, 8);
    else if (len <= 64) GAIB_SDDMM(16, 8);
    else if (len <= 128) GAIB_SDDMM(32, 8);
    else GAIB_SDDMM(64, 8);
#undef GAIB_SDDMM
    GAIB_LAUNCH_CHECK();
    return GAIB_OK;
  }
  ProfScope ps(ctx, "gat_sddmm");
  if (len <= 64)
    sddmm_kernel<1><<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->colidx, len,
                                                             d_grad, d_feat, d_out_e);
  else if (len <= 128)
    sddmm_kernel<2><<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->colidx, len,
                                                             d_grad, d_feat, d_out_e);
  else
    sddmm_kernel<4><<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->colidx, len,
                                                             d_grad, d_feat, d_out_e);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_gat_softmax_bwd_alpha(gaib_ctx* ctx, gaib_graph* g, int len,
                                          const float* d_feat, const float* d_norm_scores,
                                          const float* d_norm_scores_grad,
                                          const float* d_temp_scores, float epsilon,
                                          float* d_scores, float* d_alpha_lgrad,
                                          float* d_alpha_rgrad) {
  GAIB_CHECK(ctx && g, "gaib_gat_softmax_bwd_alpha: NULL ctx/graph");
  GAIB_CHECK(len > 0, "gaib_gat_softmax_bwd_alpha: len must be > 0");
  GAIB_CHECK(d_alpha_lgrad && d_alpha_rgrad, "gaib_gat_softmax_bwd_alpha: NULL alpha grad");
  GAIB_HIP(hipSetDevice(ctx->device));
  if (g->nv == 0) return GAIB_OK;
  GAIB_CHECK(d_feat && d_norm_scores && d_norm_scores_grad && d_temp_scores && d_scores,
             "gaib_gat_softmax_bwd_alpha: NULL pointer");
  GAIB_TRY(gaib_graph_ensure_rev(ctx, g));
  const int nblocks = (int)(g->nv < 2048 ? cdiv64(g->nv, 8) : 1024);
  const int64_t rows_per_block = cdiv64(g->nv, nblocks);
  const size_t ws_floats = (size_t)g->ne + 2 * (size_t)g->nv + (size_t)nblocks * 2 * len;
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * ws_floats));
  float* gbuf = (float*)ctx->ws;
  float* rs = gbuf + g->ne;
  float* cs = rs + g->nv;
  float* partial = cs + g->nv;
  ProfScope ps(ctx, "gat_softmax_bwd_alpha");
  softmax_bwd_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(
      g->nv, g->rowptr, d_norm_scores, d_norm_scores_grad, d_temp_scores, epsilon, d_scores, gbuf, rs);
  GAIB_LAUNCH_CHECK();
  colsum_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, g->rowptr, g->rev, gbuf, cs);
  GAIB_LAUNCH_CHECK();
  alpha_partial_kernel<<<nblocks, 256, sizeof(float) * 512, ctx->stream>>>(
      g->nv, len, d_feat, rs, cs, rows_per_block, partial);
  GAIB_LAUNCH_CHECK();
  alpha_final_kernel<<<(unsigned)cdiv64(len, 256), 256, 0, ctx->stream>>>(nblocks, len, partial,
                                                                         d_alpha_lgrad, d_alpha_rgrad);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_edge_transpose(gaib_ctx* ctx, gaib_graph* g, const float* d_in_e,
                                   float* d_out_e) {
  GAIB_CHECK(ctx && g, "gaib_edge_transpose: NULL ctx/graph");
  if (g->ne == 0) return GAIB_OK;
  GAIB_CHECK(d_in_e && d_out_e && d_in_e != d_out_e, "gaib_edge_transpose: bad pointers");
  GAIB_HIP(hipSetDevice(ctx->device));
  GAIB_TRY(gaib_graph_ensure_rev(ctx, g));
  // rev is an involution on a structurally symmetric graph: out[rev[e]] = in[e]  <=>
  // out[e] = in[rev[e]]; the gather form keeps the stores coalesced.
  edge_gather_kernel<<<(unsigned)cdiv64(g->ne, 256), 256, 0, ctx->stream>>>(g->ne, g->rev, d_in_e,
                                                                          d_out_e);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}
