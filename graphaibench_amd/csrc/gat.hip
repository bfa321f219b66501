// gat.hip -- GAT attention kernels for gfx950: edge scores + edge-softmax, SDDMM,
// softmax-backward + alpha gradients, symmetric edge transpose.  All kernels take a head count H
// (the reference is single-head, H = 1; H > 1 = H independent single-head attentions on the column
// slices [h*D/H, (h+1)*D/H), edge arrays laid out [ne][H]).
//
// replaces GAT_Aggregator::aggregate / d_aggregate (src/gnn/gconv/gat_aggregator.cpp:57-200)
// and the CUDA kernels compute_attn_score_warp, compute_scores_grad_warp,
// compute_alpha_grad_warp, symmetric_csr_transpose_kernel
// (include/gnn/graph_operations.h:191-467).
//
// Differences in HOW (results agree to fp32 rounding):
//   * a_l.h[i] and a_r.h[j] are computed once per VERTEX (O(N*D)) and gathered per edge
//     (4 B per head) instead of recomputing a_r.h[col_e] per EDGE (O(E*D), gat_aggregator.cpp:71).
//   * SDDMM is edge-parallel over 64-edge chunks (balanced on power-law rows).
//   * the alpha gradients  sum_e g_e*h[col_e]  and  sum_i (sum_e g_e)*h[i]  are regrouped by
//     vertex using the reverse-edge permutation:  alpha_r' = sum_v cs[v]*h[v],
//     alpha_l' = sum_v rs[v]*h[v]  with rs = row sums and cs = column sums of g
//     (O(E + N*D) instead of O(E*D)); reduction is a fixed two-level tree, no float atomics.
//   * the transposed-score SpMM reads p[rev(e)] directly (gaib_spmm GAIB_W_EDGE_T); the
//     reverse-edge permutation is built once per graph instead of a binary search per call.
#include "common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

// s[v,h] = <alpha[slice h], x[v, slice h]> for two alpha vectors at once.  One wave per row.
__global__ __launch_bounds__(256) void vertex_dots_kernel(int64_t nv, int len, int H, const float* x,
                                                          const float* al, const float* ar,
                                                          float* sl, float* sr) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const float* xr = x + row * (int64_t)len;
  const int dh = len / H;
  if (H > 1 && dh <= 64 && (64 % dh) == 0) {
    // head slices of 1..64 columns that tile a wave: every 64-column chunk holds whole heads, reduced inside their
    // groups of dh lanes (8 heads x 8 columns: 6 shuffles per row instead of 96)
    for (int c0 = 0; c0 < len; c0 += 64) {
      const int c = c0 + lane;
      const bool ok = c < len;
      const float v = ok ? xr[c] : 0.f;
      float pl = ok ? al[c] * v : 0.f, pr = ok ? ar[c] * v : 0.f;
      for (int o = dh >> 1; o >= 1; o >>= 1) {
        pl += __shfl_xor(pl, o, 64);
        pr += __shfl_xor(pr, o, 64);
      }
      if (ok && (lane % dh) == 0) {
        sl[row * H + c / dh] = pl;
        sr[row * H + c / dh] = pr;
      }
    }
    return;
  }
  for (int h = 0; h < H; ++h) {
    float pl = 0.f, pr = 0.f;
    for (int c = h * dh + lane; c < (h + 1) * dh; c += 64) {
      const float v = xr[c];
      pl += al[c] * v;
      pr += ar[c] * v;
    }
    pl = wave_sum(pl);
    pr = wave_sum(pr);
    if (lane == 0) {
      sl[row * H + h] = pl;
      sr[row * H + h] = pr;
    }
  }
}

// per (row, head): temp = sl[i,h] + sr[col,h]; scores = leaky_relu; norm = softmax over the row.
// (gat_aggregator.cpp:64-77; softmax math_functions.cpp:485-494: max-subtracted, expf, divide)
__global__ __launch_bounds__(256) void edge_softmax_kernel(int64_t nv, int H, const int64_t* rowptr,
                                                           const uint32_t* col, const float* sl,
                                                           const float* sr, float eps, float* temp,
                                                           float* scores, float* norm) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  if (e0 == e1) return;
  if (e1 - e0 <= 64) {
    // whole row in registers, one lane per edge
    const int64_t e = e0 + lane;
    const bool ok = e < e1;
    const int64_t c = ok ? (int64_t)col[e] : 0;
    for (int h = 0; h < H; ++h) {
      float s = -INFINITY;
      if (ok) {
        const float t = sl[row * H + h] + sr[c * H + h];
        temp[e * H + h] = t;
        s = t > 0.0f ? t : eps * t;
        scores[e * H + h] = s;
      }
      const float mx = wave_max(s);
      const float ex = ok ? expf(s - mx) : 0.f;
      const float den = wave_sum(ex);
      if (ok) norm[e * H + h] = ex / den;
    }
    return;
  }
  for (int h = 0; h < H; ++h) {
    const float s_src = sl[row * H + h];
    float mx = -INFINITY;
    for (int64_t e = e0 + lane; e < e1; e += 64) {
      const float t = s_src + sr[(int64_t)col[e] * H + h];
      temp[e * H + h] = t;
      const float s = t > 0.0f ? t : eps * t;
      scores[e * H + h] = s;
      mx = fmaxf(mx, s);
    }
    mx = wave_max(mx);
    float den = 0.f;
    for (int64_t e = e0 + lane; e < e1; e += 64) {  // each lane re-reads only its own writes
      const float ex = expf(scores[e * H + h] - mx);
      norm[e * H + h] = ex;
      den += ex;
    }
    den = wave_sum(den);
    for (int64_t e = e0 + lane; e < e1; e += 64) norm[e * H + h] = norm[e * H + h] / den;
  }
}

// SDDMM fallback: out[e,h] = <grad[i, slice h], feat[col_e, slice h]>.  One wave per row, one wave
// reduction per (edge, head).  Used for shapes the chunk kernels below do not cover.
__global__ __launch_bounds__(256) void sddmm_generic_kernel(int64_t nv, int H, const int64_t* rowptr,
                                                            const uint32_t* col, int len,
                                                            const float* grad, const float* feat,
                                                            float* out_e) {
  int row32 = (int)blockIdx.x * 4 + (int)(threadIdx.x >> 6);
  if (row32 >= nv) return;
  const int64_t row = __builtin_amdgcn_readfirstlane(row32);  // wave-uniform -> scalar loads
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  const float* gr = grad + row * (int64_t)len;
  const int dh = len / H;
  for (int64_t e = e0; e < e1; ++e) {
    const float* fr = feat + (int64_t)col[e] * len;
    for (int h = 0; h < H; ++h) {
      float p = 0.f;
      for (int c = h * dh + lane; c < (h + 1) * dh; c += 64) p += gr[c] * fr[c];
      p = wave_sum(p);
      if (lane == 0) out_e[e * H + h] = p;
    }
  }
}

// SDDMM, edge-parallel: one wave per 64-edge chunk (gaib_graph_ensure_chunks), G lanes x float4 per
// edge, so one wave instruction gathers 64/G feature rows (1 KB) and a dot product costs
// 4 FMAs + log2 shuffles.  Group k of the wave owns edges k*G .. k*G+G-1 of the chunk; in step
// j it reduces its j-th edge.
//   single head (LH == G): lane k*G+j (which sits in group k) keeps the result, so after G steps
//     lane l holds edge l and the store is coalesced.
//   H heads (LH = lanes per head = (len/H)/4 < G): the reduction stops at LH lanes; the first lane
//     of every head subgroup stores out[e,h] directly.
template <int G, int U, int LH>
__global__ __launch_bounds__(256) void sddmm_chunk_kernel(int64_t n_chunks, const uint32_t* chunk_row,
                                                          const uint32_t* chunk_ebase,
                                                          const int64_t* rowptr, const uint32_t* col,
                                                          int len, int H, const float* grad,
                                                          const float* feat, float* out_e) {
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
      for (int o = LH / 2; o > 0; o >>= 1) p += __shfl_xor(p, o, 64);
      if constexpr (LH == G) {
        if (sl == j + u) res = p;
      } else {
        const int ei = gbase + j + u;  // edge index inside the chunk handled by this group now
        if (colok && (sl & (LH - 1)) == 0 && ei < n) out_e[(eb + ei) * H + sl / LH] = p;
      }
    }
  }
  if constexpr (LH == G) {
    if (lane < n) out_e[eb + lane] = res;
  }
}

// softmax backward per (row, head) (math_functions.cpp:496-514, closed form of the :497-504
// branch) + leaky-relu' (gat_aggregator.cpp:145).  Writes ds into scores[], g into gbuf[], and the
// row sum of g into rs[].
__global__ __launch_bounds__(256) void softmax_bwd_kernel(int64_t nv, int H, const int64_t* rowptr,
                                                          const float* p, const float* dp,
                                                          const float* temp, float eps,
                                                          float* scores, float* gbuf, float* rs) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  for (int h = 0; h < H; ++h) {
    float dot = 0.f;
    for (int64_t e = e0 + lane; e < e1; e += 64) dot += p[e * H + h] * dp[e * H + h];
    dot = wave_sum(dot);
    float gs = 0.f;
    for (int64_t e = e0 + lane; e < e1; e += 64) {
      const float pe = p[e * H + h], dpe = dp[e * H + h];
      const float x = pe * (1.0f - pe) * dpe;
      const float ds = x - (dot - pe * dpe) * pe;
      scores[e * H + h] = ds;
      const float ge = ds * (temp[e * H + h] > 0.0f ? 1.0f : eps);
      gbuf[e * H + h] = ge;
      gs += ge;
    }
    gs = wave_sum(gs);
    if (lane == 0) rs[row * H + h] = gs;
  }
}

// cs[v,h] = sum_{e in row v} g[rev[e],h]  == column sum of g (structurally symmetric graph)
__global__ __launch_bounds__(256) void colsum_kernel(int64_t nv, int H, const int64_t* rowptr,
                                                     const uint32_t* rev, const float* gbuf,
                                                     float* cs) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  for (int h = 0; h < H; ++h) {
    float s = 0.f;
    for (int64_t e = e0 + lane; e < e1; e += 64) s += gbuf[(int64_t)rev[e] * H + h];
    s = wave_sum(s);
    if (lane == 0) cs[row * H + h] = s;
  }
}

// partial[b][0][c] = sum_{v in strip b} rs[v, head(c)]*x[v][c]; partial[b][1][c] likewise with cs.
// 256 threads: thread t owns column (t % cw) of every (256/cw)-th row of the strip.
__global__ __launch_bounds__(256) void alpha_partial_kernel(int64_t nv, int len, int H, const float* x,
                                                            const float* rs, const float* cs,
                                                            int64_t rows_per_block, float* partial) {
  extern __shared__ float sm[];  // [2][256]
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < nv) ? r0 + rows_per_block : nv;
  float* out = partial + (int64_t)blockIdx.x * 2 * len;
  const int dh = len / H;
  for (int c0 = 0; c0 < len; c0 += 256) {
    const int cw = (len - c0 < 256) ? (len - c0) : 256;  // columns in this pass
    const int rpp = 256 / cw;                            // rows per pass (>=1)
    const int tc = threadIdx.x % cw, tr = threadIdx.x / cw;
    const int hd = (c0 + tc) / dh;
    float al = 0.f, ar = 0.f;
    if (tr < rpp) {
      for (int64_t v = r0 + tr; v < r1; v += rpp) {
        const float xv = x[v * (int64_t)len + c0 + tc];
        al += rs[v * H + hd] * xv;
        ar += cs[v * H + hd] * xv;
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

// one wave per output element (column c of lgrad or rgrad): lanes stride over the block partials, then a fixed-order
// wave sum (one thread per column walked 1024 partials serially: 0.25 ms of pure latency)
__global__ __launch_bounds__(256) void alpha_final_kernel(int nblocks, int len, const float* partial, float* lgrad,
                                                          float* rgrad) {
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);  // [0, 2 * len)
  if (w >= 2 * len) return;
  const int lane = threadIdx.x & 63;
  float s = 0.f;
  for (int b = lane; b < nblocks; b += 64) s += partial[(int64_t)b * 2 * len + w];
  s = wave_sum(s);
  if (lane == 0) (w < len ? lgrad : rgrad)[w < len ? w : w - len] = s;
}

__global__ void edge_gather_kernel(int64_t ne, int H, const uint32_t* rev, const float* in, float* out) {
  int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= ne * H) return;
  const int64_t e = t / H;
  const int h = (int)(t - e * H);
  out[t] = in[(int64_t)rev[e] * H + h];
}

// ---- head-vectorised variants (H in {2,4,8,16}): one lane owns ALL heads of an edge, so the
// [ne][H] arrays are read and written as contiguous 4*H-byte records (the runtime-H kernels above
// walk them with a 4*H-byte stride, 8x the traffic at H = 8) and row reductions run once per row.
template <int H>
struct HeadVec {
  float v[H];
  __device__ __forceinline__ void load(const float* p) {
    if constexpr (H % 4 == 0) {
#pragma unroll
      for (int k = 0; k < H / 4; ++k) {
        const f4 t = reinterpret_cast<const f4*>(p)[k];
        v[4 * k] = t[0]; v[4 * k + 1] = t[1]; v[4 * k + 2] = t[2]; v[4 * k + 3] = t[3];
      }
    } else {
#pragma unroll
      for (int k = 0; k < H; ++k) v[k] = p[k];
    }
  }
  __device__ __forceinline__ void store(float* p) const {
    if constexpr (H % 4 == 0) {
#pragma unroll
      for (int k = 0; k < H / 4; ++k) {
        f4 t = {v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
        reinterpret_cast<f4*>(p)[k] = t;
      }
    } else {
#pragma unroll
      for (int k = 0; k < H; ++k) p[k] = v[k];
    }
  }
};

// Row-owner kernels.  BLK = false: one wave per row, 4 rows per 256-thread block, rows longer than
// heavy_thr are left to the BLK = true launch: one 1024-thread workgroup per heavy row (longest
// first), partial reductions meet in LDS.  A row's edges are strided over the owner's lanes.
constexpr int ROW_BLK_WAVES = 16;

// How a row owner's threads cover the [edge][head] records: a thread handles HV = min(H, 4) heads of an edge
// (one 16-byte access at H >= 4) and LPE = H / HV neighbouring threads share an edge, so a wave instruction
// touches whole 128-byte lines (one lane per 32-byte record, as in the first version, touched every line twice
// and left the waves stalled on memory-instruction issue: SQ_WAIT_INST_ANY 55-73 %).
template <int H>
struct HeadSplit {
  static constexpr int HV = H >= 4 ? 4 : H;
  static constexpr int LPE = H / HV;
  static_assert(HV * LPE == H && (LPE & (LPE - 1)) == 0, "H must be 1, 2, 4, 8 or 16");
};

template <bool BLK, int LPE>
struct RowOwner {
  int64_t row, e0, e1;
  int tid, lane, wave;
  int sub;       // which HV-wide slice of an edge's heads this thread handles
  int etid;      // edge slot of this thread among the owner's threads
  int ethreads;  // edges the owner's threads cover per trip
  bool valid;
  __device__ __forceinline__ RowOwner(int64_t nv, const int64_t* rowptr, int heavy_thr,
                                      const uint32_t* row_list, const uint32_t* row_order) {
    lane = threadIdx.x & 63;
    wave = threadIdx.x >> 6;
    int nthreads;
    if constexpr (BLK) {
      row = row_list[row_order[blockIdx.x]];
      tid = threadIdx.x;
      nthreads = ROW_BLK_WAVES * 64;
      valid = true;
    } else {
      row = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave;
      tid = lane;
      nthreads = 64;
      valid = row < nv;
    }
    sub = tid % LPE;
    etid = tid / LPE;
    ethreads = nthreads / LPE;
    e0 = e1 = 0;
    if (valid) {
      e0 = rowptr[row];
      e1 = rowptr[row + 1];
      if (!BLK && heavy_thr > 0 && e1 - e0 > (int64_t)heavy_thr) valid = false;
    }
  }
};

// reductions over the lanes of a wave that handle the SAME head slice (lane % LPE equal)
template <int LPE>
__device__ __forceinline__ float slice_sum(float v) {
#pragma unroll
  for (int o = 32; o >= LPE; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int LPE>
__device__ __forceinline__ float slice_max(float v) {
#pragma unroll
  for (int o = 32; o >= LPE; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// sum over the row owner's threads of one head slice; lds: [ROW_BLK_WAVES][LPE][HV] floats (BLK only).  Fixed order.
template <int HV, int LPE, bool BLK>
__device__ __forceinline__ void owner_sum(HeadVec<HV>& v, int wave, int lane, int sub, float* lds) {
#pragma unroll
  for (int h = 0; h < HV; ++h) v.v[h] = slice_sum<LPE>(v.v[h]);
  if constexpr (BLK) {
    __syncthreads();  // lds may still be read from a previous reduction
    if (lane < LPE) {  // (lane == sub here)
#pragma unroll
      for (int h = 0; h < HV; ++h) lds[(wave * LPE + sub) * HV + h] = v.v[h];
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < HV; ++h) {
      float t = lds[sub * HV + h];
      for (int w = 1; w < ROW_BLK_WAVES; ++w) t += lds[(w * LPE + sub) * HV + h];
      v.v[h] = t;
    }
  }
}

// (m, d) = (running maximum, sum of exp(s - m)) per thread -> the row's (M, D) of the thread's head slice
template <int HV, int LPE, bool BLK>
__device__ __forceinline__ void owner_softmax_stats(HeadVec<HV>& m, HeadVec<HV>& d, int wave, int lane, int sub,
                                                    float* lds) {
#pragma unroll
  for (int h = 0; h < HV; ++h) {
    const float mw = slice_max<LPE>(m.v[h]);
    const float scaled = (m.v[h] == -INFINITY) ? 0.f : d.v[h] * expf(m.v[h] - mw);
    d.v[h] = slice_sum<LPE>(scaled);
    m.v[h] = mw;
  }
  if constexpr (BLK) {
    __syncthreads();
    if (lane < LPE) {
#pragma unroll
      for (int h = 0; h < HV; ++h) {
        lds[((wave * LPE + sub) * HV + h) * 2] = m.v[h];
        lds[((wave * LPE + sub) * HV + h) * 2 + 1] = d.v[h];
      }
    }
    __syncthreads();
#pragma unroll
    for (int h = 0; h < HV; ++h) {
      float M = -INFINITY;
      for (int w = 0; w < ROW_BLK_WAVES; ++w) M = fmaxf(M, lds[((w * LPE + sub) * HV + h) * 2]);
      float D = 0.f;
      for (int w = 0; w < ROW_BLK_WAVES; ++w) {
        const float mw = lds[((w * LPE + sub) * HV + h) * 2];
        if (mw != -INFINITY) D += lds[((w * LPE + sub) * HV + h) * 2 + 1] * expf(mw - M);
      }
      m.v[h] = M;
      d.v[h] = D;
    }
  }
}

// temp = sl[i] + sr[col]; s = leaky_relu(temp); norm = softmax_row(s)   (gat_aggregator.cpp:64-77).
// Rows that fit one edge per thread group stay in registers; longer rows take two passes: an online
// (max, sum) pass that writes temp, then norm = exp(s - M) / D from the re-read temp.  132 B per edge
// at H = 8 (4 col + 32 gathered + 32 temp + 32 re-read + 32 norm); `scores` is optional (+32 B).
template <int H, bool BLK>
__global__ __launch_bounds__(BLK ? ROW_BLK_WAVES * 64 : 256) void edge_softmax_v2_kernel(
    int64_t nv, const int64_t* rowptr, const uint32_t* col, const float* sl, const float* sr, float eps,
    float* temp, float* scores, float* norm, int heavy_thr, const uint32_t* row_list, const uint32_t* row_order) {
  constexpr int HV = HeadSplit<H>::HV, LPE = HeadSplit<H>::LPE;
  __shared__ float lds[BLK ? ROW_BLK_WAVES * H * 2 : 1];
  const RowOwner<BLK, LPE> o(nv, rowptr, heavy_thr, row_list, row_order);
  if (!o.valid || o.e0 == o.e1) return;
  const int hb = o.sub * HV;  // first head of this thread's slice
  HeadVec<HV> ssrc, m, d;
  ssrc.load(sl + o.row * H + hb);
  if (!BLK && o.e1 - o.e0 <= 64 / LPE) {
    const int64_t e = o.e0 + o.etid;
    const bool ok = e < o.e1;
    HeadVec<HV> t, s;
    t.load(sr + (int64_t)(ok ? col[e] : 0u) * H + hb);
#pragma unroll
    for (int h = 0; h < HV; ++h) {
      t.v[h] = ssrc.v[h] + t.v[h];
      s.v[h] = ok ? (t.v[h] > 0.0f ? t.v[h] : eps * t.v[h]) : -INFINITY;
      const float mx = slice_max<LPE>(s.v[h]);
      const float ex = ok ? expf(s.v[h] - mx) : 0.f;
      const float den = slice_sum<LPE>(ex);
      m.v[h] = ex / den;
    }
    if (ok) {
      if (temp) t.store(temp + e * H + hb);
      if (scores) s.store(scores + e * H + hb);
      m.store(norm + e * H + hb);
    }
    return;
  }
#pragma unroll
  for (int h = 0; h < HV; ++h) { m.v[h] = -INFINITY; d.v[h] = 0.f; }
  // EU edges per thread per trip: all column ids, then all gathers, then the arithmetic
  constexpr int EU = 4;
  for (int64_t e = o.e0 + o.etid; e < o.e1; e += (int64_t)EU * o.ethreads) {
    uint32_t c[EU];
    HeadVec<HV> t[EU];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int64_t eu = e + (int64_t)u * o.ethreads;
      c[u] = eu < o.e1 ? col[eu] : 0u;
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) t[u].load(sr + (int64_t)c[u] * H + hb);
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int64_t eu = e + (int64_t)u * o.ethreads;
      if (eu < o.e1) {
        HeadVec<HV> s;
#pragma unroll
        for (int h = 0; h < HV; ++h) {
          t[u].v[h] = ssrc.v[h] + t[u].v[h];
          s.v[h] = t[u].v[h] > 0.0f ? t[u].v[h] : eps * t[u].v[h];
          // online softmax statistics with one expf: ex = exp(-|s - m|)
          const bool up = s.v[h] > m.v[h];
          const float ex = __expf(up ? m.v[h] - s.v[h] : s.v[h] - m.v[h]);
          d.v[h] = up ? d.v[h] * ex + 1.0f : d.v[h] + ex;
          m.v[h] = up ? s.v[h] : m.v[h];
        }
        if (temp) t[u].store(temp + eu * H + hb);
        if (scores) s.store(scores + eu * H + hb);
      }
    }
  }
  owner_softmax_stats<HV, LPE, BLK>(m, d, o.wave, o.lane, o.sub, lds);
#pragma unroll
  for (int h = 0; h < HV; ++h) d.v[h] = 1.0f / d.v[h];
  for (int64_t e = o.e0 + o.etid; e < o.e1; e += (int64_t)EU * o.ethreads) {
    HeadVec<HV> t[EU];
    if (temp) {  // each thread re-reads only its own writes
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int64_t eu = e + (int64_t)u * o.ethreads;
        t[u].load(temp + (eu < o.e1 ? eu : o.e0) * H + hb);
      }
    } else {
      // no temp array asked for: the pre-activation score is formed again from the per-vertex dots (4 + 4H bytes
      // per edge out of the caches instead of 4H written and 4H read back through HBM); same sum, same bits
      uint32_t c[EU];
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int64_t eu = e + (int64_t)u * o.ethreads;
        c[u] = col[eu < o.e1 ? eu : o.e0];
      }
#pragma unroll
      for (int u = 0; u < EU; ++u) t[u].load(sr + (int64_t)c[u] * H + hb);
#pragma unroll
      for (int u = 0; u < EU; ++u) {
#pragma unroll
        for (int h = 0; h < HV; ++h) t[u].v[h] = ssrc.v[h] + t[u].v[h];
      }
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int64_t eu = e + (int64_t)u * o.ethreads;
      if (eu < o.e1) {
#pragma unroll
        for (int h = 0; h < HV; ++h) {
          const float sv = t[u].v[h] > 0.0f ? t[u].v[h] : eps * t[u].v[h];
          t[u].v[h] = __expf(sv - m.v[h]) * d.v[h];  // d holds 1/D here
        }
        t[u].store(norm + eu * H + hb);
      }
    }
  }
}

// softmax backward (math_functions.cpp:496-514, closed form of the :497-504 branch) + leaky-relu'
// (gat_aggregator.cpp:145): ds = p(1-p)dp - (dot - p dp) p with dot = sum_e p dp, g = ds * lrelu'(temp).
// Writes g into gbuf, the row sum of g into rs, ds into scores when asked.  DOT = true: the row's dot comes
// from the caller (rowdot[i,h] = <grad_i, forward output_i> on slice h -- the same sum regrouped by vertex),
// which makes this ONE pass over the edge arrays.
// RE = true: no temp array; the sign of the pre-activation score comes from sl[i] + sr[col] formed again.
template <int H, bool BLK, bool DOT, bool RE>
__global__ __launch_bounds__(BLK ? ROW_BLK_WAVES * 64 : 256) void softmax_bwd_v2_kernel(
    int64_t nv, const int64_t* rowptr, const float* p, const float* dp, const float* temp, float eps,
    const float* rowdot, float* scores, float* gbuf, int pack, float* rs, int heavy_thr,
    const uint32_t* row_list, const uint32_t* row_order, const uint32_t* col, const float* sl, const float* sr) {
  constexpr int HV = HeadSplit<H>::HV, LPE = HeadSplit<H>::LPE;
  __shared__ float lds[BLK ? ROW_BLK_WAVES * H : 1];
  const RowOwner<BLK, LPE> o(nv, rowptr, heavy_thr, row_list, row_order);
  if (!o.valid) return;
  const int hb = o.sub * HV;
  HeadVec<HV> dot, gs;
#pragma unroll
  for (int h = 0; h < HV; ++h) { dot.v[h] = 0.f; gs.v[h] = 0.f; }
  if constexpr (DOT) {
    dot.load(rowdot + o.row * H + hb);
  } else {
    for (int64_t e = o.e0 + o.etid; e < o.e1; e += o.ethreads) {
      HeadVec<HV> a, b;
      a.load(p + e * H + hb);
      b.load(dp + e * H + hb);
#pragma unroll
      for (int h = 0; h < HV; ++h) dot.v[h] += a.v[h] * b.v[h];
    }
    owner_sum<HV, LPE, BLK>(dot, o.wave, o.lane, o.sub, lds);
  }
  constexpr int EU = 4;
  HeadVec<HV> ssrc;
  if constexpr (RE) ssrc.load(sl + o.row * H + hb);
  for (int64_t e = o.e0 + o.etid; e < o.e1; e += (int64_t)EU * o.ethreads) {
    HeadVec<HV> a[EU], b[EU], t[EU];
    if constexpr (RE) {
      uint32_t c[EU];
#pragma unroll
      for (int u = 0; u < EU; ++u) {
        const int64_t eu = e + (int64_t)u * o.ethreads;
        c[u] = col[eu < o.e1 ? eu : o.e0];
      }
#pragma unroll
      for (int u = 0; u < EU; ++u) t[u].load(sr + (int64_t)c[u] * H + hb);
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int64_t eu = e + (int64_t)u * o.ethreads;
      const int64_t es = eu < o.e1 ? eu : o.e0;
      a[u].load(p + es * H + hb);
      b[u].load(dp + es * H + hb);
      if constexpr (!RE) t[u].load(temp + es * H + hb);
    }
    if constexpr (RE) {
#pragma unroll
      for (int u = 0; u < EU; ++u) {
#pragma unroll
        for (int h = 0; h < HV; ++h) t[u].v[h] = ssrc.v[h] + t[u].v[h];
      }
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int64_t eu = e + (int64_t)u * o.ethreads;
      if (eu < o.e1) {
        HeadVec<HV> ds, ge;
#pragma unroll
        for (int h = 0; h < HV; ++h) {
          const float x = a[u].v[h] * (1.0f - a[u].v[h]) * b[u].v[h];
          ds.v[h] = x - (dot.v[h] - a[u].v[h] * b[u].v[h]) * a[u].v[h];
          ge.v[h] = ds.v[h] * (t[u].v[h] > 0.0f ? 1.0f : eps);
          gs.v[h] += ge.v[h];
        }
        if (scores) ds.store(scores + eu * H + hb);
        if (pack) {  // (g, p) side by side: the column-sum pass fetches both with one random access
          ge.store(gbuf + eu * 2 * H + hb);
          a[u].store(gbuf + eu * 2 * H + H + hb);
        } else {
          ge.store(gbuf + eu * H + hb);
        }
      }
    }
  }
  owner_sum<HV, LPE, BLK>(gs, o.wave, o.lane, o.sub, lds);
  if (o.tid < LPE) gs.store(rs + o.row * H + hb);
}

// cs[v] = sum_{e in row v} g[rev[e]]  == column sum of g (structurally symmetric graph); optionally also
// pT[e] = p[rev[e]] (symmetric_csr_transpose of the attention, gat_aggregator.cpp:172-175); gbuf then holds
// (g, p) records written by softmax_bwd_v2_kernel
template <int H, bool BLK>
__global__ __launch_bounds__(BLK ? ROW_BLK_WAVES * 64 : 256) void colsum_v2_kernel(
    int64_t nv, const int64_t* rowptr, const uint32_t* rev, const float* gbuf, float* cs,
    float* pT, int heavy_thr, const uint32_t* row_list, const uint32_t* row_order) {
  constexpr int HV = HeadSplit<H>::HV, LPE = HeadSplit<H>::LPE;
  __shared__ float lds[BLK ? ROW_BLK_WAVES * H : 1];
  const RowOwner<BLK, LPE> o(nv, rowptr, heavy_thr, row_list, row_order);
  if (!o.valid) return;
  const int hb = o.sub * HV;
  HeadVec<HV> s;
#pragma unroll
  for (int h = 0; h < HV; ++h) s.v[h] = 0.f;
  constexpr int EU = 4;
  const int rec = (pT ? 2 : 1) * H;  // floats per edge record in gbuf
  for (int64_t e = o.e0 + o.etid; e < o.e1; e += (int64_t)EU * o.ethreads) {
    int64_t r[EU];
    HeadVec<HV> g[EU];
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      const int64_t eu = e + (int64_t)u * o.ethreads;
      r[u] = eu < o.e1 ? (int64_t)rev[eu] : -1;
    }
#pragma unroll
    for (int u = 0; u < EU; ++u) g[u].load(gbuf + (r[u] < 0 ? 0 : r[u]) * rec + hb);
#pragma unroll
    for (int u = 0; u < EU; ++u) {
      if (r[u] >= 0) {
#pragma unroll
        for (int h = 0; h < HV; ++h) s.v[h] += g[u].v[h];
      }
    }
    if (pT) {  // the transposed attention for the gradient aggregation, while rev[e] is at hand
#pragma unroll
      for (int u = 0; u < EU; ++u) g[u].load(gbuf + (r[u] < 0 ? 0 : r[u]) * rec + H + hb);  // same line as g
#pragma unroll
      for (int u = 0; u < EU; ++u)
        if (r[u] >= 0) g[u].store(pT + (e + (int64_t)u * o.ethreads) * H + hb);
    }
  }
  owner_sum<HV, LPE, BLK>(s, o.wave, o.lane, o.sub, lds);
  if (o.tid < LPE) s.store(cs + o.row * H + hb);
}

// The same column sums over the graph's ordered 64-edge chunk list (gaib_graph_ensure_chunks): a wave takes one chunk,
// fetches the (g, p) records of its reverse edges, writes the chunk's piece of pT and one partial sum per head; a second
// kernel adds a row's partials in row order.  Row by row, the 8192 rows in flight each sweep the whole 7 GB record array
// (the reverse edge of (v -> c) sits in row c's segment); chunk by chunk in column order, everything in flight points
// into the segments of one window of rows.
template <int H>
__global__ __launch_bounds__(256) void colsum_chunk_kernel(int64_t n_chunks, const uint32_t* chunk_row,
                                                           const uint32_t* chunk_ebase, const uint32_t* chunk_start,
                                                           const int64_t* rowptr, const uint32_t* rev, const float* gbuf,
                                                           float* partial, float* pT) {
  constexpr int HV = HeadSplit<H>::HV, LPE = HeadSplit<H>::LPE;
  constexpr int EPP = 64 / LPE;  // edges per pass; a 64-edge chunk takes LPE passes
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const int sub = lane % LPE, etid = lane / LPE, hb = sub * HV;
  const int64_t row = chunk_row[c];
  const int64_t eb = chunk_ebase[c];
  const int64_t rem = rowptr[row + 1] - eb;
  const int n = rem < 64 ? (int)rem : 64;
  const int rec = (pT ? 2 : 1) * H;
  int64_t r[LPE];
  HeadVec<HV> g[LPE], s;
#pragma unroll
  for (int h = 0; h < HV; ++h) s.v[h] = 0.f;
#pragma unroll
  for (int ps = 0; ps < LPE; ++ps) {
    const int ei = ps * EPP + etid;
    r[ps] = ei < n ? (int64_t)rev[eb + ei] : -1;
  }
#pragma unroll
  for (int ps = 0; ps < LPE; ++ps) g[ps].load(gbuf + (r[ps] < 0 ? 0 : r[ps]) * rec + hb);
#pragma unroll
  for (int ps = 0; ps < LPE; ++ps) {
    if (r[ps] >= 0) {
#pragma unroll
      for (int h = 0; h < HV; ++h) s.v[h] += g[ps].v[h];
    }
  }
  if (pT) {
#pragma unroll
    for (int ps = 0; ps < LPE; ++ps) g[ps].load(gbuf + (r[ps] < 0 ? 0 : r[ps]) * rec + H + hb);  // same line as g
#pragma unroll
    for (int ps = 0; ps < LPE; ++ps)
      if (r[ps] >= 0) g[ps].store(pT + (eb + ps * EPP + etid) * H + hb);
  }
#pragma unroll
  for (int h = 0; h < HV; ++h) s.v[h] = slice_sum<LPE>(s.v[h]);
  const int64_t slot = (int64_t)chunk_start[row] + (eb - rowptr[row]) / 64;
  if (lane < LPE) s.store(partial + slot * H + hb);
}

// cs[v, h] = sum of row v's chunk partials, in row order
__global__ void colsum_reduce_kernel(int64_t nv, int H, const uint32_t* chunk_start, const float* partial, float* cs) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= nv * H) return;
  const int64_t v = t / H;
  const int h = (int)(t - v * H);
  float s = 0.f;
  for (int64_t k = chunk_start[v]; k < chunk_start[v + 1]; ++k) s += partial[k * H + h];
  cs[t] = s;
}

// rowdot[v,h] = <a[v, slice h], b[v, slice h]>.  One wave per row.
__global__ __launch_bounds__(256) void rowdot_kernel(int64_t nv, int len, int H, const float* a,
                                                     const float* b, float* out) {
  int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const float* ar = a + row * (int64_t)len;
  const float* br = b + row * (int64_t)len;
  const int dh = len / H;
  if (H > 1 && dh <= 64 && (64 % dh) == 0) {  // as in vertex_dots_kernel
    for (int c0 = 0; c0 < len; c0 += 64) {
      const int c = c0 + lane;
      const bool ok = c < len;
      float s = ok ? ar[c] * br[c] : 0.f;
      for (int o = dh >> 1; o >= 1; o >>= 1) s += __shfl_xor(s, o, 64);
      if (ok && (lane % dh) == 0) out[row * H + c / dh] = s;
    }
    return;
  }
  for (int h = 0; h < H; ++h) {
    float s = 0.f;
    for (int c = h * dh + lane; c < (h + 1) * dh; c += 64) s += ar[c] * br[c];
    s = wave_sum(s);
    if (lane == 0) out[row * H + h] = s;
  }
}

// rec[v, h] = (rowdot[v, h], row maximum, 1 / row sum, 0): what the one-sweep backward needs about a COLUMN vertex beside
// its two rows, as one 16-byte record -- 128 B = one line per vertex at 8 heads, where rowdot [nv][H] and the forward's
// statistics [nv][H][2] were two tables, two lines and two load instructions per edge
__global__ void gat_rec_kernel(int64_t n, const float* rowdot, const float2* stats, f4* rec) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) {
    const float2 st = stats[t];
    rec[t] = f4{rowdot[t], st.x, st.y, 0.f};
  }
}

// ---- lane layout of the one-sweep kernels, by row width (round 5: len = 32, 64 and 128) -------------------------------
// A lane owns 4 columns of a row, so a row takes G = len / 4 lanes (8, 16, 32) and a wave works on NG = 64 / G edges at a
// time, G steps per 64-edge chunk.  The chunk's column ids are fetched ONCE, one or two per lane, laid out so that "the
// column of (my group, step t)" is a row-share DPP inside the lane's 16-lane row (no LDS, no bpermute):
//   G = 16: 4 groups = the 4 DPP rows.  Lane (row r, position p) holds edge p*4 + r; (group r, step t) = edge t*4 + r.
//   G =  8: 8 groups, two per DPP row (half hg = 0, 1).  Same holding; (group (r, hg), step t) = edge (2t + hg)*4 + r:
//           two row-shares with constant lane numbers and a select on hg.
//   G = 32: 2 groups of two DPP rows.  (group g, step t) = edge t*2 + g; every lane of the group's two rows holds TWO ids,
//           edges p*2 + g and (p + 16)*2 + g (both rows load the same lines), so steps 0..15 come from the first, 16..31
//           from the second.
// In every layout the edges of step t are t*NG .. t*NG + NG - 1: a chunk of n edges takes ceil(n / NG) steps.
template <int G>
struct ChunkLanes {
  static_assert(G == 8 || G == 16 || G == 32, "len = 32, 64 or 128");
  static constexpr int NG = 64 / G;
  // the edge whose id this lane holds (which = 1: the second one, G = 32 only)
  static __device__ __forceinline__ int held_edge(int lane, int which) {
    const int r = lane >> 4, p = lane & 15;
    if constexpr (G == 32) return (p + 16 * which) * 2 + (r >> 1);
    else return p * 4 + r;
  }
  // the edge of (this lane's group, step t)
  static __device__ __forceinline__ int step_edge(int lane, int t) {
    if constexpr (G == 32) return t * 2 + (lane >> 5);
    else if constexpr (G == 16) return t * 4 + (lane >> 4);
    else return (2 * t + ((lane >> 3) & 1)) * 4 + (lane >> 4);
  }
  // its held value (a column id, a reverse-edge id): t is a compile-time constant after unrolling
  static __device__ __forceinline__ int step_value(int v0, int v1, int lane, int t) {
    if constexpr (G == 32) return t < 16 ? row_lane(v0, t) : row_lane(v1, t - 16);
    else if constexpr (G == 16) return row_lane(v0, t);
    else {
      const int a = row_lane(v0, 2 * t), b = row_lane(v0, 2 * t + 1);
      return ((lane >> 3) & 1) ? b : a;
    }
  }
};
// lanes_sum over an aligned group of up to 32 lanes (LH = 32: one head over a 128-wide row)
template <int LH>
__device__ __forceinline__ float lanes_sum_w(float v) {
  if constexpr (LH <= 16) return lanes_sum<LH>(v);
  else {
    v = lanes_sum<16>(v);
    return v + __shfl_xor(v, 16, 64);
  }
}

// ---- the whole edge side of GAT backward in ONE pass over the ordered 64-edge chunk list ---------------------------
// GAT_Aggregator::d_aggregate (gat_aggregator.cpp:99-200) is four sweeps over the edges: SDDMM dp_e = <grad_i, h_c>;
// softmax backward + leaky-relu' -> g_e, with the row sums rs and the column sums cs of g for the alpha gradients;
// the transpose pT_e = p[rev e]; the aggregation out_i = sum_e pT_e grad_c.  Staged (the kernels above) they move
// ~340 B per edge through HBM at 8 heads (dp written and read twice, (g, p) records written and re-read through rev,
// pT written and read).  Everything a row needs about its edge e = (i -> c) and the reverse edge (c -> i) follows from
// per-vertex quantities and ONE attention value each:
//   dp_e  = <grad_i, h_c>        g_e  = f(p_e,  dp_e,  rowdot_i, sl_i + sr_c)     -> rs_i += g_e
//   dp_r  = <grad_c, h_i>        g_r  = f(p_r,  dp_r,  rowdot_c, sl_c + sr_i)     -> cs_i += g_r   (p_r = p[rev e])
//   out_i += p_r * grad_c
// with rowdot_v = <grad_v, forward output_v> = sum_e p_e dp_e of row v (the one-pass form of softmax_bwd_v2_kernel) and
// f(p, dp, dot, t) = (p (1 - p) dp - (dot - p dp) p) * (t > 0 ? 1 : eps).  So one sweep gathers the rows h_c and grad_c
// (the two gathers SDDMM and the aggregation did separately), reads p_e (linear) and p_r (one random 4H-byte access),
// and writes nothing per edge: ~4 + 4 + 4H + 4H bytes per edge through HBM next to the two cache-resident row gathers.
// Chunk by chunk in column-block order like sddmm_chunk_kernel / spmm_chunk_kernel (rows in flight gather from one
// window of the tables); per chunk a partial output row and partial rs / cs, added per row in chunk order by
// gat_fused_reduce_kernel: deterministic, no atomics.
//
// Lanes: group k = lane / G owns edges k*G .. k*G+G-1 of the chunk, lane sl = lane % G owns 4 columns (head = sl / LH,
// LH = G / H lanes per head).  EVERYTHING is fetched in that layout, U edges per group in flight: the two rows (16 B per
// lane), rowdot of the column vertex and the two attention values p_e, p_r (4 B each), so the only dependent step is
// col / rev -> gathers, as in spmm_chunk_kernel.  A first version kept the per-(edge, head)
// scalars one EDGE per lane (H-vectors) and met the column layout through LDS: 64 different lines per wave instruction
// for each of five H-vector fetches, a dependent post-phase and 178 -> 142 VGPRs made it 10.5 ms at the reddit shape
// against 12.3 ms for the staged kernels.
// RECOMP: the attention values are not read but formed again from the forward sweep's row statistics
// (gat_fwd_fused_chunk_kernel): p_e = exp(lrelu(sl_i + sr_c) - M_i) / S_i and p_r = exp(lrelu(sl_c + sr_i) - M_c) / S_c
// with stats[v][h] = (M, 1/S) -- one 8-B gather from a 15 MB table instead of 4 B linear + 4 B random + rev per edge, and
// no [ne][H] array exists at all.
// T[v] = [h_v (len) | grad_v (len) | rec_v (4 H)]: what the backward sweep gathers per edge, as ONE row per vertex
__global__ __launch_bounds__(256) void gat_interleave_kernel(int64_t nv, int len4, int H, const f4* feat, const f4* grad,
                                                             const f4* rec, f4* T) {
  const int ldt4 = 2 * len4 + H;
  const int64_t total = nv * ldt4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t v = i / ldt4;
    const int k = (int)(i - v * ldt4);
    T[i] = k < len4 ? feat[v * len4 + k] : (k < 2 * len4 ? grad[v * len4 + k - len4] : rec[v * H + k - 2 * len4]);
  }
}

template <int G, int H, int U, bool RECOMP>
__global__ __launch_bounds__(256) void gat_bwd_fused_chunk_kernel(
    int64_t n_chunks, const uint32_t* chunk_row, const uint32_t* chunk_ebase, const uint32_t* chunk_start,
    const int64_t* rowptr, const uint32_t* col, const uint32_t* rev, int len, const float* feat, const float* grad,
    const float* p, const float2* stats, const float* rowdot, const float* alpha_l, const float* alpha_r, float eps,
    float* out_partial, float* rc_partial, const f4* rec, int phase, uint32_t own_cols, int ld, int rec_ld, int per_xcd) {
  // rec (RECOMP): (rowdot, row maximum, 1 / row sum) per (vertex, head) as one 16-byte record, see gat_rec_kernel
  // ld / rec_ld: row strides of the feat / grad tables (floats) and of the record table (16-byte records): len and H for
  // three separate tables; 2 len + 4 H and that / 4 when the three live INTERLEAVED, one [h | grad | records] row per vertex
  // (gat_interleave_kernel) -- one contiguous 640-B region per edge instead of three.
  // per_xcd > 0: workgroups are dealt to the XCDs round robin (XCD = blockIdx & 7); XCD x then walks the CONTIGUOUS range
  // [x per_xcd, (x + 1) per_xcd) of the column-block-ordered chunk list, so each L2 sees its own eighth of the columns
  // instead of all eight L2s caching the same window.
  constexpr int LH = G / H;  // lanes per head
  using CL = ChunkLanes<G>;
  constexpr int NG = CL::NG;
  int64_t blk = blockIdx.x;
  if (per_xcd > 0) blk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t c = blk * 4 + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const int sl = lane & (G - 1), gbase = lane & ~(G - 1);
  const int64_t row = chunk_row[c];
  const int64_t eb = chunk_ebase[c];
  const int64_t rb = rowptr[row];
  const int64_t rem = rowptr[row + 1] - eb;
  const int n = rem < 64 ? (int)rem : 64;

  // step t of lane group g is edge t * NG + g (ChunkLanes): a chunk of n edges takes ceil(n / NG) steps whatever n is (with
  // group g on edges 16 g .. 16 g + 15, the 17-edge tail of a row took all 16).  The lanes fetch the chunk's column ids
  // once, laid out so that "the column of my group's step t" is a DPP row-share.
  const int my_e = CL::held_edge(lane, 0), my_e1 = CL::held_edge(lane, 1);
  const int64_t el = eb + (my_e < n ? my_e : 0);
  const uint32_t cl = col[el];
  uint32_t cl1 = 0;
  int64_t el1 = eb;
  if constexpr (G == 32) {
    el1 = eb + (my_e1 < n ? my_e1 : 0);
    cl1 = col[el1];
  }
  // a rank's rows over [owned | halo] columns (a row keeps the global edge order, so halo ids sit on both sides of the
  // owned ones): phase 0 sweeps the chunks that touch owned columns only -- while the halo rows are still on the wire --,
  // phase 1 the others; -1: all
  if (phase >= 0 && ((__ballot((my_e < n && cl >= own_cols) || (G == 32 && my_e1 < n && cl1 >= own_cols)) == 0) != (phase == 0))) return;
  uint32_t rl = 0, rl1 = 0;
  if constexpr (!RECOMP) {
    rl = rev[el];
    if constexpr (G == 32) rl1 = rev[el1];
  }
  const int coff = sl * 4;  // len == 4 * G
  const int head = sl / LH;
  const f4 gi = *reinterpret_cast<const f4*>(grad + row * (int64_t)ld + coff);
  const f4 hi = *reinterpret_cast<const f4*>(feat + row * (int64_t)ld + coff);
  // the per-vertex dots a_l.h_v, a_r.h_v are formed again from the gathered rows (4 FMAs + the head's shuffle each)
  // instead of being gathered: only rowdot, which needs the vertex's forward output, comes from a table -- [nv][H]
  // floats, small enough for the L2, where a (rowdot, sl, sr) record per (vertex, head) cost a 128-B line per edge
  const f4 al4 = *reinterpret_cast<const f4*>(alpha_l + coff);
  const f4 ar4 = *reinterpret_cast<const f4*>(alpha_r + coff);
  // (instruction count is what bounds this kernel -- 2 000 VALU / LDS instructions per 64-edge chunk, 1 720 chunks per
  // SIMD: 5.7 ms of issue at the reddit shape before a byte moves -- so: dot products as FMA chains, sums over a head's
  // lanes and "column id of edge j" by DPP instead of ds_bpermute, one fast exponential per attention value)
  auto d4 = [](const f4& a, const f4& b) {
    return __builtin_fmaf(a[3], b[3], __builtin_fmaf(a[2], b[2], __builtin_fmaf(a[1], b[1], a[0] * b[0])));
  };
  const float sl_i = lanes_sum_w<LH>(d4(al4, hi));
  const float sr_i = lanes_sum_w<LH>(d4(ar4, hi));
  float rd_i;
  float2 st_i = {0.f, 0.f};
  if constexpr (RECOMP) {
    const f4 ri = rec[row * rec_ld + head];
    rd_i = ri[0];
    st_i = float2{ri[1], ri[2]};
  } else {
    rd_i = rowdot[row * H + head];
  }
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  float s_e = 0.f, s_r = 0.f;
#pragma unroll
  for (int j = 0; j < G; j += U) {
    if (j * NG >= n) break;  // (wave-uniform: no edge of the chunk is left for any group)
    f4 xg[U], xh[U];
    float pe[U], pr[U], rd[U];
    float2 stc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int ei = CL::step_edge(lane, j + u);
      const uint32_t cj = (uint32_t)CL::step_value((int)cl, (int)cl1, lane, j + u);
      xg[u] = *reinterpret_cast<const f4*>(grad + (int64_t)cj * ld + coff);
      xh[u] = *reinterpret_cast<const f4*>(feat + (int64_t)cj * ld + coff);
      if constexpr (RECOMP) {
        const f4 rc = rec[(int64_t)cj * rec_ld + head];
        rd[u] = rc[0];
        stc[u] = float2{rc[1], rc[2]};
      } else {
        rd[u] = rowdot[(int64_t)cj * H + head];
        const uint32_t rj = (uint32_t)CL::step_value((int)rl, (int)rl1, lane, j + u);
        pe[u] = p[(eb + (ei < n ? ei : 0)) * H + head];
        pr[u] = p[(int64_t)rj * H + head];
      }
    }
    __builtin_amdgcn_sched_barrier(0);  // all loads of the batch are issued before the first one is consumed
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = CL::step_edge(lane, j + u) < n;
      const float dpe = lanes_sum_w<LH>(d4(gi, xh[u]));
      const float dpr = lanes_sum_w<LH>(d4(xg[u], hi));
      const float sl_c = lanes_sum_w<LH>(d4(al4, xh[u]));
      const float sr_c = lanes_sum_w<LH>(d4(ar4, xh[u]));
      const float t_e = sl_i + sr_c, t_r = sl_c + sr_i;  // pre-activation scores of (i -> c) and (c -> i)
      float a, b;
      if constexpr (RECOMP) {
        a = __expf((t_e > 0.0f ? t_e : eps * t_e) - st_i.x) * st_i.y;
        b = __expf((t_r > 0.0f ? t_r : eps * t_r) - stc[u].x) * stc[u].y;
      } else {
        a = pe[u];
        b = pr[u];
      }
      const float dse = a * (1.0f - a) * dpe - (rd_i - a * dpe) * a;
      const float dsr = b * (1.0f - b) * dpr - (rd[u] - b * dpr) * b;
      const float ge = dse * (t_e > 0.0f ? 1.0f : eps);  // leaky-relu' at the score of (i -> c)
      const float gr = dsr * (t_r > 0.0f ? 1.0f : eps);  //              at the score of (c -> i)
      if (live) {  // (lanes past the end of a short chunk looked at the chunk's first edge: nothing of it is added)
        s_e += ge;
        s_r += gr;
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_fmaf(b, xg[u][k], acc[k]);
      }
    }
  }
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
    s_e += __shfl_xor(s_e, o, 64);
    s_r += __shfl_xor(s_r, o, 64);
  }
  const int64_t slot = (int64_t)chunk_start[row] + (eb - rb) / 64;
  if (gbase == 0) {
    *reinterpret_cast<f4*>(out_partial + slot * len + coff) = acc;
    if ((sl & (LH - 1)) == 0) {
      rc_partial[slot * 2 * H + head] = s_e;      // partial row sum of g    (-> alpha_l gradient)
      rc_partial[slot * 2 * H + H + head] = s_r;  // partial column sum of g (-> alpha_r gradient)
    }
  }
}

// ---- the same sweep on a VALU diet (round 6) ----------------------------------------------------------------------------
// Counters of gat_bwd_fused_chunk_kernel<16, 8, 4, true> at the reddit shape (profiles/r06/gat_bwd_sq.json): 2.48 G VALU
// instructions per launch = 1 324 per 64-edge chunk, SQ_ACTIVE_INST_VALU = 2.53 G quad-cycles -- 4.1 ms of pure VALU issue on
// 1 024 SIMDs at 2.4 GHz, which IS the 4.4-4.7 ms the sweep takes with every gather served by the L2 (gat_l2_ceiling.py):
// the kernel is VALU-bound.  Its ISA shows where the instructions go: hipcc packs the e-side and the r-side chains into
// v_pk_*_f32 pairs, and 461 of the 1 389 VALU instructions are v_mov_b32 marshalling operands into aligned register pairs;
// the sums over a head's lanes are v_mov_b32_dpp + add (the packed add cannot take a DPP operand) behind a zero-initialising
// move each; three 64-bit row addresses per edge cost a v_mad_u64_u32 + v_lshl_add_u64 each.  Here the data layout makes the
// pairs natural instead:
//   * the table row of a vertex interleaves h and grad ELEMENT by element: lane sl reads (h0 g0 h1 g1 | h2 g2 h3 g3), so
//     (h_k, g_k) is an aligned register pair as loaded;
//   * chain P = (dpe, dpr) = sum_k (grad_i[k], h_i[k]) * (h_c[k], grad_c[k]) and chain Q = (sr_c, sl_c) = sum_k (a_r[k], a_l[k]) *
//     (h_c[k], h_c[k]) are four packed multiply-adds each, in d4()'s order of additions (same bits), with loop-invariant left
//     operands; (t_e, t_r) = (sl_i, sr_i) + Q and everything downstream stays in pairs without a move;
//   * the sums over a head's lanes are v_add_f32_dpp, one instruction per value and stage (inline: the compiler does not fold a
//     DPP move into an add whose other operand is not the identity);
//   * one 32-bit byte offset per edge addresses all three loads of its row (tables below 4 GB; else the kernel above);
//   * the softmax backward in its short form g = p (dp - rowdot) -- the reference's p (1 - p) dp - (rowdot - p dp) p
//     (math_functions.cpp:496-514) multiplied out; one rounding fewer per term, not the same bits.
// ~41 VALU instructions per edge step instead of ~80 (898 against 1 389 in the kernel's ISA at 8 heads x 8).  RECOMP form only
// (the attention is formed again from the row statistics), heads of at most 16 lanes.
// MEASURED (reddit shape, 8 heads x 8, scripts/gat_l2_ceiling.py, profiles/r06/gat_pk_*): with every gather served by the L2
// (column ids >> 8) the backward call drops from 4.31 to 3.95 ms -- and at the REAL column ids it does not move: 6.27 against
// 6.25 ms.  There the sweep draws 43 GB per launch through the L2 -> fabric boundary at 7 TB/s (0.82 of the cache-resident gather
// rate, L2 hit rate 0.42): the VALU work had been hiding under the gathers all along.  So the kernel is an OPTION (gat_bwd_pk = 1),
// off by default -- it costs the table build (0.06 ms at the reddit shape, 0.45 ms at the products shape) and buys nothing where
// the tables do not fit the L2s; tests/test_gpu_ops.py runs both.
typedef float f2 __attribute__((ext_vector_type(2)));

// T2[v] = [(h0 g0 h1 g1 ...) 2 len | records 4 H]: h and grad of vertex v element by element, then its (rowdot, M, 1/S, 0) records
__global__ __launch_bounds__(256) void gat_interleave_pairs_kernel(int64_t nv, int len4, int H, const f4* feat, const f4* grad,
                                                                   const f4* rec, f4* T) {
  const int ldt4 = 2 * len4 + H;
  const int64_t total = nv * ldt4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t v = i / ldt4;
    const int k = (int)(i - v * ldt4);
    if (k < 2 * len4) {
      const f4 h = feat[v * len4 + (k >> 1)], g = grad[v * len4 + (k >> 1)];
      T[i] = (k & 1) ? f4{h[2], g[2], h[3], g[3]} : f4{h[0], g[0], h[1], g[1]};
    } else {
      T[i] = rec[v * H + k - 2 * len4];
    }
  }
}

// a, b, c, d <- their sums over the aligned group of LH lanes (LH = 1 .. 16 inside a 16-lane row); every lane gets them.
// v_add_f32_dpp reads its permuted operand through the DPP path: a VGPR written by the VALU instruction right before needs two
// wait states there, and the compiler's hazard recogniser does not look inside inline assembly -- hence the leading s_nop; the
// later stages read registers written four instructions earlier.
template <int LH>
__device__ __forceinline__ void lanes_sum4_dpp(float& a, float& b, float& c, float& d) {
  static_assert(LH == 1 || LH == 2 || LH == 4 || LH == 8 || LH == 16, "aligned power-of-two groups inside a 16-lane row");
#define GAIB_DPP4(CTRL)                                                                   \
  asm volatile("s_nop 1\n\t"                                                               \
               "v_add_f32_dpp %0, %0, %0 " CTRL " row_mask:0xf bank_mask:0xf\n\t"          \
               "v_add_f32_dpp %1, %1, %1 " CTRL " row_mask:0xf bank_mask:0xf\n\t"          \
               "v_add_f32_dpp %2, %2, %2 " CTRL " row_mask:0xf bank_mask:0xf\n\t"          \
               "v_add_f32_dpp %3, %3, %3 " CTRL " row_mask:0xf bank_mask:0xf"              \
               : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
  if constexpr (LH >= 2) GAIB_DPP4("quad_perm:[1,0,3,2]");
  if constexpr (LH >= 4) GAIB_DPP4("quad_perm:[2,3,0,1]");
  if constexpr (LH >= 8) GAIB_DPP4("row_half_mirror");
  if constexpr (LH >= 16) GAIB_DPP4("row_mirror");
#undef GAIB_DPP4
}

template <int G, int H, int U>
__global__ __launch_bounds__(256) void gat_bwd_fused_pk_kernel(int64_t n_chunks, const uint32_t* chunk_row, const uint32_t* chunk_ebase,
                                                               const uint32_t* chunk_start, const int64_t* rowptr, const uint32_t* col,
                                                               int len, const float* T, const float* alpha_l, const float* alpha_r,
                                                               float eps, float* out_partial, float* rc_partial, int per_xcd) {
  constexpr int LH = G / H;  // lanes per head
  static_assert(LH <= 16, "a head's lanes sit inside one 16-lane DPP row");
  using CL = ChunkLanes<G>;
  constexpr int NG = CL::NG;
  int64_t blk = blockIdx.x;
  if (per_xcd > 0) blk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t c = blk * 4 + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const int sl = lane & (G - 1), gbase = lane & ~(G - 1);
  const int64_t row = chunk_row[c];
  const int64_t eb = chunk_ebase[c];
  const int64_t rb = rowptr[row];
  const int64_t rem = rowptr[row + 1] - eb;
  const int n = rem < 64 ? (int)rem : 64;
  const int my_e = CL::held_edge(lane, 0), my_e1 = CL::held_edge(lane, 1);
  const uint32_t cl = col[eb + (my_e < n ? my_e : 0)];
  uint32_t cl1 = 0;
  if constexpr (G == 32) cl1 = col[eb + (my_e1 < n ? my_e1 : 0)];
  const int head = sl / LH;
  const uint32_t ldb = (uint32_t)(2 * len + 4 * H) * 4u;        // bytes of a table row
  const uint32_t lane_off = (uint32_t)sl * 32u;                 // this lane's (h0 g0 h1 g1 h2 g2 h3 g3)
  const uint32_t rec_off = (uint32_t)(2 * len) * 4u + (uint32_t)head * 16u;
  const char* Tb = reinterpret_cast<const char*>(T);
  auto row_at = [&](uint32_t v, uint32_t off) { return *reinterpret_cast<const f4*>(Tb + (size_t)(v * ldb + off)); };
  const uint32_t ri = (uint32_t)row;
  const f4 qi0 = row_at(ri, lane_off), qi1 = row_at(ri, lane_off + 16u), reci = row_at(ri, rec_off);
  const f4 al4 = *reinterpret_cast<const f4*>(alpha_l + sl * 4);
  const f4 ar4 = *reinterpret_cast<const f4*>(alpha_r + sl * 4);
  // loop-invariant left operands of the two chains: (grad_i[k], h_i[k]) and (a_r[k], a_l[k])
  const f2 GH[4] = {{qi0[1], qi0[0]}, {qi0[3], qi0[2]}, {qi1[1], qi1[0]}, {qi1[3], qi1[2]}};
  const f2 RL[4] = {{ar4[0], al4[0]}, {ar4[1], al4[1]}, {ar4[2], al4[2]}, {ar4[3], al4[3]}};
  // (sl_i, sr_i): the row's own dots, formed like the columns' below
  f2 SI;
  {
    float a = __builtin_fmaf(al4[3], qi1[2], __builtin_fmaf(al4[2], qi1[0], __builtin_fmaf(al4[1], qi0[2], al4[0] * qi0[0])));
    float b = __builtin_fmaf(ar4[3], qi1[2], __builtin_fmaf(ar4[2], qi1[0], __builtin_fmaf(ar4[1], qi0[2], ar4[0] * qi0[0])));
    float z0 = 0.f, z1 = 0.f;
    lanes_sum4_dpp<LH>(a, b, z0, z1);
    SI = f2{a, b};
  }
  const float rd_i = reci[0], m_i = reci[1], is_i = reci[2];
  const f2 eps2 = {eps, eps};
  f4 acc = {0.f, 0.f, 0.f, 0.f};
  f2 S = {0.f, 0.f};  // (partial row sum of g, partial column sum of g)
#pragma unroll
  for (int j = 0; j < G; j += U) {
    if (j * NG >= n) break;  // (wave-uniform: no edge of the chunk is left for any group)
    f4 q0[U], q1[U], rc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t cj = (uint32_t)CL::step_value((int)cl, (int)cl1, lane, j + u);
      const uint32_t base = cj * ldb;
      q0[u] = *reinterpret_cast<const f4*>(Tb + (size_t)(base + lane_off));
      q1[u] = *reinterpret_cast<const f4*>(Tb + (size_t)(base + lane_off + 16u));
      rc[u] = *reinterpret_cast<const f4*>(Tb + (size_t)(base + rec_off));
    }
    __builtin_amdgcn_sched_barrier(0);  // all loads of the batch are issued before the first one is consumed
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool live = CL::step_edge(lane, j + u) < n;
      const f2 e0 = {q0[u][0], q0[u][1]}, e1 = {q0[u][2], q0[u][3]}, e2 = {q1[u][0], q1[u][1]}, e3 = {q1[u][2], q1[u][3]};
      // P = (dpe, dpr) = (<grad_i, h_c>, <h_i, grad_c>)
      f2 P = GH[0] * e0;
      P = __builtin_elementwise_fma(GH[1], e1, P);
      P = __builtin_elementwise_fma(GH[2], e2, P);
      P = __builtin_elementwise_fma(GH[3], e3, P);
      // Q = (sr_c, sl_c) = (<a_r, h_c>, <a_l, h_c>)
      f2 Q = RL[0] * f2{e0[0], e0[0]};
      Q = __builtin_elementwise_fma(RL[1], f2{e1[0], e1[0]}, Q);
      Q = __builtin_elementwise_fma(RL[2], f2{e2[0], e2[0]}, Q);
      Q = __builtin_elementwise_fma(RL[3], f2{e3[0], e3[0]}, Q);
      float dpe = P[0], dpr = P[1], src = Q[0], slc = Q[1];
      lanes_sum4_dpp<LH>(dpe, dpr, src, slc);
      const f2 Tt = SI + f2{src, slc};  // pre-activation scores of (i -> c) and (c -> i)
      const f2 Tm = eps2 * Tt;
      const bool pe = Tt[0] > 0.0f, pr = Tt[1] > 0.0f;
      const float le = pe ? Tt[0] : Tm[0], lr = pr ? Tt[1] : Tm[1];
      const float a = __expf(le - m_i) * is_i;
      const float b = __expf(lr - rc[u][1]) * rc[u][2];
      // g = p (dp - rowdot) * leaky-relu'
      const f2 Gv = f2{a * (dpe - rd_i), b * (dpr - rc[u][0])} * f2{pe ? 1.0f : eps, pr ? 1.0f : eps};
      if (live) {  // (lanes past the end of a short chunk looked at the chunk's first edge: nothing of it is added)
        S += Gv;
        acc[0] = __builtin_fmaf(b, e0[1], acc[0]);
        acc[1] = __builtin_fmaf(b, e1[1], acc[1]);
        acc[2] = __builtin_fmaf(b, e2[1], acc[2]);
        acc[3] = __builtin_fmaf(b, e3[1], acc[3]);
      }
    }
  }
  float s_e = S[0], s_r = S[1];
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] += __shfl_xor(acc[k], o, 64);
    s_e += __shfl_xor(s_e, o, 64);
    s_r += __shfl_xor(s_r, o, 64);
  }
  const int64_t slot = (int64_t)chunk_start[row] + (eb - rb) / 64;
  if (gbase == 0) {
    *reinterpret_cast<f4*>(out_partial + slot * len + sl * 4) = acc;
    if ((sl & (LH - 1)) == 0) {
      rc_partial[slot * 2 * H + head] = s_e;      // partial row sum of g    (-> alpha_l gradient)
      rc_partial[slot * 2 * H + H + head] = s_r;  // partial column sum of g (-> alpha_r gradient)
    }
  }
}

// ---- forward in ONE sweep: scores, edge softmax and aggregation over the ordered chunk list ---------------------------
// GAT_Aggregator::aggregate (gat_aggregator.cpp:57-97) staged = per-vertex dots, a row-owner pass writing p [ne][H]
// (two sweeps over long rows), then the aggregation reading p.  Here a chunk's wave gathers the rows h_c once, forms
// sr_c = a_r . h_c from the gathered row, t = leaky_relu(sl_i + sr_c), and keeps an ONLINE softmax per lane group:
// running maximum m, running sum s of exp(t - m) and the running weighted row sum, rescaled by exp(m_old - m_new) when
// the maximum moves; the four groups and then the row's chunks are combined the same way (gat_fwd_reduce_kernel):
//   out_i = sum_c exp(m_c - M) acc_c / S,   S = sum_c exp(m_c - M) s_c,   M = max_c m_c
// which is the reference's max-subtracted softmax (math_functions.cpp:485-494) up to fp32 rounding.  Nothing per edge is
// written: backward forms p again from stats[v][h] = (M, 1/S) (gat_bwd_fused_chunk_kernel<RECOMP>).
constexpr float GAT_NEG = -1.0e30f;  // "no edge yet": finite, so exp(NEG - m) = 0 and NEG - NEG = 0 (not NaN)

template <int G, int H, int U>
__global__ __launch_bounds__(256) void gat_fwd_fused_chunk_kernel(
    int64_t n_chunks, const uint32_t* chunk_row, const uint32_t* chunk_ebase, const uint32_t* chunk_start,
    const int64_t* rowptr, const uint32_t* col, int len, const float* feat, const float* alpha_l, const float* alpha_r,
    float eps, float* out_partial, float2* ms_partial, int phase, uint32_t own_cols, int per_xcd) {
  constexpr int LH = G / H;
  using CL = ChunkLanes<G>;
  constexpr int NG = CL::NG;
  int64_t blk = blockIdx.x;  // (per_xcd: see gat_bwd_fused_chunk_kernel)
  if (per_xcd > 0) blk = (int64_t)(blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  const int64_t c = blk * 4 + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const int sl = lane & (G - 1), gbase = lane & ~(G - 1);
  const int64_t row = chunk_row[c];
  const int64_t eb = chunk_ebase[c];
  const int64_t rb = rowptr[row];
  const int64_t rem = rowptr[row + 1] - eb;
  const int n = rem < 64 ? (int)rem : 64;

  // step t of lane group g = edge t * NG + g; the column ids once, in DPP reach (ChunkLanes, see the backward kernel)
  const int my_e = CL::held_edge(lane, 0), my_e1 = CL::held_edge(lane, 1);
  const uint32_t cl = col[eb + (my_e < n ? my_e : 0)];
  uint32_t cl1 = 0;
  if constexpr (G == 32) cl1 = col[eb + (my_e1 < n ? my_e1 : 0)];
  if (phase >= 0 && ((__ballot((my_e < n && cl >= own_cols) || (G == 32 && my_e1 < n && cl1 >= own_cols)) == 0) != (phase == 0))) return;
  const int coff = sl * 4;
  const int head = sl / LH;
  const f4 hi = *reinterpret_cast<const f4*>(feat + row * (int64_t)len + coff);
  const f4 al4 = *reinterpret_cast<const f4*>(alpha_l + coff);
  const f4 ar4 = *reinterpret_cast<const f4*>(alpha_r + coff);
  auto d4 = [](const f4& a, const f4& b) {
    return __builtin_fmaf(a[3], b[3], __builtin_fmaf(a[2], b[2], __builtin_fmaf(a[1], b[1], a[0] * b[0])));
  };
  const float sl_i = lanes_sum_w<LH>(d4(al4, hi));  // (DPP sums / broadcasts, FMA chains: see the backward kernel)
  float m = GAT_NEG, ssum = 0.f;
  f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int j = 0; j < G; j += U) {
    if (j * NG >= n) break;  // (wave-uniform)
    f4 xh[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const uint32_t cj = (uint32_t)CL::step_value((int)cl, (int)cl1, lane, j + u);
      xh[u] = *reinterpret_cast<const f4*>(feat + (int64_t)cj * len + coff);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const float sr_c = lanes_sum_w<LH>(d4(ar4, xh[u]));
      if (CL::step_edge(lane, j + u) < n) {  // uniform per group; lanes past the end of a short chunk add nothing
        const float t0 = sl_i + sr_c;
        const float t = t0 > 0.0f ? t0 : eps * t0;
        // online softmax: one of exp(m - max), exp(t - max) is exp(0) -- ONE exponential per edge
        const float d = t - m;
        const float ex = __expf(d > 0.f ? -d : d);
        const float sc = d > 0.f ? ex : 1.f, e = d > 0.f ? 1.f : ex;
        ssum = __builtin_fmaf(ssum, sc, e);
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_fmaf(e, xh[u][k], acc[k] * sc);
        m = d > 0.f ? t : m;
      }
    }
  }
  // the four groups meet: same rescaling
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {
    const float mo = __shfl_xor(m, o, 64), so = __shfl_xor(ssum, o, 64);
    const float mn = mo > m ? mo : m;
    const float a = expf(m - mn), b = expf(mo - mn);
    ssum = ssum * a + so * b;
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = acc[k] * a + __shfl_xor(acc[k], o, 64) * b;
    m = mn;
  }
  const int64_t slot = (int64_t)chunk_start[row] + (eb - rb) / 64;
  if (gbase == 0) {
    *reinterpret_cast<f4*>(out_partial + slot * len + coff) = acc;
    if ((sl & (LH - 1)) == 0) ms_partial[slot * H + head] = float2{m, ssum};
  }
}

// sign(t_e) of every (edge, head) pre-activation score EXACTLY as the one-sweep kernels above form it (the same FMA
// chains, the same DPP sums, the same add): leaky_relu' jumps at t = 0, and a score within rounding of zero takes either
// slope in two correct fp32 evaluations -- a test that wants to compare ARITHMETIC with an fp64 evaluation of the alpha
// gradients imposes these signs on it (the way the relu mask of the oracle's forward output is imposed on the GPU's
// backward).  Test / diagnostic entry point (gaib_gat_score_signs); not on the training path.
// (a diagnostic kernel: where hipcc does not unroll its 16-/32-step loop for some shape, the column id of step j comes through a
// switch instead of a constant DPP pattern -- slower, the same values; not worth failing the -Werror=pass-failed build over)
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wpass-failed"
template <int G, int H>
__global__ __launch_bounds__(256) void gat_score_sign_kernel(int64_t n_chunks, const uint32_t* chunk_row, const uint32_t* chunk_ebase,
                                                             const int64_t* rowptr, const uint32_t* col, int len, const float* feat,
                                                             const float* alpha_l, const float* alpha_r, uint8_t* sign_out) {
  constexpr int LH = G / H;
  using CL = ChunkLanes<G>;
  const int64_t c = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (c >= n_chunks) return;
  const int lane = threadIdx.x & 63;
  const int sl = lane & (G - 1);
  const int64_t row = chunk_row[c];
  const int64_t eb = chunk_ebase[c];
  const int64_t rem = rowptr[row + 1] - eb;
  const int n = rem < 64 ? (int)rem : 64;
  const int my_e = CL::held_edge(lane, 0), my_e1 = CL::held_edge(lane, 1);
  const uint32_t cl = col[eb + (my_e < n ? my_e : 0)];
  uint32_t cl1 = 0;
  if constexpr (G == 32) cl1 = col[eb + (my_e1 < n ? my_e1 : 0)];
  const int coff = sl * 4;
  const int head = sl / LH;
  const f4 hi = *reinterpret_cast<const f4*>(feat + row * (int64_t)len + coff);
  const f4 al4 = *reinterpret_cast<const f4*>(alpha_l + coff);
  const f4 ar4 = *reinterpret_cast<const f4*>(alpha_r + coff);
  auto d4 = [](const f4& a, const f4& b) {
    return __builtin_fmaf(a[3], b[3], __builtin_fmaf(a[2], b[2], __builtin_fmaf(a[1], b[1], a[0] * b[0])));
  };
  const float sl_i = lanes_sum_w<LH>(d4(al4, hi));
#pragma unroll
  for (int j = 0; j < G; ++j) {
    if (j * CL::NG >= n) break;
    const uint32_t cj = (uint32_t)CL::step_value((int)cl, (int)cl1, lane, j);
    const f4 xh = *reinterpret_cast<const f4*>(feat + (int64_t)cj * len + coff);
    const float sr_c = lanes_sum_w<LH>(d4(ar4, xh));
    const float t_e = sl_i + sr_c;
    const int ei = CL::step_edge(lane, j);
    if (ei < n && (sl & (LH - 1)) == 0) sign_out[(eb + ei) * H + head] = t_e > 0.0f ? 1 : 0;
  }
}
#pragma clang diagnostic pop

// per row: combine the chunks' (m, s, acc) in chunk order; out = act(sum / S); stats[row][h] = (M, 1/S)
// (a row is G = len / 4 lanes of 4 columns, so the wave's NG = 64 / G lane groups take the row's chunks k, k + 1, ...,
// k + NG - 1, ... and meet at the end -- at len 64 a row with 330 chunks is 83 steps deep instead of 330, and no lane idles)
template <int G>
__global__ __launch_bounds__(256) void gat_fwd_reduce_kernel(int64_t nv, int len, int H, const uint32_t* chunk_start,
                                                             const float* out_partial, const float2* ms_partial, int relu,
                                                             float* out, float2* stats) {
  constexpr int NG = 64 / G;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int sl = lane & (G - 1), grp = lane / G;
  const int64_t c0 = chunk_start[row], c1 = chunk_start[row + 1];
  const int dh = len / H, head = (sl * 4) / dh;
  float M = GAT_NEG;
  for (int64_t k = c0 + grp; k < c1; k += NG) {
    const float mk = ms_partial[k * H + head].x;
    M = mk > M ? mk : M;
  }
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {
    const float mo = __shfl_xor(M, o, 64);
    M = mo > M ? mo : M;
  }
  float S = 0.f;
  f4 s = {0.f, 0.f, 0.f, 0.f};
  for (int64_t k = c0 + grp; k < c1; k += NG) {
    const float2 ms = ms_partial[k * H + head];
    const float w = __expf(ms.x - M);
    S += ms.y * w;
    const f4 t = *reinterpret_cast<const f4*>(out_partial + k * len + sl * 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] += t[q] * w;
  }
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {  // fixed order: deterministic
    S += __shfl_xor(S, o, 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] += __shfl_xor(s[q], o, 64);
  }
  if (grp != 0) return;
  const float inv = S > 0.f ? 1.0f / S : 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    s[q] *= inv;
    if (relu) s[q] = s[q] > 0.f ? s[q] : 0.f;
  }
  *reinterpret_cast<f4*>(out + row * (int64_t)len + sl * 4) = s;
  if ((sl * 4) % dh == 0) stats[row * H + head] = float2{M, inv};
}

// out[row] = sum of the row's chunk partials in chunk order; rs / cs [row][H] the same for the g sums
template <int G>
__global__ __launch_bounds__(256) void gat_fused_reduce_kernel(int64_t nv, int len, int H, const uint32_t* chunk_start,
                                                               const float* out_partial, const float* rc_partial,
                                                               float* out, float* rs, float* cs) {
  constexpr int NG = 64 / G;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int sl = lane & (G - 1), grp = lane / G;  // NG lane groups share the row's chunks (see the forward's)
  const int64_t c0 = chunk_start[row], c1 = chunk_start[row + 1];
  f4 s = {0.f, 0.f, 0.f, 0.f};
  // this lane's share of the 2 H row / column sums: entries sl, sl + G, sl + 2 G, sl + 3 G (2 H <= 32, G >= 8)
  float r[4] = {0.f, 0.f, 0.f, 0.f};
  const float* pp = out_partial + sl * 4;
  for (int64_t k = c0 + grp; k < c1; k += NG) {
    const f4 t = *reinterpret_cast<const f4*>(pp + k * len);
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (sl + q * G < 2 * H) r[q] += rc_partial[k * 2 * H + sl + q * G];
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] += t[q];
  }
#pragma unroll
  for (int o = G; o < 64; o <<= 1) {  // fixed order: deterministic
#pragma unroll
    for (int q = 0; q < 4; ++q) r[q] += __shfl_xor(r[q], o, 64);
#pragma unroll
    for (int q = 0; q < 4; ++q) s[q] += __shfl_xor(s[q], o, 64);
  }
  if (grp != 0) return;
  *reinterpret_cast<f4*>(out + row * (int64_t)len + sl * 4) = s;
  // entries 0..H-1: rs, H..2H-1: cs
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int e = sl + q * G;
    if (e < H) rs[row * H + e] = r[q];
    else if (e < 2 * H) cs[row * H + e - H] = r[q];
  }
}

inline unsigned rowgrid(int64_t nv) { return (unsigned)cdiv64(nv > 0 ? nv : 1, 4); }
// the row-owner kernels: gat_row_waves (1, 2 or 4) one-wave rows per workgroup
inline unsigned rowgrid_w(const gaib_ctx* ctx, int64_t nv) { return (unsigned)cdiv64(nv > 0 ? nv : 1, ctx->gat_row_waves); }

int check_heads(const char* who, int len, int heads) {
  GAIB_CHECK(len > 0, "%s: len must be > 0", who);
  GAIB_CHECK(heads >= 1 && len % heads == 0, "%s: heads (%d) must divide len (%d)", who, heads, len);
  return GAIB_OK;
}

template <int H>
int launch_edge_softmax(gaib_ctx* ctx, gaib_graph* g, const float* sl, const float* sr, float eps, float* temp,
                        float* scores, float* norm) {
  GAIB_TRY(gaib_graph_ensure_heavy(ctx, g, ctx->spmm_heavy_threshold));
  const uint32_t* rl = g->heavy_rows;
  const uint32_t* ro = g->heavy_rows ? g->heavy_rows + g->n_heavy : nullptr;
  const int thr = g->n_heavy > 0 ? g->heavy_thr : 0;
  if (g->n_heavy > 0) {  // long rows first: their tail hides under the light launch
    edge_softmax_v2_kernel<H, true><<<(unsigned)g->n_heavy, ROW_BLK_WAVES * 64, 0, ctx->stream>>>(
        g->nv, g->rowptr, g->colidx, sl, sr, eps, temp, scores, norm, thr, rl, ro);
    GAIB_LAUNCH_CHECK();
  }
  edge_softmax_v2_kernel<H, false><<<rowgrid_w(ctx, g->nv), ctx->gat_row_waves * 64, 0, ctx->stream>>>(
      g->nv, g->rowptr, g->colidx, sl, sr, eps, temp, scores, norm, thr, rl, ro);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

template <int H>
int launch_softmax_bwd(gaib_ctx* ctx, gaib_graph* g, const float* p, const float* dp, const float* temp, float eps,
                       const float* rowdot, float* scores, float* gbuf, float* rs, float* cs, float* pT,
                       const float* sl, const float* sr, float* colsum_partial) {
  GAIB_TRY(gaib_graph_ensure_heavy(ctx, g, ctx->spmm_heavy_threshold));
  const uint32_t* rl = g->heavy_rows;
  const uint32_t* ro = g->heavy_rows ? g->heavy_rows + g->n_heavy : nullptr;
  const int thr = g->n_heavy > 0 ? g->heavy_thr : 0;
  const unsigned nh = (unsigned)g->n_heavy, blk = ROW_BLK_WAVES * 64;
  const unsigned lg = rowgrid_w(ctx, g->nv), lb = (unsigned)ctx->gat_row_waves * 64;
#define GAIB_SBW_LAUNCH(DOT, RE)                                                                                   \
  do {                                                                                                             \
    if (nh)                                                                                                        \
      softmax_bwd_v2_kernel<H, true, DOT, RE><<<nh, blk, 0, ctx->stream>>>(                                        \
          g->nv, g->rowptr, p, dp, temp, eps, rowdot, scores, gbuf, pT ? 1 : 0, rs, thr, rl, ro, g->colidx, sl, sr); \
    softmax_bwd_v2_kernel<H, false, DOT, RE><<<lg, lb, 0, ctx->stream>>>(                                          \
        g->nv, g->rowptr, p, dp, temp, eps, rowdot, scores, gbuf, pT ? 1 : 0, rs, thr, rl, ro, g->colidx, sl, sr);   \
  } while (0)
  if (rowdot) {
    if (temp) GAIB_SBW_LAUNCH(true, false);
    else GAIB_SBW_LAUNCH(true, true);
  } else {
    if (temp) GAIB_SBW_LAUNCH(false, false);
    else GAIB_SBW_LAUNCH(false, true);
  }
#undef GAIB_SBW_LAUNCH
  GAIB_LAUNCH_CHECK();
  if (colsum_partial) {  // dense graphs: chunk by chunk in column order (the caller reserved the partials)
    colsum_chunk_kernel<H><<<(unsigned)cdiv64(g->n_chunks > 0 ? g->n_chunks : 1, 4), 256, 0, ctx->stream>>>(
        g->n_chunks, g->chunk_row, g->chunk_ebase, g->chunk_start, g->rowptr, g->rev, gbuf, colsum_partial, pT);
    GAIB_LAUNCH_CHECK();
    colsum_reduce_kernel<<<(unsigned)cdiv64(g->nv * H, 256), 256, 0, ctx->stream>>>(g->nv, H, g->chunk_start,
                                                                                  colsum_partial, cs);
    GAIB_LAUNCH_CHECK();
    return GAIB_OK;
  }
  if (nh) colsum_v2_kernel<H, true><<<nh, blk, 0, ctx->stream>>>(g->nv, g->rowptr, g->rev, gbuf, cs, pT, thr, rl, ro);
  colsum_v2_kernel<H, false><<<lg, lb, 0, ctx->stream>>>(g->nv, g->rowptr, g->rev, gbuf, cs, pT, thr, rl, ro);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

}  // namespace

extern "C" int gaib_gat_scores_mh(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_h,
                                  const float* d_alpha_l, const float* d_alpha_r, float epsilon,
                                  float* d_temp_scores, float* d_scores, float* d_norm_scores) {
  GAIB_CHECK(ctx && g, "gaib_gat_scores: NULL ctx/graph");
  GAIB_TRY(check_heads("gaib_gat_scores", len, heads));
  if (g->nv == 0) return GAIB_OK;
  GAIB_CHECK(d_h && d_alpha_l && d_alpha_r && d_norm_scores, "gaib_gat_scores: NULL pointer");
  // rectangular graphs (a rank's rows over [owned | halo] columns): d_h is the COLUMN table [nc x len] whose first nv
  // rows are the rows' own vectors; the per-vertex dots are taken over all nc rows
  const int64_t nt = g->nc > g->nv ? g->nc : g->nv;
  GAIB_HIP(hipSetDevice(ctx->device));
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * 2 * (size_t)nt * heads));
  float* sl = (float*)ctx->ws;
  float* sr = sl + nt * heads;
  {
    ProfScope ps(ctx, "gat_vertex_dots", (double)nt * (4.0 * len + 8.0 * heads), 4.0 * (double)nt * len);
    vertex_dots_kernel<<<rowgrid(nt), 256, 0, ctx->stream>>>(nt, len, heads, d_h, d_alpha_l, d_alpha_r, sl, sr);
  }
  GAIB_LAUNCH_CHECK();
  {
    ProfScope ps(ctx, "gat_edge_softmax", (double)g->ne * (4.0 + 4.0 * heads + 2 * 4.0 * heads) + (double)g->nv * 8.0 * heads);
    const bool al16 = (((uintptr_t)d_temp_scores | (uintptr_t)d_scores | (uintptr_t)d_norm_scores) & 15) == 0;
    int rc = GAIB_OK;
    if (heads == 1) rc = launch_edge_softmax<1>(ctx, g, sl, sr, epsilon, d_temp_scores, d_scores, d_norm_scores);
    else if (heads == 2) rc = launch_edge_softmax<2>(ctx, g, sl, sr, epsilon, d_temp_scores, d_scores, d_norm_scores);
    else if (heads == 4 && al16) rc = launch_edge_softmax<4>(ctx, g, sl, sr, epsilon, d_temp_scores, d_scores, d_norm_scores);
    else if (heads == 8 && al16) rc = launch_edge_softmax<8>(ctx, g, sl, sr, epsilon, d_temp_scores, d_scores, d_norm_scores);
    else if (heads == 16 && al16) rc = launch_edge_softmax<16>(ctx, g, sl, sr, epsilon, d_temp_scores, d_scores, d_norm_scores);
    else {
      GAIB_CHECK(d_scores && d_temp_scores, "gaib_gat_scores: this head count needs d_temp_scores and d_scores");
      edge_softmax_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(
          g->nv, heads, g->rowptr, g->colidx, sl, sr, epsilon, d_temp_scores, d_scores, d_norm_scores);
    }
    if (rc != GAIB_OK) return rc;
  }
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_gat_scores(gaib_ctx* ctx, gaib_graph* g, int len, const float* d_h,
                               const float* d_alpha_l, const float* d_alpha_r, float epsilon,
                               float* d_temp_scores, float* d_scores, float* d_norm_scores) {
  return gaib_gat_scores_mh(ctx, g, len, 1, d_h, d_alpha_l, d_alpha_r, epsilon, d_temp_scores, d_scores,
                            d_norm_scores);
}

extern "C" int gaib_sddmm_mh(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_grad,
                             const float* d_feat, float* d_out_e) {
  GAIB_CHECK(ctx && g, "gaib_sddmm: NULL ctx/graph");
  GAIB_TRY(check_heads("gaib_sddmm", len, heads));
  if (g->nv == 0 || g->ne == 0) return GAIB_OK;
  GAIB_CHECK(d_grad && d_feat && d_out_e, "gaib_sddmm: NULL pointer");
  GAIB_HIP(hipSetDevice(ctx->device));
  const int dh = len / heads;
  const int lh = dh / 4;  // lanes per head in the chunk kernels
  const bool vec_ok = (len % 4 == 0) && len <= 256 && ((((uintptr_t)d_grad | (uintptr_t)d_feat) & 15) == 0) &&
                      ctx->gat_fast && (heads == 1 || (dh % 4 == 0 && (lh & (lh - 1)) == 0));
  if (vec_ok) {
    GAIB_TRY(gaib_graph_ensure_chunks(ctx, g));
    ProfScope ps(ctx, "gat_sddmm", (double)g->ne * (4.0 + 4.0 * len + 4.0 * heads) + (double)g->nv * 4.0 * len, 2.0 * (double)g->ne * len);
    const unsigned grid = (unsigned)cdiv64(g->n_chunks > 0 ? g->n_chunks : 1, 4);
#define GAIB_SDDMM(G, U, LH)                                                                            \
  sddmm_chunk_kernel<G, U, LH><<<grid, 256, 0, ctx->stream>>>(g->n_chunks, g->chunk_row, g->chunk_ebase, \
                                                              g->rowptr, g->colidx, len, heads, d_grad, d_feat, d_out_e)
    if (heads == 1) {
      if (len <= 4) GAIB_SDDMM(1, 1, 1);
      else if (len <= 8) GAIB_SDDMM(2, 2, 2);
      else if (len <= 16) GAIB_SDDMM(4, 4, 4);
      else if (len <= 32) GAIB_SDDMM(8, 8, 8);
      else if (len <= 64) GAIB_SDDMM(16, 8, 16);
      else if (len <= 128) GAIB_SDDMM(32, 8, 32);
      else GAIB_SDDMM(64, 8, 64);
    } else {
      // G = lanes per edge (power of two covering len/4), LH = lanes per head (< G since heads > 1)
      if (len <= 8) GAIB_SDDMM(2, 2, 1);
      else if (len <= 16) { if (lh == 1) GAIB_SDDMM(4, 4, 1); else GAIB_SDDMM(4, 4, 2); }
      else if (len <= 32) { if (lh == 1) GAIB_SDDMM(8, 8, 1); else if (lh == 2) GAIB_SDDMM(8, 8, 2); else GAIB_SDDMM(8, 8, 4); }
      else if (len <= 64) {
        if (lh == 1) GAIB_SDDMM(16, 8, 1); else if (lh == 2) GAIB_SDDMM(16, 8, 2);
        else if (lh == 4) GAIB_SDDMM(16, 8, 4); else GAIB_SDDMM(16, 8, 8);
      } else if (len <= 128) {
        if (lh == 1) GAIB_SDDMM(32, 8, 1); else if (lh == 2) GAIB_SDDMM(32, 8, 2);
        else if (lh == 4) GAIB_SDDMM(32, 8, 4); else if (lh == 8) GAIB_SDDMM(32, 8, 8); else GAIB_SDDMM(32, 8, 16);
      } else {
        if (lh == 1) GAIB_SDDMM(64, 8, 1); else if (lh == 2) GAIB_SDDMM(64, 8, 2);
        else if (lh == 4) GAIB_SDDMM(64, 8, 4); else if (lh == 8) GAIB_SDDMM(64, 8, 8);
        else if (lh == 16) GAIB_SDDMM(64, 8, 16); else GAIB_SDDMM(64, 8, 32);
      }
    }
#undef GAIB_SDDMM
    GAIB_LAUNCH_CHECK();
    return GAIB_OK;
  }
  ProfScope ps(ctx, "gat_sddmm", (double)g->ne * (4.0 + 4.0 * len + 4.0 * heads) + (double)g->nv * 4.0 * len, 2.0 * (double)g->ne * len);
  sddmm_generic_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, heads, g->rowptr, g->colidx, len,
                                                               d_grad, d_feat, d_out_e);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_sddmm(gaib_ctx* ctx, gaib_graph* g, int len, const float* d_grad,
                          const float* d_feat, float* d_out_e) {
  return gaib_sddmm_mh(ctx, g, len, 1, d_grad, d_feat, d_out_e);
}

// d_temp_scores == NULL: the sign of the pre-activation score is formed again from d_feat and (d_alpha_l, d_alpha_r)
static int softmax_bwd_alpha_impl(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_feat,
                                  const float* d_norm_scores, const float* d_norm_scores_grad,
                                  const float* d_temp_scores, const float* d_alpha_l, const float* d_alpha_r,
                                  float epsilon, float* d_scores, float* d_alpha_lgrad, float* d_alpha_rgrad,
                                  const float* d_grad_rows, const float* d_fwd_out_rows, float* d_norm_scores_t) {
  GAIB_CHECK(ctx && g, "gaib_gat_softmax_bwd_alpha: NULL ctx/graph");
  GAIB_TRY(check_heads("gaib_gat_softmax_bwd_alpha", len, heads));
  GAIB_CHECK(d_alpha_lgrad && d_alpha_rgrad, "gaib_gat_softmax_bwd_alpha: NULL alpha grad");
  GAIB_HIP(hipSetDevice(ctx->device));
  if (g->nv == 0) return GAIB_OK;
  GAIB_CHECK(d_feat && d_norm_scores && d_norm_scores_grad && (d_temp_scores || (d_alpha_l && d_alpha_r)),
             "gaib_gat_softmax_bwd_alpha: NULL pointer");
  GAIB_CHECK((d_grad_rows == nullptr) == (d_fwd_out_rows == nullptr),
             "gaib_gat_softmax_bwd_alpha: d_grad_rows and d_fwd_out_rows go together");
  GAIB_TRY(gaib_graph_ensure_rev(ctx, g));
  const int nblocks = (int)(g->nv < 2048 ? cdiv64(g->nv, 8) : 1024);
  const int64_t rows_per_block = cdiv64(g->nv, nblocks);
  auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };  // keep every slab 16-byte aligned
  const size_t n_g = up4((size_t)g->ne * heads * (d_norm_scores_t ? 2 : 1)), n_v = up4((size_t)g->nv * heads);
  // graphs with a quarter of their edges in heavy rows, 1-2 heads: the column sums go chunk by chunk (reddit shape: 2.86 -> 2.67 ms single-head; at 8 heads
  // the 64-byte records gain nothing, 7.10 vs 7.11 ms).  gat_chunk_colsum: -1 = that rule, 0 never, 1 always.
  GAIB_TRY(gaib_graph_ensure_heavy(ctx, g, ctx->spmm_heavy_threshold));
  const bool v2_heads = heads == 1 || heads == 2 || heads == 4 || heads == 8 || heads == 16;
  const bool chunk_cs = v2_heads && g->ne > 0 &&
                        (ctx->gat_chunk_colsum == 1 ||
                         (ctx->gat_chunk_colsum < 0 && heads <= 2 && g->n_heavy > 0 && 4 * g->heavy_edges >= g->ne));
  if (chunk_cs) GAIB_TRY(gaib_graph_ensure_chunks(ctx, g));
  const size_t n_p = chunk_cs ? up4((size_t)g->n_chunks * heads) : 0;
  const size_t ws_floats = n_g + 5 * n_v + n_p + (size_t)nblocks * 2 * len;
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * ws_floats));
  float* gbuf = (float*)ctx->ws;
  float* rs = gbuf + n_g;
  float* cs = rs + n_v;
  float* rowdot = cs + n_v;
  float* sl = rowdot + n_v;
  float* sr = sl + n_v;
  float* cs_partial = sr + n_v;
  float* partial = cs_partial + n_p;
  ProfScope ps(ctx, "gat_softmax_bwd_alpha", (double)g->ne * (4.0 + 4.0 + 3 * 4.0 * heads) + (double)g->nv * (2 * 4.0 * len));
  if (!d_temp_scores) {
    vertex_dots_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, len, heads, d_feat, d_alpha_l, d_alpha_r, sl, sr);
    GAIB_LAUNCH_CHECK();
  }
  if (d_grad_rows) {
    rowdot_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, len, heads, d_grad_rows, d_fwd_out_rows, rowdot);
    GAIB_LAUNCH_CHECK();
  } else {
    rowdot = nullptr;
  }
  const bool al16 = (((uintptr_t)d_norm_scores | (uintptr_t)d_norm_scores_grad | (uintptr_t)d_temp_scores |
                      (uintptr_t)d_scores | (uintptr_t)gbuf | (uintptr_t)rs | (uintptr_t)cs |
                      (uintptr_t)d_norm_scores_t) & 15) == 0;
  int rc = GAIB_OK;
#define GAIB_SBW(HH) rc = launch_softmax_bwd<HH>(ctx, g, d_norm_scores, d_norm_scores_grad, d_temp_scores, epsilon, \
                                                 rowdot, d_scores, gbuf, rs, cs, d_norm_scores_t, sl, sr, \
                                                 chunk_cs ? cs_partial : nullptr)
  if (heads == 1) GAIB_SBW(1);
  else if (heads == 2) GAIB_SBW(2);
  else if (heads == 4 && al16) GAIB_SBW(4);
  else if (heads == 8 && al16) GAIB_SBW(8);
  else if (heads == 16 && al16) GAIB_SBW(16);
  else {
    GAIB_CHECK(d_scores && d_temp_scores, "gaib_gat_softmax_bwd_alpha: this head count needs d_temp_scores and d_scores");
    softmax_bwd_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, heads, g->rowptr, d_norm_scores,
                                                                d_norm_scores_grad, d_temp_scores, epsilon,
                                                                d_scores, gbuf, rs);
    colsum_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, heads, g->rowptr, g->rev, gbuf, cs);
    if (d_norm_scores_t) {
      GAIB_LAUNCH_CHECK();
      GAIB_TRY(gaib_edge_transpose_mh(ctx, g, heads, d_norm_scores, d_norm_scores_t));
    }
  }
#undef GAIB_SBW
  if (rc != GAIB_OK) return rc;
  GAIB_LAUNCH_CHECK();
  alpha_partial_kernel<<<nblocks, 256, sizeof(float) * 512, ctx->stream>>>(g->nv, len, heads, d_feat, rs, cs,
                                                                         rows_per_block, partial);
  GAIB_LAUNCH_CHECK();
  alpha_final_kernel<<<(unsigned)cdiv64(2 * (int64_t)len, 4), 256, 0, ctx->stream>>>(nblocks, len, partial, d_alpha_lgrad,
                                                                                    d_alpha_rgrad);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

// the one-sweep kernels' shapes (round 5): len = 32, 64 or 128 (8, 16 or 32 lanes x 4 columns per edge) and 1, 2, 4, 8 or 16
// heads with at least 4 columns per head -- every head width the reference's GAT runs up to its limit of 128 columns
// (gat_aggregator.cpp:57-200, global.h:58) as long as a head is a whole number of 4-column lanes
static bool gat_fused_shape(int len, int heads) {
  if (!(len == 32 || len == 64 || len == 128)) return false;
  if (!(heads == 1 || heads == 2 || heads == 4 || heads == 8 || heads == 16)) return false;
  return heads * 4 <= len;
}
// M(G, H) for the kernel instance of a shape gat_fused_shape() admits
#define GAIB_GAT_BY_HEADS(G, M, WITH16) \
  switch (heads) {                      \
    case 1: M(G, 1); break;             \
    case 2: M(G, 2); break;             \
    case 4: M(G, 4); break;             \
    case 8: M(G, 8); break;             \
    default: WITH16; break;             \
  }
#define GAIB_GAT_DISPATCH(M)                                    \
  do {                                                          \
    if (len == 32) { GAIB_GAT_BY_HEADS(8, M, (void)0) }         \
    else if (len == 64) { GAIB_GAT_BY_HEADS(16, M, M(16, 16)) } \
    else { GAIB_GAT_BY_HEADS(32, M, M(32, 16)) }                \
  } while (0)

// the packed-math sweep for the shapes it covers (a head of at most 16 lanes); the others never get here (see `pk` below)
template <int G, int H>
static void launch_bwd_pk(unsigned grid, hipStream_t st, int64_t n_chunks, const uint32_t* chunk_row, const uint32_t* chunk_ebase,
                          const uint32_t* chunk_start, const int64_t* rowptr, const uint32_t* col, int len, const float* T,
                          const float* alpha_l, const float* alpha_r, float eps, float* out_partial, float* rc_partial, int per_xcd) {
  if constexpr (G / H <= 16) {
    gat_bwd_fused_pk_kernel<G, H, 4><<<grid, 256, 0, st>>>(n_chunks, chunk_row, chunk_ebase, chunk_start, rowptr, col, len, T, alpha_l,
                                                          alpha_r, eps, out_partial, rc_partial, per_xcd);
  }
}

// The fused edge side of backward (gat_bwd_fused_chunk_kernel).  Shapes: gat_fused_shape(); otherwise, or with the option off (option
// gat_fused_bwd: 0 = never; -1 / 1 = whenever the shape fits), GAIB_ERR_UNSUPPORTED is returned and nothing was touched: the
// caller runs the staged entry points.
static bool gat_fused_applies(gaib_ctx* ctx, gaib_graph* g, int len, int heads, int knob, uintptr_t align_or, int* rc,
                              bool rect = false) {
  *rc = GAIB_OK;
  // On a rank's rectangular graph the answer must follow from rank-INVARIANT inputs alone -- len, heads, the option:
  // the one-sweep and the staged path run different collectives in backward, so a rank that has rows but no edges (an
  // empty sweep is a valid sweep) or a misaligned buffer must not take another path than its peers (the latter is an
  // error, not a reason to fall back).
  const bool heads_ok = gat_fused_shape(len, heads);
  if (rect) {
    if (heads_ok && knob != 0 && (align_or & 15) != 0) {
      gaib_set_error("one-sweep GAT on a partition: buffers must be 16-byte aligned");
      *rc = GAIB_ERR_INVALID;
      return false;
    }
    return heads_ok && knob != 0;
  }
  // Round 5: the rule is the SHAPE.  Until round 4 the automatic choice (knob < 0) also asked for a dense graph (a quarter of
  // the edges in heavy rows) over a table of <= 512 MB -- the rule of the ordered-chunk aggregation.  Measured with the kernels
  // at every width (scripts/perf_guard.py, profiles/r05/perf_guard.log): the one sweep wins wherever it applies -- reddit shape
  // 8 heads x 32: 5.06 vs 15.95 ms per layer step, x 64: 8.8 vs 17.8, x 128: 18.4 vs 26.0; products shape (sparse: one short
  // chunk per row, a 627 MB table) 1 head x 64: 19.7 vs 21.2 -- so nothing is left for the staged kernels but the widths
  // outside gat_fused_shape(), attention dropout and the explicit option.
  const bool shape_ok = heads_ok && g->nc == g->nv && g->ne > 0 && (align_or & 15) == 0;
  return shape_ok && knob != 0;
}

// Forward in one sweep (gat_fwd_fused_chunk_kernel): d_out = act(P h) and d_row_stats [nv][heads][2] = (row maximum of
// the leaky-relu'd scores, 1 / row sum of exp) -- everything backward needs to form the attention again; no [ne][heads]
// array is written.  Same cover and auto rule as gaib_gat_backward_fused (option "gat_fused_fwd"); otherwise
// GAIB_ERR_UNSUPPORTED, nothing touched, and the caller runs gaib_gat_scores_mh + gaib_spmm_mh.
// phase: -1 = the whole sweep + the per-row combination (square graphs, or a rectangular one whose table is complete);
// 0 = only the chunks over owned columns (columns < g->nv), nothing else; 1 = the remaining chunks + the combination.
// The partial results of phase 0 live in the context's workspace until phase 1: no other call on the context in between.
static int gat_forward_fused_impl(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_h, const float* d_alpha_l,
                                  const float* d_alpha_r, float epsilon, int relu, float* d_out, float* d_row_stats, int phase,
                                  bool rect, const char* who) {
  GAIB_CHECK(ctx && g, "%s: NULL ctx/graph", who);
  GAIB_TRY(check_heads(who, len, heads));
  if (g->nv == 0) {
    // a rank without rows (fewer vertices than ranks): nothing to write, its buffers may be NULL -- but WHICH path "ran" must be
    // what the ranks with rows decide, from the shape and the option alone: the two paths differ in their exchanges
    if (rect && gat_fused_shape(len, heads) && ctx->gat_fused_fwd != 0) return GAIB_OK;
    gaib_set_error("%s: not applicable to this shape / graph (len %d, heads %d)", who, len, heads);
    return GAIB_ERR_UNSUPPORTED;
  }
  GAIB_CHECK(d_h && d_alpha_l && d_alpha_r && d_out && d_row_stats && d_out != d_h, "%s: NULL or aliased pointer", who);
  GAIB_CHECK(phase >= -1 && phase <= 1, "%s: phase is -1, 0 or 1", who);
  GAIB_HIP(hipSetDevice(ctx->device));
  int rc = GAIB_OK;
  const bool use = gat_fused_applies(ctx, g, len, heads, ctx->gat_fused_fwd,
                                     (uintptr_t)d_h | (uintptr_t)d_out | (uintptr_t)d_row_stats | (uintptr_t)d_alpha_l |
                                         (uintptr_t)d_alpha_r, &rc, rect);
  if (rc != GAIB_OK) return rc;
  if (!use) {
    gaib_set_error("%s: not applicable to this shape / graph (len %d, heads %d)", who, len, heads);
    return GAIB_ERR_UNSUPPORTED;
  }
  GAIB_TRY(gaib_graph_ensure_chunks(ctx, g));
  auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
  const size_t n_op = up4((size_t)g->n_chunks * len), n_ms = up4((size_t)g->n_chunks * heads * 2);
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (n_op + n_ms)));
  float* out_partial = (float*)ctx->ws;
  float2* ms_partial = reinterpret_cast<float2*>(out_partial + n_op);
  ProfScope ps(ctx, "gat_fwd_fused", (double)g->ne * (4.0 + 4.0 * len) + (double)g->n_chunks * (4.0 * len + 8.0 * heads) * 2 + (double)g->nv * (4.0 * len + 8.0 * heads),
               4.0 * (double)g->ne * len);
  unsigned grid = (unsigned)cdiv64(g->n_chunks, 4);  // (0 on a rank whose rows have no edges: nothing to sweep)
  int per_xcd = 0;
  if (ctx->gat_chunk_xcd == 1 && grid >= 64) {
    per_xcd = (int)cdiv64(grid, 8);
    grid = (unsigned)per_xcd * 8u;
  }
#define GAIB_FF(GG, HH)                                                                                                   \
  if (grid > 0) gat_fwd_fused_chunk_kernel<GG, HH, 8><<<grid, 256, 0, ctx->stream>>>(g->n_chunks, g->chunk_row, g->chunk_ebase,          \
                                                                       g->chunk_start, g->rowptr, g->colidx, len, d_h,     \
                                                                       d_alpha_l, d_alpha_r, epsilon, out_partial, ms_partial, \
                                                                       phase, (uint32_t)g->nv, per_xcd)
  GAIB_GAT_DISPATCH(GAIB_FF);
#undef GAIB_FF
  GAIB_LAUNCH_CHECK();
  if (phase == 0) return GAIB_OK;
#define GAIB_FR(GG)                                                                                                        \
  gat_fwd_reduce_kernel<GG><<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, len, heads, g->chunk_start, out_partial, ms_partial, \
                                                                     relu ? 1 : 0, d_out, reinterpret_cast<float2*>(d_row_stats))
  if (len == 32) GAIB_FR(8);
  else if (len == 64) GAIB_FR(16);
  else GAIB_FR(32);
#undef GAIB_FR
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_gat_forward_fused(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_h,
                                      const float* d_alpha_l, const float* d_alpha_r, float epsilon, int relu,
                                      float* d_out, float* d_row_stats) {
  return gat_forward_fused_impl(ctx, g, len, heads, d_h, d_alpha_l, d_alpha_r, epsilon, relu, d_out, d_row_stats, -1, false,
                                "gaib_gat_forward_fused");
}

// the same on a rank's RECTANGULAR graph (rows = owned vertices, columns = [owned | halo]): d_tab [nc x len] holds the owned
// rows of h first, then the halo rows.  phase 0 may run while the halo rows are still arriving (it reads owned rows only).
extern "C" int gaib_gat_forward_fused_rect(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_tab,
                                           const float* d_alpha_l, const float* d_alpha_r, float epsilon, int relu,
                                           float* d_out, float* d_row_stats, int phase) {
  return gat_forward_fused_impl(ctx, g, len, heads, d_tab, d_alpha_l, d_alpha_r, epsilon, relu, d_out, d_row_stats, phase, true,
                                "gaib_gat_forward_fused_rect");
}

extern "C" int gaib_gat_backward_fused(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_feat,
                                       const float* d_grad, const float* d_fwd_out, const float* d_alpha_l,
                                       const float* d_alpha_r, const float* d_norm_scores, const float* d_row_stats,
                                       float epsilon, float* d_grad_out, float* d_alpha_lgrad, float* d_alpha_rgrad) {
  GAIB_CHECK(ctx && g, "gaib_gat_backward_fused: NULL ctx/graph");
  GAIB_TRY(check_heads("gaib_gat_backward_fused", len, heads));
  if (g->nv == 0) {  // no rows: the alpha gradients of this graph are zero, nothing else is written
    GAIB_HIP(hipSetDevice(ctx->device));
    if (d_alpha_lgrad) GAIB_HIP(hipMemsetAsync(d_alpha_lgrad, 0, sizeof(float) * len, ctx->stream));
    if (d_alpha_rgrad) GAIB_HIP(hipMemsetAsync(d_alpha_rgrad, 0, sizeof(float) * len, ctx->stream));
    return GAIB_OK;
  }
  GAIB_CHECK(d_feat && d_grad && d_fwd_out && d_alpha_l && d_alpha_r && (d_norm_scores || d_row_stats) && d_grad_out &&
                 d_alpha_lgrad && d_alpha_rgrad, "gaib_gat_backward_fused: NULL pointer");
  GAIB_CHECK(d_grad_out != d_feat && d_grad_out != d_grad, "gaib_gat_backward_fused: d_grad_out must not alias an input");
  GAIB_HIP(hipSetDevice(ctx->device));
  int rc0 = GAIB_OK;
  const bool use = gat_fused_applies(ctx, g, len, heads, ctx->gat_fused_bwd,
                                     (uintptr_t)d_feat | (uintptr_t)d_grad | (uintptr_t)d_norm_scores |
                                         (uintptr_t)d_row_stats | (uintptr_t)d_grad_out, &rc0);
  if (rc0 != GAIB_OK) return rc0;
  if (!use) {
    gaib_set_error("gaib_gat_backward_fused: not applicable to this shape / graph (len %d, heads %d)", len, heads);
    return GAIB_ERR_UNSUPPORTED;
  }
  if (!d_row_stats) GAIB_TRY(gaib_graph_ensure_rev(ctx, g));  // p[rev e] is read only when p is not formed again
  GAIB_TRY(gaib_graph_ensure_chunks(ctx, g));
  const int nblocks = (int)(g->nv < 2048 ? cdiv64(g->nv, 8) : 1024);
  const int64_t rows_per_block = cdiv64(g->nv, nblocks);
  auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
  const size_t n_v = up4((size_t)g->nv * heads);
  const size_t n_op = up4((size_t)g->n_chunks * len), n_rc = up4((size_t)g->n_chunks * 2 * heads);
  // option gat_interleave: the three per-vertex tables of the sweep as one [h | grad | records] row per vertex
  const int ldt = 2 * len + 4 * heads;
  // round 6: the packed-math sweep (gat_bwd_fused_pk_kernel) over a table whose rows interleave h and grad element by element
  // (option gat_bwd_pk = 1; off by default, see the kernel): attention formed again from the row statistics, heads of at most
  // 16 lanes, a table below 4 GB (32-bit byte offsets)
  const bool pk = d_row_stats != nullptr && ctx->gat_bwd_pk == 1 && (len / 4) / heads <= 16 &&
                  (uint64_t)g->nv * (uint64_t)ldt * 4u < ((uint64_t)1 << 32);
  const bool inter = !pk && ctx->gat_interleave == 1 && d_row_stats != nullptr;
  const size_t n_t = (inter || pk) ? up4((size_t)g->nv * ldt) + 64 : 0;  // (+ 64 floats: the table starts on a 256-B boundary)
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (7 * n_v + n_op + n_rc + (size_t)nblocks * 2 * len + n_t)));
  f4* rec = reinterpret_cast<f4*>(ctx->ws);  // [nv][H] 16-byte records (first: alignment)
  float* rowdot = (float*)ctx->ws + 4 * n_v;
  float* rs = rowdot + n_v;
  float* cs = rs + n_v;
  float* out_partial = cs + n_v;
  float* rc_partial = out_partial + n_op;
  float* partial = rc_partial + n_rc;
  float* T = reinterpret_cast<float*>(((uintptr_t)(partial + (size_t)nblocks * 2 * len) + 255) & ~(uintptr_t)255);
  ProfScope ps(ctx, "gat_bwd_fused", (double)g->ne * (4.0 + 2 * 4.0 * len + 12.0 * heads) + (double)g->n_chunks * (4.0 * len + 8.0 * heads) * 2 + (double)g->nv * 3 * 4.0 * len,
               8.0 * (double)g->ne * len);
  rowdot_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, len, heads, d_grad, d_fwd_out, rowdot);
  GAIB_LAUNCH_CHECK();
  if (d_row_stats) {
    const int64_t nrec = g->nv * (int64_t)heads;
    gat_rec_kernel<<<(unsigned)cdiv64(nrec, 256), 256, 0, ctx->stream>>>(nrec, rowdot, reinterpret_cast<const float2*>(d_row_stats),
                                                                        rec);
    GAIB_LAUNCH_CHECK();
  }
  const float *k_feat = d_feat, *k_grad = d_grad;
  const f4* k_rec = rec;
  int k_ld = len, k_rec_ld = heads;
  if (inter) {
    const int64_t tot4 = g->nv * (int64_t)(ldt / 4);
    gat_interleave_kernel<<<(unsigned)std::min<int64_t>(cdiv64(tot4, 256), (int64_t)ctx->num_cus * 16), 256, 0, ctx->stream>>>(
        g->nv, len / 4, heads, reinterpret_cast<const f4*>(d_feat), reinterpret_cast<const f4*>(d_grad), rec,
        reinterpret_cast<f4*>(T));
    GAIB_LAUNCH_CHECK();
    k_feat = T;
    k_grad = T + len;
    k_rec = reinterpret_cast<const f4*>(T + 2 * len);
    k_ld = ldt;
    k_rec_ld = ldt / 4;
  }
  unsigned grid = (unsigned)cdiv64(g->n_chunks, 4);
  int per_xcd = 0;
  if (ctx->gat_chunk_xcd == 1 && grid >= 64) {
    per_xcd = (int)cdiv64(grid, 8);
    grid = (unsigned)per_xcd * 8u;
  }
  if (pk) {
    const int64_t tot4 = g->nv * (int64_t)(ldt / 4);
    gat_interleave_pairs_kernel<<<(unsigned)std::min<int64_t>(cdiv64(tot4, 256), (int64_t)ctx->num_cus * 16), 256, 0, ctx->stream>>>(
        g->nv, len / 4, heads, reinterpret_cast<const f4*>(d_feat), reinterpret_cast<const f4*>(d_grad), rec,
        reinterpret_cast<f4*>(T));
    GAIB_LAUNCH_CHECK();
#define GAIB_FBP(GG, HH)                                                                                                        \
  launch_bwd_pk<GG, HH>(grid, ctx->stream, g->n_chunks, g->chunk_row, g->chunk_ebase, g->chunk_start, g->rowptr, g->colidx, len, T, \
                        d_alpha_l, d_alpha_r, epsilon, out_partial, rc_partial, per_xcd)
    GAIB_GAT_DISPATCH(GAIB_FBP);
#undef GAIB_FBP
    GAIB_LAUNCH_CHECK();
  } else {
  // edges in flight per group: 8 or 4 (option gat_fused_unroll)
#define GAIB_FB_U(GG, HH, UU, RC)                                                                                          \
  gat_bwd_fused_chunk_kernel<GG, HH, UU, RC><<<grid, 256, 0, ctx->stream>>>(                                               \
      g->n_chunks, g->chunk_row, g->chunk_ebase, g->chunk_start, g->rowptr, g->colidx, g->rev, len, k_feat, k_grad,        \
      d_norm_scores, reinterpret_cast<const float2*>(d_row_stats), rowdot, d_alpha_l, d_alpha_r, epsilon, out_partial,     \
      rc_partial, k_rec, -1, 0u, k_ld, k_rec_ld, per_xcd)
  // (8 edges in flight per group is an option of the 64-wide form only, where it was measured)
#define GAIB_FB(GG, HH)                                                      \
  do {                                                                       \
    if (d_row_stats) {                                                       \
      if (GG == 16 && ctx->gat_fused_unroll == 8) GAIB_FB_U(16, HH, 8, true); \
      else GAIB_FB_U(GG, HH, 4, true);                                       \
    } else {                                                                 \
      if (GG == 16 && ctx->gat_fused_unroll == 8) GAIB_FB_U(16, HH, 8, false); \
      else GAIB_FB_U(GG, HH, 4, false);                                      \
    }                                                                        \
  } while (0)
  GAIB_GAT_DISPATCH(GAIB_FB);
#undef GAIB_FB
#undef GAIB_FB_U
  GAIB_LAUNCH_CHECK();
  }
#define GAIB_FRD(GG)                                                                                                   \
  gat_fused_reduce_kernel<GG><<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, len, heads, g->chunk_start, out_partial, \
                                                                       rc_partial, d_grad_out, rs, cs)
  if (len == 32) GAIB_FRD(8);
  else if (len == 64) GAIB_FRD(16);
  else GAIB_FRD(32);
  GAIB_LAUNCH_CHECK();
  alpha_partial_kernel<<<nblocks, 256, sizeof(float) * 512, ctx->stream>>>(g->nv, len, heads, d_feat, rs, cs,
                                                                         rows_per_block, partial);
  GAIB_LAUNCH_CHECK();
  alpha_final_kernel<<<(unsigned)cdiv64(2 * (int64_t)len, 4), 256, 0, ctx->stream>>>(nblocks, len, partial, d_alpha_lgrad,
                                                                                    d_alpha_rgrad);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_gat_score_signs(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_h, const float* d_alpha_l,
                                    const float* d_alpha_r, uint8_t* d_sign_out) {
  GAIB_CHECK(ctx && g && d_h && d_alpha_l && d_alpha_r && d_sign_out, "gaib_gat_score_signs: NULL argument");
  GAIB_TRY(check_heads("gaib_gat_score_signs", len, heads));
  GAIB_CHECK(gat_fused_shape(len, heads),
             "gaib_gat_score_signs: the one-sweep kernels' shapes only (len 32, 64 or 128; 1, 2, 4, 8 or 16 heads of >= 4 columns)");
  GAIB_HIP(hipSetDevice(ctx->device));
  if (g->ne == 0) return GAIB_OK;
  GAIB_TRY(gaib_graph_ensure_chunks(ctx, g));
  const unsigned grid = (unsigned)cdiv64(g->n_chunks, 4);
#define GAIB_SS(GG, HH)                                                                                                 \
  gat_score_sign_kernel<GG, HH><<<grid, 256, 0, ctx->stream>>>(g->n_chunks, g->chunk_row, g->chunk_ebase, g->rowptr,     \
                                                               g->colidx, len, d_h, d_alpha_l, d_alpha_r, d_sign_out)
  GAIB_GAT_DISPATCH(GAIB_SS);
#undef GAIB_SS
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

// ---- the one-sweep backward on a rank's RECTANGULAR graph -------------------------------------------------
// Everything row i needs about its edge (i -> c) and the reverse edge (c -> i) follows from per-VERTEX quantities of i and
// c (h, grad, and the record (rowdot, row maximum, 1 / row sum)), and in a structurally symmetric graph the in-edges of i
// are the reverses of its out-edges: so the rank that owns row i computes rs_i, cs_i and the aggregated gradient of i from
// i's own edge list alone, given the halo vertices' h rows (still there from forward), grad rows and records.  No
// transposed structure, no permuted [ne][H] copies, no reverse exchange of partial rows: two forward-direction halo
// exchanges (grad rows, records) and one sweep.
//   gaib_gat_backward_rec: the owned vertices' records  rec[v][h] = (<grad_v, out_v>_h, M_vh, 1/S_vh, 0)
//   gaib_gat_backward_fused_rect: d_feat_tab / d_grad_tab [nc x len], d_rec_tab [nc][heads][4] with the owned rows first;
//     phase as in gaib_gat_forward_fused_rect (0 reads owned rows only).  The alpha gradients cover the OWNED rows: the
//     caller sums them over the ranks.
extern "C" int gaib_gat_backward_rec(gaib_ctx* ctx, int64_t nv, int len, int heads, const float* d_grad, const float* d_fwd_out,
                                     const float* d_row_stats, float* d_rec) {
  GAIB_CHECK(ctx && (nv == 0 || (d_grad && d_fwd_out && d_row_stats && d_rec)), "gaib_gat_backward_rec: NULL argument");
  GAIB_TRY(check_heads("gaib_gat_backward_rec", len, heads));
  if (nv <= 0) return GAIB_OK;
  GAIB_HIP(hipSetDevice(ctx->device));
  const size_t n_v = ((size_t)nv * heads + 3) & ~(size_t)3;
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * n_v));
  float* rowdot = (float*)ctx->ws;
  rowdot_kernel<<<rowgrid(nv), 256, 0, ctx->stream>>>(nv, len, heads, d_grad, d_fwd_out, rowdot);
  GAIB_LAUNCH_CHECK();
  const int64_t nrec = nv * (int64_t)heads;
  gat_rec_kernel<<<(unsigned)cdiv64(nrec, 256), 256, 0, ctx->stream>>>(nrec, rowdot, reinterpret_cast<const float2*>(d_row_stats),
                                                                      reinterpret_cast<f4*>(d_rec));
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_gat_backward_fused_rect(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_feat_tab,
                                            const float* d_grad_tab, const float* d_rec_tab, const float* d_alpha_l,
                                            const float* d_alpha_r, float epsilon, float* d_grad_out, float* d_alpha_lgrad,
                                            float* d_alpha_rgrad, int phase) {
  GAIB_CHECK(ctx && g, "gaib_gat_backward_fused_rect: NULL ctx/graph");
  GAIB_TRY(check_heads("gaib_gat_backward_fused_rect", len, heads));
  if (g->nv == 0) {  // a rank without rows: zero alpha gradients (they are summed over the ranks afterwards), nothing else
    GAIB_CHECK(phase >= -1 && phase <= 1, "gaib_gat_backward_fused_rect: phase is -1, 0 or 1");
    if (!(gat_fused_shape(len, heads) && ctx->gat_fused_bwd != 0)) {
      gaib_set_error("gaib_gat_backward_fused_rect: not applicable to this shape / graph (len %d, heads %d)", len, heads);
      return GAIB_ERR_UNSUPPORTED;  // (the decision of the ranks WITH rows: shape and option alone, see the forward)
    }
    if (phase != 0) {
      GAIB_HIP(hipSetDevice(ctx->device));
      if (d_alpha_lgrad) GAIB_HIP(hipMemsetAsync(d_alpha_lgrad, 0, sizeof(float) * len, ctx->stream));
      if (d_alpha_rgrad) GAIB_HIP(hipMemsetAsync(d_alpha_rgrad, 0, sizeof(float) * len, ctx->stream));
    }
    return GAIB_OK;
  }
  GAIB_CHECK(d_feat_tab && d_grad_tab && d_rec_tab && d_alpha_l && d_alpha_r && d_grad_out && d_alpha_lgrad && d_alpha_rgrad,
             "gaib_gat_backward_fused_rect: NULL pointer");
  GAIB_CHECK(d_grad_out != d_feat_tab && d_grad_out != d_grad_tab, "gaib_gat_backward_fused_rect: d_grad_out must not alias an input");
  GAIB_CHECK(phase >= -1 && phase <= 1, "gaib_gat_backward_fused_rect: phase is -1, 0 or 1");
  GAIB_HIP(hipSetDevice(ctx->device));
  int rc0 = GAIB_OK;
  const bool use = gat_fused_applies(ctx, g, len, heads, ctx->gat_fused_bwd,
                                     (uintptr_t)d_feat_tab | (uintptr_t)d_grad_tab | (uintptr_t)d_rec_tab | (uintptr_t)d_grad_out,
                                     &rc0, true);
  if (rc0 != GAIB_OK) return rc0;
  if (!use) {
    gaib_set_error("gaib_gat_backward_fused_rect: not applicable to this shape / graph (len %d, heads %d)", len, heads);
    return GAIB_ERR_UNSUPPORTED;
  }
  GAIB_TRY(gaib_graph_ensure_chunks(ctx, g));
  const int nblocks = (int)(g->nv < 2048 ? cdiv64(g->nv, 8) : 1024);
  const int64_t rows_per_block = cdiv64(g->nv, nblocks);
  auto up4 = [](size_t n) { return (n + 3) & ~(size_t)3; };
  const size_t n_v = up4((size_t)g->nv * heads);
  const size_t n_op = up4((size_t)g->n_chunks * len), n_rc = up4((size_t)g->n_chunks * 2 * heads);
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (2 * n_v + n_op + n_rc + (size_t)nblocks * 2 * len)));
  float* rs = (float*)ctx->ws;
  float* cs = rs + n_v;
  float* out_partial = cs + n_v;
  float* rc_partial = out_partial + n_op;
  float* partial = rc_partial + n_rc;
  ProfScope ps(ctx, "gat_bwd_fused", (double)g->ne * (4.0 + 2 * 4.0 * len + 12.0 * heads) + (double)g->n_chunks * (4.0 * len + 8.0 * heads) * 2 + (double)g->nv * 3 * 4.0 * len,
               8.0 * (double)g->ne * len);
  const unsigned grid = (unsigned)cdiv64(g->n_chunks, 4);
#define GAIB_FBR(GG, HH)                                                                                                   \
  if (grid > 0) gat_bwd_fused_chunk_kernel<GG, HH, 4, true><<<grid, 256, 0, ctx->stream>>>(                                              \
      g->n_chunks, g->chunk_row, g->chunk_ebase, g->chunk_start, g->rowptr, g->colidx, nullptr, len, d_feat_tab, d_grad_tab, \
      nullptr, nullptr, nullptr, d_alpha_l, d_alpha_r, epsilon, out_partial, rc_partial,                                   \
      reinterpret_cast<const f4*>(d_rec_tab), phase, (uint32_t)g->nv, len, HH, 0)
  GAIB_GAT_DISPATCH(GAIB_FBR);
#undef GAIB_FBR
  GAIB_LAUNCH_CHECK();
  if (phase == 0) return GAIB_OK;
  if (len == 32) GAIB_FRD(8);
  else if (len == 64) GAIB_FRD(16);
  else GAIB_FRD(32);
  GAIB_LAUNCH_CHECK();
  alpha_partial_kernel<<<nblocks, 256, sizeof(float) * 512, ctx->stream>>>(g->nv, len, heads, d_feat_tab, rs, cs,
                                                                         rows_per_block, partial);
  GAIB_LAUNCH_CHECK();
  alpha_final_kernel<<<(unsigned)cdiv64(2 * (int64_t)len, 4), 256, 0, ctx->stream>>>(nblocks, len, partial, d_alpha_lgrad,
                                                                                    d_alpha_rgrad);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

// ---- the pieces of GAT backward that work on RECTANGULAR graphs (a rank's share of a partition) ------
// On a partition the reverse edge of (i -> c) lives on the rank that owns c, so the reverse-edge permutation of the
// square path is replaced by the rank's TRANSPOSED local structure (rows = owned + halo vertices, columns = owned rows,
// built by the host with the edge permutation CSC position -> CSR edge id):
//   gaib_gat_softmax_bwd_rows  per-row softmax backward + leaky-relu': g_e and the row sums rs (no column side)
//   gaib_edge_gather_perm      out_e[k] = in_e[perm[k]]   (g and p in transposed order)
//   gaib_edge_rowsum           out[v] = sum over row v of an edge array   (column sums of g = row sums over the transpose)
//   gaib_gat_alpha_grads       alpha_l' = sum_v rs[v] x[v], alpha_r' = sum_v cs[v] x[v]   (fixed two-level reduction)
// and the transposed aggregation is gaib_spmm_mh on the transposed graph followed by gaib_halo_reduce.
extern "C" int gaib_gat_softmax_bwd_rows(gaib_ctx* ctx, gaib_graph* g, int heads, const float* d_norm_scores,
                                         const float* d_norm_scores_grad, const float* d_temp_scores, float epsilon,
                                         float* d_g_out, float* d_rs_out) {
  GAIB_CHECK(ctx && g, "gaib_gat_softmax_bwd_rows: NULL ctx/graph");
  GAIB_CHECK(heads >= 1, "gaib_gat_softmax_bwd_rows: heads must be >= 1");
  if (g->nv == 0) return GAIB_OK;  // (a rank without rows: its per-edge arrays may be NULL)
  GAIB_CHECK(d_norm_scores && d_norm_scores_grad && d_temp_scores && d_g_out && d_rs_out, "gaib_gat_softmax_bwd_rows: NULL argument");
  GAIB_HIP(hipSetDevice(ctx->device));
  const bool al16 = (((uintptr_t)d_norm_scores | (uintptr_t)d_norm_scores_grad | (uintptr_t)d_temp_scores |
                      (uintptr_t)d_g_out | (uintptr_t)d_rs_out) & 15) == 0;
  ProfScope ps(ctx, "gat_softmax_bwd_alpha", (double)g->ne * (4.0 + 4.0 + 3 * 4.0 * heads));
  GAIB_TRY(gaib_graph_ensure_heavy(ctx, g, ctx->spmm_heavy_threshold));
  const uint32_t* rl = g->heavy_rows;
  const uint32_t* ro = g->heavy_rows ? g->heavy_rows + g->n_heavy : nullptr;
  const int thr = g->n_heavy > 0 ? g->heavy_thr : 0;
  const unsigned nh = (unsigned)g->n_heavy, blk = ROW_BLK_WAVES * 64;
  const unsigned lg = rowgrid_w(ctx, g->nv), lb = (unsigned)ctx->gat_row_waves * 64;
#define GAIB_ROWS(HH)                                                                                                    \
  do {                                                                                                                   \
    if (nh)                                                                                                              \
      softmax_bwd_v2_kernel<HH, true, false, false><<<nh, blk, 0, ctx->stream>>>(                                        \
          g->nv, g->rowptr, d_norm_scores, d_norm_scores_grad, d_temp_scores, epsilon, nullptr, nullptr, d_g_out, 0,      \
          d_rs_out, thr, rl, ro, g->colidx, nullptr, nullptr);                                                           \
    softmax_bwd_v2_kernel<HH, false, false, false><<<lg, lb, 0, ctx->stream>>>(                                          \
        g->nv, g->rowptr, d_norm_scores, d_norm_scores_grad, d_temp_scores, epsilon, nullptr, nullptr, d_g_out, 0,        \
        d_rs_out, thr, rl, ro, g->colidx, nullptr, nullptr);                                                             \
  } while (0)
  if (heads == 1) GAIB_ROWS(1);
  else if (heads == 2) GAIB_ROWS(2);
  else if (heads == 4 && al16) GAIB_ROWS(4);
  else if (heads == 8 && al16) GAIB_ROWS(8);
  else if (heads == 16 && al16) GAIB_ROWS(16);
  else {
    // generic head counts: the kernel also writes ds; park it in the workspace
    GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (size_t)g->ne * heads));
    softmax_bwd_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, heads, g->rowptr, d_norm_scores, d_norm_scores_grad,
                                                                d_temp_scores, epsilon, (float*)ctx->ws, d_g_out, d_rs_out);
  }
#undef GAIB_ROWS
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_edge_gather_perm(gaib_ctx* ctx, int64_t ne, int heads, const uint32_t* d_perm, const float* d_in_e,
                                     float* d_out_e) {
  GAIB_CHECK(ctx && heads >= 1 && ne >= 0, "gaib_edge_gather_perm: bad argument");
  if (ne == 0) return GAIB_OK;
  GAIB_CHECK(d_perm && d_in_e && d_out_e && d_in_e != d_out_e, "gaib_edge_gather_perm: NULL or aliased pointer");
  GAIB_HIP(hipSetDevice(ctx->device));
  edge_gather_kernel<<<(unsigned)cdiv64(ne * heads, 256), 256, 0, ctx->stream>>>(ne, heads, d_perm, d_in_e, d_out_e);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

__global__ __launch_bounds__(256) void edge_rowsum_kernel(int64_t nv, int H, const int64_t* rowptr, const float* in_e,
                                                          float* out) {
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= nv) return;
  const int lane = threadIdx.x & 63;
  const int64_t e0 = rowptr[row], e1 = rowptr[row + 1];
  for (int h = 0; h < H; ++h) {
    float s = 0.f;
    for (int64_t e = e0 + lane; e < e1; e += 64) s += in_e[e * H + h];
    s = wave_sum(s);
    if (lane == 0) out[row * H + h] = s;
  }
}

extern "C" int gaib_edge_rowsum(gaib_ctx* ctx, gaib_graph* g, int heads, const float* d_in_e, float* d_out_rows) {
  GAIB_CHECK(ctx && g && heads >= 1, "gaib_edge_rowsum: bad argument");
  if (g->nv == 0) return GAIB_OK;  // (a rank without rows: the arrays may be NULL)
  GAIB_CHECK(d_out_rows && (d_in_e || g->ne == 0), "gaib_edge_rowsum: NULL pointer");
  GAIB_HIP(hipSetDevice(ctx->device));
  edge_rowsum_kernel<<<rowgrid(g->nv), 256, 0, ctx->stream>>>(g->nv, heads, g->rowptr, d_in_e, d_out_rows);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_gat_alpha_grads(gaib_ctx* ctx, int64_t nv, int len, int heads, const float* d_x, const float* d_rs,
                                    const float* d_cs, float* d_alpha_lgrad, float* d_alpha_rgrad) {
  GAIB_CHECK(ctx && d_alpha_lgrad && d_alpha_rgrad && nv >= 0, "gaib_gat_alpha_grads: bad argument");
  GAIB_TRY(check_heads("gaib_gat_alpha_grads", len, heads));
  GAIB_HIP(hipSetDevice(ctx->device));
  if (nv == 0) {  // no vertices (a rank without rows): both gradients are zero
    GAIB_HIP(hipMemsetAsync(d_alpha_lgrad, 0, sizeof(float) * len, ctx->stream));
    GAIB_HIP(hipMemsetAsync(d_alpha_rgrad, 0, sizeof(float) * len, ctx->stream));
    return GAIB_OK;
  }
  GAIB_CHECK(d_x && d_rs && d_cs, "gaib_gat_alpha_grads: NULL pointer");
  const int nblocks = (int)(nv < 2048 ? cdiv64(nv, 8) : 1024);
  const int64_t rows_per_block = cdiv64(nv, nblocks);
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (size_t)nblocks * 2 * len));
  float* partial = (float*)ctx->ws;
  alpha_partial_kernel<<<nblocks, 256, sizeof(float) * 512, ctx->stream>>>(nv, len, heads, d_x, d_rs, d_cs,
                                                                         rows_per_block, partial);
  GAIB_LAUNCH_CHECK();
  alpha_final_kernel<<<(unsigned)cdiv64(2 * (int64_t)len, 4), 256, 0, ctx->stream>>>(nblocks, len, partial, d_alpha_lgrad,
                                                                                    d_alpha_rgrad);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_gat_softmax_bwd_alpha_ex(gaib_ctx* ctx, gaib_graph* g, int len, int heads,
                                             const float* d_feat, const float* d_norm_scores,
                                             const float* d_norm_scores_grad,
                                             const float* d_temp_scores, float epsilon,
                                             float* d_scores, float* d_alpha_lgrad,
                                             float* d_alpha_rgrad, const float* d_grad_rows,
                                             const float* d_fwd_out_rows, float* d_norm_scores_t) {
  GAIB_CHECK(d_temp_scores, "gaib_gat_softmax_bwd_alpha: NULL pointer");
  return softmax_bwd_alpha_impl(ctx, g, len, heads, d_feat, d_norm_scores, d_norm_scores_grad, d_temp_scores, nullptr,
                                nullptr, epsilon, d_scores, d_alpha_lgrad, d_alpha_rgrad, d_grad_rows, d_fwd_out_rows,
                                d_norm_scores_t);
}

extern "C" int gaib_gat_softmax_bwd_alpha_re(gaib_ctx* ctx, gaib_graph* g, int len, int heads,
                                             const float* d_feat, const float* d_alpha_l, const float* d_alpha_r,
                                             const float* d_norm_scores, const float* d_norm_scores_grad,
                                             float epsilon, float* d_scores, float* d_alpha_lgrad,
                                             float* d_alpha_rgrad, const float* d_grad_rows,
                                             const float* d_fwd_out_rows, float* d_norm_scores_t) {
  GAIB_CHECK(d_alpha_l && d_alpha_r, "gaib_gat_softmax_bwd_alpha_re: NULL alpha");
  GAIB_CHECK(heads == 1 || heads == 2 || heads == 4 || heads == 8 || heads == 16,
             "gaib_gat_softmax_bwd_alpha_re: heads must be 1, 2, 4, 8 or 16");
  return softmax_bwd_alpha_impl(ctx, g, len, heads, d_feat, d_norm_scores, d_norm_scores_grad, nullptr, d_alpha_l,
                                d_alpha_r, epsilon, d_scores, d_alpha_lgrad, d_alpha_rgrad, d_grad_rows, d_fwd_out_rows,
                                d_norm_scores_t);
}

extern "C" int gaib_gat_softmax_bwd_alpha_mh(gaib_ctx* ctx, gaib_graph* g, int len, int heads,
                                             const float* d_feat, const float* d_norm_scores,
                                             const float* d_norm_scores_grad,
                                             const float* d_temp_scores, float epsilon,
                                             float* d_scores, float* d_alpha_lgrad,
                                             float* d_alpha_rgrad) {
  return gaib_gat_softmax_bwd_alpha_ex(ctx, g, len, heads, d_feat, d_norm_scores, d_norm_scores_grad, d_temp_scores,
                                       epsilon, d_scores, d_alpha_lgrad, d_alpha_rgrad, nullptr, nullptr, nullptr);
}

extern "C" int gaib_gat_softmax_bwd_alpha(gaib_ctx* ctx, gaib_graph* g, int len, const float* d_feat,
                                          const float* d_norm_scores,
                                          const float* d_norm_scores_grad,
                                          const float* d_temp_scores, float epsilon,
                                          float* d_scores, float* d_alpha_lgrad,
                                          float* d_alpha_rgrad) {
  return gaib_gat_softmax_bwd_alpha_mh(ctx, g, len, 1, d_feat, d_norm_scores, d_norm_scores_grad, d_temp_scores,
                                       epsilon, d_scores, d_alpha_lgrad, d_alpha_rgrad);
}

extern "C" int gaib_edge_transpose_mh(gaib_ctx* ctx, gaib_graph* g, int heads, const float* d_in_e,
                                      float* d_out_e) {
  GAIB_CHECK(ctx && g, "gaib_edge_transpose: NULL ctx/graph");
  GAIB_CHECK(heads >= 1, "gaib_edge_transpose: heads must be >= 1");
  if (g->ne == 0) return GAIB_OK;
  GAIB_CHECK(d_in_e && d_out_e && d_in_e != d_out_e, "gaib_edge_transpose: bad pointers");
  GAIB_HIP(hipSetDevice(ctx->device));
  GAIB_TRY(gaib_graph_ensure_rev(ctx, g));
  // rev is an involution on a structurally symmetric graph: out[rev[e]] = in[e]  <=>
  // out[e] = in[rev[e]]; the gather form keeps the stores coalesced.
  edge_gather_kernel<<<(unsigned)cdiv64(g->ne * heads, 256), 256, 0, ctx->stream>>>(g->ne, heads, g->rev,
                                                                                  d_in_e, d_out_e);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_edge_transpose(gaib_ctx* ctx, gaib_graph* g, const float* d_in_e,
                                   float* d_out_e) {
  return gaib_edge_transpose_mh(ctx, g, 1, d_in_e, d_out_e);
}
