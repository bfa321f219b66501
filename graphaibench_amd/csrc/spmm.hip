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
#include "common.h"

namespace {

struct SpmmArgs {
  const int64_t* rowptr;
  const uint32_t* col;
  const float* rw;      // per-row weight    (WMODE 0)
  const float* ew;      // per-edge weight   (WMODE 1, 2)
  const uint32_t* rev;  // reverse edge ids  (WMODE 2: w = ew[rev[e]])
  const float* in;
  float* out;
  int64_t ld;  // row stride of in/out (floats)
  int ncols;   // columns handled by this launch (<= 64*VEC*CT), starting at in/out
  int n_rows;
  int heavy_thr;
  const uint32_t* row_list;  // heavy kernel only
  int nblocks;               // light kernels: logical number of row blocks
  int per_xcd;               // ceil(nblocks/8) when swizzled, 0 otherwise
  uint32_t in_bytes;         // BUF kernels: size of the feature table (< 4 GB)
  const uint32_t* col_flagged;  // GM 3: column ids with the top bit set on cold columns
  int accumulate;            // out += instead of out = (second half of a split aggregation)
  int relu;                  // clamp at 0 on store (activation fused)
  int heads;                 // WMODE 3/4: edge weights are [ne][heads]; head of a column = col / dh
  int dh;
};

template <int VEC> struct VecT;
template <> struct VecT<1> { typedef float type; };
template <> struct VecT<2> { typedef float type __attribute__((ext_vector_type(2))); };
template <> struct VecT<4> { typedef float type __attribute__((ext_vector_type(4))); };

template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vzero() {
  typename VecT<VEC>::type z;
  if constexpr (VEC == 1) z = 0.f;
  else
    for (int i = 0; i < VEC; ++i) z[i] = 0.f;
  return z;
}
template <int VEC>
__device__ __forceinline__ typename VecT<VEC>::type vrelu(typename VecT<VEC>::type v) {
  if constexpr (VEC == 1) return v > 0.f ? v : 0.f;
  else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) v[i] = v[i] > 0.f ? v[i] : 0.f;
    return v;
  }
}
template <int VEC>
__device__ __forceinline__ void vacc(typename VecT<VEC>::type& acc, float w,
                                     const typename VecT<VEC>::type& x) {
  // separate multiply and add: the reference does scale() then vadd_cpu()
  // (math_functions.cpp:336-356, 266-283); this file is built with -ffp-contract=off.
  if constexpr (VEC == 1) {
    float t = w * x;
    acc = acc + t;
  } else {
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      float t = w * x[i];
      acc[i] = acc[i] + t;
    }
  }
}

__device__ __forceinline__ int logical_block(const SpmmArgs& a) {
  int b = blockIdx.x;
  if (a.per_xcd > 0) b = (b & 7) * a.per_xcd + (b >> 3);
  return b;
}

// WMODE: 0 per-row weight | 1 per-edge | 2 per-edge through the reverse permutation |
//        3 per-(edge, head) | 4 per-(edge, head) through the reverse permutation
template <int WMODE>
__device__ __forceinline__ float load_edge_w(const SpmmArgs& a, int64_t e, int head = 0) {
  if constexpr (WMODE == 1) return a.ew[e];
  else if constexpr (WMODE == 2) return a.ew[a.rev[e]];
  else if constexpr (WMODE == 3) return a.ew[e * a.heads + head];
  else if constexpr (WMODE == 4) return a.ew[(int64_t)a.rev[e] * a.heads + head];
  else return 0.f;
}

typedef unsigned u2_t __attribute__((ext_vector_type(2)));
typedef unsigned u4_t __attribute__((ext_vector_type(4)));

// One feature-row gather.  BUF: `buffer_load_dwordxN v, voff, s[rsrc], soff offen` -- the row
// base (col * row bytes) is a 32-bit SGPR offset against one descriptor for the whole table,
// so a gather in flight costs only its VEC destination VGPRs (no 64-bit VGPR address pair).
// Needs the table to be < 4 GB; larger tables use 64-bit global_load addresses.
// GM (gather mode): 0 = 64-bit global_load; 1 = buffer_load, default cache policy; 2 = buffer_load nt
// (streaming) for every gather; 3 = buffer_load, nt only for COLD columns (top bit of the column id
// set by gaib_graph_ensure_hot_flags), so the few thousand hub rows keep their place in the 4 MB L2.
template <int VEC, int GM>
struct RowGather {
  static constexpr bool BUF = GM != 0;
  __amdgpu_buffer_rsrc_t rsrc;
  const char* inb;
  int64_t ldb;
  __device__ __forceinline__ RowGather(const SpmmArgs& a) {
    inb = reinterpret_cast<const char*>(a.in);
    ldb = a.ld * 4;
    if constexpr (BUF) rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
  }
  __device__ __forceinline__ typename VecT<VEC>::type load(uint32_t cj, uint32_t voff) const {
    typedef typename VecT<VEC>::type vec_t;
    if constexpr (GM == 3) {
      const uint32_t c = cj & 0x7fffffffu;
      if (cj & 0x80000000u) return load_buf<2>(c, voff);  // wave-uniform branch (cj is scalar)
      return load_buf<0>(c, voff);
    } else if constexpr (GM == 2) {
      return load_buf<2>(cj, voff);
    } else if constexpr (GM == 1) {
      return load_buf<0>(cj, voff);
    } else {
      const char* rowp = inb + (int64_t)cj * ldb;  // scalar base
      return *reinterpret_cast<const vec_t*>(rowp + voff);
    }
  }
  template <int AUX>
  __device__ __forceinline__ typename VecT<VEC>::type load_buf(uint32_t cj, uint32_t voff) const {
    typedef typename VecT<VEC>::type vec_t;
    const int soff = (int)(cj * (uint32_t)ldb);
    if constexpr (VEC == 1) {
      return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)voff, soff, AUX));
    } else if constexpr (VEC == 2) {
      u2_t r = __builtin_amdgcn_raw_buffer_load_b64(rsrc, (int)voff, soff, AUX);
      vec_t v;
      v[0] = __uint_as_float(r[0]);
      v[1] = __uint_as_float(r[1]);
      return v;
    } else {
      u4_t r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff, soff, AUX);
      vec_t v;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = __uint_as_float(r[i]);
      return v;
    }
  }
};

// ---- the shared per-wave edge loop: accumulate edges [eb, ee) of one row -------------
// chunk_stride: distance between this wave's 64-edge chunks (64 for a whole row, 64*W when
// W waves share a row).
// voff[ct] is the lane's BYTE offset inside a feature row; lanes whose columns fall outside
// the row are pointed at offset 0, so every gather is unconditional (a predicated load makes
// hipcc branch on EXEC and drain vmcnt after each one); what they accumulate is never stored.
template <int VEC, int CT, int WMODE, int U, int BUF>
__device__ __forceinline__ void wave_accumulate(const SpmmArgs& a, int lane, int64_t eb, int64_t ee,
                                                int64_t chunk_stride, float roww,
                                                const uint32_t (&voff)[CT],
                                                typename VecT<VEC>::type (&acc)[CT]) {
  typedef typename VecT<VEC>::type vec_t;
  const RowGather<VEC, BUF> gather(a);
  vec_t x[U][CT];  // gather destinations; the tail's piece p lives in x[p .. 2p-1]
  constexpr bool MH = WMODE >= 3;  // multi-head: every lane fetches the weight of ITS head itself
  int hd[CT];
  float wv[MH ? U : 1][CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) hd[ct] = MH ? (int)((voff[ct] >> 2) / (uint32_t)a.dh) : 0;
  for (int64_t base = eb; base < ee; base += chunk_stride) {
    const int64_t rem = ee - base;
    const int n = rem < 64 ? (int)rem : 64;  // wave-uniform
    uint32_t c = 0;
    float w = 0.f;
    if (lane < n) {
      c = a.col[base + lane];
      if constexpr (WMODE == 1 || WMODE == 2) w = load_edge_w<WMODE>(a, base + lane);
    }
    int j = 0;
    // full batches: U independent row gathers in flight, straight-line code
    for (; j + U <= n; j += U) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, j + u);
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          x[u][ct] = gather.load(cj, voff[ct]);
          if constexpr (MH) wv[u][ct] = load_edge_w<WMODE>(a, base + j + u, hd[ct]);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float wj = (WMODE == 0) ? roww : (MH ? 0.f : readlane_f(w, j + u));
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
          float wsel = wj;
          if constexpr (MH) wsel = wv[u][ct];
          vacc<VEC>(acc[ct], wsel, x[u][ct]);
        }
      }
    }
    // tail: r = n - j < U edges, done as power-of-two pieces U/2, U/4, .., 1 (CSR order kept):
    // first every piece's gathers are issued, then every piece is accumulated.
    const int r = n - j;
    if (r > 0) {
      int jj = j;
#pragma unroll
      for (int p = U / 2; p >= 1; p >>= 1) {
        if (r & p) {
#pragma unroll
          for (int u = 0; u < p; ++u) {
            const uint32_t cj = (uint32_t)__builtin_amdgcn_readlane((int)c, jj + u);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) x[p + u][ct] = gather.load(cj, voff[ct]);
          }
          jj += p;
        }
      }
      jj = j;
#pragma unroll
      for (int p = U / 2; p >= 1; p >>= 1) {
        if (r & p) {
#pragma unroll
          for (int u = 0; u < p; ++u) {
            const float wj = (WMODE == 0) ? roww : (MH ? 0.f : readlane_f(w, jj + u));
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
              // (the tail is short: its per-head weights are fetched at the point of use)
              float wh = wj;
              if constexpr (MH) wh = load_edge_w<WMODE>(a, base + jj + u, hd[ct]);
              vacc<VEC>(acc[ct], wh, x[p + u][ct]);
            }
          }
          jj += p;
        }
      }
    }
  }
}

// ---- light rows, one wave per row ------------------------------------------------------
template <int VEC, int CT, int WMODE, int U, int BUF>
__global__ __launch_bounds__(256) void spmm_w64_kernel(SpmmArgs a) {
  typedef typename VecT<VEC>::type vec_t;
  const int lane = threadIdx.x & 63;
  const int lb = logical_block(a);
  if (lb >= a.nblocks) return;
  int row = lb * 4 + (threadIdx.x >> 6);
  if (row >= a.n_rows) return;
  row = __builtin_amdgcn_readfirstlane(row);
  const int64_t e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
  if (e1 - e0 > (int64_t)a.heavy_thr) return;  // done by spmm_heavy_kernel
  bool colok[CT];
  uint32_t voff[CT];
  vec_t acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    colok[ct] = (lane + ct * 64) * VEC < a.ncols;
    voff[ct] = colok[ct] ? (uint32_t)((lane + ct * 64) * VEC * 4) : 0u;
    acc[ct] = vzero<VEC>();
  }
  const float roww = (WMODE == 0) ? a.rw[row] : 0.f;
  float* o = a.out + (int64_t)row * a.ld + lane * VEC;
  if (a.accumulate) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      if (colok[ct]) acc[ct] = *reinterpret_cast<const vec_t*>(o + ct * 64 * VEC);
  }
  wave_accumulate<VEC, CT, WMODE, U, BUF>(a, lane, e0, e1, 64, roww, voff, acc);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
    if (colok[ct]) *reinterpret_cast<vec_t*>(o + ct * 64 * VEC) = a.relu ? vrelu<VEC>(acc[ct]) : acc[ct];
}

// ---- heavy rows, one 1024-thread workgroup per row ------------------------------------
constexpr int HEAVY_WAVES = 16;
template <int VEC, int CT, int WMODE, int U, int BUF>
__global__ __launch_bounds__(HEAVY_WAVES * 64) void spmm_heavy_kernel(SpmmArgs a) {
  typedef typename VecT<VEC>::type vec_t;
  extern __shared__ __attribute__((aligned(16))) float red[];  // [HEAVY_WAVES][CT*64*VEC]
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int row = (int)a.row_list[blockIdx.x];
  const int64_t e0 = a.rowptr[row], e1 = a.rowptr[row + 1];
  uint32_t voff[CT];
  vec_t acc[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    const bool ok = (lane + ct * 64) * VEC < a.ncols;
    voff[ct] = ok ? (uint32_t)((lane + ct * 64) * VEC * 4) : 0u;
    acc[ct] = vzero<VEC>();
  }
  const float roww = (WMODE == 0) ? a.rw[row] : 0.f;
  wave_accumulate<VEC, CT, WMODE, U, BUF>(a, lane, e0 + (int64_t)wave * 64, e1,
                                     (int64_t)HEAVY_WAVES * 64, roww, voff, acc);
  constexpr int W = CT * 64 * VEC;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
    *reinterpret_cast<vec_t*>(&red[wave * W + (ct * 64 + lane) * VEC]) = acc[ct];
  __syncthreads();
  for (int c = threadIdx.x; c < a.ncols; c += HEAVY_WAVES * 64) {
    float s = a.accumulate ? a.out[(int64_t)row * a.ld + c] + red[c] : red[c];
#pragma unroll
    for (int w = 1; w < HEAVY_WAVES; ++w) s = s + red[w * W + c];
    a.out[(int64_t)row * a.ld + c] = (a.relu && !(s > 0.f)) ? 0.f : s;
  }
}

// ---- light rows, narrow features: 64/G rows per wave ----------------------------------
template <int VEC, int G, int WMODE>
__global__ __launch_bounds__(256) void spmm_sub_kernel(SpmmArgs a) {
  typedef typename VecT<VEC>::type vec_t;
  constexpr int RPW = 64 / G;
  constexpr int U = 4;
  const int lane = threadIdx.x & 63;
  const int sub = lane / G, sl = lane % G;
  const int lb = logical_block(a);
  if (lb >= a.nblocks) return;
  const int64_t row = ((int64_t)lb * 4 + (threadIdx.x >> 6)) * RPW + sub;
  int64_t e0 = 0, e1 = 0;
  bool active = row < a.n_rows;
  if (active) {
    e0 = a.rowptr[row];
    e1 = a.rowptr[row + 1];
    if (e1 - e0 > (int64_t)a.heavy_thr) { active = false; e1 = e0; }
  }
  const bool colok = sl * VEC < a.ncols;
  vec_t acc = vzero<VEC>();
  if (a.accumulate && active && colok) acc = *reinterpret_cast<const vec_t*>(a.out + row * a.ld + sl * VEC);
  const float roww = (WMODE == 0 && active) ? a.rw[row] : 0.f;
  const int head = (WMODE >= 3 && colok) ? (sl * VEC) / a.dh : 0;
  const float* inl = a.in + sl * VEC;
  for (int64_t e = e0; e < e1; e += U) {
    // every lane of the group reads the same colidx/weight address (hardware broadcast)
    uint32_t cj[U];
    float wj[U];
    vec_t x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = e + u < e1;
      cj[u] = ok ? a.col[e + u] : 0u;
      wj[u] = (WMODE == 0) ? roww : (ok ? load_edge_w<WMODE>(a, e + u, head) : 0.f);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool ok = e + u < e1;
      x[u] = (ok && colok) ? *reinterpret_cast<const vec_t*>(inl + (int64_t)cj[u] * a.ld) : vzero<VEC>();
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (e + u < e1) vacc<VEC>(acc, wj[u], x[u]);
  }
  if (active && colok) *reinterpret_cast<vec_t*>(a.out + row * a.ld + sl * VEC) = a.relu ? vrelu<VEC>(acc) : acc;
}

// ---- dispatch --------------------------------------------------------------------------
template <int VEC, int CT, int WMODE, int U, int BUF>
int launch_w64_u(gaib_ctx* ctx, const gaib_graph* g, SpmmArgs a) {
  // heavy rows first (few, long): their tail hides under the light kernel's start
  if (g->n_heavy > 0) {
    SpmmArgs h = a;
    h.row_list = g->heavy_rows;
    size_t lds = sizeof(float) * HEAVY_WAVES * CT * 64 * VEC;
    ProfScope ps(ctx, "spmm_heavy");
    spmm_heavy_kernel<VEC, CT, WMODE, U, BUF><<<dim3((unsigned)g->n_heavy), HEAVY_WAVES * 64, lds,
                                               ctx->stream>>>(h);
    GAIB_LAUNCH_CHECK();
  }
  a.nblocks = (int)cdiv64(a.n_rows, 4);
  unsigned grid = (unsigned)a.nblocks;
  if (ctx->spmm_xcd_swizzle && a.nblocks >= 64) {
    a.per_xcd = (int)cdiv64(a.nblocks, 8);
    grid = (unsigned)a.per_xcd * 8u;
  }
  if (grid > 0) {
    ProfScope ps(ctx, "spmm_light");
    spmm_w64_kernel<VEC, CT, WMODE, U, BUF><<<dim3(grid), 256, 0, ctx->stream>>>(a);
    GAIB_LAUNCH_CHECK();
  }
  return GAIB_OK;
}

template <int VEC, int CT, int WMODE>
int launch_w64(gaib_ctx* ctx, const gaib_graph* g, SpmmArgs a) {
  // gathers in flight per wave: sized so the destination registers stay <= 32 VGPRs
  constexpr int U = (VEC * CT >= 8) ? 4 : (VEC * CT >= 4 ? 8 : 16);
  int gm = (a.in_bytes != 0 && ctx->spmm_addr_mode != 2) ? 1 : 0;
  if (gm == 1 && ctx->spmm_gather_mode == 2) gm = 2;
  if (gm == 1 && ctx->spmm_gather_mode == 3 && a.col_flagged) {
    gm = 3;
    a.col = a.col_flagged;
  }
  const bool u8 = ctx->spmm_unroll == 8 && U > 8;
  switch (gm) {
    case 0: return u8 ? launch_w64_u<VEC, CT, WMODE, 8, 0>(ctx, g, a) : launch_w64_u<VEC, CT, WMODE, U, 0>(ctx, g, a);
    case 2: return u8 ? launch_w64_u<VEC, CT, WMODE, 8, 2>(ctx, g, a) : launch_w64_u<VEC, CT, WMODE, U, 2>(ctx, g, a);
    case 3: return u8 ? launch_w64_u<VEC, CT, WMODE, 8, 3>(ctx, g, a) : launch_w64_u<VEC, CT, WMODE, U, 3>(ctx, g, a);
    default: return u8 ? launch_w64_u<VEC, CT, WMODE, 8, 1>(ctx, g, a) : launch_w64_u<VEC, CT, WMODE, U, 1>(ctx, g, a);
  }
}

template <int VEC, int G, int WMODE>
int launch_sub(gaib_ctx* ctx, const gaib_graph* g, SpmmArgs a) {
  if (g->n_heavy > 0) {
    SpmmArgs h = a;
    h.row_list = g->heavy_rows;
    size_t lds = sizeof(float) * HEAVY_WAVES * 64 * VEC;
    ProfScope ps(ctx, "spmm_heavy");
    spmm_heavy_kernel<VEC, 1, WMODE, 8, 0><<<dim3((unsigned)g->n_heavy), HEAVY_WAVES * 64, lds,
                                                 ctx->stream>>>(h);
    GAIB_LAUNCH_CHECK();
  }
  constexpr int RPW = 64 / G;
  a.nblocks = (int)cdiv64(a.n_rows, 4 * RPW);
  unsigned grid = (unsigned)a.nblocks;
  if (ctx->spmm_xcd_swizzle && a.nblocks >= 64) {
    a.per_xcd = (int)cdiv64(a.nblocks, 8);
    grid = (unsigned)a.per_xcd * 8u;
  }
  if (grid > 0) {
    ProfScope ps(ctx, "spmm_sub");
    spmm_sub_kernel<VEC, G, WMODE><<<dim3(grid), 256, 0, ctx->stream>>>(a);
    GAIB_LAUNCH_CHECK();
  }
  return GAIB_OK;
}

template <int VEC, int WMODE>
int dispatch_ct(gaib_ctx* ctx, const gaib_graph* g, const SpmmArgs& a, int lanes) {
  if (lanes <= 64) return launch_w64<VEC, 1, WMODE>(ctx, g, a);
  if (lanes <= 128) return launch_w64<VEC, 2, WMODE>(ctx, g, a);
  return launch_w64<VEC, 4, WMODE>(ctx, g, a);
}

template <int VEC, int WMODE>
int dispatch_sub(gaib_ctx* ctx, const gaib_graph* g, const SpmmArgs& a, int lanes) {
  if (lanes <= 1) return launch_sub<VEC, 1, WMODE>(ctx, g, a);
  if (lanes <= 2) return launch_sub<VEC, 2, WMODE>(ctx, g, a);
  if (lanes <= 4) return launch_sub<VEC, 4, WMODE>(ctx, g, a);
  if (lanes <= 8) return launch_sub<VEC, 8, WMODE>(ctx, g, a);
  if (lanes <= 16) return launch_sub<VEC, 16, WMODE>(ctx, g, a);
  return launch_sub<VEC, 32, WMODE>(ctx, g, a);
}

template <int WMODE>
int dispatch_vec(gaib_ctx* ctx, const gaib_graph* g, const SpmmArgs& a0, int len) {
  // widest vector the row stride and base pointers allow
  const uintptr_t al = (uintptr_t)a0.in | (uintptr_t)a0.out;
  int vmax = 1;
  if (len % 4 == 0 && (al & 15) == 0) vmax = 4;
  else if (len % 2 == 0 && (al & 7) == 0) vmax = 2;
  if (WMODE >= 3) {  // a lane's VEC columns must sit in one head
    while (vmax > 1 && a0.dh % vmax != 0) vmax >>= 1;
  }
  int variant = ctx->spmm_variant;
  // variant: 0 auto | 1 force w64 with VEC=1 | 2 force w64 VEC=2 | 4 force w64 VEC=4 |
  //          32 force sub-wave G=32 path with the widest vector (two 128-wide rows per wave)
  const int lanes_max = (len + vmax - 1) / vmax;
  bool use_sub = lanes_max < 32;
  int vec = vmax;
  if (!use_sub) {
    // smallest vector that still fits the row in one 64-lane pass (more lanes busy per load)
    if (len <= 64) vec = 1;
    else if (len <= 128 && vmax >= 2) vec = 2;
    else vec = vmax;
    if (WMODE >= 3 && a0.dh % vec != 0) vec = vmax;
  }
  if (variant == 1) { use_sub = false; vec = 1; }
  if (variant == 2 && vmax >= 2) { use_sub = false; vec = 2; }
  if (variant == 4 && vmax >= 4) { use_sub = false; vec = 4; }
  if (variant == 32 && lanes_max <= 32) { use_sub = true; vec = vmax; }
  if (use_sub) {
    SpmmArgs a = a0;
    a.ncols = len;
    const int lanes = (len + vec - 1) / vec;
    if (vec == 4) return dispatch_sub<4, WMODE>(ctx, g, a, lanes);
    if (vec == 2) return dispatch_sub<2, WMODE>(ctx, g, a, lanes);
    return dispatch_sub<1, WMODE>(ctx, g, a, lanes);
  }
  // one launch covers up to 256 lanes' worth of columns; wider rows are done in column slabs
  const int slab = 256 * vec;
  for (int c0 = 0; c0 < len; c0 += slab) {
    SpmmArgs a = a0;
    a.in = a0.in + c0;
    a.out = a0.out + c0;
    a.ncols = (len - c0 < slab) ? (len - c0) : slab;
    const int lanes = (a.ncols + vec - 1) / vec;
    int rc;
    if (vec == 4) rc = dispatch_ct<4, WMODE>(ctx, g, a, lanes);
    else if (vec == 2) rc = dispatch_ct<2, WMODE>(ctx, g, a, lanes);
    else rc = dispatch_ct<1, WMODE>(ctx, g, a, lanes);
    if (rc != GAIB_OK) return rc;
  }
  return GAIB_OK;
}

}  // namespace

static int spmm_impl(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len,
                     const float* d_in, float* d_out, int flags, int heads = 1) {
  const int accumulate = (flags & GAIB_ACCUMULATE) ? 1 : 0;
  GAIB_CHECK(ctx && g, "gaib_spmm: NULL ctx/graph");
  GAIB_CHECK(len >= 0, "gaib_spmm: len < 0");
  GAIB_CHECK(ctx->device == g->device, "gaib_spmm: graph lives on device %d, ctx on %d", g->device,
             ctx->device);
  if (len == 0 || g->nv == 0) return GAIB_OK;
  GAIB_CHECK(d_in && d_out, "gaib_spmm: NULL feature pointer");
  GAIB_CHECK(d_in != d_out, "gaib_spmm: in and out must not alias");
  GAIB_HIP(hipSetDevice(ctx->device));
  GAIB_TRY(gaib_graph_ensure_heavy(ctx, g, ctx->spmm_heavy_threshold));
  SpmmArgs a;
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
  a.nblocks = 0;
  a.per_xcd = 0;
  a.accumulate = accumulate;
  a.relu = (flags & GAIB_RELU) ? 1 : 0;
  a.heads = heads;
  a.dh = heads > 0 ? len / heads : len;
  a.col_flagged = nullptr;
  if (ctx->spmm_gather_mode == 3 && g->nc == g->nv && !g->col_vdata) {
    GAIB_TRY(gaib_graph_ensure_hot_flags(ctx, g, len));
    a.col_flagged = g->colidx_flagged;
  }
  // feature table = nc rows of len floats; the 32-bit buffer path needs it below 4 GB
  const int64_t table_bytes = g->nc * (int64_t)len * 4;
  a.in_bytes = table_bytes < ((int64_t)1 << 32) ? (uint32_t)table_bytes : 0u;
  switch (weight_kind) {
    case GAIB_W_GCN:
      GAIB_TRY(gaib_graph_ensure_w_gcn(ctx, g));
      a.ew = g->w_gcn;
      return dispatch_vec<1>(ctx, g, a, len);
    case GAIB_W_MEAN:
      GAIB_TRY(gaib_graph_ensure_inv_deg(ctx, g));
      a.rw = g->inv_deg;
      return dispatch_vec<0>(ctx, g, a, len);
    case GAIB_W_MEAN_T:
      GAIB_TRY(gaib_graph_ensure_w_mean_t(ctx, g));
      a.ew = g->w_mean_t;
      return dispatch_vec<1>(ctx, g, a, len);
    case GAIB_W_EDGE:
      GAIB_CHECK(d_edge_w || g->ne == 0, "gaib_spmm: GAIB_W_EDGE needs d_edge_w");
      a.ew = d_edge_w;
      if (heads > 1) return dispatch_vec<3>(ctx, g, a, len);
      return dispatch_vec<1>(ctx, g, a, len);
    case GAIB_W_EDGE_T:
      GAIB_CHECK(d_edge_w || g->ne == 0, "gaib_spmm: GAIB_W_EDGE_T needs d_edge_w");
      GAIB_TRY(gaib_graph_ensure_rev(ctx, g));
      a.ew = d_edge_w;
      a.rev = g->rev;
      if (heads > 1) return dispatch_vec<4>(ctx, g, a, len);
      return dispatch_vec<2>(ctx, g, a, len);
    default:
      gaib_set_error("gaib_spmm: unknown weight_kind %d", weight_kind);
      return GAIB_ERR_INVALID;
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

extern "C" int gaib_spmm_mh(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w,
                            int heads, int len, const float* d_in, float* d_out, int flags) {
  GAIB_CHECK(heads >= 1 && len % heads == 0, "gaib_spmm_mh: heads (%d) must divide len (%d)", heads, len);
  GAIB_CHECK(heads == 1 || weight_kind == GAIB_W_EDGE || weight_kind == GAIB_W_EDGE_T,
             "gaib_spmm_mh: per-head weights need GAIB_W_EDGE or GAIB_W_EDGE_T");
  return spmm_impl(ctx, g, weight_kind, d_edge_w, len, d_in, d_out, flags, heads);
}
