// spmm_core.h -- device code shared by the aggregation kernels (spmm.hip, spmm_gemm.hip):
// launch arguments, the feature-row gather and the per-wave edge loop.
#pragma once
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
  int64_t ld;   // row stride of the gathered table `in` (floats)
  int64_t ldo;  // row stride of out, of the partial sums continued in accumulate mode and of the fused path's row operands
  int ncols;   // columns handled by this launch (<= 64*VEC*CT), starting at in/out
  int n_rows;
  int heavy_thr;
  const uint32_t* row_list;  // heavy kernel only: row ids (ascending) ...
  const uint32_t* row_order; // ... and the slot each workgroup takes (longest rows first)
  int nblocks;               // light kernels: logical number of row blocks
  int per_xcd;               // ceil(nblocks/8) when swizzled, 0 otherwise
  int xcd_chunk;             // > 0: an XCD takes chunks of this many consecutive row blocks in round robin over the XCDs
                             // (0: one contiguous range of row blocks per XCD)
  uint32_t in_bytes;         // BUF kernels: size of the feature table (< 4 GB)
  const uint32_t* col_flagged;  // GM 3: column ids with the top bit set on cold columns
  int accumulate;            // out += instead of out = (second half of a split aggregation)
  int relu;                  // clamp at 0 on store (activation fused)
  int heads;                 // WMODE 3/4: edge weights are [ne][heads]; head of a column = col / dh
  int dh;
  int compact;               // heavy kernel: row k of row_list is written to out row k (fused path's scratch)
  // PART kernels (row classes of a vertex-range partition, spmm_part.hip) -- ignored by the others:
  const uint32_t* row_map;   // row r of the graph is row row_map[r] of out / the continued partial sums / rows2 / y (NULL: r itself)
  const float* in2;          // column ids >= n_first index this second table (row id - n_first): the halo table behind the
  uint32_t n_first;          //   rank's own rows.  in2 == NULL: one table, n_first = 0xffffffff
  uint32_t in2_bytes;        // BUF kernels: size of the second table (< 4 GB)
};

// the row of the caller's matrices that row r of the graph stands for
template <bool PART>
__device__ __forceinline__ int64_t out_row(const SpmmArgs& a, int64_t r) {
  if constexpr (PART) return a.row_map ? (int64_t)a.row_map[r] : r;
  else return r;
}

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

// blockIdx -> row block.  Workgroups are dealt to the 8 XCDs round robin (XCD = blockIdx & 7), each XCD has its own L2.
//   per_xcd > 0, xcd_chunk == 0: XCD x owns the contiguous range [x * per_xcd, (x + 1) * per_xcd) of row blocks --
//     consecutive row blocks share an L2, but a graph whose long rows sit together (a degree-sorted numbering) loads
//     one XCD with most of the edges (scripts/locality_study.py --order degree: 2.5x slower);
//   xcd_chunk = C: XCD x takes the chunks x, x + 8, x + 16, ... of C consecutive row blocks -- still C consecutive
//     row blocks per L2 at a time, and the edges spread over the XCDs at C-block granularity.
__device__ __forceinline__ int logical_block(const SpmmArgs& a) {
  int b = blockIdx.x;
  if (a.per_xcd > 0) {
    const int x = b & 7, k = b >> 3;
    if (a.xcd_chunk > 0) b = ((k / a.xcd_chunk) * 8 + x) * a.xcd_chunk + (k % a.xcd_chunk);
    else b = x * a.per_xcd + k;
  }
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
// PART: two tables -- column ids below n_first index `in`, the others `in2` (a rank's own rows and its halo table, which
// live in different allocations); the choice is a scalar select on the (wave-uniform) column id.
template <int VEC, int GM, bool PART = false>
struct RowGather {
  static constexpr bool BUF = GM != 0;
  __amdgpu_buffer_rsrc_t rsrc, rsrc2;
  const char *inb, *inb2;  // inb2 is biased by -n_first rows: row base = inb2 + col * ldb
  int64_t ldb;
  uint32_t n_first;
  __device__ __forceinline__ RowGather(const SpmmArgs& a) {
    inb = reinterpret_cast<const char*>(a.in);
    ldb = a.ld * 4;
    if constexpr (BUF) rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.in, 0, (int)a.in_bytes, 0x00020000);
    if constexpr (PART) {
      n_first = a.in2 ? a.n_first : 0xffffffffu;
      inb2 = reinterpret_cast<const char*>(a.in2) - (int64_t)a.n_first * ldb;
      if constexpr (BUF) rsrc2 = __builtin_amdgcn_make_buffer_rsrc((void*)a.in2, 0, (int)a.in2_bytes, 0x00020000);
    }
  }
  __device__ __forceinline__ typename VecT<VEC>::type load(uint32_t cj, uint32_t voff) const {
    typedef typename VecT<VEC>::type vec_t;
    if constexpr (PART) {
      static_assert(GM == 0 || GM == 1, "two-table gathers: plain buffer or global loads");
      const bool second = cj >= n_first;  // scalar
      if constexpr (GM == 1) {
        const int soff = (int)((second ? cj - n_first : cj) * (uint32_t)ldb);
        return load_rsrc<0>(second ? rsrc2 : rsrc, soff, voff);  // (scalar selects: straight-line code)
      } else {
        const char* rowp = (second ? inb2 : inb) + (int64_t)cj * ldb;
        return *reinterpret_cast<const vec_t*>(rowp + voff);
      }
    } else if constexpr (GM == 3) {
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
    return load_rsrc<AUX>(rsrc, (int)(cj * (uint32_t)ldb), voff);
  }
  template <int AUX>
  __device__ __forceinline__ typename VecT<VEC>::type load_rsrc(__amdgpu_buffer_rsrc_t rsrc, int soff, uint32_t voff) const {
    typedef typename VecT<VEC>::type vec_t;
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
// PRE: the column ids (and weights) of the row's FIRST chunk were requested by the caller ahead of time (c_first / w_first,
// lane l = edge eb + l, 0 past the row's end) -- the fused kernel asks for row r + 1's while row r's gathers are in flight.
template <int VEC, int CT, int WMODE, int U, int BUF, bool PART = false, bool PRE = false>
__device__ __forceinline__ void wave_accumulate(const SpmmArgs& a, int lane, int64_t eb, int64_t ee,
                                                int64_t chunk_stride, float roww,
                                                const uint32_t (&voff)[CT],
                                                typename VecT<VEC>::type (&acc)[CT], uint32_t c_first = 0u,
                                                float w_first = 0.f) {
  typedef typename VecT<VEC>::type vec_t;
  const RowGather<VEC, BUF, PART> gather(a);
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
    if (PRE && base == eb) {  // (wave-uniform)
      c = c_first;
      w = w_first;
    } else if (lane < n) {
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
      // all U gathers are issued before the first one is consumed: without the fence the scheduler may interleave
      // loads and uses to save registers (seen in the two-product fused kernel: vmcnt(1) after every load)
      __builtin_amdgcn_sched_barrier(0);
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
      __builtin_amdgcn_sched_barrier(0);
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

}  // namespace
