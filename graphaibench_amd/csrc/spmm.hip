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
#include "spmm_core.h"

namespace {

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
  float* o = a.out + (int64_t)row * a.ldo + lane * VEC;
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
  const int slot = (int)a.row_order[blockIdx.x];
  const int row = (int)a.row_list[slot];
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
  float* orow = a.out + (a.compact ? (int64_t)slot : (int64_t)row) * a.ldo;
  for (int c = threadIdx.x; c < a.ncols; c += HEAVY_WAVES * 64) {
    float s = a.accumulate ? orow[c] + red[c] : red[c];
#pragma unroll
    for (int w = 1; w < HEAVY_WAVES; ++w) s = s + red[w * W + c];
    orow[c] = (a.relu && !(s > 0.f)) ? 0.f : s;
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
  if (a.accumulate && active && colok) acc = *reinterpret_cast<const vec_t*>(a.out + row * a.ldo + sl * VEC);
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
  if (active && colok) *reinterpret_cast<vec_t*>(a.out + row * a.ldo + sl * VEC) = a.relu ? vrelu<VEC>(acc) : acc;
}

// ---- dispatch --------------------------------------------------------------------------
template <int VEC, int CT, int WMODE, int U, int BUF>
int launch_w64_u(gaib_ctx* ctx, const gaib_graph* g, SpmmArgs a) {
  // heavy rows first (few, long): their tail hides under the light kernel's start
  if (g->n_heavy > 0) {
    SpmmArgs h = a;
    h.row_list = g->heavy_rows;
    h.row_order = g->heavy_rows + g->n_heavy;
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
    if (ctx->spmm_xcd_swizzle == 2 && a.nblocks >= 8 * 64) {  // chunks of 64 row blocks (256 rows) round robin
      a.xcd_chunk = 64;
      a.per_xcd = (int)(cdiv64(a.per_xcd, 64) * 64);
    }
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
    h.row_order = g->heavy_rows + g->n_heavy;
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
  // Measured on the products-shaped graph (scripts/microbench.py, every width from 4 to 256): one row per wave
  // with 16 gathers in flight beats the packed sub-wave kernel at EVERY width (D=64: 4.3 vs 5.6 ms, D=32: 2.4
  // vs 3.8, D=4: 2.2 vs 3.2), and among the one-row kernels the narrowest lane vector that covers the row in
  // at most two passes wins (D=64: 4 B lanes 4.3 ms, 16 B lanes 5.3; D=256: 8 B lanes 17.4, 16 B lanes 17.9).
  // The sub-wave kernel stays reachable as variant 32.
  const int lanes_max = (len + vmax - 1) / vmax;
  bool use_sub = false;
  int vec = vmax;
  if (len <= 64) vec = 1;
  else if (len <= 256 && vmax >= 2) vec = 2;
  else if (len <= 128) vec = 1;
  if (WMODE >= 3 && a0.dh % vec != 0) vec = 1;
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

// ---- aggregation fused with the dense product ------------------------------------------
//   agg[i,:] = sum_e w_e * in[col_e,:]          (the aggregation above, one wave per row)
//   y[i,:]   = act(agg[i,:] . op(W))             on the matrix cores, inside the same wave
// One persistent 1024-thread workgroup per CU.  op(W) is staged ONCE into LDS, k-contiguous
// (wl[n][k]); after that the 16 waves never synchronise again.  A wave takes 16-row tiles off
// a global counter; finished rows are parked eight at a time in the wave's LDS strip [8][K+4]
// and read back in MFMA operand order (lane l: A[i = l&15][k = (l>>4)*K/4 + s] at step s), then
// multiplied with op(W) from LDS: v_mfma_f32_16x16x4_f32, one 16x16 output tile at a time.
// Why LDS and not L2 for op(W): the vector-memory path of a CU is in order, so a weight load
// issued between gathers waits ~5 us behind them (measured: +1.5 ms per pass at products scale);
// from LDS the dense product costs only the y store.
// Heavy rows are aggregated first by spmm_heavy_kernel into a compact scratch and picked up here.
struct FuseArgs {
  const float* wt;            // [n_out][K]
  float* y;                   // [n_rows][ldy]
  int64_t ldy;
  int n_out;                  // multiple of 16
  int relu;
  const float* heavy_agg;     // [n_heavy][K]
  const uint32_t* heavy_rows; // ascending
  int n_heavy;
  int* tile_counter;          // zeroed before the launch
  const float* agg_in;        // accumulate mode: partial sums to continue (same layout as the agg rows)
  const float* wt2;           // DUAL: second weight matrix [n_out][K], k-contiguous
  const float* rows2;         // DUAL: second row operand [n_rows][ncols]:  y += rows2[i,:] . op(W2)
  int ldw;                    // row stride of wt / wt2 in floats (== ncols unless the launch handles a K-slab of a wider matrix)
  int y_accum;                // y += instead of y = (the second K-slab of a 129..256-wide aggregation; not with DUAL)
  int tile_xcd;               // 0: one global counter; n > 0: tiles off eight per-XCD counters over interleaved chunks of 2^(n-1) tiles
};

// option spmm_tile_xcd -> FuseArgs::tile_xcd (0 = global counter, else log2(chunk length in tiles) + 1).
// -1 (default): by the graph's numbering -- XCD-affine chunks of 1024 tiles when at least a quarter of the edges stay
// within 32 768 ids of their row, else the global counter.  Measured (scripts/ab_tile_xcd.py, products size, D = 128): a
// numbering with planted locality 5.94 -> 5.01 ms at 512-1024 tiles per chunk (6.2 ms at 16-128: an XCD's 512 waves in
// flight then span 65 536 rows); a random numbering 7.60 -> 7.80 ms with ANY chunk length, and 8.1 ms with eight counters
// WITHOUT the XCD affinity -- i.e. on a random order the one shared in-order window over the streamed arrays is worth 3 %,
// hence the rule instead of one setting.  1 = the round-2 form (16-tile chunks); n = chunk length, rounded down to 2^k.
static int tile_xcd_arg(gaib_ctx* ctx, gaib_graph* g) {
  int v = ctx->spmm_tile_xcd;
  if (v < 0) {
    if (gaib_graph_ensure_locality(ctx, g) != GAIB_OK) return 0;
    v = g->near_frac >= 0.25f ? 1024 : 0;
  }
  if (v <= 0) return 0;
  if (v == 1) v = 16;
  int sh = 0;
  while ((2 << sh) <= v) ++sh;
  return sh + 1;
}

typedef float f32x4_t __attribute__((ext_vector_type(4)));
constexpr int FUSE_ROWS = 16;
constexpr int FUSE_WAVES = 16;

// STRIP = rows a wave parks in LDS at a time (8; 2 with two weight matrices);
// DUAL: y = act(agg . op(W) + rows2 . op(W2)) -- the self term of a SAGE layer (sage_layer.cpp:22,50) in the same pass
// FLAT: short rows (the halo-column half of a partitioned graph has 3-5 edges per row).  Row by row, a wave then has
// one column-id load and a handful of gathers in flight and waits two memory latencies per row (measured 2.3 ms for
// 12 M edges over 2.4 M rows, 1.3 ms of traffic).  Here the edges of a strip's rows are ONE stream: column ids and
// weights are loaded 64 edges at a time, U gathers are in flight whatever rows they belong to, and the running sum
// moves to the next row when the edge index passes a row boundary (wave-uniform control).  Same edge order, same sums.
template <int VEC, int WMODE, int U, int GM, int STRIP, bool DUAL, bool FLAT = false, bool YACC = false>
__global__ __launch_bounds__(FUSE_WAVES * 64) void spmm_gemm_kernel(SpmmArgs a, FuseArgs f) {
  typedef typename VecT<VEC>::type vec_t;
  constexpr int K = 64 * VEC;  // padded inner dimension; a.ncols (<= K) columns are real
  constexpr int KQ = K / 4;
  constexpr int LDT = K + 4;
  constexpr int HALF = STRIP;
  constexpr int NPASS = FUSE_ROWS / STRIP;
  extern __shared__ __attribute__((aligned(16))) float fuse_lds[];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int n_pad = (f.n_out + 15) & ~15;
  float* wl = fuse_lds;                                            // [n_pad][LDT], zero padded
  float* wl2 = fuse_lds + n_pad * LDT;                             // DUAL: the second matrix
  float* tile = fuse_lds + (DUAL ? 2 : 1) * n_pad * LDT + wave * (HALF * LDT);  // [HALF][LDT]
  for (int t = threadIdx.x; t < n_pad * K; t += FUSE_WAVES * 64) {
    const int n = t / K, k = t % K;
    const bool in = n < f.n_out && k < a.ncols;
    wl[n * LDT + k] = in ? f.wt[(int64_t)n * f.ldw + k] : 0.f;
    if constexpr (DUAL) wl2[n * LDT + k] = in ? f.wt2[(int64_t)n * f.ldw + k] : 0.f;
  }
  __syncthreads();  // the only workgroup barrier
  const int i = lane & 15, kq = lane >> 4;
  const bool colok = lane * VEC < a.ncols;
  const uint32_t voff[1] = {colok ? (uint32_t)(lane * VEC * 4) : 0u};
  const int ntiles = (a.n_rows + FUSE_ROWS - 1) / FUSE_ROWS;
  // FLAT: tiles are cheap (a few edges per row), and one atomic per tile on one address becomes the floor (153 k
  // atomics = 0.4 ms at 2.4 M rows).  Guided chunks instead: a wave takes (tiles left) / (4 x waves) tiles at a time,
  // at most 8, down to single tiles at the end.
  // XCD-affine supply (option spmm_tile_xcd = chunk length in tiles): tiles come off EIGHT counters, one per XCD
  // (workgroups are dealt to the XCDs round robin: XCD = blockIdx & 7).  XCD x owns the chunks x, x + 8, x + 16, ... of
  // 2^tsh consecutive tiles: consecutive rows of a graph with locality in its numbering meet in ONE L2 instead of being
  // spread over all eight, and all XCDs advance through the rows at the same pace, so a numbering with its long rows at
  // one end stays balanced.  The chunk has to be LONG: an XCD's 512 waves hold 512 tiles = 8 192 rows at any time, and
  // with 16-tile chunks those are 32 chunks spread over 65 536 rows of the numbering (round 2: natural order 7.0 ms, no
  // better than the global counter's 6.7); with chunks of 512 tiles they are one run of consecutive rows.  An XCD whose
  // chunks are used up steals from the XCD that has the most tiles left.
  const int tsh = f.tile_xcd - 1;  // log2 of the chunk length in tiles (tile_xcd = 0: one global counter)
  int own = f.tile_xcd ? (int)(blockIdx.x & 7) : 0;  // the XCD whose counter this wave is drawing from
  int k_next = 0, k_left = 0;
  const int nwaves4 = ((int)gridDim.x * FUSE_WAVES * 4) / 8 > 0 ? ((int)gridDim.x * FUSE_WAVES * 4) / 8 : 1;
  const int per_xcd_tiles = (ntiles + 7) / 8;  // about what one XCD's chunks hold
  const int n_chunks = f.tile_xcd ? (ntiles + (1 << tsh) - 1) >> tsh : 0;
  for (;;) {
    if (k_left == 0) {
      int want = 1;
      if constexpr (FLAT) {  // guided: (tiles this XCD has left) / (4 x its waves), at most 8, single tiles at the end
        const int left = f.tile_xcd ? per_xcd_tiles - k_next : (ntiles - k_next) / 8;
        want = (left > 0 ? left : 0) / nwaves4;
        want = want < 1 ? 1 : (want > 8 ? 8 : want);
      }
      int k0 = 0;
      if (lane == 0) k0 = atomicAdd(f.tile_counter + own, want);
      k0 = __builtin_amdgcn_readfirstlane(k0);
      if (!f.tile_xcd) {  // one global counter: tiles in order
        if (k0 >= ntiles) break;
      } else if ((((k0 >> tsh) * 8 + own) << tsh) >= ntiles) {
        // this XCD's chunks are used up: steal from the XCD that has the most tiles left (lane x looks at counter x;
        // counters only grow, so a look that says "nothing left anywhere" is final)
        int left = 0;
        if (lane < 8) {
          const int mine = n_chunks > lane ? ((n_chunks - lane + 7) >> 3) << tsh : 0;  // tiles in XCD `lane`'s chunks
          left = mine - __hip_atomic_load(f.tile_counter + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        int best = -1, best_left = 0;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
          const int lx = __builtin_amdgcn_readlane(left, x);
          if (lx > best_left) best_left = lx, best = x;
        }
        if (best < 0) break;
        own = best;
        k_next = 0;
        continue;
      }
      k_next = k0;
      k_left = want;
    }
    const int kk = k_next++;
    --k_left;
    const int t = f.tile_xcd ? ((((kk >> tsh) * 8 + own) << tsh) + (kk & ((1 << tsh) - 1))) : kk;
    if (t >= ntiles) continue;  // the ragged end of the last chunk
    const int row0 = t * FUSE_ROWS;
    // the 17 row boundaries of this tile, lane r holds rowptr[row0 + r]
    int rpi = row0 + (lane < FUSE_ROWS ? lane : FUSE_ROWS);
    if (rpi > a.n_rows) rpi = a.n_rows;
    const int64_t rp = a.rowptr[rpi];
    const int rp_lo = (int)(uint32_t)(rp & 0xffffffffll), rp_hi = (int)(rp >> 32);
    float af[KQ];
#pragma unroll
    for (int s = 0; s < KQ; ++s) af[s] = 0.f;
    unsigned long long heavy_mask = 0;  // FLAT: bit r = row r of the tile is a heavy row
    float rwv = 0.f;                    // FLAT, WMODE 0: lane r holds the row weight of row r
    if constexpr (FLAT) {
      const int nb = lane < 63 ? lane + 1 : 63;
      const int64_t rp_next = ((int64_t)__shfl(rp_hi, nb) << 32) | (uint32_t)__shfl(rp_lo, nb);
      heavy_mask = __ballot(lane < FUSE_ROWS && rp_next - rp > (int64_t)a.heavy_thr);
      if constexpr (WMODE == 0) {
        int rwi = row0 + (lane < FUSE_ROWS ? lane : 0);
        if (rwi >= a.n_rows) rwi = a.n_rows - 1;
        rwv = a.rw[rwi];
      }
    }
    for (int h = 0; h < NPASS; ++h) {
      bool flat_done = false;
      if constexpr (FLAT) {
        if (((heavy_mask >> (h * HALF)) & ((1ull << HALF) - 1)) == 0) {
          flat_done = true;
          const int rbase = h * HALF;
          auto rp_at = [&](int rr) -> int64_t {
            return ((int64_t)__builtin_amdgcn_readlane(rp_hi, rr) << 32) | (uint32_t)__builtin_amdgcn_readlane(rp_lo, rr);
          };
          const int64_t e_lo = rp_at(rbase), e_hi = rp_at(rbase + HALF);
          const RowGather<VEC, GM> gather(a);
          float* trow_w = tile + lane * VEC;  // this lane's columns of strip row 0
          if (f.agg_in) {
            // accumulate mode: the strip starts out as the partial sums of its rows (all requests first)
            vec_t t[HALF];
#pragma unroll
            for (int r2 = 0; r2 < HALF; ++r2) {
              int row = row0 + rbase + r2;
              if (row >= a.n_rows) row = a.n_rows - 1;
              t[r2] = *reinterpret_cast<const vec_t*>(f.agg_in + (int64_t)row * a.ldo + (colok ? lane * VEC : 0));
            }
#pragma unroll
            for (int r2 = 0; r2 < HALF; ++r2)
              *reinterpret_cast<vec_t*>(trow_w + r2 * LDT) = colok ? t[r2] : vzero<VEC>();
          }
          int r = 0;
          int64_t row_end = rp_at(rbase + 1);
          vec_t acc = f.agg_in ? *reinterpret_cast<const vec_t*>(trow_w) : vzero<VEC>();
          float roww = (WMODE == 0) ? readlane_f(rwv, rbase) : 0.f;
          auto flush = [&]() {  // row r is complete: store it, park it, open row r + 1
            const int row = row0 + rbase + r;
            if (row < a.n_rows && a.out && colok)
              *reinterpret_cast<vec_t*>(a.out + (int64_t)row * a.ldo + lane * VEC) = acc;
            *reinterpret_cast<vec_t*>(trow_w + r * LDT) = colok ? acc : vzero<VEC>();
            ++r;
            if (r < HALF) {
              row_end = rp_at(rbase + r + 1);
              acc = f.agg_in ? *reinterpret_cast<const vec_t*>(trow_w + r * LDT) : vzero<VEC>();
              if constexpr (WMODE == 0) roww = readlane_f(rwv, rbase + r);
            }
          };
          for (int64_t base = e_lo; base < e_hi; base += 64) {
            const int64_t rem = e_hi - base;
            const int n = rem < 64 ? (int)rem : 64;
            uint32_t c = 0;
            float w = 0.f;
            if (lane < n) {
              c = a.col[base + lane];
              if constexpr (WMODE == 1 || WMODE == 2) w = load_edge_w<WMODE>(a, base + lane);
            }
            vec_t x[U];
            int j = 0;
            for (; j + U <= n; j += U) {
#pragma unroll
              for (int u = 0; u < U; ++u)
                x[u] = gather.load((uint32_t)__builtin_amdgcn_readlane((int)c, j + u), voff[0]);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int u = 0; u < U; ++u) {
                while (base + j + u == row_end) flush();
                vacc<VEC>(acc, (WMODE == 0) ? roww : readlane_f(w, j + u), x[u]);
              }
            }
            const int rest = n - j;
            if (rest > 0) {  // power-of-two pieces, all requests first (as in wave_accumulate)
              int jj = j;
#pragma unroll
              for (int p = U / 2; p >= 1; p >>= 1) {
                if (rest & p) {
#pragma unroll
                  for (int u = 0; u < p; ++u)
                    x[p + u] = gather.load((uint32_t)__builtin_amdgcn_readlane((int)c, jj + u), voff[0]);
                  jj += p;
                }
              }
              __builtin_amdgcn_sched_barrier(0);
              jj = j;
#pragma unroll
              for (int p = U / 2; p >= 1; p >>= 1) {
                if (rest & p) {
#pragma unroll
                  for (int u = 0; u < p; ++u) {
                    while (base + jj + u == row_end) flush();
                    vacc<VEC>(acc, (WMODE == 0) ? roww : readlane_f(w, jj + u), x[p + u]);
                  }
                  jj += p;
                }
              }
            }
          }
          while (r < HALF) flush();  // the row in progress and the empty rows behind the last edge
        }
      }
      for (int r = 0; r < HALF && !flat_done; ++r) {
        const int rr = h * HALF + r;
        const int row = row0 + rr;
        vec_t acc[1];
        acc[0] = vzero<VEC>();
        if (row < a.n_rows) {
          if (f.agg_in && colok)
            acc[0] = *reinterpret_cast<const vec_t*>(f.agg_in + (int64_t)row * a.ldo + lane * VEC);
          const int64_t e0 = ((int64_t)__builtin_amdgcn_readlane(rp_hi, rr) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane(rp_lo, rr);
          const int64_t e1 = ((int64_t)__builtin_amdgcn_readlane(rp_hi, rr + 1) << 32) |
                             (uint32_t)__builtin_amdgcn_readlane(rp_lo, rr + 1);
          if (e1 - e0 > (int64_t)a.heavy_thr) {
            int lo = 0, hi = f.n_heavy - 1;
            while (lo < hi) {
              const int mid = (lo + hi) >> 1;
              if (f.heavy_rows[mid] < (uint32_t)row) lo = mid + 1;
              else hi = mid;
            }
            if (colok) {
              const vec_t hv = *reinterpret_cast<const vec_t*>(f.heavy_agg + (int64_t)lo * a.ldo + lane * VEC);
              if constexpr (VEC == 1) acc[0] = f.agg_in ? acc[0] + hv : hv;
              else {
#pragma unroll
                for (int e = 0; e < VEC; ++e) acc[0][e] = f.agg_in ? acc[0][e] + hv[e] : hv[e];
              }
            }
          } else {
            const float roww = (WMODE == 0) ? a.rw[row] : 0.f;
            wave_accumulate<VEC, 1, WMODE, U, GM>(a, lane, e0, e1, 64, roww, voff, acc);
          }
          if (a.out && colok) *reinterpret_cast<vec_t*>(a.out + (int64_t)row * a.ldo + lane * VEC) = acc[0];
        }
        // lanes beyond the real columns gathered column 0 (see wave_accumulate): they must enter the product as 0
        if (!colok) acc[0] = vzero<VEC>();
        *reinterpret_cast<vec_t*>(tile + r * LDT + lane * VEC) = acc[0];
      }
      // LDS operations of one wave complete in order; the fences keep the compiler from moving
      // the fragment reads above the row stores (and the next half's stores above the reads)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const bool mine = (i / HALF) == h;  // lanes whose A row sits in this strip
      const float* trow = tile + (i % HALF) * LDT + kq * KQ;
#pragma unroll
      for (int s4 = 0; s4 < KQ / 4; ++s4) {
        const f32x4_t tv = *reinterpret_cast<const f32x4_t*>(trow + 4 * s4);
#pragma unroll
        for (int e = 0; e < 4; ++e) af[4 * s4 + e] = mine ? tv[e] : af[4 * s4 + e];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const float* wbase = wl + i * LDT + kq * KQ;
    auto mfma_tile = [&](const float* wr, f32x4_t c) {
#pragma unroll
      for (int s4 = 0; s4 < KQ / 4; ++s4) {
        const f32x4_t b = *reinterpret_cast<const f32x4_t*>(wr + 4 * s4);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * s4 + 0], b[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * s4 + 1], b[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * s4 + 2], b[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x4f32(af[4 * s4 + 3], b[3], c, 0, 0, 0);
      }
      return c;
    };
    auto store_tile = [&](int n0, const f32x4_t& c) {
      if (n0 + i < f.n_out) {
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int row = row0 + 4 * kq + reg;  // C/D layout: row = 4*(lane>>4) + reg, col = lane&15
          if (row < a.n_rows) {
            float v = c[reg];
            if constexpr (YACC) v += f.y[(int64_t)row * f.ldy + n0 + i];  // the second K-slab of a wide aggregation
            if (f.relu) v = v > 0.f ? v : 0.f;
            f.y[(int64_t)row * f.ldy + n0 + i] = v;
          }
        }
      }
    };
    if constexpr (!DUAL && YACC) {
      // y += : the tile's OLD values are requested all at once, before the matrix-core phase, instead of one dependent
      // load per element right before its store (the second K-slab ran 8.15 ms against 7.6 for the first: 0.5 ms of
      // exposed latency per launch).  n_out <= 256: at most 16 output tiles x 4 values per lane; the gather registers are
      // dead by now.
      // Two batches of 8 tiles (32 registers each): all 16 at once spilled three registers at the kernel's 128-VGPR cap.
      constexpr int NTY = 8;
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        if (nb * NTY * 16 >= n_pad) break;
        f32x4_t yo[NTY];
#pragma unroll
        for (int q = 0; q < NTY; ++q) {
          const int nt = nb * NTY + q;
          yo[q] = f32x4_t{0.f, 0.f, 0.f, 0.f};
          if (nt * 16 < n_pad && nt * 16 + i < f.n_out) {
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
              const int row = row0 + 4 * kq + reg;
              yo[q][reg] = f.y[(int64_t)(row < a.n_rows ? row : 0) * f.ldy + nt * 16 + i];
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NTY; ++q) {
          const int nt = nb * NTY + q;
          if (nt * 16 < n_pad) {
            const f32x4_t c = mfma_tile(wbase + nt * 16 * LDT, yo[q]);  // the old values seed the accumulators
            const int n0 = nt * 16;
            if (n0 + i < f.n_out) {
#pragma unroll
              for (int reg = 0; reg < 4; ++reg) {
                const int row = row0 + 4 * kq + reg;
                if (row < a.n_rows) {
                  float v = c[reg];
                  if (f.relu) v = v > 0.f ? v : 0.f;
                  f.y[(int64_t)row * f.ldy + n0 + i] = v;
                }
              }
            }
          }
        }
      }
    } else if constexpr (!DUAL) {
      for (int n0 = 0; n0 < n_pad; n0 += 16) {
        const f32x4_t c = mfma_tile(wbase + n0 * LDT, f32x4_t{0.f, 0.f, 0.f, 0.f});
        store_tile(n0, c);
      }
    } else {
      // second product with the tile's own rows of rows2.  Nothing of it is live while the gathers run (64 more
      // registers there made the compiler serialise them): the 16 rows are requested now, coalesced like gathered
      // rows, the first product runs on the matrix cores while they travel, then they take the same trip through
      // the LDS strip into operand order (reusing af) and the second chain continues the same accumulators.
      constexpr int NT = 8;  // n_pad <= 128 on this path (checked by the launcher)
      vec_t xs[FUSE_ROWS];
#pragma unroll
      for (int r = 0; r < FUSE_ROWS; ++r) {
        const int row = row0 + r;
        const int64_t rs = row < a.n_rows ? row : 0;
        xs[r] = *reinterpret_cast<const vec_t*>(f.rows2 + rs * a.ldo + (colok ? lane * VEC : 0));
      }
      f32x4_t c[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        c[nt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        if (nt * 16 < n_pad) c[nt] = mfma_tile(wbase + nt * 16 * LDT, c[nt]);
      }
#pragma unroll
      for (int h = 0; h < NPASS; ++h) {
#pragma unroll
        for (int r = 0; r < HALF; ++r) {
          const bool ok = colok && (row0 + h * HALF + r < a.n_rows);
          *reinterpret_cast<vec_t*>(tile + r * LDT + lane * VEC) = ok ? xs[h * HALF + r] : vzero<VEC>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const bool mine = (i / HALF) == h;
        const float* trow = tile + (i % HALF) * LDT + kq * KQ;
#pragma unroll
        for (int s4 = 0; s4 < KQ / 4; ++s4) {
          const f32x4_t tv = *reinterpret_cast<const f32x4_t*>(trow + 4 * s4);
#pragma unroll
          for (int e = 0; e < 4; ++e) af[4 * s4 + e] = mine ? tv[e] : af[4 * s4 + e];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nt * 16 < n_pad) {
          c[nt] = mfma_tile(wbase + n_pad * LDT + nt * 16 * LDT, c[nt]);  // same position in wl2
          store_tile(nt * 16, c[nt]);
        }
      }
    }
  }
}

__global__ void transpose_small_kernel(int rows, int cols, const float* in, float* out) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;  // out[c][r] = in[r][c]
  if (t < rows * cols) {
    const int c = t / rows, r = t % rows;
    out[(int64_t)c * rows + r] = in[(int64_t)r * cols + c];
  }
}

// LDS of one fused workgroup: one or two weight matrices [n_pad][K+4] + 16 row strips [strip][K+4]
inline size_t fuse_lds_bytes(int kpad, int n_out, bool dual, int strip) {
  const size_t n_pad = (size_t)((n_out + 15) & ~15);
  return sizeof(float) * (size_t)(kpad + 4) * ((dual ? 2 : 1) * n_pad + (size_t)FUSE_WAVES * strip);
}
inline int fuse_strip_rows(int kpad, int n_out, bool dual) {
  // two products: 2-row strips (measured: the strip height costs nothing) and at most eight 16-wide output tiles
  // (their accumulators stay in registers between the two products)
  if (dual) return (n_out <= 128 && fuse_lds_bytes(kpad, n_out, dual, 2) <= 160 * 1024) ? 2 : 0;
  if (fuse_lds_bytes(kpad, n_out, dual, 8) <= 160 * 1024) return 8;
  // wide outputs (a 128-column slab of op(W) for 256 outputs is 135 KB): 2-row strips
  return fuse_lds_bytes(kpad, n_out, dual, 2) <= 160 * 1024 ? 2 : 0;
}

template <int VEC, int WMODE>
int launch_fused(gaib_ctx* ctx, const gaib_graph* g, SpmmArgs a, FuseArgs f, float* heavy_scratch) {
  constexpr int U = 16;
  constexpr int K = 64 * VEC;
  const bool buf = a.in_bytes != 0 && ctx->spmm_addr_mode != 2;
  if (g->n_heavy > 0) {
    SpmmArgs h = a;
    h.row_list = g->heavy_rows;
    h.row_order = g->heavy_rows + g->n_heavy;
    h.out = heavy_scratch;
    h.compact = 1;  // (row k of the scratch has the stride of the output rows, h.ldo)
    h.relu = 0;
    h.accumulate = 0;
    size_t lds = sizeof(float) * HEAVY_WAVES * 64 * VEC;
    ProfScope ps(ctx, "spmm_heavy");
    if (buf) spmm_heavy_kernel<VEC, 1, WMODE, U, 1><<<dim3((unsigned)g->n_heavy), HEAVY_WAVES * 64, lds, ctx->stream>>>(h);
    else spmm_heavy_kernel<VEC, 1, WMODE, U, 0><<<dim3((unsigned)g->n_heavy), HEAVY_WAVES * 64, lds, ctx->stream>>>(h);
    GAIB_LAUNCH_CHECK();
  }
  const bool dual = f.wt2 != nullptr;
  const int strip = fuse_strip_rows(K, f.n_out, dual);  // 8, 2 or 0 (does not fit: the caller checked)
  const size_t lds = fuse_lds_bytes(K, f.n_out, dual, strip);
  const int64_t ntiles = cdiv64(a.n_rows, FUSE_ROWS);
  const int cus = ctx->spmm_fuse_cus > 0 ? ctx->spmm_fuse_cus : ctx->num_cus;
  const unsigned grid = (unsigned)std::min<int64_t>(cus, cdiv64(ntiles, FUSE_WAVES));
  GAIB_HIP(hipMemsetAsync(f.tile_counter, 0, 8 * sizeof(int), ctx->stream));  // one counter per XCD
  ProfScope ps(ctx, "spmm_gemm_fused");
  // more than 64 KB of dynamic LDS has to be asked for
  // (the edge-stream form keeps 8 gathers in flight, not 16: with 16 the operand fragments of the dense product
  // spill and are reloaded inside the MFMA loop)
#define GAIB_FUSED_LAUNCH_Y(GM, STRIP, DUAL, FLAT, YACC)                                                              \
  do {                                                                                                                \
    constexpr int UU = FLAT ? 8 : U;                                                                                  \
    GAIB_HIP(hipFuncSetAttribute((const void*)spmm_gemm_kernel<VEC, WMODE, UU, GM, STRIP, DUAL, FLAT, YACC>,          \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                            \
    spmm_gemm_kernel<VEC, WMODE, UU, GM, STRIP, DUAL, FLAT, YACC><<<dim3(grid), FUSE_WAVES * 64, lds, ctx->stream>>>( \
        a, f);                                                                                                        \
  } while (0)
#define GAIB_FUSED_LAUNCH(GM, STRIP, DUAL, FLAT) GAIB_FUSED_LAUNCH_Y(GM, STRIP, DUAL, FLAT, false)
  // short rows (fewer than 12 edges per row on average: halo-column halves, citation graphs): the edge-stream form
  // (scripts/ab_flat.py: -34 % at 3 edges per row, -20 % at 5, even at 12, +2 % at 30)
  const bool flat = !dual && strip == 8 && !f.y_accum &&
                    (ctx->spmm_flat == 1 || (ctx->spmm_flat < 0 && g->ne < 12 * (int64_t)a.n_rows));
  if (f.y_accum) {  // the second K-slab of a 129..256-column aggregation (VEC == 2 only; never dual or flat)
    if constexpr (VEC == 2) {
      if (buf) {
        if (strip == 2) GAIB_FUSED_LAUNCH_Y(1, 2, false, false, true);
        else GAIB_FUSED_LAUNCH_Y(1, 8, false, false, true);
      } else {
        if (strip == 2) GAIB_FUSED_LAUNCH_Y(0, 2, false, false, true);
        else GAIB_FUSED_LAUNCH_Y(0, 8, false, false, true);
      }
    }
  } else if (buf) {
    if (dual) GAIB_FUSED_LAUNCH(1, 2, true, false);
    else if (strip == 2) GAIB_FUSED_LAUNCH(1, 2, false, false);
    else if (flat) GAIB_FUSED_LAUNCH(1, 8, false, true);
    else GAIB_FUSED_LAUNCH(1, 8, false, false);
  } else {
    if (dual) GAIB_FUSED_LAUNCH(0, 2, true, false);
    else if (strip == 2) GAIB_FUSED_LAUNCH(0, 2, false, false);
    else if (flat) GAIB_FUSED_LAUNCH(0, 8, false, true);
    else GAIB_FUSED_LAUNCH(0, 8, false, false);
  }
#undef GAIB_FUSED_LAUNCH
#undef GAIB_FUSED_LAUNCH_Y
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

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
    ProfScope ps(ctx, "spmm_chunk");
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
  ProfScope ps(ctx, "spmm_chunk_reduce");
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
static int spmm_setup(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len,
                      const float* d_in, float* d_out, int flags, int heads, SpmmArgs* pa, int* wmode) {
  SpmmArgs& a = *pa;
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
  if (ctx->spmm_gather_mode == 3 && g->nc == g->nv && !g->col_vdata) {
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
    if (ctx->spmm_pad && !ctx->forked && len > 16 && stride != bytes && g->ne > 4 * g->nc &&
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
  const int64_t table_bytes = g->nc * a.ld * 4;
  a.in_bytes = table_bytes < ((int64_t)1 << 32) ? (uint32_t)table_bytes : 0u;
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

static int spmm_impl(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len,
                     const float* d_in, float* d_out, int flags, int heads = 1) {
  GAIB_CHECK(ctx && g, "gaib_spmm: NULL ctx/graph");
  GAIB_CHECK(len >= 0, "gaib_spmm: len < 0");
  GAIB_CHECK(ctx->device == g->device, "gaib_spmm: graph lives on device %d, ctx on %d", g->device,
             ctx->device);
  if (len == 0 || g->nv == 0) return GAIB_OK;
  GAIB_CHECK(d_in && d_out, "gaib_spmm: NULL feature pointer");
  GAIB_CHECK(d_in != d_out, "gaib_spmm: in and out must not alias");
  SpmmArgs a;
  int wmode = 0;
  GAIB_TRY(spmm_setup(ctx, g, weight_kind, d_edge_w, len, d_in, d_out, flags, heads, &a, &wmode));
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

// agg = A.in ; out = act(agg . op(W) [+ rows2 . op(W2)]).  Fused on the matrix cores when the shape allows,
// otherwise gaib_spmm followed by gaib_sgemm (same results up to summation order).
static int spmm_gemm_impl(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len_in,
                          const float* d_in, float* d_agg, const float* d_W, int transW, const float* d_rows2,
                          const float* d_W2, int len_out, float* d_out, int flags) {
  GAIB_CHECK(ctx && g, "gaib_spmm_gemm: NULL ctx/graph");
  GAIB_CHECK(len_in >= 0 && len_out >= 0, "gaib_spmm_gemm: negative length");
  GAIB_CHECK(ctx->device == g->device, "gaib_spmm_gemm: graph lives on device %d, ctx on %d", g->device,
             ctx->device);
  GAIB_CHECK((flags & ~(GAIB_RELU | GAIB_AGG_SCRATCH | GAIB_ACCUMULATE)) == 0, "gaib_spmm_gemm: unsupported flags %d",
             flags);
  GAIB_CHECK((d_rows2 == nullptr) == (d_W2 == nullptr), "gaib_spmm_gemm2: rows2 and W2 go together");
  if (g->nv == 0 || len_out == 0) return GAIB_OK;
  GAIB_CHECK(d_in && d_agg && d_W && d_out, "gaib_spmm_gemm: NULL pointer");
  GAIB_CHECK(d_in != d_agg && d_agg != d_out && d_in != d_out && d_rows2 != d_out && d_rows2 != d_agg,
             "gaib_spmm_gemm: buffers must not alias");
  const bool dual = d_rows2 != nullptr;
  const uintptr_t al = (uintptr_t)d_in | (uintptr_t)d_agg;
  // the weight matrices [len_out x (len_in+4)] and 16 row strips must fit the CU's 160 KB of LDS;
  // 65..128 columns need 8-byte lanes
  const int kpad = len_in <= 64 ? 64 : 128;
  const bool lanes_ok = len_in <= 64 ? true : (len_in % 2 == 0 && (al & 7) == 0);
  // dense graphs over a cache-sized table aggregate faster by ordered chunks (spmm_chunk_kernel) than row by row
  // inside the fused kernel: two kernels there
  const bool dense = ctx->spmm_chunked != 0 && len_in % 4 == 0 && (al & 15) == 0 &&
                     (ctx->spmm_chunked == 1 || chunk_rule(ctx, g, len_in));
  const bool shape_ok = ctx->spmm_fuse != 0 && !dense && len_in >= 1 && len_in <= 128 && lanes_ok && g->ne > 0 &&
                        g->nv >= 1 &&
                        (weight_kind == GAIB_W_GCN || weight_kind == GAIB_W_MEAN ||
                         weight_kind == GAIB_W_MEAN_T || weight_kind == GAIB_W_EDGE);
  // two products whose matrices do not fit LDS together (SAGE's 100 -> 256 input layer): the neighbour product still
  // rides on the aggregation, the self term follows as an accumulating GEMM that also applies the activation
  if (shape_ok && dual && fuse_strip_rows(kpad, len_out, true) == 0 && fuse_strip_rows(kpad, len_out, false) != 0) {
    GAIB_TRY(spmm_gemm_impl(ctx, g, weight_kind, d_edge_w, len_in, d_in, d_agg, d_W, transW, nullptr, nullptr, len_out, d_out,
                            flags & ~GAIB_RELU));
    return gaib_sgemm_ex(ctx, 0, transW, g->nv, len_out, len_in, d_rows2, d_W2,
                         GAIB_ACCUMULATE | ((flags & GAIB_RELU) ? GAIB_RELU : 0), d_out);
  }
  const bool fusable = shape_ok && fuse_strip_rows(kpad, len_out, dual) != 0;
  // 129..256 columns (the hidden width 256 of scripts/run-sage-products.sh): op(W) does not fit LDS next to the strips,
  // so the aggregation runs as two 128-column K-slabs through the same kernel, each with its [len_out x 128] slab of
  // op(W) in LDS: slab 0 writes y = agg[:, :128] . op(W)[:128, :], slab 1 adds agg[:, 128:] . op(W)[128:, :] (and applies
  // the activation).  The gathered bytes are the same as one 1-KB row gather per edge; colidx / weights are streamed
  // twice and y is read-modify-written once -- against a separate pass over agg and a 2 x 2.45 M x 256 x 256 GEMM.
  // A second product (SAGE's self term) follows as an accumulating GEMM.
  const bool kslab = ctx->spmm_fuse != 0 && !dense && len_in > 128 && len_in <= 256 && len_in % 2 == 0 && (al & 7) == 0 &&
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
  GAIB_TRY(spmm_setup(ctx, g, weight_kind, d_edge_w, len_in, d_in, d_agg, 0, 1, &a, &wmode));
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

extern "C" int gaib_spmm_mh(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w,
                            int heads, int len, const float* d_in, float* d_out, int flags) {
  GAIB_CHECK(heads >= 1 && len % heads == 0, "gaib_spmm_mh: heads (%d) must divide len (%d)", heads, len);
  GAIB_CHECK(heads == 1 || weight_kind == GAIB_W_EDGE || weight_kind == GAIB_W_EDGE_T,
             "gaib_spmm_mh: per-head weights need GAIB_W_EDGE or GAIB_W_EDGE_T");
  return spmm_impl(ctx, g, weight_kind, d_edge_w, len, d_in, d_out, flags, heads);
}
