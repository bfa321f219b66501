// spmm_kernels.h -- the aggregation kernels and their launchers, shared by spmm.hip (whole graphs) and spmm_part.hip (the
// row classes of a vertex-range partition).  Every kernel takes a compile-time PART flag:
//   PART = false  the graph's row r is row r of the caller's matrices, one feature table (spmm.hip)
//   PART = true   the graph is a COMPACT subset of a rank's rows (gaib_graph_split_classes): row r stands for row
//                 row_map[r] of out / agg / rows2 / y, and column ids >= n_first index a second table (the halo table
//                 behind the rank's own rows) -- see SpmmArgs.  Same sums, same order.
#pragma once
#include <algorithm>
#include "spmm_core.h"

namespace {

// ---- light rows, one wave per row ------------------------------------------------------
template <int VEC, int CT, int WMODE, int U, int BUF, bool PART = false>
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
  float* o = a.out + out_row<PART>(a, row) * a.ldo + lane * VEC;
  if (a.accumulate) {
#pragma unroll
    for (int ct = 0; ct < CT; ++ct)
      if (colok[ct]) acc[ct] = *reinterpret_cast<const vec_t*>(o + ct * 64 * VEC);
  }
  wave_accumulate<VEC, CT, WMODE, U, BUF, PART>(a, lane, e0, e1, 64, roww, voff, acc);
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
    if (colok[ct]) *reinterpret_cast<vec_t*>(o + ct * 64 * VEC) = a.relu ? vrelu<VEC>(acc[ct]) : acc[ct];
}

// ---- heavy rows, one 1024-thread workgroup per row ------------------------------------
constexpr int HEAVY_WAVES = 16;
template <int VEC, int CT, int WMODE, int U, int BUF, bool PART = false>
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
  wave_accumulate<VEC, CT, WMODE, U, BUF, PART>(a, lane, e0 + (int64_t)wave * 64, e1,
                                           (int64_t)HEAVY_WAVES * 64, roww, voff, acc);
  constexpr int W = CT * 64 * VEC;
#pragma unroll
  for (int ct = 0; ct < CT; ++ct)
    *reinterpret_cast<vec_t*>(&red[wave * W + (ct * 64 + lane) * VEC]) = acc[ct];
  __syncthreads();
  float* orow = a.out + (a.compact ? (int64_t)slot : out_row<PART>(a, row)) * a.ldo;
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
template <int VEC, int CT, int WMODE, int U, int BUF, bool PART = false>
int launch_w64_u(gaib_ctx* ctx, const gaib_graph* g, SpmmArgs a) {
  // heavy rows first (few, long): their tail hides under the light kernel's start
  if (g->n_heavy > 0) {
    SpmmArgs h = a;
    h.row_list = g->heavy_rows;
    h.row_order = g->heavy_rows + g->n_heavy;
    size_t lds = sizeof(float) * HEAVY_WAVES * CT * 64 * VEC;
    ProfScope ps(ctx, "spmm_heavy", gaib_alg_spmm_bytes((double)g->heavy_edges, (double)g->n_heavy, a.ncols, WMODE == 0 ? 0 : 4,
                                                         a.accumulate ? 2 : 1), 2.0 * g->heavy_edges * a.ncols, a.ncols);
    // (a 1024-thread workgroup leaves a wave 128 VGPRs: the multi-head edge-weight modes -- a weight per head and edge next to
    // the gathers -- keep half as many rows in flight there, or they spill 52-61 registers; 16 waves per row hide the rest)
    constexpr int UH = (WMODE >= 3 && U * CT * VEC > 16) ? (U / 2 > 4 ? U / 2 : 4) : U;
    spmm_heavy_kernel<VEC, CT, WMODE, UH, BUF, PART><<<dim3((unsigned)g->n_heavy), HEAVY_WAVES * 64, lds,
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
    // (row classes of a partition are timed under keys of their own: interior / owned-column pass, halo-column pass, one pass)
    const double e_l = (double)g->ne - (g->n_heavy > 0 ? (double)g->heavy_edges : 0.0), r_l = (double)a.n_rows - (double)g->n_heavy;
    ProfScope ps(ctx, !PART ? "spmm_light" : (a.in2 ? "part_light_2t" : (a.accumulate ? "part_light_acc" : "part_light")),
                 gaib_alg_spmm_bytes(e_l, r_l, a.ncols, WMODE == 0 ? 0 : 4, a.accumulate ? 2 : 1), 2.0 * e_l * a.ncols, a.ncols);
    spmm_w64_kernel<VEC, CT, WMODE, U, BUF, PART><<<dim3(grid), 256, 0, ctx->stream>>>(a);
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
  // option spmm_unroll = 8: eight gathers in flight where the default is sixteen.  (Only THERE is the 8-deep form instantiated:
  // as a dead branch next to U = 4 it gave the wide-lane kernels eight 8-register gathers they never run -- and spilled
  // 82-236 registers doing it, round 6's spill check)
  if constexpr (U > 8) {
    if (ctx->spmm_unroll == 8) {
      switch (gm) {
        case 0: return launch_w64_u<VEC, CT, WMODE, 8, 0>(ctx, g, a);
        case 2: return launch_w64_u<VEC, CT, WMODE, 8, 2>(ctx, g, a);
        case 3: return launch_w64_u<VEC, CT, WMODE, 8, 3>(ctx, g, a);
        default: return launch_w64_u<VEC, CT, WMODE, 8, 1>(ctx, g, a);
      }
    }
  }
  switch (gm) {
    case 0: return launch_w64_u<VEC, CT, WMODE, U, 0>(ctx, g, a);
    case 2: return launch_w64_u<VEC, CT, WMODE, U, 2>(ctx, g, a);
    case 3: return launch_w64_u<VEC, CT, WMODE, U, 3>(ctx, g, a);
    default: return launch_w64_u<VEC, CT, WMODE, U, 1>(ctx, g, a);
  }
}

template <int VEC, int G, int WMODE>
int launch_sub(gaib_ctx* ctx, const gaib_graph* g, SpmmArgs a) {
  if (g->n_heavy > 0) {
    SpmmArgs h = a;
    h.row_list = g->heavy_rows;
    h.row_order = g->heavy_rows + g->n_heavy;
    size_t lds = sizeof(float) * HEAVY_WAVES * 64 * VEC;
    ProfScope ps(ctx, "spmm_heavy", gaib_alg_spmm_bytes((double)g->heavy_edges, (double)g->n_heavy, a.ncols, WMODE == 0 ? 0 : 4,
                                                         a.accumulate ? 2 : 1), 2.0 * g->heavy_edges * a.ncols, a.ncols);
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
    const double e_l = (double)g->ne - (g->n_heavy > 0 ? (double)g->heavy_edges : 0.0), r_l = (double)a.n_rows - (double)g->n_heavy;
    ProfScope ps(ctx, "spmm_sub", gaib_alg_spmm_bytes(e_l, r_l, a.ncols, WMODE == 0 ? 0 : 4, a.accumulate ? 2 : 1), 2.0 * e_l * a.ncols, a.ncols);
    spmm_sub_kernel<VEC, G, WMODE><<<dim3(grid), 256, 0, ctx->stream>>>(a);
    GAIB_LAUNCH_CHECK();
  }
  return GAIB_OK;
}

template <int VEC, int WMODE>
int dispatch_ct(gaib_ctx* ctx, const gaib_graph* g, const SpmmArgs& a, int lanes) {
  if (lanes <= 64) return launch_w64<VEC, 1, WMODE>(ctx, g, a);
  if (lanes <= 128) return launch_w64<VEC, 2, WMODE>(ctx, g, a);
  // 16-byte lanes never take four column tiles: 16 accumulator floats per lane next to 4 gathers of 16 floats each spilled
  // 205-354 VGPRs in every such kernel (round 5's code-object notes); rows of 513 .. 1024 columns run as 512-column slabs
  // of two tiles instead (dispatch_vec), each slab walking the row's edge list once more -- 8 B per edge next to 2 KB of row
  if constexpr (VEC == 4) {
    gaib_set_error("spmm: %d 16-byte lanes in one launch (internal: the slab width is 128 lanes at VEC = 4)", lanes);
    return GAIB_ERR_INVALID;
  } else {
    return launch_w64<VEC, 4, WMODE>(ctx, g, a);
  }
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
  // one launch covers up to 256 lanes' worth of columns (128 lanes of 16 B: see dispatch_ct); wider rows are done in column slabs
  const int slab = (vec == 4 ? 128 : 256) * vec;
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
  int overlaps_transfer;      // GAIB_OVERLAPS_TRANSFER: leave ctx->comm_reserve_cus CUs to the transport's kernels
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
// RING (with FLAT): the strip's edge stream as a software pipeline instead of batches.  Batch by batch a wave holds U
// gathers in flight, waits for all of them, consumes them, issues the next U: the bytes in flight follow a sawtooth
// between U and 0 rows, and at 16 waves per CU that average is what bounds the halo-column half of a partitioned
// aggregation (3-5 edges per row: 5.2 TB/s of the bytes it moves where the row kernels reach 7.5).  Here edge e's gather
// is re-issued into the register edge e - U has just been consumed from: U rows are in flight from the strip's first edge to
// its last.  The column ids of the strip's first 64 edges are requested BEFORE the partial sums the strip continues (the
// gathers hang on the former only), those of the next 64 while the current ones are consumed.  Same edge order, same sums.
// AFFINE: the XCD-affine tile supply is compiled in.  A template flag, not a run-time test: with both supplies in one loop the
// global-counter form -- the headline's, a random vertex order -- ran 0.5 % slower than before the affine supply existed
// (7.597 -> 7.635 ms per launch, bisected to that change on one box: scripts/drift_ab.sh, profiles/r04/drift_*.jsonl).
template <int VEC, int WMODE, int U, int GM, int STRIP, bool DUAL, bool FLAT = false, bool YACC = false, bool PART = false,
          bool RING = false, bool AFFINE = false, bool PREF = false>
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
  // aggregate rows leave the kernel non-temporally: nothing here reads them again, and as plain stores they sit in the L2
  // between the gathered rows (measured: the halo-column half 3.79 -> 3.60 ms per step, the headline kernel unchanged at
  // 7.54-7.55 ms; product rows stored the same way change nothing either way)
  auto store_row = [&](vec_t* p, const vec_t& v) { __builtin_nontemporal_store(v, p); };
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
  const int tsh = AFFINE ? f.tile_xcd - 1 : 0;  // log2 of the chunk length in tiles
  int own = AFFINE ? (int)(blockIdx.x & 7) : 0;  // the XCD whose counter this wave is drawing from
  int k_next = 0, k_left = 0;
  const int nwaves4 = ((int)gridDim.x * FUSE_WAVES * 4) / 8 > 0 ? ((int)gridDim.x * FUSE_WAVES * 4) / 8 : 1;
  const int per_xcd_tiles = (ntiles + 7) / 8;  // about what one XCD's chunks hold
  const int n_chunks = AFFINE ? (ntiles + (1 << tsh) - 1) >> tsh : 0;
  for (;;) {
    if (k_left == 0) {
      int want = 1;
      if constexpr (FLAT) {  // guided: (tiles this XCD has left) / (4 x its waves), at most 8, single tiles at the end
        const int left = AFFINE ? per_xcd_tiles - k_next : (ntiles - k_next) / 8;
        want = (left > 0 ? left : 0) / nwaves4;
        want = want < 1 ? 1 : (want > 8 ? 8 : want);
      }
      int k0 = 0;
      if (lane == 0) k0 = atomicAdd(f.tile_counter + own, want);
      k0 = __builtin_amdgcn_readfirstlane(k0);
      if constexpr (!AFFINE) {  // one global counter: tiles in order
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
    const int t = AFFINE ? ((((kk >> tsh) * 8 + own) << tsh) + (kk & ((1 << tsh) - 1))) : kk;
    if (t >= ntiles) continue;  // the ragged end of the last chunk
    const int row0 = t * FUSE_ROWS;
    // PART: lane r holds the row of the caller's matrices that tile row r stands for (rows past the end: the last one's)
    int rid = 0;
    if constexpr (PART) {
      int rc = row0 + (lane < FUSE_ROWS ? lane : 0);
      if (rc >= a.n_rows) rc = a.n_rows - 1;
      rid = a.row_map ? (int)a.row_map[rc] : rc;
    }
    auto orow = [&](int rr) -> int64_t {  // rr wave-uniform
      if constexpr (PART) return (int64_t)__builtin_amdgcn_readlane(rid, rr);
      else return (int64_t)(row0 + rr);
    };
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
    // PRE (the instantiations for numberings with locality, AFFINE): the column ids and weights of the NEXT row's first 64
    // edges are requested while this row's gathers are in flight.  Row by row a wave waits one trip to the id stream (HBM:
    // the ids are read once) and then one per batch of gathers; where the gathered rows come from the L2 the first is a
    // third of a row's time (communities of 2 048 rows, 1 MB, run at the same 47 ps per edge as communities of 16 384:
    // scripts/locality_ceiling.py).  On a random order the gathers themselves fill the fabric: not compiled in there.
    constexpr bool PRE = PREF && !FLAT && (WMODE == 0 || WMODE == 1);
    uint32_t c_pre = 0;
    float w_pre = 0.f;
    auto prefetch_ids = [&](int rr) {  // rr wave-uniform, < FUSE_ROWS
      const int64_t e0 = ((int64_t)__builtin_amdgcn_readlane(rp_hi, rr) << 32) | (uint32_t)__builtin_amdgcn_readlane(rp_lo, rr);
      const int64_t e1 = ((int64_t)__builtin_amdgcn_readlane(rp_hi, rr + 1) << 32) | (uint32_t)__builtin_amdgcn_readlane(rp_lo, rr + 1);
      c_pre = 0;
      w_pre = 0.f;
      if (e0 + lane < e1) {
        c_pre = a.col[e0 + lane];
        if constexpr (WMODE == 1) w_pre = load_edge_w<WMODE>(a, e0 + lane);
      }
    };
    if constexpr (PRE) prefetch_ids(0);
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
          const RowGather<VEC, GM, PART> gather(a);
          float* trow_w = tile + lane * VEC;  // this lane's columns of strip row 0
          // RING: ids and weights of the strip's first (and second) 64 edges, lane l holds edge 64 q + l
          const int total = (int)(e_hi - e_lo);  // (a strip without heavy rows: at most HALF x heavy_thr edges)
          uint32_t c_cur = 0, c_nxt = 0;
          float w_cur = 0.f, w_nxt = 0.f;
          auto load_ids = [&](int q, uint32_t& c, float& w) {
            const int64_t e = e_lo + 64 * (int64_t)q + lane;
            c = 0;
            w = 0.f;
            if (e < e_hi) {
              c = a.col[e];
              if constexpr (WMODE == 1 || WMODE == 2) w = load_edge_w<WMODE>(a, e);
            }
          };
          if constexpr (RING) {
            load_ids(0, c_cur, w_cur);
            if (total > 64) load_ids(1, c_nxt, w_nxt);
          }
          if (f.agg_in) {
            // accumulate mode: the strip starts out as the partial sums of its rows (all requests first)
            vec_t t[HALF];
#pragma unroll
            for (int r2 = 0; r2 < HALF; ++r2) {
              int64_t row = row0 + rbase + r2;
              if (row >= a.n_rows) row = a.n_rows - 1;
              if constexpr (PART) row = orow(rbase + r2);
              t[r2] = *reinterpret_cast<const vec_t*>(f.agg_in + row * a.ldo + (colok ? lane * VEC : 0));
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
              store_row(reinterpret_cast<vec_t*>(a.out + orow(rbase + r) * a.ldo + lane * VEC), acc);
            *reinterpret_cast<vec_t*>(trow_w + r * LDT) = colok ? acc : vzero<VEC>();
            ++r;
            if (r < HALF) {
              row_end = rp_at(rbase + r + 1);
              acc = f.agg_in ? *reinterpret_cast<const vec_t*>(trow_w + r * LDT) : vzero<VEC>();
              if constexpr (WMODE == 0) roww = readlane_f(rwv, rbase + r);
            }
          };
          if constexpr (RING) {
            vec_t x[U];
            if (total > 0) {
#pragma unroll
              for (int u = 0; u < U; ++u)  // (edges past the strip's end read row 0 of the table: never consumed)
                x[u] = gather.load((uint32_t)__builtin_amdgcn_readlane((int)c_cur, u), voff[0]);
            }
            for (int k = 0; k < total; k += U) {
              const int kc = k & 63;
              if (kc == 0 && k > 0) {  // entering the next 64 edges: their ids are here, request the ones after them
                c_cur = c_nxt;
                w_cur = w_nxt;
                if (k + 64 < total) load_ids((k >> 6) + 1, c_nxt, w_nxt);
              }
              const bool wrap = kc == 64 - U;  // the refills of this batch belong to the next 64 edges
              const uint32_t c_src = wrap ? c_nxt : c_cur;
              if (k + 2 * U <= total) {  // a full batch with a full batch behind it: straight-line
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  while (e_lo + k + u == row_end) flush();
                  vacc<VEC>(acc, (WMODE == 0) ? roww : readlane_f(w_cur, kc + u), x[u]);
                  x[u] = gather.load((uint32_t)__builtin_amdgcn_readlane((int)c_src, (kc + u + U) & 63), voff[0]);
                }
              } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                  if (k + u < total) {
                    while (e_lo + k + u == row_end) flush();
                    vacc<VEC>(acc, (WMODE == 0) ? roww : readlane_f(w_cur, kc + u), x[u]);
                  }
                  if (k + u + U < total)
                    x[u] = gather.load((uint32_t)__builtin_amdgcn_readlane((int)c_src, (kc + u + U) & 63), voff[0]);
                }
              }
            }
          }
          for (int64_t base = e_lo; base < e_hi && !RING; base += 64) {
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
        const uint32_t c_now = c_pre;
        const float w_now = w_pre;
        if constexpr (PRE) {
          if (rr + 1 < FUSE_ROWS) prefetch_ids(rr + 1);
        }
        if (row < a.n_rows) {
          if (f.agg_in && colok)
            acc[0] = *reinterpret_cast<const vec_t*>(f.agg_in + orow(rr) * a.ldo + lane * VEC);
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
            wave_accumulate<VEC, 1, WMODE, U, GM, PART, PRE>(a, lane, e0, e1, 64, roww, voff, acc, c_now, w_now);
          }
          if (a.out && colok) store_row(reinterpret_cast<vec_t*>(a.out + orow(rr) * a.ldo + lane * VEC), acc[0]);
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
    // the four rows of y this lane stores (C/D layout: row = 4*(lane>>4) + reg); PART: through the row map
    int64_t yrow[4];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      if constexpr (PART) yrow[reg] = (int64_t)__shfl(rid, 4 * kq + reg, 64);
      else yrow[reg] = (int64_t)(row0 + 4 * kq + reg);
    }
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
            if constexpr (YACC) v += f.y[yrow[reg] * f.ldy + n0 + i];  // the second K-slab of a wide aggregation
            if (f.relu) v = v > 0.f ? v : 0.f;
            f.y[yrow[reg] * f.ldy + n0 + i] = v;
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
              yo[q][reg] = f.y[(row < a.n_rows ? yrow[reg] : (int64_t)0) * f.ldy + nt * 16 + i];
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
                  f.y[yrow[reg] * f.ldy + n0 + i] = v;
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
        int64_t rs = row < a.n_rows ? row : 0;
        if constexpr (PART) rs = orow(r);
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

template <int VEC, int WMODE, bool PART = false>
int launch_fused(gaib_ctx* ctx, const gaib_graph* g, SpmmArgs a, FuseArgs f, float* heavy_scratch) {
  constexpr int U = 16;
  constexpr int K = 64 * VEC;
  // buffer loads need every table below 4 GB
  const bool buf = a.in_bytes != 0 && ctx->spmm_addr_mode != 2 && (!PART || !a.in2 || a.in2_bytes != 0);
  if (g->n_heavy > 0) {
    SpmmArgs h = a;
    h.row_list = g->heavy_rows;
    h.row_order = g->heavy_rows + g->n_heavy;
    h.out = heavy_scratch;
    h.compact = 1;  // (row k of the scratch has the stride of the output rows, h.ldo)
    h.relu = 0;
    h.accumulate = 0;
    size_t lds = sizeof(float) * HEAVY_WAVES * 64 * VEC;
    ProfScope ps(ctx, "spmm_heavy", gaib_alg_spmm_bytes((double)g->heavy_edges, (double)g->n_heavy, a.ncols, WMODE == 0 ? 0 : 4, 1),
                 2.0 * g->heavy_edges * a.ncols, a.ncols);
    if (buf) spmm_heavy_kernel<VEC, 1, WMODE, U, 1, PART><<<dim3((unsigned)g->n_heavy), HEAVY_WAVES * 64, lds, ctx->stream>>>(h);
    else spmm_heavy_kernel<VEC, 1, WMODE, U, 0, PART><<<dim3((unsigned)g->n_heavy), HEAVY_WAVES * 64, lds, ctx->stream>>>(h);
    GAIB_LAUNCH_CHECK();
  }
  if constexpr (PART) f.tile_xcd = 0;
  const bool dual = f.wt2 != nullptr;
  const int strip = fuse_strip_rows(K, f.n_out, dual);  // 8, 2 or 0 (does not fit: the caller checked)
  const size_t lds = fuse_lds_bytes(K, f.n_out, dual, strip);
  const int64_t ntiles = cdiv64(a.n_rows, FUSE_ROWS);
  int cus = ctx->spmm_fuse_cus > 0 ? ctx->spmm_fuse_cus : ctx->num_cus;
  // a halo exchange is in flight on the communication stream (the caller said so): its kernels need CUs to land on, and a
  // persistent workgroup of this kernel owns its CU's registers until the last tile (measured: 224 of 256 CUs cost 1.5 %)
  if (f.overlaps_transfer && gaib_comm_reserve(ctx) > 0) cus = std::max(cus - gaib_comm_reserve(ctx), std::min(cus, 64));
  const unsigned grid = (unsigned)std::min<int64_t>(cus, cdiv64(ntiles, FUSE_WAVES));
  GAIB_HIP(hipMemsetAsync(f.tile_counter, 0, 8 * sizeof(int), ctx->stream));  // one counter per XCD
  // SURVEY 8(d) for the aggregation part (the heavy rows' edges are the heavy kernel's) + what the riding product moves: the
  // aggregate rows stored (unless scratch) or continued (agg_in), y written (read too on a later K-slab), the second row operand
  const double e_l = (double)g->ne - (g->n_heavy > 0 ? (double)g->heavy_edges : 0.0), r_all = (double)a.n_rows;
  const double fused_bytes = gaib_alg_spmm_bytes(e_l, r_all, a.ncols, WMODE == 0 ? 0 : 4, (a.out ? 1 : 0) + (f.agg_in ? 1 : 0) + (dual ? 1 : 0)) +
                             r_all * 4.0 * f.n_out * (f.y_accum ? 2 : 1);
  const double fused_flops = 2.0 * e_l * a.ncols + 2.0 * r_all * a.ncols * f.n_out * (dual ? 2 : 1);
  ProfScope ps(ctx, !PART ? "spmm_gemm_fused" : (a.in2 ? "part_fused_2t" : (f.agg_in ? "part_fused_acc" : "part_fused")), fused_bytes,
               fused_flops, a.ncols);
  // more than 64 KB of dynamic LDS has to be asked for
  // (the edge-stream form keeps 8 gathers in flight, not 16: with 16 the operand fragments of the dense product
  // spill and are reloaded inside the MFMA loop)
#define GAIB_FUSED_LAUNCH_P(GM, STRIP, DUAL, FLAT, YACC, RING, AFF, PRE)                                              \
  do {                                                                                                                \
    constexpr int UU = FLAT ? 8 : U;                                                                                  \
    GAIB_HIP(hipFuncSetAttribute(                                                                                     \
        (const void*)spmm_gemm_kernel<VEC, WMODE, UU, GM, STRIP, DUAL, FLAT, YACC, PART, RING, AFF, PRE>,             \
        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                                                     \
    spmm_gemm_kernel<VEC, WMODE, UU, GM, STRIP, DUAL, FLAT, YACC, PART, RING, AFF, PRE>                               \
        <<<dim3(grid), FUSE_WAVES * 64, lds, ctx->stream>>>(a, f);                                                    \
  } while (0)
#define GAIB_FUSED_LAUNCH_A(GM, STRIP, DUAL, FLAT, YACC, RING, AFF)                                                   \
  do {                                                                                                                \
    if constexpr (AFF && !FLAT) {                                                                                     \
      if (ctx->spmm_prefetch_ids) GAIB_FUSED_LAUNCH_P(GM, STRIP, DUAL, FLAT, YACC, RING, AFF, true);                  \
      else GAIB_FUSED_LAUNCH_P(GM, STRIP, DUAL, FLAT, YACC, RING, AFF, false);                                        \
    } else {                                                                                                          \
      GAIB_FUSED_LAUNCH_P(GM, STRIP, DUAL, FLAT, YACC, RING, AFF, false);                                             \
    }                                                                                                                 \
  } while (0)
  // (row classes of a partition are rectangular graphs: their numbering is never measured as local, no affine variants)
#define GAIB_FUSED_LAUNCH_R(GM, STRIP, DUAL, FLAT, YACC, RING)                                                        \
  do {                                                                                                                \
    if constexpr (!PART) {                                                                                            \
      if (f.tile_xcd) GAIB_FUSED_LAUNCH_A(GM, STRIP, DUAL, FLAT, YACC, RING, true);                                   \
      else GAIB_FUSED_LAUNCH_A(GM, STRIP, DUAL, FLAT, YACC, RING, false);                                             \
    } else {                                                                                                          \
      GAIB_FUSED_LAUNCH_A(GM, STRIP, DUAL, FLAT, YACC, RING, false);                                                  \
    }                                                                                                                 \
  } while (0)
#define GAIB_FUSED_LAUNCH_Y(GM, STRIP, DUAL, FLAT, YACC) GAIB_FUSED_LAUNCH_R(GM, STRIP, DUAL, FLAT, YACC, false)
#define GAIB_FUSED_LAUNCH(GM, STRIP, DUAL, FLAT) GAIB_FUSED_LAUNCH_Y(GM, STRIP, DUAL, FLAT, false)
  // short rows (fewer than 12 edges per row on average: halo-column halves, citation graphs): the edge-stream form
  // (scripts/ab_flat.py: -34 % at 3 edges per row, -20 % at 5, even at 12, +2 % at 30)
  const bool flat = !dual && strip == 8 && !f.y_accum &&
                    (ctx->spmm_flat >= 1 || (ctx->spmm_flat < 0 && g->ne < 12 * (int64_t)a.n_rows));
  const bool ring = flat && ctx->spmm_flat_ring != 0;  // the edge stream as a software pipeline (see RING)
  if (f.y_accum) {  // the second K-slab of a 129..256-column aggregation (VEC == 2 only; never dual or flat; not on row classes)
    if constexpr (VEC == 2 && !PART) {
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
    else if (flat && ring) GAIB_FUSED_LAUNCH_R(1, 8, false, true, false, true);
    else if (flat) GAIB_FUSED_LAUNCH(1, 8, false, true);
    else GAIB_FUSED_LAUNCH(1, 8, false, false);
  } else {
    if (dual) GAIB_FUSED_LAUNCH(0, 2, true, false);
    else if (strip == 2) GAIB_FUSED_LAUNCH(0, 2, false, false);
    else if (flat && ring) GAIB_FUSED_LAUNCH_R(0, 8, false, true, false, true);
    else if (flat) GAIB_FUSED_LAUNCH(0, 8, false, true);
    else GAIB_FUSED_LAUNCH(0, 8, false, false);
  }
#undef GAIB_FUSED_LAUNCH_A
#undef GAIB_FUSED_LAUNCH_P
#undef GAIB_FUSED_LAUNCH_R
#undef GAIB_FUSED_LAUNCH
#undef GAIB_FUSED_LAUNCH_Y
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

}  // namespace
