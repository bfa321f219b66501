// common.h -- internal definitions shared by the HIP translation units behind include/gaib.h.
// gfx950 (MI355X / CDNA4) only: wave64, 256 CUs in 8 XCDs, 160 KB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <string.h>
#include <vector>
#include "gaib.h"

#define GAIB_WAVE 64

void gaib_set_error(const char* fmt, ...);

#define GAIB_HIP(call)                                                                    \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) {                                                               \
      gaib_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, hipGetErrorString(e_)); \
      return GAIB_ERR_HIP;                                                                \
    }                                                                                     \
  } while (0)

#define GAIB_CHECK(cond, ...)       \
  do {                              \
    if (!(cond)) {                  \
      gaib_set_error(__VA_ARGS__);  \
      return GAIB_ERR_INVALID;      \
    }                               \
  } while (0)

#define GAIB_TRY(call)         \
  do {                         \
    int rc_ = (call);          \
    if (rc_ != GAIB_OK) return rc_; \
  } while (0)

#define GAIB_LAUNCH_CHECK() GAIB_HIP(hipGetLastError())

// Calls that wait for the stream or (re)allocate cannot be recorded into a HIP graph: inside a capture they are an
// error that says so, instead of a hipErrorStreamCaptureUnsupported from somewhere below.
#define GAIB_NOT_WHILE_CAPTURING(ctx, what)                                                                        \
  GAIB_CHECK(!(ctx)->capturing, "%s inside gaib_capture_begin/end: it waits for the stream or allocates -- run the " \
                                "sequence once before capturing it (buffers and lazily built tables then exist)", what)


#define GAIB_FLAT_RING_DEFAULT 1  // spmm_flat_ring's default (see spmm_kernels.h RING)

struct gaib_ctx {
  int device;
  hipStream_t stream;
  int num_cus;
  // growable scratch (split-K partials, per-vertex scores, ...); never shrinks
  void* ws;
  size_t ws_bytes;
  // padded copy of an aggregation's input table (spmm.hip: rows re-strided to whole 64-B pieces); never shrinks
  void* pad;
  size_t pad_bytes;
  // side stream (gaib_stream_fork/join) with its own scratch
  hipStream_t main_stream, side_stream;
  hipEvent_t ev_fork, ev_join;
  void* ws_side;
  size_t ws_side_bytes;
  int forked;
  // tuning knobs
  int spmm_heavy_threshold;  // rows with more edges go to the workgroup-per-row kernel
  int spmm_variant;          // 0 = auto, see spmm.hip
  int spmm_xcd_swizzle;      // row blocks -> XCDs: 2 (default) = chunks of 256 rows round robin, 1 = one contiguous range per XCD, 0 = none
  int spmm_chunked;          // dense graphs: aggregation by ordered 64-edge chunks + per-row reduction: -1 auto, 0 never, 1 always
  int spmm_pad;              // 1 = re-stride odd-width input tables where that saves >10 % of the gathered lines, 0 = never
  int spmm_fuse;             // 1 = gaib_spmm_gemm may fuse the dense product into the aggregation
  int spmm_flat;             // fused kernel, edge-stream form for short rows: -1 = by average degree, 0 = never, 1 = always
  int spmm_flat_ring;        // edge-stream form: 1 (default) = software-pipelined gathers (U always in flight), 0 = batches of U
  int comm_reserve_cus;      // CUs the persistent fused kernel leaves free while a halo exchange is in flight (GAIB_OVERLAPS_TRANSFER):
                             // -1 = unset (the communicator's default applies), >= 0 = the caller's choice (option / GAIB_COMM_RESERVE_CUS),
                             // an explicit 0 included; clamped so that the fused kernel keeps at least 64 CUs
  int comm_reserve_default;  // what gaib_comm_init found right for its transport: 32 under RCCL with more than one rank (its send /
                             // recv kernels need CUs to land on: the fused kernel holds every register of the CUs it sits on until
                             // its last tile), 0 on the peer-to-peer pull (copy engines) and before any communicator exists
  int spmm_fuse_cus;         // fused kernel: persistent workgroups (= CUs it occupies); 0 = all CUs.  Fewer leave whole CUs to a kernel on another stream
  int spmm_tile_xcd;         // fused kernel's tile supply: -1 (default) = by the graph's numbering (XCD-affine chunks of 1024 tiles where the numbering has locality, else one global counter), 0 = global counter, n > 0 = XCD-affine chunks of n tiles
  int spmm_prefetch_ids;     // fused row form on a numbering with locality (affine supply): 1 = next row's column ids requested a row ahead, 0 = per row
  int spmm_unroll;           // 0 = auto, 8 = cap gathers in flight per wave at 8
  int spmm_addr_mode;        // 0 = auto (buffer loads when the table is < 4 GB), 2 = force 64-bit global
  int spmm_gather_mode;      // 0/1 default cache policy, 2 = nt gathers, 3 = nt for cold columns only
  int spmm_hot_bytes;        // L2 budget for the hot rows of gather mode 3
  int sgemm_variant;         // 0 = auto
  int gat_fast;              // reserved
  int gat_chunk_colsum;      // GAT backward column sums by ordered chunks: -1 = dense graphs, 0 never, 1 always
  int gat_chunk_sort;        // 1 = SDDMM edge chunks ordered by column block (set before the graph's first SDDMM)
  int gat_row_waves;         // rows (= waves) per workgroup in the GAT row-owner kernels: 1, 2 or 4
  int gat_fused_bwd;         // the one-pass edge side of GAT backward: -1 = dense graphs (aggregation's rule), 0 never, 1 whenever the shape fits
  int gat_fused_fwd;         // the one-sweep forward (scores + online softmax + aggregation): -1 = dense graphs, 0 never, 1 whenever the shape fits
  int gat_fused_unroll;      // gathers in flight per lane and table in the fused backward sweep: 4 (default; measured 10.7 vs 10.9 ms) or 8
  int gat_bwd_pk;            // one-sweep GAT backward: 1 = the packed-math kernel over the element-interleaved table (round 6), 0 (default) = the round-5 kernel
  int gat_interleave;        // one-sweep GAT backward: 1 = gather from ONE interleaved [h | grad | records] row per vertex (built per call), 0 = three tables
  int gat_chunk_xcd;         // one-sweep GAT kernels: 1 = every XCD walks a contiguous eighth of the column-block-ordered chunk list, 0 = round robin
  int graph_rev_search;      // 1 = reverse-edge permutation by per-edge binary search (the reference's way) instead of the sort
  hipStream_t owned_stream;  // gaib_ctx_own_stream: a stream the context created (destroyed with it), else NULL
  int capturing;             // 1 between gaib_capture_begin and gaib_capture_end: calls are recorded into a HIP graph, nothing runs
  // recorded sequences (gaib_exec) freeze the ws / pad pointers of their capture in their kernel nodes: while any of
  // them is alive a workspace that has to grow is RETIRED (kept allocated) instead of freed, and released with the
  // last exec or the context
  int live_execs;
  std::vector<void*> retired;
  // in-stream kernel timing (gaib_prof_*)
  int prof_on;
  // bytes / flops: the ALGORITHMIC work of the launch as SURVEY.md 8(d) prices it (0: not stated by the site)
  // tag: the shape of the launch where the price depends on it -- the row width of a gather kernel ("47": a row that is no
  // whole number of 128-B lines costs more than its bytes), M x N x K of a dense product -- the table lists launches per
  // "key@tag"; empty: one line per key
  struct ProfRec { const char* key; hipEvent_t a, b; double bytes, flops; char tag[28]; };
  std::vector<ProfRec> prof;
};

// CUs the fused aggregation leaves to the transport while an exchange is in flight: the caller's choice where there is one
// (explicit 0 included), else the default of the communicator that was created on this context
static inline int gaib_comm_reserve(const gaib_ctx* ctx) {
  const int r = ctx->comm_reserve_cus >= 0 ? ctx->comm_reserve_cus : ctx->comm_reserve_default;
  const int cap = ctx->num_cus > 64 ? ctx->num_cus - 64 : 0;
  return r < 0 ? 0 : (r > cap ? cap : r);
}

// RAII: event pair around a kernel launch when profiling is on
// the roofs an epoch record prices a launch against (MI355X_MICROARCH.md: HBM3E 8.0 TB/s; fp32 MFMA 157.3 TFLOP/s dense)
constexpr double GAIB_HBM_PEAK_BPS = 8.0e12;
constexpr double GAIB_MFMA_F32_PEAK_FLOPS = 157.3e12;
// algorithmic bytes of one aggregation launch over `edges` edges into `rows` rows of `cols` columns --
// every gathered row counted (4 cols), the column id (4) and the per-edge weight where there is one, `stored` [rows x cols]
// matrices written (or read back), the row pointers (8 B here: rowptr is int64)
static inline double gaib_alg_spmm_bytes(double edges, double rows, double cols, double w_bytes_per_edge, double stored) {
  return edges * (4.0 * cols + 4.0 + w_bytes_per_edge) + stored * rows * 4.0 * cols + (rows + 1.0) * 8.0;
}

struct ProfScope {
  gaib_ctx* c;
  size_t idx;
  ProfScope(gaib_ctx* ctx, const char* key, double bytes = 0.0, double flops = 0.0, int cols = 0, const char* tag = nullptr)
      : c(ctx), idx((size_t)-1) {
    if (!c->prof_on || c->capturing) return;
    gaib_ctx::ProfRec r;
    r.key = key;
    r.bytes = bytes;
    r.flops = flops;
    r.tag[0] = 0;
    if (tag) snprintf(r.tag, sizeof(r.tag), "%s", tag);
    else if (cols > 0) snprintf(r.tag, sizeof(r.tag), "%d", cols);
    if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess) return;
    (void)hipEventRecord(r.a, c->stream);
    c->prof.push_back(r);
    idx = c->prof.size() - 1;
  }
  ~ProfScope() {
    if (idx != (size_t)-1) (void)hipEventRecord(c->prof[idx].b, c->stream);
  }
};

int gaib_ws_reserve(gaib_ctx* ctx, size_t bytes);
int gaib_pad_reserve(gaib_ctx* ctx, size_t bytes);

struct gaib_graph {
  int device;
  int64_t nv, ne;
  int64_t nc;        // number of columns == rows of the feature table (nv unless rectangular)
  int64_t* rowptr;   // [nv+1]
  uint32_t* colidx;  // [ne]
  float* vdata;      // [nv] deg^-1/2            (compute_vertex_data)
  float* edata;      // [ne] 1/(sqrt d_i sqrt d_j) (compute_edge_data)
  float* inv_deg;    // [nv] (float)(1.0/(float)deg)            lazily
  float* col_vdata;    // [nc] set_vertex_norm only (else the column side uses vdata)
  float* col_inv_deg;  // [nc] set_vertex_norm only (else inv_deg)
  float* w_gcn;      // [ne] vdata[i]*vdata[col]                lazily
  float* w_mean_t;   // [ne] inv_deg[col]                       lazily
  uint32_t* rev;     // [ne] index of the reverse edge          lazily
  // rows with degree > heavy_thr (built for the threshold the list was made with)
  // 64-edge chunks of the edge list (edge-parallel kernels: SDDMM), built lazily on the host
  uint32_t* chunk_row;    // [n_chunks] row of each chunk
  uint32_t* chunk_ebase;  // [n_chunks] first edge of each chunk
  uint32_t* chunk_start;  // [nv + 1] number of chunks in the rows before row v (chunk k of a row, in row order)
  int64_t n_chunks;
  uint32_t* colidx_flagged;  // gather mode 3: colidx with bit 31 set on cold (low-degree) columns
  int64_t hot_threshold;     // degree threshold the flags were built with (-1 = none)
  int64_t hot_rows;          // size of the hot set the flags were built for
  uint32_t* heavy_rows;
  int64_t n_heavy;
  int64_t heavy_edges;
  int heavy_thr;
  int64_t max_degree;
  int64_t dev_bytes;
  // row classes of a vertex-range partition (gaib_graph_split_classes): this graph holds a SUBSET of a rank's rows, compact;
  // row r stands for row row_map[r] of the caller's [n_out_rows x len] matrices.  NULL: an ordinary graph.
  uint32_t* row_map;
  int64_t n_out_rows;
  int rows_unsorted;  // 1: a row's column ids are not ascending (gaib_graph_reorder keeps the edge ORDER of every row);
                      // the reverse-edge permutation needs sorted rows: gaib_graph_sort_rows first
  float near_frac;  // share of (sampled) edges whose column id lies within 32 768 of the row id: locality of the numbering; < 0 = not measured yet
};

int gaib_graph_ensure_locality(gaib_ctx* ctx, gaib_graph* g);

int gaib_graph_ensure_inv_deg(gaib_ctx* ctx, gaib_graph* g);
int gaib_graph_ensure_w_gcn(gaib_ctx* ctx, gaib_graph* g);
int gaib_graph_ensure_w_mean_t(gaib_ctx* ctx, gaib_graph* g);
int gaib_graph_ensure_rev(gaib_ctx* ctx, gaib_graph* g);
int gaib_graph_ensure_heavy(gaib_ctx* ctx, gaib_graph* g, int thr);
int gaib_graph_ensure_chunks(gaib_ctx* ctx, gaib_graph* g);
int gaib_graph_ensure_hot_flags(gaib_ctx* ctx, gaib_graph* g, int len);

static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// wave-level helpers -------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ unsigned wave_sum_u32(unsigned v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (unsigned)__shfl_xor((int)v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// ---- DPP (data-parallel primitives: a lane permutation folded into a VALU instruction, no LDS trip) ------------------
// __shfl / __shfl_xor compile to ds_bpermute_b32 -- an LDS instruction plus its wait -- also where the pattern is fixed at
// compile time.  Inside a 16-lane row the fixed patterns are free: quad_perm, row_half_mirror, row_mirror for sums over
// aligned groups of 2 / 4 / 8 / 16 lanes, row_newbcast for "lane n of my row".
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// sum over the aligned group of LH consecutive lanes (LH = 1, 2, 4, 8, 16) this lane belongs to; every lane gets it
template <int LH>
__device__ __forceinline__ float lanes_sum(float v) {
  static_assert(LH == 1 || LH == 2 || LH == 4 || LH == 8 || LH == 16, "aligned power-of-two groups inside a 16-lane row");
  if constexpr (LH >= 2) v += dpp_f32<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (LH >= 4) v += dpp_f32<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (LH >= 8) v += dpp_f32<0x141>(v);   // row_half_mirror: the other quad of the half row
  if constexpr (LH >= 16) v += dpp_f32<0x140>(v);  // row_mirror: the other half row
  return v;
}
// lane n (0..15, a compile-time constant after unrolling) of this lane's 16-lane row
__device__ __forceinline__ int row_lane(int v, int n) {
#define GAIB_RB(N) case N: return __builtin_amdgcn_update_dpp(0, v, 0x150 + N, 0xf, 0xf, false);
  switch (n & 15) {
    GAIB_RB(0) GAIB_RB(1) GAIB_RB(2) GAIB_RB(3) GAIB_RB(4) GAIB_RB(5) GAIB_RB(6) GAIB_RB(7)
    GAIB_RB(8) GAIB_RB(9) GAIB_RB(10) GAIB_RB(11) GAIB_RB(12) GAIB_RB(13) GAIB_RB(14) default: GAIB_RB(15)
  }
#undef GAIB_RB
}
__device__ __forceinline__ float dot4_fma(const float (&a)[4], const float (&b)[4]) {
  return __builtin_fmaf(a[3], b[3], __builtin_fmaf(a[2], b[2], __builtin_fmaf(a[1], b[1], a[0] * b[0])));
}
__device__ __forceinline__ float readlane_f(float v, int l) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
