/*
 * gaib.h -- C ABI of the MI355X (gfx950) GNN aggregation engine.
 *
 * This is the drop-in boundary.  The reference has no FFI layer: its
 * "operator API" is a set of C++ declarations whose definitions the GPU Makefile target
 * takes from .cu files (src/gnn/Makefile:56-77).  The host-side C++ mirror of those
 * declarations lives in include/{gnn,layers,utils}/ of this repo and is implemented
 * ENTIRELY on top of the functions below; every entry point cites the reference
 * interface (file:line under /root/reference) it replaces.
 *
 * Conventions
 *   - plain C types only; all pointers named d_* are DEVICE pointers (HBM), h_* host.
 *   - every function returns GAIB_OK (0) or a negative gaib_status; gaib_last_error()
 *     gives the message.  The C++ mirror turns non-zero into the reference's
 *     print-and-exit (include/utils/cutils.h:18-28,133-174).
 *   - all work is enqueued on the context's HIP stream; nothing synchronises unless
 *     the name says so (the reference syncs after every launch: cutils.h:18-28).
 *   - feature matrices are row-major fp32 [nv x len]; row offsets are 64-bit (the
 *     reference's uint32 `dst*len`, graph_operations.h:100,105, overflows at N*D>2^32).
 */
#ifndef GAIB_H
#define GAIB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  GAIB_OK = 0,
  GAIB_ERR_INVALID = -1, /* bad argument / shape */
  GAIB_ERR_HIP = -2,     /* a HIP runtime call failed */
  GAIB_ERR_NOMEM = -3,
  GAIB_ERR_ASYMMETRIC = -4, /* edge_transpose: reverse edge missing (math_functions.cpp:70 assert) */
  GAIB_ERR_UNSUPPORTED = -5,
  GAIB_ERR_COMM = -6 /* a collective failed, a peer reported a failure or did not arrive in time */
} gaib_status;

typedef struct gaib_ctx gaib_ctx;     /* device + stream + workspace              */
typedef struct gaib_graph gaib_graph; /* CSR in HBM + normalisers + row schedules */

const char* gaib_last_error(void);
const char* gaib_version(void);

/* ---- context -------------------------------------------------------------------------
 * replaces the static cuBLAS/cuSPARSE/cuRAND handle holder `gpu_context`
 * (include/gnn/gpu_context.h:4-16, src/utilities/random.cpp:62-80).
 * `stream` is a hipStream_t (NULL = the device's null stream); it is borrowed. */
int gaib_device_count(int* h_count); /* visible devices (initialises the HIP runtime: rank processes only, never a launcher) */
int gaib_ctx_create(int device, void* stream, gaib_ctx** out);
int gaib_ctx_destroy(gaib_ctx* ctx);
int gaib_ctx_set_stream(gaib_ctx* ctx, void* stream);
/* switch the context to a (non-blocking) stream of its own, created once and destroyed with the context: for callers
 * without a stream to lend -- the null stream cannot be recorded by gaib_capture_begin */
int gaib_ctx_own_stream(gaib_ctx* ctx);
int gaib_sync(gaib_ctx* ctx); /* CudaTest()'s cudaDeviceSynchronize, cutils.h:18-28 */
/* Side stream for work that is independent of the calls that follow it (the weight-gradient GEMM
 * next to the aggregation of the input gradient: an MFMA-bound and an HBM-bound kernel overlap).
 *   gaib_side_begin: everything enqueued so far is a dependency of the side stream; calls go to the
 *                    side stream (with its own scratch) from here ...
 *   gaib_side_end:   ... to here; calls go to the main stream again, which does NOT wait.
 *   gaib_side_wait:  the main stream waits for the side section.  Buffers the side section reads or
 *                    writes must not be written on the main stream between end and wait.
 * One side section at a time (begin after begin without wait is an error). */
int gaib_side_begin(gaib_ctx* ctx);
int gaib_side_end(gaib_ctx* ctx);
int gaib_side_wait(gaib_ctx* ctx);
/* HIP graphs: record a call sequence once, replay it with one launch.  The 2-layer, 16-column models of the
 * reference's shipped datasets (cora / citeseer, inputs/) are launch bound -- an epoch is ~40 kernels of a few
 * microseconds -- where the reference's GPU build pays a cudaDeviceSynchronize per op (cutils.h:18-28).
 *   gaib_capture_begin: every call on this context is RECORDED on its stream from here (nothing runs) ...
 *   gaib_capture_end:   ... to here; *out replays the sequence.  gaib_capture_abort drops an open capture.
 * Inside a capture, calls that wait for the stream or allocate return GAIB_ERR_INVALID with a message (gaib_sync,
 * gaib_memcpy_h2d/d2h, metrics with host results, a workspace or lazily built graph table that does not exist
 * yet): run the sequence once eagerly first.  Pointers and by-value scalars are frozen in the recording; per-replay
 * state lives in device memory (gaib_adam_step_dev, gaib_masked_*_dev, gaib_memcpy_d2h_async).  The context must
 * own a real stream (not the null stream), without a side section or kernel timing switched on.
 * Scratch: a recording holds the addresses of the context's internal workspaces.  While any gaib_exec of a context
 * is alive, a workspace that has to grow (a later, larger call outside the replay) is kept allocated next to its
 * replacement and released with the last gaib_exec -- replays stay valid.  A gaib_exec is launched on the context it
 * was recorded on, and destroyed before it (gaib_ctx_destroy refuses otherwise). */
typedef struct gaib_exec gaib_exec;
int gaib_capture_begin(gaib_ctx* ctx);
int gaib_capture_end(gaib_ctx* ctx, gaib_exec** out);
int gaib_capture_abort(gaib_ctx* ctx);
int gaib_exec_launch(gaib_ctx* ctx, gaib_exec* exec);
int gaib_exec_elapsed_ms(gaib_exec* exec, float* h_ms); /* device time of the last launch; after gaib_sync */
int64_t gaib_exec_nodes(const gaib_exec* exec); /* graph nodes recorded (kernels, memsets, copies) */
int gaib_exec_destroy(gaib_exec* exec);

/* ---- memory: float/uint/uint8_malloc_device, *_free_device, copy_*_device, copy_float_host,
 * init_const_gpu  (include/utils/math_functions.hh:161-173, math_functions.cu:12-14;
 * templates malloc_device<T>/copy_async_device<T>, include/utils/cutils.h:193-202) ---- */
int gaib_malloc(gaib_ctx* ctx, size_t bytes, void** d_ptr);
int gaib_free(gaib_ctx* ctx, void* d_ptr);
int gaib_memcpy_h2d(gaib_ctx* ctx, void* d_dst, const void* h_src, size_t bytes);
int gaib_memcpy_d2h(gaib_ctx* ctx, void* h_dst, const void* d_src, size_t bytes); /* syncs */
int gaib_memcpy_d2d(gaib_ctx* ctx, void* d_dst, const void* d_src, size_t bytes);
/* pinned host memory and a stream-ordered read-back into it (valid after the next gaib_sync) */
int gaib_host_alloc(gaib_ctx* ctx, size_t bytes, void** h_ptr);
int gaib_host_free(gaib_ctx* ctx, void* h_ptr);
int gaib_memcpy_d2h_async(gaib_ctx* ctx, void* h_pinned_dst, const void* d_src, size_t bytes);
int gaib_fill_f32(gaib_ctx* ctx, int64_t n, float value, float* d_x);
/* x <- alpha * x  (scale, math_functions.cpp:336-356 / scal_gpu; the partitioned trainer rescales the loss gradient
 * from 1/(local range) to 1/(global range), softmax_loss_layer.cpp:31) */
int gaib_scale_f32(gaib_ctx* ctx, int64_t n, float alpha, float* d_x);

/* ---- graph: LearningGraph's device half (include/gnn/lgraph.h:20-277) ------------------
 * gaib_graph_create  = alloc_on_device + copy_to_gpu (src/gnn/lgraph.cu:51-92).
 *   rowptr: nv+1 entries of `rowptr_bits` (32: index_t as LearningGraph holds it; 64: as
 *   graph.vertex.bin stores it, reader.cpp:446-454); colidx: ne uint32 (graph.edge.bin).
 *   src_on_device != 0 means the two arrays are already in HBM (they are copied).
 *   Rows must be sorted by column id (lgraph.h:185 / math_functions.cpp:32-44 need it). */
int gaib_graph_create(gaib_ctx* ctx, int64_t nv, int64_t ne, const void* rowptr, int rowptr_bits,
                      const uint32_t* colidx, int src_on_device, gaib_graph** out);
/* Rectangular variant for a vertex-range partition (modelled on
 * PartitionedGraph::edgecut_induced_partition1D, src/partitioner/graph_partition.cc:128-178):
 * nv owned rows, column ids index a feature table of nc >= nv rows (owned rows first, halo
 * rows after).  add_selfloop / edge_transpose / GAT need a square graph. */
int gaib_graph_create_rect(gaib_ctx* ctx, int64_t nv, int64_t nc, int64_t ne, const void* rowptr,
                           int rowptr_bits, const uint32_t* colidx, int src_on_device,
                           gaib_graph** out);
/* Partitioned graphs: a halo column's local degree is truncated, so the normalisers come from
 * the GLOBAL graph: d_row_vdata / d_row_inv_deg [nv] (NULL = derive from this graph's rowptr; pass
 * them when the rows hold only part of their edges, i.e. the owned-/halo-column split graphs),
 * d_col_vdata [nc] (deg^-1/2), d_col_inv_deg [nc] (1/deg).  Arrays are copied. */
int gaib_graph_set_vertex_norm(gaib_ctx* ctx, gaib_graph* g, const float* d_row_vdata,
                               const float* d_row_inv_deg, const float* d_col_vdata,
                               const float* d_col_inv_deg);
int gaib_graph_destroy(gaib_graph* g);
/* LearningGraph::add_selfloop (lgraph.h:185-218) as a device-side rebuild. */
int gaib_graph_add_selfloop(gaib_ctx* ctx, const gaib_graph* g, gaib_graph** out);
/* Opt-in relabelling: the same graph under a new vertex numbering computed on the device from the graph alone
 * (GAIB_ORDER_DEGREE: hubs first; GAIB_ORDER_BFS: breadth-first levels from the highest-degree vertex; GAIB_ORDER_CM:
 * Cuthill-McKee inside those levels).  A numbering with
 * locality is worth up to 1.2-1.6x to the aggregation; the library never relabels on its own.  Every
 * row keeps the ORDER of its edges, so aggregation outputs are bit-identical once un-permuted.
 *   d_new_of_old [nv] (required), d_old_of_new [nv] (may be NULL): int64 device arrays the call fills.
 *   features in the new numbering:  gaib_gather_rows(ctx, nv, d_old_of_new, len, x_old, x_new)
 *   outputs back in the old one:     gaib_gather_rows(ctx, nv, d_new_of_old, len, y_new, y_old)
 * The reference has no counterpart (its reader keeps the file's numbering, reader.cpp:414-457). */
#define GAIB_ORDER_DEGREE 1
#define GAIB_ORDER_BFS 2
#define GAIB_ORDER_CM 3 /* the BFS levels, inside a level by the position of the first parent (Cuthill-McKee) */
int gaib_graph_reorder(gaib_ctx* ctx, gaib_graph* g, int method, gaib_graph** out, int64_t* d_new_of_old,
                       int64_t* d_old_of_new);
/* A relabelled graph's rows keep their edge order, i.e. their column ids are NOT ascending: it is an aggregation-only
 * graph -- gaib_edge_transpose / GAIB_W_EDGE_T / the GAT backward entry points, which derive the reverse-edge permutation
 * from sorted rows (math_functions.cpp:32-44), refuse it with GAIB_ERR_UNSUPPORTED.  gaib_graph_sort_rows sorts every row's
 * column ids (in place; per-edge caches are rebuilt on demand): everything runs on it, and aggregation results are then equal
 * to the original numbering's up to fp32 summation order instead of bit for bit. */
int gaib_graph_sort_rows(gaib_ctx* ctx, gaib_graph* g);
int64_t gaib_graph_nv(const gaib_graph* g);
int64_t gaib_graph_ne(const gaib_graph* g);
int64_t gaib_graph_nc(const gaib_graph* g); /* columns = rows of the feature table its column ids index (nv unless rectangular) */
const int64_t* gaib_graph_rowptr(const gaib_graph* g);  /* device, int64[nv+1] */
const uint32_t* gaib_graph_colidx(const gaib_graph* g); /* device, uint32[ne]  */
/* compute_vertex_data (lgraph.cpp:22-34 / lgraph.cu:94-105): deg^-1/2, 0 for isolated. */
int gaib_graph_compute_vertex_data(gaib_ctx* ctx, gaib_graph* g);
const float* gaib_graph_vertex_data(const gaib_graph* g); /* device float[nv] or NULL */
/* compute_edge_data (lgraph.cpp:6-20 / lgraph.cu:107-140): 1/(sqrt(d_i)*sqrt(d_j)). */
int gaib_graph_compute_edge_data(gaib_ctx* ctx, gaib_graph* g);
const float* gaib_graph_edge_data(const gaib_graph* g); /* device float[ne] or NULL */
/* device bytes held by the graph (CSR + normalisers + schedules) */
int64_t gaib_graph_device_bytes(const gaib_graph* g);

/* ---- sparse aggregation (the hot loop) -------------------------------------------------
 * out[i,:] = sum_{e in row i} w_e * in[col_e,:]      (out is fully overwritten)
 *   GAIB_W_GCN    w_e = vd[i]*vd[col_e]      GCN_Aggregator::aggregate/d_aggregate
 *                                            (gcn_aggregator.cpp:23-77; .cu:14-51)
 *   GAIB_W_MEAN   w_e = 1/deg(i)             SAGE_Aggregator::aggregate   (sage_aggregator.cpp:7-30)
 *   GAIB_W_MEAN_T w_e = 1/deg(col_e)         SAGE_Aggregator::d_aggregate (sage_aggregator.cpp:32-54)
 *   GAIB_W_EDGE   w_e = d_edge_w[e]          update_all (gat_aggregator.cpp:26-45), spmm()
 *                                            (math_functions.cpp:207-220; cusparseSpMM .cu:407-410)
 *   GAIB_W_EDGE_T w_e = d_edge_w[rev(e)]     symmetric_csr_transpose + update_all fused
 *                                            (gat_aggregator.cpp:175,198)
 * Rows of degree <= the heavy-row threshold are summed in CSR order with a separate
 * multiply and add (bit-identical to the OpenMP loop); longer rows are split over the
 * waves of a workgroup and combined through LDS in a fixed order. */
typedef enum {
  GAIB_W_GCN = 0,
  GAIB_W_MEAN = 1,
  GAIB_W_MEAN_T = 2,
  GAIB_W_EDGE = 3,
  GAIB_W_EDGE_T = 4
} gaib_weight_kind;
int gaib_spmm(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len,
              const float* d_in, float* d_out);
/* out[i,:] += ... : the second half of a split aggregation (owned-column edges first with
 * gaib_spmm, halo-column edges added once the halo rows have arrived). */
int gaib_spmm_acc(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len,
                  const float* d_in, float* d_out);
/* general form: flags = GAIB_ACCUMULATE (out += ...) | GAIB_RELU (relu_gpu fused into the store:
 * the layers apply the activation right after the aggregation / GEMM, gcn_layer.cpp:27) */
#define GAIB_ACCUMULATE 1
#define GAIB_RELU 2
#define GAIB_AGG_SCRATCH 4 /* gaib_spmm_gemm: d_agg is scratch, its contents after the call are unspecified */
#define GAIB_OVERLAPS_TRANSFER 8 /* gaib_spmm_gemm*: a halo exchange is in flight on the communicator's stream while this call
                                  * runs (between gaib_halo_exchange_begin and _end).  The fused kernel is one persistent
                                  * workgroup per CU that holds the CU's registers until its last tile; under RCCL it then leaves
                                  * "comm_reserve_cus" CUs to the send / recv kernels: the option (or GAIB_COMM_RESERVE_CUS) where one is set
                                  * -- an explicit 0 included, -1 = unset --, else gaib_comm_init's default for its transport (32
                                  * under RCCL with > 1 rank, 0 on the peer-to-peer pull); at most num_cus - 64 */
int gaib_spmm_ex(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len,
                 const float* d_in, float* d_out, int flags);
/* multi-head attention weights: d_edge_w is [ne][heads]; column c uses head c / (len/heads).
 * weight_kind must be GAIB_W_EDGE or GAIB_W_EDGE_T when heads > 1. */
int gaib_spmm_mh(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int heads,
                 int len, const float* d_in, float* d_out, int flags);

/* ---- GAT attention pieces ---------------------------------------------------------------
 * gaib_gat_scores: GAT_Aggregator::aggregate's score pass (gat_aggregator.cpp:60-92;
 *   compute_attn_score_warp, graph_operations.h:250-337): per edge
 *   temp = a_l.h[i] + a_r.h[col_e]; scores = leaky_relu(temp, eps); norm = row softmax.
 *   The d_* edge arrays [ne] are written; d_scores may be NULL for 1, 2, 4, 8 or 16 heads (it is
 *   leaky_relu(temp) and nothing in backward reads it), and so may d_temp_scores: backward reads only
 *   its sign, which gaib_gat_softmax_bwd_alpha_re forms again from the per-vertex dots. */
int gaib_gat_scores(gaib_ctx* ctx, gaib_graph* g, int len, const float* d_h, const float* d_alpha_l,
                    const float* d_alpha_r, float epsilon, float* d_temp_scores, float* d_scores,
                    float* d_norm_scores);
/* Multi-head forms (BASELINE config "reddit GAT 2-layer 8-head"; the reference is single-head,
 * include/gnn/aggregator.h:62-65): `heads` independent attentions on the column slices
 * [h*len/heads, (h+1)*len/heads); alpha vectors stay [len]; every edge array is [ne][heads].
 * heads == 1 is exactly the single-head entry point. */
int gaib_gat_scores_mh(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_h,
                       const float* d_alpha_l, const float* d_alpha_r, float epsilon,
                       float* d_temp_scores, float* d_scores, float* d_norm_scores);
int gaib_sddmm_mh(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_grad,
                  const float* d_feat, float* d_out_e);
int gaib_gat_softmax_bwd_alpha_mh(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_feat,
                                  const float* d_norm_scores, const float* d_norm_scores_grad,
                                  const float* d_temp_scores, float epsilon, float* d_scores,
                                  float* d_alpha_lgrad, float* d_alpha_rgrad);
int gaib_edge_transpose_mh(gaib_ctx* ctx, gaib_graph* g, int heads, const float* d_in_e, float* d_out_e);
/* SDDMM (gat_aggregator.cpp:106-113; compute_scores_grad_warp, graph_operations.h:191-223):
 *   d_out_e[e] = <grad[i,:], feat[col_e,:]> */
int gaib_sddmm(gaib_ctx* ctx, gaib_graph* g, int len, const float* d_grad, const float* d_feat,
               float* d_out_e);
/* softmax backward + leaky-relu' + alpha gradients (gat_aggregator.cpp:121-167;
 *   compute_alpha_grad_warp, graph_operations.h:396-467).  d_scores[e] is overwritten with
 *   d(softmax) like the reference does; d_alpha_lgrad/d_alpha_rgrad [len] are overwritten.
 *   Deterministic (no float atomics). */
int gaib_gat_softmax_bwd_alpha(gaib_ctx* ctx, gaib_graph* g, int len, const float* d_feat,
                               const float* d_norm_scores, const float* d_norm_scores_grad,
                               const float* d_temp_scores, float epsilon, float* d_scores,
                               float* d_alpha_lgrad, float* d_alpha_rgrad);
/* general form: H heads; d_scores (the softmax-input gradient ds) may be NULL when the caller does not read it;
 * d_grad_rows / d_fwd_out_rows (both or neither, [nv x len]): the gradient and the forward output of the
 * aggregation -- sum_e p_e dp_e is then taken per vertex as <grad_i, out_i> and the edge arrays are read once;
 * d_norm_scores_t (optional, [ne][heads]) receives the transposed attention p[rev(e)] (symmetric_csr_transpose,
 * gat_aggregator.cpp:172-175), produced in the pass that already walks rev: the gradient aggregation can then read
 * it linearly (GAIB_W_EDGE) instead of through the permutation (GAIB_W_EDGE_T). */
int gaib_gat_softmax_bwd_alpha_ex(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_feat,
                                  const float* d_norm_scores, const float* d_norm_scores_grad,
                                  const float* d_temp_scores, float epsilon, float* d_scores,
                                  float* d_alpha_lgrad, float* d_alpha_rgrad, const float* d_grad_rows,
                                  const float* d_fwd_out_rows, float* d_norm_scores_t);
/* the same without the temp_scores array (1, 2, 4, 8 or 16 heads): leaky-relu' needs only the sign of
 * temp = a_l.h[i] + a_r.h[col_e], formed again from d_feat and the alpha vectors (4 + 4*heads bytes per edge out of the
 * caches instead of 4*heads written in forward and read here through HBM).  Same results bit for bit. */
int gaib_gat_softmax_bwd_alpha_re(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_feat,
                                  const float* d_alpha_l, const float* d_alpha_r, const float* d_norm_scores,
                                  const float* d_norm_scores_grad, float epsilon, float* d_scores,
                                  float* d_alpha_lgrad, float* d_alpha_rgrad, const float* d_grad_rows,
                                  const float* d_fwd_out_rows, float* d_norm_scores_t);
/* The whole edge side of GAT_Aggregator::d_aggregate (gat_aggregator.cpp:99-200: SDDMM, softmax backward + leaky-relu',
 * alpha gradients, transpose, gradient aggregation) in ONE sweep over the edges.  d_fwd_out is the aggregation's forward
 * output (sum_e p_e dp_e of a row == <grad_i, out_i>); d_grad_out [nv x len] must not alias an input.  Nothing per edge is
 * written: the per-edge arrays dp / ds / p^T of the staged entry points do not exist on this path.  The attention comes
 * from d_norm_scores [ne][heads], or -- when d_row_stats (gaib_gat_forward_fused) is given -- is formed again.  Applies to
 * len == 32, 64 or 128 (8, 16 or 32 lanes x 4 columns per edge; the reference's GAT covers len <= 128, global.h:58) with 1, 2, 4,
 * 8 or 16 heads of at least 4 columns, on any graph (option "gat_fused_bwd": -1 or 1 = whenever the shape fits, 0 = never);
 * otherwise returns GAIB_ERR_UNSUPPORTED without touching anything and the caller uses the staged entry points. */
int gaib_gat_backward_fused(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_feat, const float* d_grad,
                            const float* d_fwd_out, const float* d_alpha_l, const float* d_alpha_r,
                            const float* d_norm_scores, const float* d_row_stats, float epsilon, float* d_grad_out,
                            float* d_alpha_lgrad, float* d_alpha_rgrad);
/* GAT_Aggregator::aggregate (gat_aggregator.cpp:57-97) in ONE sweep: scores, edge softmax (online: running maximum and
 * sum per chunk, combined per row) and the attention-weighted aggregation; d_out = act(P h), and d_row_stats
 * [nv][heads][2] = (row maximum of the leaky-relu'd scores, 1 / row sum of exp) is everything backward needs to form the
 * attention again: pass it to gaib_gat_backward_fused as d_row_stats (d_norm_scores may then be NULL) and no [ne][heads]
 * array exists at all.  Same cover as gaib_gat_backward_fused (option "gat_fused_fwd"); GAIB_ERR_UNSUPPORTED otherwise. */
int gaib_gat_forward_fused(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_h, const float* d_alpha_l,
                           const float* d_alpha_r, float epsilon, int relu, float* d_out, float* d_row_stats);
/* Test / diagnostic: d_sign_out [ne][heads] (uint8) = (t_e > 0) of every pre-activation score a_l.h[i] + a_r.h[col_e]
 * EXACTLY as the one-sweep kernels form it.  leaky_relu' jumps at 0, so a score within rounding of zero takes either
 * slope in two correct fp32 evaluations; a comparison of the alpha gradients with an fp64 evaluation imposes these signs
 * on it to measure arithmetic only (the counterpart of running backward on the oracle's relu mask).  d_h [nc x len]. */
int gaib_gat_score_signs(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_h, const float* d_alpha_l,
                         const float* d_alpha_r, uint8_t* d_sign_out);
/* The one-sweep forward / backward on a rank's RECTANGULAR graph (rows = owned vertices, columns = [owned | halo]; tables
 * [nc x len] with the owned rows first).  In a structurally symmetric graph the in-edges of an owned vertex are the
 * reverses of its out-edges, so rs_i, cs_i and the aggregated gradient of i follow from i's own edge list given the halo
 * vertices' h rows, grad rows and records rec[v][h] = (<grad_v, out_v>_h, row maximum, 1 / row sum, 0): two
 * forward-direction halo exchanges (grad rows, records = 4 * heads floats per vertex) and one sweep -- no transposed
 * structure, no reverse exchange.  phase: -1 = everything; 0 = only the chunks over owned columns (may run while the halo
 * rows are still arriving: it reads owned rows only); 1 = the remaining chunks + the per-row combination (phase 0's
 * partial results wait in the context's workspace: no other call on the context in between).  The alpha gradients cover
 * the owned rows; the caller sums them over the ranks.  Same shape cover as gaib_gat_backward_fused (len 32 / 64 / 128; 1, 2,
 * 4, 8 or 16 heads of >= 4 columns; options gat_fused_fwd / gat_fused_bwd = 0 switch them off), else GAIB_ERR_UNSUPPORTED.
 * A graph WITHOUT rows (a rank of a partition with fewer vertices than ranks) is accepted with NULL buffers: the answer --
 * GAIB_OK or GAIB_ERR_UNSUPPORTED -- then follows from len, heads and the option alone, i.e. it is what the ranks that do have
 * rows get (the two paths differ in their halo exchanges: all ranks must take the same one), nothing is written except zero
 * alpha gradients. */
int gaib_gat_forward_fused_rect(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_tab, const float* d_alpha_l,
                                const float* d_alpha_r, float epsilon, int relu, float* d_out, float* d_row_stats, int phase);
int gaib_gat_backward_rec(gaib_ctx* ctx, int64_t nv, int len, int heads, const float* d_grad, const float* d_fwd_out,
                          const float* d_row_stats, float* d_rec);
int gaib_gat_backward_fused_rect(gaib_ctx* ctx, gaib_graph* g, int len, int heads, const float* d_feat_tab,
                                 const float* d_grad_tab, const float* d_rec_tab, const float* d_alpha_l,
                                 const float* d_alpha_r, float epsilon, float* d_grad_out, float* d_alpha_lgrad,
                                 float* d_alpha_rgrad, int phase);
/* GAT backward on a RECTANGULAR graph (a rank's rows over [owned | halo] columns), where the reverse
 * edge of (i -> c) lives on another rank and the reverse-edge permutation is replaced by the rank's transposed local
 * structure (rows = owned + halo vertices, columns = owned rows; include/gnn/partition.h).  gaib_gat_scores_mh and
 * gaib_sddmm_mh take rectangular graphs as they are (d_h / d_feat = the [nc x len] column table).
 *   gaib_gat_softmax_bwd_rows: the row side of gat_aggregator.cpp:121-167 -- g_e = d(softmax) * leaky-relu' into
 *     d_g_out [ne][heads], its row sums into d_rs_out [nv][heads];
 *   gaib_edge_gather_perm: d_out_e[k] = d_in_e[perm[k]] for [ne][heads] edge arrays (g and p in transposed order);
 *   gaib_edge_rowsum: d_out_rows[v] = sum of an edge array over row v (column sums of g = row sums over the transpose);
 *   gaib_gat_alpha_grads: alpha_l' = sum_v rs[v] x[v], alpha_r' = sum_v cs[v] x[v] over nv rows (graph_operations.h:
 *     396-467 regrouped by vertex; fixed two-level reduction, no atomics).
 * The transposed aggregation is gaib_spmm_mh on the transposed graph followed by gaib_halo_reduce. */
int gaib_gat_softmax_bwd_rows(gaib_ctx* ctx, gaib_graph* g, int heads, const float* d_norm_scores,
                              const float* d_norm_scores_grad, const float* d_temp_scores, float epsilon,
                              float* d_g_out, float* d_rs_out);
int gaib_edge_gather_perm(gaib_ctx* ctx, int64_t ne, int heads, const uint32_t* d_perm, const float* d_in_e,
                          float* d_out_e);
int gaib_edge_rowsum(gaib_ctx* ctx, gaib_graph* g, int heads, const float* d_in_e, float* d_out_rows);
int gaib_gat_alpha_grads(gaib_ctx* ctx, int64_t nv, int len, int heads, const float* d_x, const float* d_rs,
                         const float* d_cs, float* d_alpha_lgrad, float* d_alpha_rgrad);
/* symmetric_csr_transpose (math_functions.cpp:46-74; csr2csc math_functions.cu:345-358):
 *   d_out_e[rev(e)] = d_in_e[e].  The reverse-edge permutation is built once per graph. */
int gaib_edge_transpose(gaib_ctx* ctx, gaib_graph* g, const float* d_in_e, float* d_out_e);

/* ---- row classes of a vertex-range partition ---------------------------------------------------------
 * The reference partitioner's local graph marks the owned (master) rows and appends the halo vertices behind them
 * (src/partitioner/graph_partition.cc:70-80,128-178, include/graph_partition.h:21-22,36-37).  An owned row whose edges all
 * stay inside the range -- an INTERIOR row -- needs nothing from the exchange; only BOUNDARY rows wait for the halo rows.
 * gaib_graph_split_classes cuts a rank's owned-column graph and halo-column graph (same rows, gaib_graph_set_vertex_norm
 * applied to both) into class graphs.  Each is COMPACT -- row k is the k-th row of the class, ascending -- and carries a
 * row map: every aggregation entry point (gaib_spmm*, gaib_spmm_gemm*) treats row k of such a graph as row map[k] of its
 * [n_rows x len] output / partial-sum / rows2 matrices, so the classes of one partition fill disjoint rows of the same
 * matrices.  Edge order inside a row is kept: the sums are those of the graphs the classes were cut from.
 *   *interior : rows without halo-column edges                               columns: owned
 *   *bnd_own  : the boundary rows' owned-column edges                        columns: owned
 *   *bnd_halo : the boundary rows' halo-column edges                         columns: halo
 *   *bnd_full : both, per row [owned-column ..., halo-column ...]              columns: [owned | halo]: halo id + n_own
 * Any of the four out pointers may be NULL (not built).  h_n_boundary / h_boundary_edges (optional): boundary rows and
 * their edges.  Class graphs take GAIB_W_GCN / _MEAN / _MEAN_T / single-head _EDGE weights.
 * flags: GAIB_SPLIT_ALL_BOUNDARY = every row counts as a boundary row (*interior gets no rows): where next to no row is
 * interior, one pass over all rows of bnd_full beats two launches. */
#define GAIB_SPLIT_ALL_BOUNDARY 1
int gaib_graph_split_classes(gaib_ctx* ctx, const gaib_graph* g_own, const gaib_graph* g_halo, gaib_graph** interior,
                             gaib_graph** bnd_own, gaib_graph** bnd_halo, gaib_graph** bnd_full, int64_t* h_n_boundary,
                             int64_t* h_boundary_edges, int flags);
/* ---- pieces of a halo-column graph (round 6; the exchange in time slices: gaib_halo_set_pieces below) ------------------
 * Cuts a rank's halo-column graph (gaib_graph_create_rect + gaib_graph_set_vertex_norm; also a *bnd_halo row class) into
 * n_pieces graphs over the SAME rows, column space and normalisers: piece k holds, row by row and in the row's edge order,
 * the edges whose column lies in one of piece k's column ranges -- range j = [h_range_begin[j], h_range_end[j]) of the
 * column space belongs to piece h_range_piece[j]; ranges are disjoint and every edge's column lies in one (else
 * GAIB_ERR_INVALID).  With the ranges of gaib_halo_piece_ranges, piece k reads exactly the rows of the halo table that have
 * landed after gaib_halo_exchange_wait_piece(k): aggregate piece 0, 1, ... in GAIB_ACCUMULATE mode as they arrive (the last
 * one may carry the dense product) -- a row's terms are then added piece by piece, inside a piece in column order: with ONE
 * peer the order of the uncut graph (same bits on rows below the heavy threshold), with several the piece-major order of
 * the same terms.  out: n_pieces graphs (the caller destroys each). */
#define GAIB_GRAPH_MAX_PIECES 16
int gaib_graph_split_pieces(gaib_ctx* ctx, const gaib_graph* g, int n_pieces, int n_ranges, const int64_t* h_range_begin,
                            const int64_t* h_range_end, const int* h_range_piece, gaib_graph** out);
/* the row map by hand (uint32 [nv] device array, copied; n_out_rows = rows of the matrices it indexes; NULL removes it) */
int gaib_graph_set_row_map(gaib_ctx* ctx, gaib_graph* g, const uint32_t* d_row_map, int64_t n_out_rows);
const uint32_t* gaib_graph_row_map(const gaib_graph* g); /* device, uint32[nv], or NULL */
/* Aggregation over TWO feature tables: column ids below n_first index d_in, the others d_in2 (row id - n_first) -- a
 * rank's own rows and the halo table behind them, two allocations read side by side in one pass over [owned | halo]
 * (bnd_full above, n_first = n_own).  Otherwise gaib_spmm_ex / gaib_spmm_gemm(2) (d_rows2 / d_W2: both or NULL). */
int gaib_spmm_2t(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len, const float* d_in,
                 const float* d_in2, int64_t n_first, float* d_out, int flags);
int gaib_spmm_gemm_2t(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w, int len_in, const float* d_in,
                      const float* d_in2, int64_t n_first, float* d_agg, const float* d_W, int transW, const float* d_rows2,
                      const float* d_W2, int len_out, float* d_out, int flags);
/* 1 if gaib_spmm_gemm(2) would run this shape as ONE kernel (shape only: widths, weight kind, the option "spmm_fuse").
 * The classes of a partition fill one output: all of them take the fused kernel or none does -- on a class graph
 * gaib_spmm_gemm* returns GAIB_ERR_UNSUPPORTED for a shape this says 0 to, and the caller aggregates class by class
 * and multiplies all rows at once. */
int gaib_spmm_gemm_fusable(gaib_ctx* ctx, int weight_kind, int len_in, int len_out, int dual);

/* ---- aggregation fused with the dense product: the `din <= dout` branch of the layers
 * (gcn_layer.cpp:19-24, sage_layer.cpp:25-34): agg = A.in (gaib_spmm semantics, [nv x len_in]),
 * out = act(agg . op(W)), op(W) = W [len_in x len_out] or, with transW, W^T for W [len_out x len_in]
 * (the input-gradient product of backward).  For len_in in {64, 128} and len_out % 16 == 0 the product
 * runs on the matrix cores inside the aggregating wave (no second pass over agg); other shapes run
 * gaib_spmm + gaib_sgemm.  flags: GAIB_RELU, GAIB_AGG_SCRATCH, GAIB_ACCUMULATE (d_agg holds partial sums that
 * this call continues -- the halo half of a partitioned aggregation).  d_agg must always be a valid
 * [nv x len_in] buffer.  Option "spmm_fuse" = 0 forces the two-kernel path. */
int gaib_spmm_gemm(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w,
                   int len_in, const float* d_in, float* d_agg, const float* d_W, int transW,
                   int len_out, float* d_out, int flags);

/* the same with a second, row-local product in the store: out = act(agg . op(W) + rows2 . op(W2)), rows2 [nv x len_in],
 * W2 shaped like W -- the self term of a SAGE layer (sage_layer.cpp:22 forward, :50 backward) without its own pass. */
int gaib_spmm_gemm2(gaib_ctx* ctx, gaib_graph* g, int weight_kind, const float* d_edge_w,
                    int len_in, const float* d_in, float* d_agg, const float* d_W, int transW,
                    const float* d_rows2, const float* d_W2, int len_out, float* d_out, int flags);

/* ---- dense update: matmul -> sgemm_gpu -> cublasSgemm (math_functions.cu:321-343) --------
 * row-major C[M x N] = op(A)[M x K] . op(B)[K x N]  (+ C if accum).  fp32 MFMA. */
int gaib_sgemm(gaib_ctx* ctx, int transA, int transB, int64_t M, int64_t N, int64_t K,
               const float* d_A, const float* d_B, int accum, float* d_C);
/* flags = GAIB_ACCUMULATE | GAIB_RELU (activation fused into the epilogue) */
int gaib_sgemm_ex(gaib_ctx* ctx, int transA, int transB, int64_t M, int64_t N, int64_t K,
                  const float* d_A, const float* d_B, int flags, float* d_C);

/* weight gradient with the layer's d_relu folded in (d_relu_gpu on grad_in followed by matmul(transA),
 * gcn_layer.cpp:33-52): d_G <- d_G where d_mask > 0 else 0 (IN PLACE, what d_relu_gpu leaves behind) and
 * C[M x N] (=|+=) A^T . G for A [K x M], G / mask [K x N].  One pass over G instead of two. */
int gaib_sgemm_drelu(gaib_ctx* ctx, int64_t M, int64_t N, int64_t K, const float* d_A, float* d_G,
                     const float* d_mask, int accum, float* d_C);

/* ---- elementwise: relu_gpu / d_relu_gpu (math_functions.cu:242-268), dropout mask replay
 * d_dropout_gpu (:134-146) ---- */
int gaib_relu(gaib_ctx* ctx, int64_t n, const float* d_in, float* d_out);
int gaib_d_relu(gaib_ctx* ctx, int64_t n, const float* d_in_diff, const float* d_data,
                float* d_out_diff);
int gaib_dropout(gaib_ctx* ctx, int64_t n, float scale, float drop_rate, uint64_t seed,
                 const float* d_in, uint8_t* d_masks, float* d_out);
int gaib_d_dropout(gaib_ctx* ctx, int64_t n, float scale, const float* d_in,
                   const uint8_t* d_masks, float* d_out);

/* bias_mv (include/utils/math_functions.hh:36; math_functions.cu:207-221): x[i, j] += b[j] for x [n x len] */
int gaib_bias_add(gaib_ctx* ctx, int64_t n, int len, float* d_x, const float* d_b);
/* reduce_sum (math_functions.hh:37-38; math_functions.cpp:246-262, .cu:224-238): a[j] = sum_i x[i, j], overwritten.
 * Deterministic (two levels in a fixed order; the reference's CUDA kernel adds into a[j] from all threads unguarded). */
int gaib_colsum(gaib_ctx* ctx, int64_t n, int len, const float* d_x, float* d_a);
/* rng_uniform_gpu / gpu_rng_uniform (math_functions.hh:156,174; math_functions.cu:39-50): r[i] uniform on [a, b) from the
 * library's counter RNG (element i of stream `seed`).  The reference draws cuRAND XORWOW numbers: streams differ. */
int gaib_rng_uniform(gaib_ctx* ctx, int64_t n, float a, float b, uint64_t seed, float* d_r);
/* csr2csc (math_functions.hh:45; cusparseCsr2cscEx2, math_functions.cu:345-358): the transpose of a general
 * [nrows x ncols] CSR matrix with 32-bit offsets, all arrays in device memory: d_rowptrT [ncols + 1], d_colidxT [nnz]
 * (row ids, ascending inside a column), d_valuesT [nnz] (d_values / d_valuesT may both be NULL: structure only).  Syncs. */
int gaib_csr2csc(gaib_ctx* ctx, int nrows, int ncols, int nnz, const float* d_values, const int* d_rowptr,
                 const int* d_colidx, float* d_valuesT, int* d_rowptrT, int* d_colidxT);

/* ---- loss / metrics: softmax_cross_entropy_gpu, d_softmax_cross_entropy_gpu,
 * masked_avg_loss_gpu, masked_accuracy_single (math_functions.cu:516-564,749-761,886-942) ---- */
int gaib_softmax_xent(gaib_ctx* ctx, int num_cls, int64_t begin, int64_t end,
                      const float* d_in, const uint8_t* d_masks, const uint8_t* d_labels,
                      float* d_loss, float* d_out);
int gaib_d_softmax_xent(gaib_ctx* ctx, int num_cls, int64_t begin, int64_t end,
                        const uint8_t* d_masks, const uint8_t* d_labels, const float* d_out,
                        float* d_diff);
/* multi-label head: sigmoid + cross entropy with [n x num_cls] 0/1 labels, its gradient, and the micro F1
 * that masked_accuracy_multi returns (src/layers/sigmoid_loss_layer.cpp:4-33, sigmoid_loss_layer.cu:4-17,
 * math_functions.cpp:517-521,553-559,580-621; math_functions.cu:583-637,944-1044).  Vertices of the range that
 * the mask excludes get loss 0; h_counts (optional) receives tp, fp, fn. */
int gaib_sigmoid_xent(gaib_ctx* ctx, int num_cls, int64_t begin, int64_t end,
                      const float* d_in, const uint8_t* d_masks, const uint8_t* d_labels,
                      float* d_loss, float* d_out);
int gaib_d_sigmoid_xent(gaib_ctx* ctx, int num_cls, int64_t begin, int64_t end,
                        const uint8_t* d_masks, const uint8_t* d_labels,
                        const float* d_out, float* d_diff);
int gaib_masked_f1_micro(gaib_ctx* ctx, int64_t begin, int64_t end, int num_cls,
                         const uint8_t* d_masks, const float* d_preds, const uint8_t* d_labels,
                         float* h_result, int64_t* h_counts);
int gaib_masked_avg_loss(gaib_ctx* ctx, int64_t begin, int64_t end, const uint8_t* d_masks,
                         const float* d_loss, float* h_result); /* syncs */
int gaib_masked_accuracy_single(gaib_ctx* ctx, int64_t begin, int64_t end, int num_cls,
                                const uint8_t* d_masks, const float* d_preds,
                                const uint8_t* d_labels, float* h_result); /* syncs */
/* The same reductions with the result left in DEVICE memory (d_result[0]; d_counts[3] = tp, fp, fn): no host wait,
 * so they can be part of a recorded sequence (gaib_capture_begin).  Block partials are added in the order of the
 * host forms: identical bits. */
int gaib_masked_avg_loss_dev(gaib_ctx* ctx, int64_t begin, int64_t end, const uint8_t* d_masks,
                             const float* d_loss, float* d_result);
int gaib_masked_accuracy_single_dev(gaib_ctx* ctx, int64_t begin, int64_t end, int num_cls,
                                    const uint8_t* d_masks, const float* d_preds,
                                    const uint8_t* d_labels, float* d_result);
int gaib_masked_f1_counts_dev(gaib_ctx* ctx, int64_t begin, int64_t end, int num_cls,
                              const uint8_t* d_masks, const float* d_preds, const uint8_t* d_labels,
                              uint64_t* d_counts);

/* ---- l2norm / d_l2norm (math_functions.cu:158-196) ---- */
int gaib_l2norm(gaib_ctx* ctx, int64_t n, int dim, const float* d_in, float* d_out);
int gaib_d_l2norm(gaib_ctx* ctx, int64_t n, int dim, const float* d_feat_in,
                  const float* d_grad_in, float* d_grad_out);

/* ---- optimizer: adam::update_gpu -> update_kernel (src/utilities/optimizer.cu:5-36).
 * The caller owns m/v state and the beta powers (optimizer.h:99-116). ---- */
int gaib_adam_step(gaib_ctx* ctx, int64_t n, const float* d_dW, float* d_W, float* d_m,
                   float* d_v, float alpha, float b1, float b2, float b1_t, float b2_t, float eps);
/* The same step with the beta powers in device memory: d_pow[0] = b1^t, d_pow[1] = b2^t are read by the step and
 * then advanced ON THE DEVICE (the host's `b1_t *= b1; b2_t *= b2`, optimizer.cpp:34-35 -- the same float products),
 * so a recorded step replays correctly.  Initialise d_pow to {b1, b2}. */
int gaib_adam_step_dev(gaib_ctx* ctx, int64_t n, const float* d_dW, float* d_W, float* d_m, float* d_v,
                       float alpha, float b1, float b2, float eps, float* d_pow);

/* ---- multi-GPU helpers (no reference counterpart) --------------------------
 * pack rows for the halo exchange: d_out[k,:] = d_in[d_idx[k],:] */
int gaib_gather_rows(gaib_ctx* ctx, int64_t n_idx, const int64_t* d_idx, int len,
                     const float* d_in, float* d_out);
/* out[dst_idx[k], :] = in[src_idx[k], :], dst rows distinct.  The halo plans pack with it in SOURCE order (src ascending):
 * a row that goes to several peers is read from HBM once (its repeats hit the cache) instead of once per peer. */
int gaib_gather_scatter_rows(gaib_ctx* ctx, int64_t n_idx, const int64_t* d_src_idx, const int64_t* d_dst_idx, int len,
                             const float* d_in, float* d_out);

/* ---- collectives of the vertex-range partitioned path ("halo_exchange(handle,D,buf)",
 * "allreduce(buf,n)"; 8e) -------------------------------------------------------------------------------------
 * One process per GPU.  The reference has no multi-GPU GNN; the partition scheme is its
 * PartitionedGraph::edgecut_induced_partition1D (src/partitioner/graph_partition.cc:128-178), the host pattern it uses
 * for multi-GPU work is one host thread per device with peer copies (src/triangle/multigpu_induced.cu:31-84).
 *
 * Transports:
 *   GAIB_COMM_RCCL  RCCL (ncclSend/ncclRecv groups, ncclAllReduce) on a communication stream; ONE GPU PER RANK
 *                   (RCCL refuses two ranks on one device: gaib_comm_init then fails on every rank); stream-ordered.
 *   GAIB_COMM_IPC   peer-to-peer pull through hipIpc handles published in POSIX shared memory (one node; xGMI
 *                   between GPUs, plain device copies when ranks share a GPU -- tests); host-synchronous hand-overs.
 *                   A send buffer above GAIB_IPC_CHUNK_BYTES (default 512 MiB) is cut into separately exported chunks
 *                   of whole rows (at most 64): hipIpcOpenMemHandle does not return for allocations above 2 GiB.  No
 *                   allocation above GAIB_IPC_EXPORT_LIMIT_BYTES (1.5 GiB) is exported: GAIB_ERR_UNSUPPORTED instead
 *                   (gaib_halo_reduce stages its partial rows in chunks the same way).  GAIB_COMM_DEBUG=1 prints the steps
 *                   of every exchange with time stamps on stderr.
 * gaib_comm_unique_id is called by ONE rank; the caller carries the GAIB_COMM_ID_BYTES to the other ranks (a file,
 * MPI, torch.distributed's store ...).  gaib_comm_init is collective.  Every collective below must be called by
 * every rank in the same order.  A rank that fails or does not arrive within GAIB_COMM_TIMEOUT_S (default 120 s)
 * makes the waiting ranks return GAIB_ERR_COMM instead of hanging. */
typedef struct gaib_comm gaib_comm;
typedef struct gaib_halo gaib_halo;
#define GAIB_COMM_ID_BYTES 128
#define GAIB_COMM_RCCL 0
#define GAIB_COMM_IPC 1
/* local and cheap: GAIB_OK if this process can use the transport at all (RCCL: librccl.so.1 loads and has every entry
 * point used here).  What the ranks agree on -- together with "every rank has a device of its own" -- BEFORE any of
 * them enters gaib_comm_init(GAIB_COMM_RCCL): ncclCommInitRank is collective and has no deadline. */
int gaib_comm_transport_available(int transport);
int gaib_comm_unique_id(int transport, void* h_id);
int gaib_comm_init(gaib_ctx* ctx, int rank, int nranks, const void* h_id, int transport, gaib_comm** out);
int gaib_comm_destroy(gaib_comm* comm);
int gaib_comm_rank(const gaib_comm* comm);
int gaib_comm_size(const gaib_comm* comm); /* RCCL: what ncclCommCount reports for the communicator that was built */
int gaib_comm_barrier(gaib_comm* comm); /* syncs the compute stream first */
/* in-place sum over ranks on the context's stream: the weight / alpha gradients of a layer (<= 256 KB each);
 * every rank ends with the same bits */
int gaib_allreduce_f32(gaib_comm* comm, float* d_buf, int64_t n);
/* host scalars (masked loss sums, accuracy counts): in-place sum over ranks, n <= 1024; syncs */
int gaib_allreduce_host_f64(gaib_comm* comm, double* h_buf, int n);
/* A halo plan: which of this rank's rows go to which peer and how many rows arrive from each peer.
 *   h_send_counts[r] rows go to rank r: their local row ids are send_idx[sum(h_send_counts[:r]) ...] (host or device
 *   array, copied); h_recv_counts[r] rows arrive from rank r; both 0 for the rank itself.  The table an exchange fills
 *   is [sum(h_recv_counts) x len], grouped by source rank in rank order -- the order of the halo column ids of
 *   gaib_graph_create_rect's column space after the owned rows.  Collective (create and destroy, in the same order
 *   on every rank); at most 8 plans alive per communicator, a destroyed plan's slot is reused.  One plan serves any
 *   mix of row lengths and directions: its buffers grow as needed and peers re-map them when they do. */
int gaib_halo_create(gaib_comm* comm, const int64_t* h_send_counts, const int64_t* send_idx, int idx_on_device,
                     const int64_t* h_recv_counts, gaib_halo** out);
int gaib_halo_destroy(gaib_halo* halo);
int64_t gaib_halo_rows(const gaib_halo* halo);      /* rows of the halo table */
int64_t gaib_halo_send_rows(const gaib_halo* halo); /* rows this rank packs per exchange */
int64_t gaib_halo_link_rows(const gaib_halo* halo); /* the most rows one peer pair moves per exchange, either direction:
                                                      * what one xGMI link carries (every pair has its own) */
int64_t gaib_halo_bytes_sent(const gaib_halo* halo);
/* how the plan's rows left so far: exchanges that ran the pack kernel, sends that went STRAIGHT from the caller's matrix, and the
 * number of peers whose send list is one run of consecutive rows (RCCL transport: such a peer is sent from d_rows itself; if
 * every peer is, the plan never packs -- the N-way cut of a graph whose ranges need all of each other's rows) */
int gaib_halo_send_stats(const gaib_halo* halo, int64_t* h_packs, int64_t* h_direct_sends, int* h_direct_peers);
/* one exchange = begin (pack the requested rows of d_rows [n_own x len] on the compute stream, start moving them)
 * ... independent work on the compute stream (the owned-column edges of the aggregation) ... end (the compute stream
 * continues only after the halo rows have arrived; *d_table stays valid until the next begin on this plan).
 * d_rows must stay valid and UNMODIFIED until gaib_halo_exchange_end: over RCCL a peer whose send list is one run of
 * consecutive rows is sent straight from d_rows on the communication stream (gaib_halo_send_stats), not from a packed
 * copy.  (A plan that mixes direct and packed peers still packs -- and reserves buffer space for -- all of its rows.) */
int gaib_halo_exchange_begin(gaib_halo* halo, int len, const float* d_rows);
int gaib_halo_exchange_end(gaib_halo* halo, const float** d_table);
/* An exchange in K time slices ("pieces", round 6): slice k of a peer pair's R rows is rows [R k / K, R (k + 1) / K) of
 * that pair's segment (gaib_halo_piece_slice: sender and receiver cut the same R the same way), slice k of EVERY pair
 * travels together (one ncclGroup / one round of peer pulls), so every link is busy throughout and piece k has landed after
 * ~(k + 1) / K of the exchange.  Between begin and end, gaib_halo_exchange_wait_piece(k) makes the compute stream wait for
 * piece k only (stream-ordered, the host does not wait): the caller's halo-column pass over piece k's columns
 * (gaib_graph_split_pieces with gaib_halo_piece_ranges) then runs while the later slices are still on the wire.
 * gaib_halo_exchange_end stays due after the last piece.  The table's layout does not change with K.
 * gaib_halo_set_pieces: 1 <= n_pieces <= 16, THE SAME ON EVERY RANK (the peer-to-peer transport checks it, RCCL would
 * mismatch its counts), not while an exchange is in flight; default 1.  gaib_halo_piece_ranges: the column ranges
 * [begin, end) of the table that piece k fills (at most one per peer; h_n = how many, at most cap). */
int gaib_halo_set_pieces(gaib_halo* halo, int n_pieces);
/* the library's choice for a partition of n_global vertices into `world` ranges, a function of figures every rank holds (so
 * the ranks agree without a collective): GAIB_HALO_PIECES if set (1 .. 16), else 4 from 131 072 rows per range, 2 from
 * 32 768, else 1.  make_partitioned_graph and dist.py set it on their plans. */
int gaib_halo_default_pieces(int64_t n_global, int world);
int gaib_halo_pieces(const gaib_halo* halo);
int gaib_halo_piece_slice(int64_t rows, int n_pieces, int piece, int64_t* h_lo, int64_t* h_hi); /* pure arithmetic, no device */
int gaib_halo_piece_ranges(const gaib_halo* halo, int piece, int cap, int64_t* h_begin, int64_t* h_end, int* h_n);
int gaib_halo_exchange_wait_piece(gaib_halo* halo, int piece, const float** d_table);

/* the reverse of an exchange: d_halo_rows [halo rows x len] (the table's layout) holds this rank's partial sums for its
 * HALO vertices; they travel back to the owners, which add them to their own rows: d_rows[send_idx[k], :] += arrived[k, :],
 * peer by peer in rank order (deterministic).  The transposed aggregation of GAT backward on a partition: the term
 * p_(i->c) * grad_i of a row i that another rank owns.  Collective; not while an exchange of the same plan is in flight. */
int gaib_halo_reduce(gaib_halo* halo, int len, const float* d_halo_rows, float* d_rows);

/* ---- in-stream kernel timing (measurement only) -----------------------------------------------
 * When enabled, the aggregation entry points bracket each kernel launch with a HIP event pair
 * on the context's stream (no host sync).  gaib_prof_get syncs the stream and returns the launch
 * count and summed device time for a key: "spmm_light", "spmm_heavy", "spmm_sub", "sgemm". */
int gaib_prof_enable(gaib_ctx* ctx, int on);
int gaib_prof_reset(gaib_ctx* ctx);
int gaib_prof_get(gaib_ctx* ctx, const char* key, int64_t* h_count, double* h_total_ms);
/* the same plus the launches' ALGORITHMIC work as the call sites state it (bytes of the aggregation /
 * edge kernels, flops and operand bytes of the dense products; 0 where a site states none) and h_roof_ms = the time those
 * launches would take at the chip's roofs, per launch max(bytes / 8 TB/s, flops / 157.3 TFLOP/s).  gaib_prof_table: one text
 * line "key[@shape] count total_ms alg_bytes flops roof_ms" per key and launch shape that has records -- shape = the row width
 * of a gather kernel ("spmm_light@47"), M x N x K of a dense product ("sgemm@2449029x128x100") -- (at most cap bytes incl. the
 * terminating 0 are written; *h_needed = what the whole table takes); the epoch records of bench.py are built from it */
int gaib_prof_get_work(gaib_ctx* ctx, const char* key, int64_t* h_count, double* h_total_ms, double* h_bytes, double* h_flops,
                       double* h_roof_ms);
int gaib_prof_table(gaib_ctx* ctx, char* h_buf, size_t cap, size_t* h_needed);
/* heavy-row split of a graph at the context's current threshold: rows / edges handled by the
 * workgroup-per-row kernel, and the maximum degree */
/* locality of the vertex numbering: the share of (sampled) edges whose column id lies within 32 768 of the row id
 * (0 for rectangular graphs and graphs below 262 144 vertices).  The fused aggregation's tile supply follows it: from
 * 0.25 on, tiles are dealt XCD-affine in long chunks (option "spmm_tile_xcd" = -1), else off one global counter; the
 * affine form also requests a row's column ids one row ahead (option "spmm_prefetch_ids", default 1; same sums). */
int gaib_graph_locality(gaib_ctx* ctx, gaib_graph* g, float* h_near_frac);
int gaib_graph_stats(gaib_ctx* ctx, gaib_graph* g, int64_t* h_n_heavy, int64_t* h_heavy_edges,
                     int64_t* h_max_degree);

/* ---- bandwidth probes (measurement only) -------------------------------------------------------
 * gaib_probe_stream_copy: a 16-B-per-lane copy kernel of `bytes` bytes (rounded down to 16), `iters` launches
 *   on the context's stream between one HIP event pair; *h_gbs = (bytes read + bytes written) * iters / time.
 *   This is the chip's achievable streaming rate that bench.py reports next to the 8 TB/s spec figure
 *   (SURVEY.md 8d "confirm with a stream-copy kernel").
 * gaib_probe_peer_copy: hipMemcpyPeerAsync of `bytes` from device src_dev to device dst_dev, `iters` copies
 *   (and as many in the opposite direction at the same time on a second stream when bidir != 0);
 *   *h_gbs = bytes moved per direction * iters / time.  The xGMI link probe
 *   (template: /root/reference/src/test/test_nvlink.cu:37-77).  Needs two visible devices. */
int gaib_probe_stream_copy(gaib_ctx* ctx, size_t bytes, int iters, double* h_gbs);
int gaib_probe_peer_copy(int src_dev, int dst_dev, size_t bytes, int iters, int bidir, double* h_gbs);

/* ---- tuning knobs (benchmarks only; defaults are what ships) ---- */
int gaib_set_option(gaib_ctx* ctx, const char* key, int64_t value);
/* what a record wants to name: "comm_reserve_cus" (CUs the fused kernel leaves to the transport: the EFFECTIVE figure -- option,
 * environment or the communicator's default, clamped; "comm_reserve_cus_raw": what the caller set, -1 = unset),
 * "spmm_fuse_cus", "spmm_flat_ring", "num_cus" */
int gaib_get_option(gaib_ctx* ctx, const char* key, int64_t* h_value);

#ifdef __cplusplus
}
#endif
#endif /* GAIB_H */
