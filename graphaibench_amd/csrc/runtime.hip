// runtime.hip -- context, memory and option entry points of include/gaib.h.
// Replaces the reference's gpu_context statics (include/gnn/gpu_context.h:4-16,
// src/utilities/random.cpp:62-80) and the malloc/copy helpers
// (include/utils/math_functions.hh:161-173, include/utils/cutils.h:193-202).
#include <algorithm>
#include <stdarg.h>
#include <string.h>
#include <utility>
#include "common.h"
#include <stdlib.h>

static thread_local char g_err[1024] = "";

void gaib_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* gaib_last_error(void) { return g_err; }
extern "C" const char* gaib_version(void) { return "graphaibench_amd 0.1 (gfx950)"; }

extern "C" int gaib_device_count(int* h_count) {
  GAIB_CHECK(h_count, "gaib_device_count: h_count is NULL");
  *h_count = 0;
  GAIB_HIP(hipGetDeviceCount(h_count));
  return GAIB_OK;
}

extern "C" int gaib_ctx_create(int device, void* stream, gaib_ctx** out) {
  GAIB_CHECK(out != nullptr, "gaib_ctx_create: out is NULL");
  int ndev = 0;
  GAIB_HIP(hipGetDeviceCount(&ndev));
  GAIB_CHECK(device >= 0 && device < ndev, "gaib_ctx_create: device %d out of range (%d)", device, ndev);
  GAIB_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  GAIB_HIP(hipGetDeviceProperties(&prop, device));
  gaib_ctx* c = new gaib_ctx();
  c->device = device;
  c->stream = (hipStream_t)stream;
  c->num_cus = prop.multiProcessorCount;
  c->ws = nullptr;
  c->ws_bytes = 0;
  c->pad = nullptr;
  c->pad_bytes = 0;
  c->main_stream = c->stream;
  c->side_stream = nullptr;
  c->ev_fork = c->ev_join = nullptr;
  c->ws_side = nullptr;
  c->ws_side_bytes = 0;
  c->forked = 0;
  c->spmm_heavy_threshold = 1024;
  c->spmm_variant = 0;
  c->spmm_xcd_swizzle = 2;
  c->spmm_unroll = 0;
  c->spmm_tile_xcd = -1;
  c->spmm_prefetch_ids = 1;
  c->spmm_fuse_cus = 0;
  c->spmm_fuse = 1;
  c->spmm_pad = 1;
  c->spmm_chunked = -1;
  c->spmm_flat = -1;
  c->spmm_addr_mode = 0;
  c->spmm_gather_mode = 0;
  c->spmm_hot_bytes = 3 << 20;
  c->sgemm_variant = 0;
  if (const char* e = getenv("GAIB_SGEMM_VARIANT")) c->sgemm_variant = atoi(e);  // (A/B of a whole trainer run: 61 = without sgemm_skinny.hip)
  c->gat_fast = 1;
  c->gat_row_waves = 4;
  c->gat_chunk_sort = 1;
  c->gat_chunk_colsum = -1;
  c->graph_rev_search = 0;
  c->gat_fused_bwd = -1;
  c->gat_fused_fwd = -1;
  c->gat_fused_unroll = 4;
  c->spmm_flat_ring = GAIB_FLAT_RING_DEFAULT;
  c->comm_reserve_cus = -1;  // unset: the communicator's default applies (an explicit 0 stays 0)
  c->comm_reserve_default = 0;
  c->gat_interleave = 0;
  c->gat_bwd_pk = 0;  // measured, round 6: the packed-math sweep saves a third of the VALU instructions and nothing at the real
                      // column ids (6.27 vs 6.25 ms at the reddit shape: the sweep is bound by the L2 -> fabric gather stream there)
  c->gat_chunk_xcd = 0;
  c->prof_on = 0;
  c->capturing = 0;
  c->live_execs = 0;
  c->owned_stream = nullptr;
  *out = c;
  return GAIB_OK;
}

extern "C" int gaib_ctx_destroy(gaib_ctx* ctx) {
  if (!ctx) return GAIB_OK;
  GAIB_CHECK(ctx->live_execs == 0, "gaib_ctx_destroy: %d recorded sequence(s) of this context are still alive "
                                   "(gaib_exec_destroy them first: they hold its scratch pointers)", ctx->live_execs);
  (void)hipSetDevice(ctx->device);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->pad) (void)hipFree(ctx->pad);
  if (ctx->ws_side) (void)hipFree(ctx->ws_side);
  for (void* p : ctx->retired) (void)hipFree(p);
  if (ctx->side_stream) (void)hipStreamDestroy(ctx->side_stream);
  if (ctx->owned_stream) (void)hipStreamDestroy(ctx->owned_stream);
  if (ctx->ev_fork) (void)hipEventDestroy(ctx->ev_fork);
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  delete ctx;
  return GAIB_OK;
}

extern "C" int gaib_ctx_set_stream(gaib_ctx* ctx, void* stream) {
  GAIB_CHECK(ctx, "gaib_ctx_set_stream: ctx is NULL");
  GAIB_CHECK(!ctx->forked, "gaib_ctx_set_stream: a side section is open");
  ctx->stream = (hipStream_t)stream;
  ctx->main_stream = ctx->stream;
  return GAIB_OK;
}

// A stream of the context's own (non-blocking, i.e. not ordered against the null stream): what a caller without a
// stream of its own -- the C++ drivers -- needs before gaib_capture_begin, since the null stream cannot be captured.
extern "C" int gaib_ctx_own_stream(gaib_ctx* ctx) {
  GAIB_CHECK(ctx, "gaib_ctx_own_stream: ctx is NULL");
  GAIB_CHECK(!ctx->forked && !ctx->capturing, "gaib_ctx_own_stream: a side section or capture is open");
  if (ctx->owned_stream && ctx->stream == ctx->owned_stream) return GAIB_OK;
  GAIB_HIP(hipSetDevice(ctx->device));
  GAIB_HIP(hipStreamSynchronize(ctx->stream));  // what was enqueued on the old stream is done before the switch
  if (!ctx->owned_stream) GAIB_HIP(hipStreamCreateWithFlags(&ctx->owned_stream, hipStreamNonBlocking));
  ctx->stream = ctx->main_stream = ctx->owned_stream;
  return GAIB_OK;
}

extern "C" int gaib_side_begin(gaib_ctx* ctx) {
  GAIB_CHECK(ctx, "gaib_side_begin: ctx is NULL");
  GAIB_CHECK(ctx->forked == 0, "gaib_side_begin: a side section is already open or not waited for");
  GAIB_HIP(hipSetDevice(ctx->device));
  if (!ctx->side_stream) {
    GAIB_HIP(hipStreamCreateWithFlags(&ctx->side_stream, hipStreamNonBlocking));
    GAIB_HIP(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    GAIB_HIP(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  }
  GAIB_HIP(hipEventRecord(ctx->ev_fork, ctx->main_stream));
  GAIB_HIP(hipStreamWaitEvent(ctx->side_stream, ctx->ev_fork, 0));
  ctx->stream = ctx->side_stream;
  std::swap(ctx->ws, ctx->ws_side);
  std::swap(ctx->ws_bytes, ctx->ws_side_bytes);
  ctx->forked = 1;
  return GAIB_OK;
}

extern "C" int gaib_side_end(gaib_ctx* ctx) {
  GAIB_CHECK(ctx, "gaib_side_end: ctx is NULL");
  GAIB_CHECK(ctx->forked == 1, "gaib_side_end: no open side section");
  GAIB_HIP(hipEventRecord(ctx->ev_join, ctx->side_stream));
  ctx->stream = ctx->main_stream;
  std::swap(ctx->ws, ctx->ws_side);
  std::swap(ctx->ws_bytes, ctx->ws_side_bytes);
  ctx->forked = 2;
  return GAIB_OK;
}

extern "C" int gaib_side_wait(gaib_ctx* ctx) {
  GAIB_CHECK(ctx, "gaib_side_wait: ctx is NULL");
  GAIB_CHECK(ctx->forked == 2, "gaib_side_wait: no ended side section");
  GAIB_HIP(hipStreamWaitEvent(ctx->main_stream, ctx->ev_join, 0));
  ctx->forked = 0;
  return GAIB_OK;
}

extern "C" int gaib_sync(gaib_ctx* ctx) {
  GAIB_CHECK(ctx, "gaib_sync: ctx is NULL");
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_sync");
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  return GAIB_OK;
}

// A buffer whose address may sit in the kernel nodes of a recorded sequence must outlive that sequence: with execs
// alive it is parked on ctx->retired (a replay keeps using it as ITS scratch, calls made outside the replay use the
// new one -- scratch carries nothing from call to call), otherwise freed.
static int release_or_retire(gaib_ctx* ctx, void* p) {
  if (!p) return GAIB_OK;
  if (ctx->live_execs > 0) {
    ctx->retired.push_back(p);
    return GAIB_OK;
  }
  GAIB_HIP(hipFree(p));
  return GAIB_OK;
}

int gaib_ws_reserve(gaib_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return GAIB_OK;
  GAIB_NOT_WHILE_CAPTURING(ctx, "growing the workspace");
  // the old buffer may still be in use by enqueued kernels
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  GAIB_TRY(release_or_retire(ctx, ctx->ws));
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  size_t want = bytes + (bytes >> 2);
  hipError_t e = hipMalloc(&ctx->ws, want);
  if (e != hipSuccess) {
    gaib_set_error("workspace hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    return GAIB_ERR_NOMEM;
  }
  ctx->ws_bytes = want;
  return GAIB_OK;
}

int gaib_pad_reserve(gaib_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->pad_bytes) return GAIB_OK;
  GAIB_NOT_WHILE_CAPTURING(ctx, "growing the padded-table buffer");
  GAIB_HIP(hipStreamSynchronize(ctx->stream));  // the old buffer may still be read by enqueued kernels
  GAIB_TRY(release_or_retire(ctx, ctx->pad));
  ctx->pad = nullptr;
  ctx->pad_bytes = 0;
  hipError_t e = hipMalloc(&ctx->pad, bytes);
  if (e != hipSuccess) {
    gaib_set_error("padded-table hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return GAIB_ERR_NOMEM;
  }
  ctx->pad_bytes = bytes;
  return GAIB_OK;
}

extern "C" int gaib_malloc(gaib_ctx* ctx, size_t bytes, void** d_ptr) {
  GAIB_CHECK(ctx && d_ptr, "gaib_malloc: NULL argument");
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_malloc");
  GAIB_HIP(hipSetDevice(ctx->device));
  *d_ptr = nullptr;
  if (bytes == 0) return GAIB_OK;
  hipError_t e = hipMalloc(d_ptr, bytes);
  if (e != hipSuccess) {
    gaib_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return GAIB_ERR_NOMEM;
  }
  return GAIB_OK;
}

extern "C" int gaib_free(gaib_ctx* ctx, void* d_ptr) {
  GAIB_CHECK(ctx, "gaib_free: ctx is NULL");
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_free");
  if (d_ptr) GAIB_HIP(hipFree(d_ptr));
  return GAIB_OK;
}

extern "C" int gaib_memcpy_h2d(gaib_ctx* ctx, void* d_dst, const void* h_src, size_t bytes) {
  GAIB_CHECK(ctx, "gaib_memcpy_h2d: ctx is NULL");
  if (bytes == 0) return GAIB_OK;
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_memcpy_h2d");
  GAIB_HIP(hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
  // pageable source: make the call safe to return from (source may be freed by caller)
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  return GAIB_OK;
}

extern "C" int gaib_memcpy_d2h(gaib_ctx* ctx, void* h_dst, const void* d_src, size_t bytes) {
  GAIB_CHECK(ctx, "gaib_memcpy_d2h: ctx is NULL");
  if (bytes == 0) return GAIB_OK;
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_memcpy_d2h");
  GAIB_HIP(hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  return GAIB_OK;
}

extern "C" int gaib_memcpy_d2d(gaib_ctx* ctx, void* d_dst, const void* d_src, size_t bytes) {
  GAIB_CHECK(ctx, "gaib_memcpy_d2d: ctx is NULL");
  if (bytes == 0) return GAIB_OK;
  GAIB_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return GAIB_OK;
}

// ---- pinned host memory + stream-ordered read-back (results of a captured sequence) ---------------------------------
extern "C" int gaib_host_alloc(gaib_ctx* ctx, size_t bytes, void** h_ptr) {
  GAIB_CHECK(ctx && h_ptr, "gaib_host_alloc: NULL argument");
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_host_alloc");
  GAIB_HIP(hipSetDevice(ctx->device));
  *h_ptr = nullptr;
  if (bytes == 0) return GAIB_OK;
  hipError_t e = hipHostMalloc(h_ptr, bytes, hipHostMallocDefault);
  if (e != hipSuccess) {
    gaib_set_error("hipHostMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return GAIB_ERR_NOMEM;
  }
  memset(*h_ptr, 0, bytes);
  return GAIB_OK;
}

extern "C" int gaib_host_free(gaib_ctx* ctx, void* h_ptr) {
  GAIB_CHECK(ctx, "gaib_host_free: ctx is NULL");
  if (h_ptr) GAIB_HIP(hipHostFree(h_ptr));
  return GAIB_OK;
}

extern "C" int gaib_memcpy_d2h_async(gaib_ctx* ctx, void* h_pinned_dst, const void* d_src, size_t bytes) {
  GAIB_CHECK(ctx && (bytes == 0 || (h_pinned_dst && d_src)), "gaib_memcpy_d2h_async: NULL argument");
  if (bytes == 0) return GAIB_OK;
  GAIB_HIP(hipMemcpyAsync(h_pinned_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  return GAIB_OK;
}

// ---- HIP graphs: a recorded call sequence replayed with one launch ------------------------------------------------
// The models of BASELINE configs 1-2 (cora / citeseer: 2 layers, 16 hidden columns, ~13 k edges) are launch bound: an
// epoch is ~40 kernels of a few microseconds each, and the host spends longer enqueueing one than the GPU running it.
// Between gaib_capture_begin and gaib_capture_end every call on this context is RECORDED on its stream (relaxed capture
// mode: only this stream, other threads are not affected), nothing runs; gaib_exec_launch replays the whole sequence.
// Pointers and scalar arguments are frozen at capture time: what changes from replay to replay must live in device
// memory (gaib_adam_step_dev keeps the beta powers there; gaib_masked_*_dev leave their results there).
struct gaib_exec {
  gaib_ctx* ctx;  // the context it was recorded on (must outlive it: its retired workspaces are released with the last exec)
  hipGraph_t graph;
  hipGraphExec_t exec;
  int device;
  size_t nodes;
  hipEvent_t ev0, ev1;  // around the last launch (gaib_exec_elapsed_ms)
  int launched;
};

extern "C" int gaib_capture_begin(gaib_ctx* ctx) {
  GAIB_CHECK(ctx, "gaib_capture_begin: ctx is NULL");
  GAIB_CHECK(!ctx->capturing, "gaib_capture_begin: a capture is already open");
  GAIB_CHECK(!ctx->forked, "gaib_capture_begin: a side section is open");
  GAIB_CHECK(ctx->stream != nullptr, "gaib_capture_begin: the context runs on the null stream, which cannot be captured "
                                     "(create the context on a stream of its own)");
  GAIB_CHECK(!ctx->prof_on, "gaib_capture_begin: in-stream kernel timing is on (gaib_prof_enable)");
  GAIB_HIP(hipSetDevice(ctx->device));
  GAIB_HIP(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
  ctx->capturing = 1;
  return GAIB_OK;
}

extern "C" int gaib_capture_abort(gaib_ctx* ctx) {
  GAIB_CHECK(ctx, "gaib_capture_abort: ctx is NULL");
  if (!ctx->capturing) return GAIB_OK;
  hipGraph_t g = nullptr;
  (void)hipStreamEndCapture(ctx->stream, &g);
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  ctx->capturing = 0;
  return GAIB_OK;
}

extern "C" int gaib_capture_end(gaib_ctx* ctx, gaib_exec** out) {
  GAIB_CHECK(ctx && out, "gaib_capture_end: NULL argument");
  GAIB_CHECK(ctx->capturing, "gaib_capture_end: no open capture");
  *out = nullptr;
  hipGraph_t g = nullptr;
  hipError_t e = hipStreamEndCapture(ctx->stream, &g);
  ctx->capturing = 0;
  if (e != hipSuccess || !g) {
    gaib_set_error("gaib_capture_end: hipStreamEndCapture: %s (a call inside the capture cannot be recorded)",
                   hipGetErrorString(e));
    if (g) (void)hipGraphDestroy(g);
    return GAIB_ERR_HIP;
  }
  gaib_exec* x = new gaib_exec();
  x->ctx = ctx;
  x->graph = g;
  x->exec = nullptr;
  x->device = ctx->device;
  x->nodes = 0;
  x->ev0 = x->ev1 = nullptr;
  x->launched = 0;
  (void)hipGraphGetNodes(g, nullptr, &x->nodes);
  e = hipGraphInstantiate(&x->exec, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    gaib_set_error("gaib_capture_end: hipGraphInstantiate: %s", hipGetErrorString(e));
    (void)hipGraphDestroy(g);
    delete x;
    return GAIB_ERR_HIP;
  }
  ctx->live_execs++;
  *out = x;
  return GAIB_OK;
}

extern "C" int64_t gaib_exec_nodes(const gaib_exec* x) { return x ? (int64_t)x->nodes : 0; }

extern "C" int gaib_exec_launch(gaib_ctx* ctx, gaib_exec* x) {
  GAIB_CHECK(ctx && x, "gaib_exec_launch: NULL argument");
  GAIB_CHECK(!ctx->capturing, "gaib_exec_launch: inside a capture");
  GAIB_CHECK(x->ctx == ctx, "gaib_exec_launch: the sequence was recorded on another context (its scratch pointers belong there)");
  if (!x->ev0) {
    GAIB_HIP(hipEventCreate(&x->ev0));
    GAIB_HIP(hipEventCreate(&x->ev1));
  }
  GAIB_HIP(hipEventRecord(x->ev0, ctx->stream));
  GAIB_HIP(hipGraphLaunch(x->exec, ctx->stream));
  GAIB_HIP(hipEventRecord(x->ev1, ctx->stream));
  x->launched = 1;
  return GAIB_OK;
}

// device time of the last launch (after the stream reached its end: gaib_sync)
extern "C" int gaib_exec_elapsed_ms(gaib_exec* x, float* h_ms) {
  GAIB_CHECK(x && h_ms, "gaib_exec_elapsed_ms: NULL argument");
  GAIB_CHECK(x->launched, "gaib_exec_elapsed_ms: never launched");
  GAIB_HIP(hipEventElapsedTime(h_ms, x->ev0, x->ev1));
  return GAIB_OK;
}

extern "C" int gaib_exec_destroy(gaib_exec* x) {
  if (!x) return GAIB_OK;
  if (x->ev0) (void)hipEventDestroy(x->ev0);
  if (x->ev1) (void)hipEventDestroy(x->ev1);
  if (x->exec) (void)hipGraphExecDestroy(x->exec);
  if (x->graph) (void)hipGraphDestroy(x->graph);
  gaib_ctx* ctx = x->ctx;
  if (ctx && --ctx->live_execs == 0 && !ctx->retired.empty()) {
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);  // a replay that used them may still be running
    for (void* p : ctx->retired) (void)hipFree(p);
    ctx->retired.clear();
  }
  delete x;
  return GAIB_OK;
}

extern "C" int gaib_prof_enable(gaib_ctx* ctx, int on) {
  GAIB_CHECK(ctx, "gaib_prof_enable: ctx is NULL");
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_prof_enable");
  ctx->prof_on = on ? 1 : 0;
  return GAIB_OK;
}

extern "C" int gaib_prof_reset(gaib_ctx* ctx) {
  GAIB_CHECK(ctx, "gaib_prof_reset: ctx is NULL");
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  for (auto& r : ctx->prof) {
    (void)hipEventDestroy(r.a);
    (void)hipEventDestroy(r.b);
  }
  ctx->prof.clear();
  return GAIB_OK;
}

extern "C" int gaib_prof_get(gaib_ctx* ctx, const char* key, int64_t* h_count, double* h_total_ms) {
  GAIB_CHECK(ctx && key && h_count && h_total_ms, "gaib_prof_get: NULL argument");
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  int64_t n = 0;
  double ms = 0.0;
  for (auto& r : ctx->prof) {
    if (strcmp(r.key, key) != 0) continue;
    float t = 0.f;
    GAIB_HIP(hipEventElapsedTime(&t, r.a, r.b));
    ms += t;
    n++;
  }
  *h_count = n;
  *h_total_ms = ms;
  return GAIB_OK;
}

// the same with the launches' algorithmic work (the figures the sites state: SURVEY.md 8(d)) and the time the launches would
// take at the chip's roofs -- per launch max(bytes / 8 TB/s, flops / 157.3 TFLOP/s), summed
extern "C" int gaib_prof_get_work(gaib_ctx* ctx, const char* key, int64_t* h_count, double* h_total_ms, double* h_bytes,
                                  double* h_flops, double* h_roof_ms) {
  GAIB_CHECK(ctx && key && h_count && h_total_ms && h_bytes && h_flops && h_roof_ms, "gaib_prof_get_work: NULL argument");
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  int64_t n = 0;
  double ms = 0.0, by = 0.0, fl = 0.0, roof = 0.0;
  for (auto& r : ctx->prof) {
    if (strcmp(r.key, key) != 0) continue;
    float t = 0.f;
    GAIB_HIP(hipEventElapsedTime(&t, r.a, r.b));
    ms += t;
    by += r.bytes;
    fl += r.flops;
    roof += 1e3 * std::max(r.bytes / GAIB_HBM_PEAK_BPS, r.flops / GAIB_MFMA_F32_PEAK_FLOPS);
    n++;
  }
  *h_count = n;
  *h_total_ms = ms;
  *h_bytes = by;
  *h_flops = fl;
  *h_roof_ms = roof;
  return GAIB_OK;
}

// every key that has records, as text: one line "key[@tag] count total_ms alg_bytes flops roof_ms" per (key, shape tag: the row
// width of a gather kernel, M x N x K of a dense product), in order of first appearance.  Returns the number of bytes the whole table needs (incl. the terminating
// 0) through *h_needed; writes at most `cap` bytes.  (The trainer prints it on GAIB_PROF_TABLE; bench.py's epoch workloads
// parse it.)
extern "C" int gaib_prof_table(gaib_ctx* ctx, char* h_buf, size_t cap, size_t* h_needed) {
  GAIB_CHECK(ctx && h_needed && (h_buf || cap == 0), "gaib_prof_table: NULL argument");
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  struct Row { const char* key; const char* tag; int64_t n; double ms, by, fl, roof; };
  std::vector<Row> rows;
  for (auto& r : ctx->prof) {
    Row* at = nullptr;
    for (auto& q : rows)
      if (strcmp(q.tag, r.tag) == 0 && strcmp(q.key, r.key) == 0) at = &q;
    if (!at) {
      rows.push_back(Row{r.key, r.tag, 0, 0, 0, 0, 0});
      at = &rows.back();
    }
    float t = 0.f;
    GAIB_HIP(hipEventElapsedTime(&t, r.a, r.b));
    at->n++;
    at->ms += t;
    at->by += r.bytes;
    at->fl += r.flops;
    at->roof += 1e3 * std::max(r.bytes / GAIB_HBM_PEAK_BPS, r.flops / GAIB_MFMA_F32_PEAK_FLOPS);
  }
  std::string out;
  char line[320];
  for (auto& q : rows) {
    if (q.tag[0]) snprintf(line, sizeof(line), "%s@%s %lld %.6f %.0f %.0f %.9f\n", q.key, q.tag, (long long)q.n, q.ms, q.by, q.fl, q.roof);
    else snprintf(line, sizeof(line), "%s %lld %.6f %.0f %.0f %.9f\n", q.key, (long long)q.n, q.ms, q.by, q.fl, q.roof);
    out += line;
  }
  *h_needed = out.size() + 1;
  if (cap > 0) {
    const size_t w = out.size() < cap - 1 ? out.size() : cap - 1;
    memcpy(h_buf, out.data(), w);
    h_buf[w] = 0;
  }
  return GAIB_OK;
}

extern "C" int gaib_get_option(gaib_ctx* ctx, const char* key, int64_t* h_value) {
  GAIB_CHECK(ctx && key && h_value, "gaib_get_option: NULL argument");
  if (!strcmp(key, "comm_reserve_cus")) *h_value = gaib_comm_reserve(ctx);  // the EFFECTIVE figure
  else if (!strcmp(key, "comm_reserve_cus_raw")) *h_value = ctx->comm_reserve_cus;  // what the caller set (-1: unset)
  else if (!strcmp(key, "spmm_fuse_cus")) *h_value = ctx->spmm_fuse_cus;
  else if (!strcmp(key, "spmm_flat_ring")) *h_value = ctx->spmm_flat_ring;
  else if (!strcmp(key, "num_cus")) *h_value = ctx->num_cus;
  else {
    gaib_set_error("gaib_get_option: no readable option '%s'", key);
    return GAIB_ERR_INVALID;
  }
  return GAIB_OK;
}

extern "C" int gaib_set_option(gaib_ctx* ctx, const char* key, int64_t value) {
  GAIB_CHECK(ctx && key, "gaib_set_option: NULL argument");
  if (!strcmp(key, "spmm_heavy_threshold")) {
    GAIB_CHECK(value >= 1 && value <= (1 << 30), "spmm_heavy_threshold out of range");
    ctx->spmm_heavy_threshold = (int)value;
  } else if (!strcmp(key, "spmm_variant"))
    ctx->spmm_variant = (int)value;
  else if (!strcmp(key, "spmm_xcd_swizzle"))
    ctx->spmm_xcd_swizzle = (int)value;
  else if (!strcmp(key, "spmm_fuse"))
    ctx->spmm_fuse = (int)value;
  else if (!strcmp(key, "spmm_chunked"))
    ctx->spmm_chunked = (int)value;
  else if (!strcmp(key, "spmm_pad"))
    ctx->spmm_pad = (int)value;
  else if (!strcmp(key, "spmm_flat"))
    ctx->spmm_flat = (int)value;
  else if (!strcmp(key, "spmm_fuse_cus")) {
    GAIB_CHECK(value >= 0 && value <= ctx->num_cus, "spmm_fuse_cus: 0 (all) .. %d", ctx->num_cus);
    ctx->spmm_fuse_cus = (int)value;
  } else if (!strcmp(key, "spmm_tile_xcd"))
    ctx->spmm_tile_xcd = (int)value;
  else if (!strcmp(key, "spmm_prefetch_ids"))
    ctx->spmm_prefetch_ids = value != 0;
  else if (!strcmp(key, "spmm_unroll"))
    ctx->spmm_unroll = (int)value;
  else if (!strcmp(key, "spmm_addr_mode"))
    ctx->spmm_addr_mode = (int)value;
  else if (!strcmp(key, "spmm_gather_mode"))
    ctx->spmm_gather_mode = (int)value;
  else if (!strcmp(key, "spmm_hot_bytes"))
    ctx->spmm_hot_bytes = (int)value;
  else if (!strcmp(key, "sgemm_variant"))
    ctx->sgemm_variant = (int)value;
  else if (!strcmp(key, "gat_fast"))
    ctx->gat_fast = (int)value;
  else if (!strcmp(key, "gat_chunk_colsum"))
    ctx->gat_chunk_colsum = (int)value;
  else if (!strcmp(key, "gat_chunk_sort"))
    ctx->gat_chunk_sort = (int)value;
  else if (!strcmp(key, "gat_fused_fwd"))
    ctx->gat_fused_fwd = (int)value;
  else if (!strcmp(key, "gat_fused_unroll"))
    ctx->gat_fused_unroll = (int)value;
  else if (!strcmp(key, "comm_reserve_cus"))  // -1: back to "unset"; else clamped so that the fused kernel keeps >= 64 CUs
    ctx->comm_reserve_cus = value < 0 ? -1 : (int)std::min<int64_t>(value, std::max(0, ctx->num_cus - 64));
  else if (!strcmp(key, "spmm_flat_ring"))
    ctx->spmm_flat_ring = value < 0 ? GAIB_FLAT_RING_DEFAULT : (int)value;  // (-1: back to the default)
  else if (!strcmp(key, "gat_interleave"))
    ctx->gat_interleave = (int)value;
  else if (!strcmp(key, "gat_bwd_pk"))
    ctx->gat_bwd_pk = (int)value;
  else if (!strcmp(key, "gat_chunk_xcd"))
    ctx->gat_chunk_xcd = (int)value;
  else if (!strcmp(key, "gat_fused_bwd"))
    ctx->gat_fused_bwd = (int)value;
  else if (!strcmp(key, "graph_rev_search"))
    ctx->graph_rev_search = (int)value;
  else if (!strcmp(key, "gat_row_waves")) {
    GAIB_CHECK(value == 1 || value == 2 || value == 4, "gat_row_waves must be 1, 2 or 4");
    ctx->gat_row_waves = (int)value;
  }
  else {
    gaib_set_error("gaib_set_option: unknown key '%s'", key);
    return GAIB_ERR_INVALID;
  }
  return GAIB_OK;
}
