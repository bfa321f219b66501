// comm.hip -- the collectives of the partitioned path behind include/gaib.h:
// communicator, halo-row exchange (one all-to-all(v) of feature rows per aggregation), all-reduce of the
// weight gradients.  The reference has no multi-GPU GNN; its multi-GPU host pattern is one host thread per
// device + peer copies (src/triangle/multigpu_induced.cu:31-84) over the vertex-range partition of
// src/partitioner/graph_partition.cc:128-178.  Here: one PROCESS per GPU and two transports behind one interface:
//
//   GAIB_COMM_RCCL  ncclSend/ncclRecv groups + ncclAllReduce on a communication stream next to the compute
//                   stream (RCCL is dlopen'ed: the library has no link-time dependency on it, and inside a
//                   torch process the already loaded librccl.so.1 is the one that gets used).  Stream-ordered:
//                   the host never waits.
//   GAIB_COMM_IPC   peer-to-peer PULL: every rank publishes the hipIpc handle of its packed send buffer in a
//                   POSIX shared-memory segment; after a host barrier each rank copies the rows it needs straight
//                   out of its peers' send buffers (device-to-device; over xGMI between GPUs of one node).  Works
//                   for several ranks on ONE GPU as well (tests), needs no RCCL.  Host-synchronous at the two
//                   hand-over points of an exchange (after the pack, after the pull); the owned-edge aggregation
//                   enqueued in between still overlaps the copies.
//
// Failure behaviour: every wait on a peer has a deadline (GAIB_COMM_TIMEOUT_S, default 120 s) and a shared error
// flag; a rank that fails or times out raises the flag, every other rank returns GAIB_ERR_COMM from its next wait
// instead of spinning forever.  The C++ mirror turns that into the reference's print-and-exit.
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdlib.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <atomic>
#include <new>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>
#include <mutex>
#include <utility>
#include "common.h"

#define GAIB_COMM_MAX_RANKS 16
#define GAIB_COMM_MAX_HALOS 8
#define GAIB_HALO_MAX_PIECES 16  // time slices of one exchange (gaib_halo_set_pieces)
// elementwise.hip: the source-ordered pack into destination row ADDRESSES (a send buffer that is several allocations)
int gaib_gather_rows_to_addresses(gaib_ctx* ctx, int64_t n_idx, const int64_t* d_src_idx, const int64_t* d_dst_addr, int len,
                                  const float* d_in);
// IPC: a send buffer is cut into separately allocated (and separately exported) chunks -- hipIpcOpenMemHandle of an
// allocation above 2 GiB does not return on this runtime (measured: a 1.86 GB buffer opens at once, a 2.42 GB one hangs all
// ranks; bench.py --gpus 3 --cut-fraction 0.3 on one device) -- and a receiver opens only the chunks its segment touches
#define GAIB_IPC_MAX_CHUNKS 64
#define GAIB_COMM_REDUCE_FLOATS (64 * 1024)  // per-rank all-reduce staging slot in the shm segment (256 KB)

namespace {

struct RcclApi {
  void* dl = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
};
RcclApi g_rccl;

int rccl_load() {
  if (g_rccl.dl) return GAIB_OK;
  const char* cands[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  // GAIB_RCCL_LIB names THE library to bind (another RCCL build; tests/fake_rccl's strict double, which carries the
  // same calls between processes that share one GPU): no fallback to the system's when it does not load
  const char* named = getenv("GAIB_RCCL_LIB");
  if (named && *named) {
    h = dlopen(named, RTLD_NOW | RTLD_LOCAL);
  } else {
    for (const char* c : cands)
      if ((h = dlopen(c, RTLD_NOW | RTLD_LOCAL))) break;
  }
  if (!h) {
    gaib_set_error("gaib_comm: cannot dlopen %s (%s)", named && *named ? named : "librccl.so.1", dlerror());
    return GAIB_ERR_UNSUPPORTED;
  }
#define GAIB_SYM(field, name)                                        \
  g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);             \
  if (!g_rccl.field) {                                               \
    gaib_set_error("gaib_comm: librccl lacks %s", name);             \
    dlclose(h);                                                      \
    return GAIB_ERR_UNSUPPORTED;                                     \
  }
  GAIB_SYM(GetUniqueId, "ncclGetUniqueId")
  GAIB_SYM(CommInitRank, "ncclCommInitRank")
  GAIB_SYM(CommDestroy, "ncclCommDestroy")
  GAIB_SYM(CommCount, "ncclCommCount")
  GAIB_SYM(CommUserRank, "ncclCommUserRank")
  GAIB_SYM(GetErrorString, "ncclGetErrorString")
  GAIB_SYM(AllReduce, "ncclAllReduce")
  GAIB_SYM(Send, "ncclSend")
  GAIB_SYM(Recv, "ncclRecv")
  GAIB_SYM(GroupStart, "ncclGroupStart")
  GAIB_SYM(GroupEnd, "ncclGroupEnd")
#undef GAIB_SYM
  g_rccl.dl = h;
  return GAIB_OK;
}

#define GAIB_NCCL(call)                                                                                   \
  do {                                                                                                    \
    ncclResult_t r_ = (call);                                                                             \
    if (r_ != ncclSuccess) {                                                                              \
      gaib_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #call, g_rccl.GetErrorString(r_));           \
      return GAIB_ERR_COMM;                                                                               \
    }                                                                                                     \
  } while (0)

// ---- shared-memory segment of the IPC transport --------------------------------------------------------------
struct ShmSlot {  // one rank's published send buffer of one halo plan
  uint64_t gen;   // bumped whenever the buffer was (re)allocated: peers re-open the handle
  uint64_t capacity_bytes;
  int32_t device;
  int32_t n_pieces;  // time slices the owner cuts an exchange into (gaib_halo_set_pieces): the puller's must be the same
  hipIpcMemHandle_t handle;
  int64_t send_off[GAIB_COMM_MAX_RANKS + 1];  // row offsets of the per-destination groups inside the buffer
  // a send buffer above the chunk size: rows [j * chunk_rows, (j + 1) * chunk_rows) live in allocation j (handle = chunk 0)
  int32_t n_chunks;
  int32_t pad2;
  int64_t chunk_rows;
  hipIpcMemHandle_t handle_x[GAIB_IPC_MAX_CHUNKS - 1];
  // the reverse direction (gaib_halo_reduce): the rank's halo table, grouped by OWNER rank
  uint64_t gen_t;
  hipIpcMemHandle_t handle_t;
  int64_t recv_off[GAIB_COMM_MAX_RANKS + 1];
  // ... staged in chunks when it is above the chunk size (handle_t = chunk 0)
  int32_t n_chunks_t;
  int32_t pad3;
  int64_t chunk_rows_t;
  hipIpcMemHandle_t handle_tx[GAIB_IPC_MAX_CHUNKS - 1];
};
struct ShmSeg {
  std::atomic<uint32_t> magic;
  uint32_t nranks;
  std::atomic<uint32_t> bar_count;
  std::atomic<uint32_t> bar_sense;
  std::atomic<uint32_t> error;
  uint32_t pad[3];
  ShmSlot slot[GAIB_COMM_MAX_HALOS][GAIB_COMM_MAX_RANKS];
  float reduce[GAIB_COMM_MAX_RANKS][GAIB_COMM_REDUCE_FLOATS];
};
const uint32_t kMagic = 0x47414942u;  // "GAIB"

double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}
// GAIB_COMM_DEBUG=1: the steps of an exchange with time stamps on stderr (diagnosis of a stuck transport)
bool comm_debug() {
  static const bool on = getenv("GAIB_COMM_DEBUG") && atoi(getenv("GAIB_COMM_DEBUG")) != 0;
  return on;
}
#define GAIB_COMM_DBG(c, ...)                                        \
  do {                                                               \
    if (comm_debug()) {                                              \
      fprintf(stderr, "[gaib_comm r%d %.3f] ", (c)->rank, now_s()); \
      fprintf(stderr, __VA_ARGS__);                                  \
      fprintf(stderr, "\n");                                         \
      fflush(stderr);                                                \
    }                                                                \
  } while (0)
// size of one chunk of an IPC send buffer, and the largest allocation this transport will export at all
size_t ipc_chunk_bytes() {
  const char* e = getenv("GAIB_IPC_CHUNK_BYTES");  // (tests cut small buffers into many chunks)
  const long long v = e ? atoll(e) : 0;
  return v > 0 ? (size_t)v : (size_t)512 << 20;
}
size_t ipc_export_limit();
// chunk size for `total` bytes: the configured one, larger where GAIB_IPC_MAX_CHUNKS of them would not hold the buffer (config 5 in a
// random vertex order sends 41 GB per rank) -- up to the export limit
size_t ipc_chunk_bytes_for(size_t total) {
  size_t c = ipc_chunk_bytes();
  const size_t need = (total + GAIB_IPC_MAX_CHUNKS - 1) / GAIB_IPC_MAX_CHUNKS;
  if (need > c) c = std::min(ipc_export_limit(), (need + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1));
  return c;
}
size_t ipc_export_limit() {
  const char* e = getenv("GAIB_IPC_EXPORT_LIMIT_BYTES");
  const long long v = e ? atoll(e) : 0;
  return v > 0 ? (size_t)v : (size_t)1536 << 20;
}
double timeout_s() {
  const char* e = getenv("GAIB_COMM_TIMEOUT_S");
  double v = e ? atof(e) : 120.0;
  return v > 0 ? v : 120.0;
}

}  // namespace

struct gaib_comm {
  gaib_ctx* ctx;
  int rank, nranks, transport;
  hipStream_t cstream;  // communication stream
  hipEvent_t ev_ready, ev_done;
  ncclComm_t nccl;
  // IPC transport
  ShmSeg* seg;
  char shm_name[80];
  uint32_t sense;
  uint32_t halo_slots;  // bitmap of the slot rows in use (plans are created and destroyed collectively: same bits on every rank)
  float* h_stage;  // pinned staging for the IPC all-reduce
  // IPC: every buffer this communicator ever exported.  hipFree does not give an exported buffer's memory back to the
  // device (measured: 8 x {hipMalloc 2 MiB, hipIpcGetMemHandle, hipFree} = 16 MiB less free memory, also when no peer ever
  // opened the handle; scripts/ipc_leak_probe.py), so a process that builds and drops partitions would lose its halo
  // buffers each time.  They are therefore pooled: a plan takes its send buffer / halo table from here (best fit), hands
  // them back when it is destroyed (after the peers closed their mappings and a barrier), the handle of a buffer is
  // exported once, and hipFree is left to gaib_comm_destroy.  The RCCL transport draws from the same pool (no export
  // there; it saves the synchronising hipMalloc / hipFree pair when partitions are rebuilt, e.g. per sampled subgraph).
  struct IpcBuf {
    void* p;
    size_t cap;
    hipIpcMemHandle_t handle;
    bool exported, in_use;
  };
  std::vector<IpcBuf>* ipc_bufs;  // (a pointer: the struct is zero-filled after construction; EVERY buffer is in it -- a
                                  // buffer outside the pool would be hipFree'd while exported, the leak the pool exists against)
  ~gaib_comm() { delete ipc_bufs; }
};

struct gaib_halo {
  gaib_comm* c;
  int id;  // slot row in the shm segment
  int64_t send_counts[GAIB_COMM_MAX_RANKS], recv_counts[GAIB_COMM_MAX_RANKS];
  int64_t send_off[GAIB_COMM_MAX_RANKS + 1], recv_off[GAIB_COMM_MAX_RANKS + 1];
  int64_t* d_send_idx;
  // the pack in SOURCE order: the (row, slot) pairs of the send list sorted by row.  A row that several peers list
  // (2.9 x on average at 8 ranks, 6.6 x on a random vertex order) is then read from HBM once, its repeats hit the cache:
  // 1.26 -> 0.83 ms and 3.11 -> 1.79 ms for the two ends of the bench's partition axis (scripts/ab_pack.py)
  int64_t *d_pack_row, *d_pack_slot;
  int64_t* d_pack_addr;  // chunked send buffer: the address of every pair's destination row (rebuilt per exchange: 20 us)
  float* sendbuf;
  size_t send_cap;
  float* table;
  size_t table_cap;
  // IPC, send buffers above the chunk size: chunk 0 is sendbuf, chunks 1.. live here (every chunk its own allocation)
  float* send_x[GAIB_IPC_MAX_CHUNKS - 1];
  size_t send_x_cap[GAIB_IPC_MAX_CHUNKS - 1];
  uint64_t send_x_serial[GAIB_IPC_MAX_CHUNKS - 1], pub_x_serial[GAIB_IPC_MAX_CHUNKS - 1];
  int pub_n_chunks;
  int64_t pub_chunk_rows;
  // gaib_halo_reduce lands what arrives here (never exported; the send buffer stays what an exchange made it)
  float* landing;
  size_t landing_cap;
  uint64_t landing_serial;
  // IPC, reverse direction with a halo table above the chunk size: the partial rows are staged in chunks of their own
  // (the table itself is one allocation by contract and would be exported whole)
  float* stage_x[GAIB_IPC_MAX_CHUNKS];
  size_t stage_x_cap[GAIB_IPC_MAX_CHUNKS];
  uint64_t stage_x_serial[GAIB_IPC_MAX_CHUNKS], pub_stage_serial[GAIB_IPC_MAX_CHUNKS];
  int pub_n_chunks_t;
  int64_t pub_chunk_rows_t;
  // IPC: which ALLOCATION of each buffer the peers hold a handle of.  Either entry point may reallocate either buffer
  // (an exchange reserves the table, a reduce may stage in it; chunks come and go with the row length), so "did MY reserve() reallocate"
  // is not the question -- "is the published allocation still the current one" is.
  uint64_t send_serial, table_serial;          // bumped by every (re)allocation
  uint64_t pub_send_serial, pub_table_serial;  // allocation the handle in the segment belongs to (0 = none)
  // IPC: a buffer that has to grow is RETIRED, not freed, until the plan is destroyed.  Peers hold a mapping of it
  // (hipIpcOpenMemHandle) and close that mapping only when they meet the new handle at the next hand-over; freeing the
  // memory first made their hipIpcCloseMemHandle act on memory that no longer existed, and -- depending on where the next
  // allocations landed -- later pulls through the re-opened mapping read wrong rows at non-zero offsets (found by
  // test_ipc_halo_buffers_regrow_between_exchange_and_reduce[3] when plan creation began to allocate scratch of its own).
  // gaib_halo_destroy closes every mapping on every rank, passes a barrier, and only then frees.
  std::vector<void*> retired;  // (growable: a full list must never push a possibly-mapped buffer back into the pool early)
  int pending_len;
  struct Peer {
    uint64_t gen;
    void* base;
    void* base_x[GAIB_IPC_MAX_CHUNKS - 1];  // chunks 1.. of the peer's send buffer, opened when first needed
    uint64_t gen_t;  // the peer's halo table (reverse direction)
    void* base_t;
    void* base_tx[GAIB_IPC_MAX_CHUNKS - 1];
  } peer[GAIB_COMM_MAX_RANKS];
  int64_t bytes_sent;
  // RCCL, round 5: a peer whose send list is one run of consecutive rows (first, first + 1, ...) -- every peer of a partition
  // whose ranges need (nearly) all of each other's rows, the N-way cut of a graph on a random numbering: dist.py then asks for
  // the whole range -- is sent straight from the caller's matrix; if EVERY peer is, the plan never packs and needs no send
  // buffer (the pack was 0.43 of the 3.6 ms of rank 0's step at N = 8, profiles/r05/shard/).  -1: packed as before.
  int64_t direct_first[GAIB_COMM_MAX_RANKS];
  int all_direct;  // every peer with rows to send is direct (and there is one)
  int64_t packs, direct_sends;  // exchanges that ran the pack kernel / sends that went straight from the caller's matrix
  // round 6: an exchange in K time slices ("pieces").  Slice k of a peer pair's R rows is rows [R k / K, R (k + 1) / K) of that
  // pair's segment -- sender and receiver cut the same R the same way -- and slice k of EVERY pair travels together (one
  // ncclGroup / one round of pulls), so every link is busy all the time and piece k has landed after ~ (k + 1) / K of the
  // exchange.  ev_piece[k] is recorded on the communication stream behind slice k: the caller's halo-column pass over piece
  // k's columns (gaib_halo_exchange_wait_piece) runs while slices k + 1 ... are still on the wire.  The table's layout does
  // not change (grouped by source rank): a piece is up to nranks - 1 column ranges of it (gaib_halo_piece_ranges).
  int n_pieces;
  hipEvent_t ev_piece[GAIB_HALO_MAX_PIECES];
  int64_t piece_waits;
};

// rows [lo, hi) of a peer pair's R-row segment that travel in slice k of K
static inline void piece_slice(int64_t rows, int n_pieces, int piece, int64_t* lo, int64_t* hi) {
  *lo = rows * piece / n_pieces;
  *hi = rows * (piece + 1) / n_pieces;
}

namespace {

// central sense-reversing barrier in the segment, with deadline and error flag
int shm_barrier(gaib_comm* c, const char* what) {
  ShmSeg* s = c->seg;
  if (c->nranks == 1) return GAIB_OK;
  const uint32_t my = (c->sense ^= 1u);
  if (s->bar_count.fetch_add(1) + 1 == (uint32_t)c->nranks) {
    s->bar_count.store(0);
    s->bar_sense.store(my);
    return GAIB_OK;
  }
  const double deadline = now_s() + timeout_s();
  unsigned spins = 0;
  while (s->bar_sense.load() != my) {
    if (s->error.load()) {
      gaib_set_error("gaib_comm(rank %d): a peer reported a failure while this rank waited in %s", c->rank, what);
      return GAIB_ERR_COMM;
    }
    if ((++spins & 1023u) == 0) {
      if (now_s() > deadline) {
        s->error.store(1);
        gaib_set_error("gaib_comm(rank %d): timed out after %.0f s waiting for the other ranks in %s", c->rank,
                       timeout_s(), what);
        return GAIB_ERR_COMM;
      }
      usleep(50);
    }
  }
  return GAIB_OK;
}

// addr[k] = address of row slot[k] of a send buffer cut into chunks of chunk_rows rows
struct ChunkBases {
  float* p[GAIB_IPC_MAX_CHUNKS];
};
__global__ void chunk_row_address_kernel(int64_t n, const int64_t* slot, ChunkBases b, int64_t chunk_rows, int64_t row_floats,
                                         int64_t* addr) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int64_t s = slot[k], j = s / chunk_rows;
  addr[k] = (int64_t)(uintptr_t)(b.p[j] + (s - j * chunk_rows) * row_floats);
}

__global__ void iota_i64_kernel(int64_t n, int64_t* x) {
  const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t < n) x[t] = t;
}

int fail(gaib_comm* c, int rc) {  // tell the peers, keep the message
  if (c && c->seg) c->seg->error.store(1);
  return rc;
}

// IPC buffers come out of (and go back into) the communicator's pool -- see gaib_comm::ipc_bufs
gaib_comm::IpcBuf* pool_find(gaib_comm* c, const void* p) {
  for (gaib_comm::IpcBuf& b : *c->ipc_bufs)
    if (b.p == p) return &b;
  return nullptr;
}
void pool_release(gaib_comm* c, void* p) {
  if (!p) return;
  gaib_comm::IpcBuf* b = pool_find(c, p);
  if (b) b->in_use = false;  // (every buffer reserve() hands out is in the pool; hipFree is left to gaib_comm_destroy)
}
// the handle of a pooled buffer: exported once
hipError_t pool_handle(gaib_comm* c, void* p, hipIpcMemHandle_t* out) {
  gaib_comm::IpcBuf* b = pool_find(c, p);
  if (b && b->exported) {
    *out = b->handle;
    return hipSuccess;
  }
  hipError_t e = hipIpcGetMemHandle(out, p);
  if (e == hipSuccess && b) {
    b->handle = *out;
    b->exported = true;
  }
  return e;
}

// max_cap: never hand out a pooled buffer larger than this (buffers that will be exported: the IPC export limit)
int reserve(float** p, size_t* cap, uint64_t* serial, size_t bytes, hipStream_t s, gaib_halo* keep = nullptr,
            size_t max_cap = ~(size_t)0) {
  if (bytes <= *cap && *p) return 0;
  if (*p) {
    GAIB_HIP(hipStreamSynchronize(s));
    if (keep && keep->c->transport == GAIB_COMM_IPC)
      keep->retired.push_back(*p);  // IPC: peers may still have it mapped (see gaib_halo::retired)
    else if (keep)
      pool_release(keep->c, *p);
    else
      GAIB_HIP(hipFree(*p));
    *p = nullptr;
    *cap = 0;
  }
  // whole 2-MiB pieces: these buffers are exported through hipIpcGetMemHandle, and a small allocation can share its
  // backing object with other small allocations of the process -- exporting a second one out of the same object failed
  // with "invalid argument" (and, before buffers were retired, handed peers a mapping that read wrong rows)
  const size_t gran = (size_t)2 << 20;
  size_t want = ((bytes < 1 ? 1 : bytes) + gran - 1) / gran * gran;
  if (keep) {  // best fit among the pooled buffers nobody uses
    gaib_comm* c = keep->c;
    gaib_comm::IpcBuf* best = nullptr;
    for (gaib_comm::IpcBuf& b : *c->ipc_bufs)
      if (!b.in_use && b.cap >= want && b.cap <= max_cap && (!best || b.cap < best->cap)) best = &b;
    if (best && best->cap <= 2 * want) {  // (not a buffer far larger than asked for: the next large plan wants it)
      best->in_use = true;
      *p = (float*)best->p;
      *cap = best->cap;
      ++*serial;
      return 1;
    }
  }
  hipError_t e = hipMalloc((void**)p, want);
  if (e != hipSuccess) {
    gaib_set_error("gaib_halo: hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
    return GAIB_ERR_NOMEM;
  }
  if (keep) {
    gaib_comm::IpcBuf b;
    memset((void*)&b, 0, sizeof(b));
    b.p = *p;
    b.cap = want;
    b.in_use = true;
    keep->c->ipc_bufs->push_back(b);
  }
  *cap = want;
  ++*serial;
  return 1;  // (re)allocated
}

}  // namespace

// can this process use the transport at all?  (RCCL: the library loads and has every entry point.)  Cheap and local:
// what ranks agree on BEFORE anybody enters the collective, deadline-less ncclCommInitRank.
extern "C" int gaib_comm_transport_available(int transport) {
  if (transport == GAIB_COMM_IPC) return GAIB_OK;
  GAIB_CHECK(transport == GAIB_COMM_RCCL, "gaib_comm_transport_available: unknown transport %d", transport);
  return rccl_load();
}

extern "C" int gaib_comm_unique_id(int transport, void* h_id) {
  GAIB_CHECK(h_id, "gaib_comm_unique_id: h_id is NULL");
  memset(h_id, 0, GAIB_COMM_ID_BYTES);
  if (transport == GAIB_COMM_RCCL) {
    GAIB_TRY(rccl_load());
    ncclUniqueId id;
    GAIB_NCCL(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == GAIB_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(h_id, &id, sizeof(id));
    return GAIB_OK;
  }
  GAIB_CHECK(transport == GAIB_COMM_IPC, "gaib_comm_unique_id: unknown transport %d", transport);
  int fd = open("/dev/urandom", O_RDONLY);
  GAIB_CHECK(fd >= 0, "gaib_comm_unique_id: /dev/urandom: %s", strerror(errno));
  ssize_t n = read(fd, h_id, 16);
  close(fd);
  GAIB_CHECK(n == 16, "gaib_comm_unique_id: short read from /dev/urandom");
  return GAIB_OK;
}

// RCCL communicators of more than one rank alive per context: while there is one, the fused aggregation leaves CUs to RCCL's
// kernels by default (ctx->comm_reserve_default = 32); a peer-to-peer pull communicator created NEXT to it (bench.py's transport
// A/B) does not change that, and the default returns to 0 when the last of them is destroyed.  (Kept here, not in gaib_ctx.)
static int rccl_alive(gaib_ctx* ctx, int delta) {
  static std::mutex mu;
  static std::vector<std::pair<gaib_ctx*, int>> alive;
  std::lock_guard<std::mutex> lock(mu);
  for (auto& e : alive)
    if (e.first == ctx) {
      e.second = e.second + delta > 0 ? e.second + delta : 0;
      return e.second;
    }
  alive.push_back({ctx, delta > 0 ? delta : 0});
  return alive.back().second;
}

// GAIB_COMM_RESERVE_CUS: the caller's choice from the environment (validated: a whole number >= 0; anything else is ignored
// with a line on stderr); an option set before gaib_comm_init wins
static void reserve_from_env(gaib_ctx* ctx) {
  const char* e = getenv("GAIB_COMM_RESERVE_CUS");
  if (!e || ctx->comm_reserve_cus >= 0) return;
  char* end = nullptr;
  const long v = strtol(e, &end, 10);
  if (end == e || *end != '\0' || v < 0 || v > 4096) {
    fprintf(stderr, "[gaib] GAIB_COMM_RESERVE_CUS='%s' ignored (want a whole number >= 0)\n", e);
    return;
  }
  ctx->comm_reserve_cus = (int)v;  // (clamped where it is used: gaib_comm_reserve)
}

extern "C" int gaib_comm_init(gaib_ctx* ctx, int rank, int nranks, const void* h_id, int transport, gaib_comm** out) {
  GAIB_CHECK(ctx && h_id && out, "gaib_comm_init: NULL argument");
  GAIB_CHECK(nranks >= 1 && nranks <= GAIB_COMM_MAX_RANKS && rank >= 0 && rank < nranks,
             "gaib_comm_init: rank %d of %d (at most %d ranks)", rank, nranks, GAIB_COMM_MAX_RANKS);
  GAIB_CHECK(transport == GAIB_COMM_RCCL || transport == GAIB_COMM_IPC, "gaib_comm_init: unknown transport %d", transport);
  GAIB_HIP(hipSetDevice(ctx->device));
  gaib_comm* c = new (std::nothrow) gaib_comm();
  GAIB_CHECK(c, "gaib_comm_init: out of memory");
  memset((void*)c, 0, sizeof(*c));
  c->ipc_bufs = new std::vector<gaib_comm::IpcBuf>();
  c->ctx = ctx;
  c->rank = rank;
  c->nranks = nranks;
  c->transport = transport;
  hipError_t e = hipStreamCreateWithFlags(&c->cstream, hipStreamNonBlocking);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_ready, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_done, hipEventDisableTiming);
  if (e != hipSuccess) {
    gaib_set_error("gaib_comm_init: %s", hipGetErrorString(e));
    delete c;
    return GAIB_ERR_HIP;
  }
  if (transport == GAIB_COMM_RCCL) {
    int rc = rccl_load();
    if (rc != GAIB_OK) {
      delete c;
      return rc;
    }
    ncclUniqueId id;
    memcpy(&id, h_id, sizeof(id));
    // one GPU per rank: RCCL itself refuses two ranks on one device ("duplicate GPU")
    ncclResult_t r = g_rccl.CommInitRank(&c->nccl, nranks, id, rank);
    if (r != ncclSuccess) {
      gaib_set_error("gaib_comm_init: ncclCommInitRank(rank %d of %d, device %d) -> %s", rank, nranks, ctx->device,
                     g_rccl.GetErrorString(r));
      delete c;
      return GAIB_ERR_COMM;
    }
    // what RCCL itself says it built: gaib_comm_size reports THIS (bench.py's config.rccl_ranks), not the argument
    int cnt = -1, me = -1;
    if (g_rccl.CommCount(c->nccl, &cnt) != ncclSuccess || g_rccl.CommUserRank(c->nccl, &me) != ncclSuccess || cnt != nranks ||
        me != rank) {
      gaib_set_error("gaib_comm_init: RCCL built a communicator of %d ranks (this one: %d), asked for rank %d of %d", cnt, me,
                     rank, nranks);
      (void)g_rccl.CommDestroy(c->nccl);
      delete c;
      return GAIB_ERR_COMM;
    }
    c->nranks = cnt;
    // RCCL's send / recv kernels need CUs to land on while the persistent fused aggregation runs (GAIB_OVERLAPS_TRANSFER):
    // one eighth of the chip, measured to cost that kernel 1.5 %; GAIB_COMM_RESERVE_CUS / the option override
    if (cnt > 1) {
      rccl_alive(ctx, +1);
      ctx->comm_reserve_default = 32;
    }
    reserve_from_env(ctx);
    *out = c;
    return GAIB_OK;
  }
  reserve_from_env(ctx);  // (the peer-to-peer pull runs on copy engines: it asks for no CUs and leaves the default alone)
  // ---- IPC: map (rank 0: create) the segment named after the id ----
  const unsigned char* b = (const unsigned char*)h_id;
  snprintf(c->shm_name, sizeof(c->shm_name), "/gaib_%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x", b[0], b[1], b[2],
           b[3], b[4], b[5], b[6], b[7], b[8], b[9], b[10], b[11]);
  int fd = -1;
  const double deadline = now_s() + timeout_s();
  if (rank == 0) {
    shm_unlink(c->shm_name);
    fd = shm_open(c->shm_name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd >= 0 && ftruncate(fd, sizeof(ShmSeg)) != 0) {
      close(fd);
      fd = -1;
    }
  } else {
    while ((fd = shm_open(c->shm_name, O_RDWR, 0600)) < 0 && now_s() < deadline) usleep(1000);
    struct stat st;
    while (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size < sizeof(ShmSeg) && now_s() < deadline) usleep(1000);
  }
  if (fd < 0) {
    gaib_set_error("gaib_comm_init(rank %d): shm_open(%s): %s", rank, c->shm_name, strerror(errno));
    delete c;
    return GAIB_ERR_COMM;
  }
  void* m = mmap(nullptr, sizeof(ShmSeg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    gaib_set_error("gaib_comm_init(rank %d): mmap(%s): %s", rank, c->shm_name, strerror(errno));
    delete c;
    return GAIB_ERR_COMM;
  }
  c->seg = (ShmSeg*)m;
  if (rank == 0) {
    c->seg->nranks = (uint32_t)nranks;
    c->seg->bar_count.store(0);
    c->seg->bar_sense.store(0);
    c->seg->error.store(0);
    c->seg->magic.store(kMagic);
  } else {
    while (c->seg->magic.load() != kMagic && now_s() < deadline) usleep(1000);
    if (c->seg->magic.load() != kMagic || c->seg->nranks != (uint32_t)nranks) {
      gaib_set_error("gaib_comm_init(rank %d): segment %s not initialised by rank 0 for %d ranks", rank, c->shm_name, nranks);
      munmap(m, sizeof(ShmSeg));
      delete c;
      return GAIB_ERR_COMM;
    }
  }
  e = hipHostMalloc((void**)&c->h_stage, sizeof(float) * GAIB_COMM_REDUCE_FLOATS, hipHostMallocDefault);
  if (e != hipSuccess) {
    gaib_set_error("gaib_comm_init: hipHostMalloc: %s", hipGetErrorString(e));
    munmap(m, sizeof(ShmSeg));
    delete c;
    return GAIB_ERR_HIP;
  }
  int rc = shm_barrier(c, "gaib_comm_init");
  if (rc != GAIB_OK) return fail(c, rc);
  if (rank == 0) shm_unlink(c->shm_name);  // everybody has it mapped: the name can go (nothing left behind on a crash)
  *out = c;
  return GAIB_OK;
}

extern "C" int gaib_comm_rank(const gaib_comm* c) { return c ? c->rank : -1; }
extern "C" int gaib_comm_size(const gaib_comm* c) { return c ? c->nranks : 0; }

extern "C" int gaib_comm_destroy(gaib_comm* c) {
  if (!c) return GAIB_OK;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->cstream);
  if (c->transport == GAIB_COMM_RCCL && c->nccl) (void)g_rccl.CommDestroy(c->nccl);
  if (c->transport == GAIB_COMM_RCCL && c->nranks > 1 && c->nccl && rccl_alive(c->ctx, -1) == 0)
    c->ctx->comm_reserve_default = 0;  // its send / recv kernels are gone with the last such communicator
  if (c->h_stage) (void)hipHostFree(c->h_stage);
  for (gaib_comm::IpcBuf& b : *c->ipc_bufs) (void)hipFree(b.p);  // (every plan is gone: gaib_halo_destroy comes first)
  if (c->seg) munmap(c->seg, sizeof(ShmSeg));
  (void)hipEventDestroy(c->ev_ready);
  (void)hipEventDestroy(c->ev_done);
  (void)hipStreamDestroy(c->cstream);
  delete c;
  return GAIB_OK;
}

extern "C" int gaib_allreduce_f32(gaib_comm* c, float* d_buf, int64_t n);

extern "C" int gaib_comm_barrier(gaib_comm* c) {
  GAIB_CHECK(c, "gaib_comm_barrier: comm is NULL");
  GAIB_HIP(hipStreamSynchronize(c->ctx->stream));
  if (c->transport == GAIB_COMM_IPC) return shm_barrier(c, "gaib_comm_barrier");
  // RCCL: a 1-element all-reduce, issued like every other collective of this communicator (on its communication stream)
  GAIB_TRY(gaib_ws_reserve(c->ctx, 256));
  GAIB_HIP(hipMemsetAsync(c->ctx->ws, 0, sizeof(float), c->ctx->stream));
  GAIB_TRY(gaib_allreduce_f32(c, (float*)c->ctx->ws, 1));
  GAIB_HIP(hipStreamSynchronize(c->ctx->stream));
  return GAIB_OK;
}

// weight gradients: in-place sum over ranks, identical bits on every rank (IPC: slots added in rank order)
extern "C" int gaib_allreduce_f32(gaib_comm* c, float* d_buf, int64_t n) {
  GAIB_CHECK(c && (d_buf || n == 0), "gaib_allreduce_f32: NULL argument");
  GAIB_CHECK(n >= 0, "gaib_allreduce_f32: n < 0");
  if (n == 0 || (c->nranks == 1 && c->transport == GAIB_COMM_IPC)) return GAIB_OK;  // RCCL runs also with one rank
  GAIB_HIP(hipSetDevice(c->ctx->device));
  hipStream_t s = c->ctx->stream;
  if (c->transport == GAIB_COMM_RCCL) {
    // the communicator is driven from ONE stream (the communication stream); the compute stream hands over and
    // takes back through events, the host does not wait
    GAIB_HIP(hipEventRecord(c->ev_ready, s));
    GAIB_HIP(hipStreamWaitEvent(c->cstream, c->ev_ready, 0));
    GAIB_NCCL(g_rccl.AllReduce(d_buf, d_buf, (size_t)n, ncclFloat32, ncclSum, c->nccl, c->cstream));
    GAIB_HIP(hipEventRecord(c->ev_done, c->cstream));
    GAIB_HIP(hipStreamWaitEvent(s, c->ev_done, 0));
    return GAIB_OK;
  }
  ShmSeg* seg = c->seg;
  for (int64_t off = 0; off < n; off += GAIB_COMM_REDUCE_FLOATS) {
    const int64_t cnt = n - off < GAIB_COMM_REDUCE_FLOATS ? n - off : GAIB_COMM_REDUCE_FLOATS;
    hipError_t e = hipMemcpyAsync(c->h_stage, d_buf + off, sizeof(float) * cnt, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      gaib_set_error("gaib_allreduce_f32: %s", hipGetErrorString(e));
      return fail(c, GAIB_ERR_HIP);
    }
    memcpy(seg->reduce[c->rank], c->h_stage, sizeof(float) * cnt);
    int rc = shm_barrier(c, "gaib_allreduce_f32 (publish)");
    if (rc != GAIB_OK) return rc;
    for (int64_t i = 0; i < cnt; i++) {
      float acc = seg->reduce[0][i];
      for (int r = 1; r < c->nranks; r++) acc += seg->reduce[r][i];
      c->h_stage[i] = acc;
    }
    rc = shm_barrier(c, "gaib_allreduce_f32 (consume)");  // nobody overwrites a slot that is still being read
    if (rc != GAIB_OK) return rc;
    e = hipMemcpyAsync(d_buf + off, c->h_stage, sizeof(float) * cnt, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) {
      gaib_set_error("gaib_allreduce_f32: %s", hipGetErrorString(e));
      return fail(c, GAIB_ERR_HIP);
    }
  }
  return GAIB_OK;
}

// scalars that live on the host (loss, accuracy counts): sum over ranks
extern "C" int gaib_allreduce_host_f64(gaib_comm* c, double* h_buf, int n) {
  GAIB_CHECK(c && h_buf && n >= 0 && n <= 1024, "gaib_allreduce_host_f64: bad argument (n <= 1024)");
  if (n == 0 || c->nranks == 1) return GAIB_OK;
  if (c->transport == GAIB_COMM_IPC) {
    double* slot = (double*)c->seg->reduce[c->rank];
    memcpy(slot, h_buf, sizeof(double) * n);
    int rc = shm_barrier(c, "gaib_allreduce_host_f64 (publish)");
    if (rc != GAIB_OK) return rc;
    for (int i = 0; i < n; i++) {
      double acc = 0.0;
      for (int r = 0; r < c->nranks; r++) acc += ((const double*)c->seg->reduce[r])[i];
      h_buf[i] = acc;
    }
    return shm_barrier(c, "gaib_allreduce_host_f64 (consume)");
  }
  // RCCL: through a small device buffer as 2 floats per double would lose bits; use ncclFloat64
  GAIB_HIP(hipSetDevice(c->ctx->device));
  GAIB_TRY(gaib_ws_reserve(c->ctx, sizeof(double) * 1024));
  double* d = (double*)c->ctx->ws;
  hipStream_t s = c->ctx->stream;
  GAIB_HIP(hipMemcpyAsync(d, h_buf, sizeof(double) * n, hipMemcpyHostToDevice, s));
  GAIB_HIP(hipEventRecord(c->ev_ready, s));
  GAIB_HIP(hipStreamWaitEvent(c->cstream, c->ev_ready, 0));
  GAIB_NCCL(g_rccl.AllReduce(d, d, (size_t)n, ncclFloat64, ncclSum, c->nccl, c->cstream));
  GAIB_HIP(hipEventRecord(c->ev_done, c->cstream));
  GAIB_HIP(hipStreamWaitEvent(s, c->ev_done, 0));
  GAIB_HIP(hipMemcpyAsync(h_buf, d, sizeof(double) * n, hipMemcpyDeviceToHost, s));
  GAIB_HIP(hipStreamSynchronize(s));
  return GAIB_OK;
}

// ---- halo plan --------------------------------------------------------------------------------------------------
extern "C" int gaib_halo_create(gaib_comm* c, const int64_t* h_send_counts, const int64_t* send_idx, int idx_on_device,
                                const int64_t* h_recv_counts, gaib_halo** out) {
  GAIB_CHECK(c && h_send_counts && h_recv_counts && out, "gaib_halo_create: NULL argument");
  int slot = 0;
  while (slot < GAIB_COMM_MAX_HALOS && (c->halo_slots >> slot & 1u)) slot++;
  GAIB_CHECK(slot < GAIB_COMM_MAX_HALOS, "gaib_halo_create: at most %d halo plans alive per communicator", GAIB_COMM_MAX_HALOS);
  GAIB_HIP(hipSetDevice(c->ctx->device));
  gaib_halo* h = new (std::nothrow) gaib_halo();
  GAIB_CHECK(h, "gaib_halo_create: out of memory");  // (value-initialised: every plain member is zero, `retired` is empty)
  h->c = c;
  h->n_pieces = 1;
  h->id = slot;  // taken (bit set) only once every argument check has passed
  h->send_off[0] = h->recv_off[0] = 0;
  for (int r = 0; r < c->nranks; r++) {
    if (h_send_counts[r] < 0 || h_recv_counts[r] < 0 || (r == c->rank && (h_send_counts[r] || h_recv_counts[r]))) {
      gaib_set_error("gaib_halo_create: counts must be >= 0 and 0 for the rank itself (rank %d, peer %d)", c->rank, r);
      delete h;
      return GAIB_ERR_INVALID;
    }
    h->send_counts[r] = h_send_counts[r];
    h->recv_counts[r] = h_recv_counts[r];
    h->send_off[r + 1] = h->send_off[r] + h_send_counts[r];
    h->recv_off[r + 1] = h->recv_off[r] + h_recv_counts[r];
  }
  const int64_t n_send = h->send_off[c->nranks];
  for (int r = 0; r < GAIB_COMM_MAX_RANKS; r++) h->direct_first[r] = -1;
  GAIB_CHECK(n_send == 0 || send_idx, "gaib_halo_create: send_idx is NULL");
  if (n_send) {
    hipError_t e = hipMalloc((void**)&h->d_send_idx, sizeof(int64_t) * n_send);
    if (e == hipSuccess)
      e = hipMemcpyAsync(h->d_send_idx, send_idx, sizeof(int64_t) * n_send,
                         idx_on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, c->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->ctx->stream);
    if (e != hipSuccess) {
      gaib_set_error("gaib_halo_create: %s", hipGetErrorString(e));
      if (h->d_send_idx) (void)hipFree(h->d_send_idx);
      delete h;
      return GAIB_ERR_HIP;
    }
    // which peers' lists are one run of consecutive rows (checked on the host, once)
    {
      std::vector<int64_t> hidx((size_t)n_send);
      e = hipMemcpy(hidx.data(), h->d_send_idx, sizeof(int64_t) * n_send, hipMemcpyDeviceToHost);
      int any = 0, all = 1;
      for (int r = 0; r < c->nranks; r++) {
        h->direct_first[r] = -1;
        const int64_t n_r = h->send_counts[r], o = h->send_off[r];
        if (e != hipSuccess || n_r == 0 || c->transport != GAIB_COMM_RCCL || getenv("GAIB_NO_DIRECT_SEND")) {
          if (n_r) all = 0;
          continue;
        }
        bool run = true;
        for (int64_t k = 1; k < n_r && run; ++k) run = hidx[o + k] == hidx[o] + k;
        if (run && hidx[o] >= 0) {
          h->direct_first[r] = hidx[o];
          any = 1;
        } else {
          all = 0;
        }
      }
      h->all_direct = any && all;
      (void)hipGetLastError();
    }
    // (row, slot) pairs sorted by row (stable radix sort of the row ids with the slot numbers as values); if anything
    // here fails the plan packs in destination order, as before
    int64_t *iota = nullptr, *srow = nullptr, *sslot = nullptr;
    void* tmp = nullptr;
    size_t tmp_bytes = 0;
    e = hipMalloc((void**)&iota, sizeof(int64_t) * n_send);
    if (e == hipSuccess) e = hipMalloc((void**)&srow, sizeof(int64_t) * n_send);
    if (e == hipSuccess) e = hipMalloc((void**)&sslot, sizeof(int64_t) * n_send);
    if (e == hipSuccess) {
      iota_i64_kernel<<<(unsigned)cdiv64(n_send, 256), 256, 0, c->ctx->stream>>>(n_send, iota);
      e = hipGetLastError();
    }
    if (e == hipSuccess)
      e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, h->d_send_idx, srow, iota, sslot, n_send, 0, 48, c->ctx->stream);
    if (e == hipSuccess) e = hipMalloc(&tmp, tmp_bytes ? tmp_bytes : 16);
    if (e == hipSuccess)
      e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, h->d_send_idx, srow, iota, sslot, n_send, 0, 48, c->ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->ctx->stream);
    if (tmp) (void)hipFree(tmp);
    if (iota) (void)hipFree(iota);
    if (e == hipSuccess) {
      h->d_pack_row = srow;
      h->d_pack_slot = sslot;
    } else {
      (void)hipGetLastError();
      if (srow) (void)hipFree(srow);
      if (sslot) (void)hipFree(sslot);
    }
  }
  c->halo_slots |= 1u << slot;
  *out = h;
  return GAIB_OK;
}

extern "C" int gaib_halo_destroy(gaib_halo* h) {
  if (!h) return GAIB_OK;
  gaib_comm* c = h->c;
  (void)hipSetDevice(c->ctx->device);
  (void)hipStreamSynchronize(c->cstream);
  (void)hipStreamSynchronize(c->ctx->stream);
  if (c->transport == GAIB_COMM_IPC) {
    for (int r = 0; r < c->nranks; r++) {
      if (h->peer[r].base) (void)hipIpcCloseMemHandle(h->peer[r].base);
      for (void* q : h->peer[r].base_x)
        if (q) (void)hipIpcCloseMemHandle(q);
      if (h->peer[r].base_t) (void)hipIpcCloseMemHandle(h->peer[r].base_t);
      for (void* q : h->peer[r].base_tx)
        if (q) (void)hipIpcCloseMemHandle(q);
    }
    // nobody may still be pulling from the send buffer that is about to be freed
    if (c->seg && !c->seg->error.load()) (void)shm_barrier(c, "gaib_halo_destroy");
  }
  if (h->d_send_idx) (void)hipFree(h->d_send_idx);
  if (h->d_pack_row) (void)hipFree(h->d_pack_row);
  if (h->d_pack_slot) (void)hipFree(h->d_pack_slot);
  if (h->d_pack_addr) (void)hipFree(h->d_pack_addr);
  for (hipEvent_t ev : h->ev_piece)
    if (ev) (void)hipEventDestroy(ev);
  pool_release(c, h->sendbuf);  // back into the communicator's pool (see gaib_comm::ipc_bufs)
  for (float* q : h->send_x) pool_release(c, q);
  pool_release(c, h->landing);
  for (float* q : h->stage_x) pool_release(c, q);
  pool_release(c, h->table);
  for (void* q : h->retired) pool_release(c, q);
  c->halo_slots &= ~(1u << h->id);  // after the barrier above: the slot row can serve the next plan
  delete h;
  return GAIB_OK;
}

extern "C" int64_t gaib_halo_rows(const gaib_halo* h) { return h ? h->recv_off[h->c->nranks] : 0; }
extern "C" int64_t gaib_halo_send_rows(const gaib_halo* h) { return h ? h->send_off[h->c->nranks] : 0; }
extern "C" int64_t gaib_halo_link_rows(const gaib_halo* h) {
  if (!h) return 0;
  int64_t m = 0;
  for (int r = 0; r < h->c->nranks; ++r) {
    if (h->send_counts[r] > m) m = h->send_counts[r];
    if (h->recv_counts[r] > m) m = h->recv_counts[r];
  }
  return m;
}
extern "C" int64_t gaib_halo_bytes_sent(const gaib_halo* h) { return h ? h->bytes_sent : 0; }

extern "C" int gaib_halo_send_stats(const gaib_halo* h, int64_t* h_packs, int64_t* h_direct_sends, int* h_direct_peers) {
  GAIB_CHECK(h && h_packs && h_direct_sends && h_direct_peers, "gaib_halo_send_stats: NULL argument");
  *h_packs = h->packs;
  *h_direct_sends = h->direct_sends;
  int n = 0;
  for (int r = 0; r < h->c->nranks; r++) n += h->direct_first[r] >= 0 ? 1 : 0;
  *h_direct_peers = n;
  return GAIB_OK;
}

// ---- an exchange in time slices ("pieces", round 6) ----
extern "C" int gaib_halo_piece_slice(int64_t rows, int n_pieces, int piece, int64_t* h_lo, int64_t* h_hi) {
  GAIB_CHECK(h_lo && h_hi, "gaib_halo_piece_slice: NULL argument");
  GAIB_CHECK(rows >= 0 && n_pieces >= 1 && n_pieces <= GAIB_HALO_MAX_PIECES && piece >= 0 && piece < n_pieces,
             "gaib_halo_piece_slice: piece %d of %d over %lld rows (at most %d pieces)", piece, n_pieces, (long long)rows,
             GAIB_HALO_MAX_PIECES);
  piece_slice(rows, n_pieces, piece, h_lo, h_hi);
  return GAIB_OK;
}

// how many slices a partition of n_global vertices over `world` ranges cuts its exchanges into -- a function of figures every
// rank holds (so all ranks agree without a collective): GAIB_HALO_PIECES if set; else by the rows one peer pair can move at
// most (a whole range): from 131 072 rows (64 MB of 512-B rows) 4 slices, from 32 768 two, below that one -- a slice should
// stay far above the latency of its group launch, and every further piece costs one more read + write of the rows' partial sums
extern "C" int gaib_halo_default_pieces(int64_t n_global, int world) {
  const char* e = getenv("GAIB_HALO_PIECES");
  if (e && *e) {
    char* end = nullptr;
    const long v = strtol(e, &end, 10);
    if (end != e && *end == '\0' && v >= 1 && v <= GAIB_HALO_MAX_PIECES) return (int)v;
    fprintf(stderr, "[gaib] GAIB_HALO_PIECES='%s' ignored (want 1 .. %d)\n", e, GAIB_HALO_MAX_PIECES);
  }
  if (world < 2 || n_global <= 0) return 1;
  const int64_t rows = n_global / world;
  return rows >= 131072 ? 4 : (rows >= 32768 ? 2 : 1);
}

extern "C" int gaib_halo_set_pieces(gaib_halo* h, int n_pieces) {
  GAIB_CHECK(h, "gaib_halo_set_pieces: halo is NULL");
  GAIB_CHECK(n_pieces >= 1 && n_pieces <= GAIB_HALO_MAX_PIECES, "gaib_halo_set_pieces: %d pieces (1 .. %d)", n_pieces,
             GAIB_HALO_MAX_PIECES);
  GAIB_CHECK(h->pending_len == 0, "gaib_halo_set_pieces: an exchange is in flight on this plan");
  GAIB_HIP(hipSetDevice(h->c->ctx->device));
  for (int k = 0; k < n_pieces; ++k)
    if (!h->ev_piece[k]) GAIB_HIP(hipEventCreateWithFlags(&h->ev_piece[k], hipEventDisableTiming));
  h->n_pieces = n_pieces;
  return GAIB_OK;
}
extern "C" int gaib_halo_pieces(const gaib_halo* h) { return h ? h->n_pieces : 0; }

extern "C" int gaib_halo_piece_ranges(const gaib_halo* h, int piece, int cap, int64_t* h_begin, int64_t* h_end, int* h_n) {
  GAIB_CHECK(h && h_begin && h_end && h_n, "gaib_halo_piece_ranges: NULL argument");
  GAIB_CHECK(piece >= 0 && piece < h->n_pieces, "gaib_halo_piece_ranges: piece %d of %d", piece, h->n_pieces);
  int n = 0;
  for (int r = 0; r < h->c->nranks; ++r) {
    int64_t lo, hi;
    piece_slice(h->recv_counts[r], h->n_pieces, piece, &lo, &hi);
    if (hi <= lo) continue;
    GAIB_CHECK(n < cap, "gaib_halo_piece_ranges: more than %d ranges", cap);
    h_begin[n] = h->recv_off[r] + lo;
    h_end[n] = h->recv_off[r] + hi;
    ++n;
  }
  *h_n = n;
  return GAIB_OK;
}

// 1. pack the owned rows the peers asked for (compute stream), 2. start moving them.  Every rank calls it for every
// exchange (also one that neither sends nor receives).  Returns at once on RCCL; on IPC after the peers' packs are
// done and this rank's pulls are enqueued.
extern "C" int gaib_halo_exchange_begin(gaib_halo* h, int len, const float* d_rows) {
  GAIB_CHECK(h && len >= 1, "gaib_halo_exchange_begin: bad argument");
  GAIB_CHECK(h->pending_len == 0, "gaib_halo_exchange_begin: the previous exchange was not ended");
  gaib_comm* c = h->c;
  gaib_ctx* ctx = c->ctx;
  GAIB_HIP(hipSetDevice(ctx->device));
  const int64_t n_send = h->send_off[c->nranks], n_recv = h->recv_off[c->nranks];
  GAIB_CHECK(n_send == 0 || d_rows, "gaib_halo_exchange_begin: d_rows is NULL");
  const size_t row_bytes = sizeof(float) * (size_t)len;
  gaib_halo* keep = h;  // (both transports draw from the communicator's pool; only IPC retires, see reserve)
  // IPC: a send buffer above the chunk size is cut into chunks of whole rows, each its own allocation (GAIB_IPC_MAX_CHUNKS)
  const size_t chunk_bytes = ipc_chunk_bytes_for(row_bytes * (size_t)n_send);
  const bool chunked = c->transport == GAIB_COMM_IPC && row_bytes * (size_t)n_send > chunk_bytes;
  const int64_t chunk_rows = chunked ? std::max<int64_t>(1, (int64_t)(chunk_bytes / row_bytes)) : n_send;
  const int n_chunks = chunked ? (int)cdiv64(n_send, chunk_rows) : 1;
  if (n_chunks > GAIB_IPC_MAX_CHUNKS) {
    gaib_set_error("gaib_halo_exchange_begin(rank %d): %lld rows of %zu B need %d chunks of %zu B, at most %d (GAIB_IPC_CHUNK_BYTES)",
                   c->rank, (long long)n_send, row_bytes, n_chunks, chunk_bytes, GAIB_IPC_MAX_CHUNKS);
    return fail(c, GAIB_ERR_UNSUPPORTED);
  }
  auto chunk_ptr = [&](int j) -> float* { return j == 0 ? h->sendbuf : h->send_x[j - 1]; };
  auto rows_in_chunk = [&](int j) -> int64_t { return std::min<int64_t>(chunk_rows, n_send - (int64_t)j * chunk_rows); };
  const size_t exp_cap = c->transport == GAIB_COMM_IPC ? ipc_export_limit() : ~(size_t)0;  // (IPC exports every send allocation)
  const bool no_pack = c->transport == GAIB_COMM_RCCL && h->all_direct;  // every peer reads the caller's matrix itself
  int ra = no_pack ? 0 : reserve(&h->sendbuf, &h->send_cap, &h->send_serial, row_bytes * (size_t)(chunked ? chunk_rows : n_send),
                                 ctx->stream, keep, exp_cap);
  if (ra < 0) return fail(c, ra);
  for (int j = 1; j < n_chunks; ++j) {
    int rx = reserve(&h->send_x[j - 1], &h->send_x_cap[j - 1], &h->send_x_serial[j - 1], row_bytes * (size_t)rows_in_chunk(j),
                     ctx->stream, keep, exp_cap);
    if (rx < 0) return fail(c, rx);
  }
  int rb = reserve(&h->table, &h->table_cap, &h->table_serial, row_bytes * (size_t)n_recv, ctx->stream, keep);
  if (rb < 0) return fail(c, rb);
  if (n_send && no_pack) {
    // nothing to pack
  } else if (n_send && !chunked) {
    int rc = h->d_pack_row ? gaib_gather_scatter_rows(ctx, n_send, h->d_pack_row, h->d_pack_slot, len, d_rows, h->sendbuf)
                           : gaib_gather_rows(ctx, n_send, h->d_send_idx, len, d_rows, h->sendbuf);
    if (rc != GAIB_OK) return fail(c, rc);
    h->packs++;
  } else if (n_send) {
    // the source-ordered pack into the chunks: every (row, slot) pair's destination as an address; where that form does
    // not apply (odd row lengths, no sorted pairs) chunk by chunk in destination order
    int rc = GAIB_ERR_UNSUPPORTED;
    if (h->d_pack_row) {
      if (!h->d_pack_addr && hipMalloc((void**)&h->d_pack_addr, sizeof(int64_t) * (size_t)n_send) != hipSuccess) {
        (void)hipGetLastError();
        h->d_pack_addr = nullptr;
      }
      if (h->d_pack_addr) {
        ChunkBases b;
        for (int j = 0; j < GAIB_IPC_MAX_CHUNKS; ++j) b.p[j] = j < n_chunks ? chunk_ptr(j) : nullptr;
        chunk_row_address_kernel<<<(unsigned)cdiv64(n_send, 256), 256, 0, ctx->stream>>>(n_send, h->d_pack_slot, b, chunk_rows, len,
                                                                                        h->d_pack_addr);
        GAIB_LAUNCH_CHECK();
        rc = gaib_gather_rows_to_addresses(ctx, n_send, h->d_pack_row, h->d_pack_addr, len, d_rows);
        if (rc != GAIB_OK && rc != GAIB_ERR_UNSUPPORTED) return fail(c, rc);
      }
    }
    for (int j = 0; j < n_chunks && rc == GAIB_ERR_UNSUPPORTED; ++j) {
      int r2 = gaib_gather_rows(ctx, rows_in_chunk(j), h->d_send_idx + (int64_t)j * chunk_rows, len, d_rows, chunk_ptr(j));
      if (r2 != GAIB_OK) return fail(c, r2);
    }
    h->packs++;
  }
  h->bytes_sent += (int64_t)row_bytes * n_send;
  h->pending_len = len;
  GAIB_HIP(hipEventRecord(c->ev_ready, ctx->stream));
  if (c->transport == GAIB_COMM_RCCL) {
    GAIB_HIP(hipStreamWaitEvent(c->cstream, c->ev_ready, 0));
    if (c->nranks > 1) {
      // one group per time slice (K = 1: the whole exchange): slice k of every peer pair travels together, and the event
      // behind it tells the compute stream that piece k's columns of the table are there
      const int K = h->n_pieces;
      for (int r = 0; r < c->nranks; r++)
        if (h->send_counts[r] && h->direct_first[r] >= 0) h->direct_sends++;
      for (int k = 0; k < K; ++k) {
        GAIB_NCCL(g_rccl.GroupStart());
        for (int r = 0; r < c->nranks; r++) {
          int64_t lo, hi;
          piece_slice(h->send_counts[r], K, k, &lo, &hi);
          if (hi > lo) {
            // a run of consecutive rows goes straight from the caller's matrix (it stays untouched until gaib_halo_exchange_end:
            // the aggregation in between only reads it); otherwise from the packed send buffer
            const bool direct = h->direct_first[r] >= 0;
            const float* src = direct ? d_rows + (h->direct_first[r] + lo) * len : h->sendbuf + (h->send_off[r] + lo) * len;
            GAIB_NCCL(g_rccl.Send(src, (size_t)((hi - lo) * len), ncclFloat32, r, c->nccl, c->cstream));
          }
          piece_slice(h->recv_counts[r], K, k, &lo, &hi);
          if (hi > lo)
            GAIB_NCCL(g_rccl.Recv(h->table + (h->recv_off[r] + lo) * len, (size_t)((hi - lo) * len), ncclFloat32, r, c->nccl,
                                  c->cstream));
        }
        GAIB_NCCL(g_rccl.GroupEnd());
        if (K > 1) GAIB_HIP(hipEventRecord(h->ev_piece[k], c->cstream));
      }
    } else {
      for (int k = 0; k < h->n_pieces && h->n_pieces > 1; ++k) GAIB_HIP(hipEventRecord(h->ev_piece[k], c->cstream));
    }
    GAIB_HIP(hipEventRecord(c->ev_done, c->cstream));
    return GAIB_OK;
  }
  // ---- IPC pull ----
  ShmSlot* mine = &c->seg->slot[h->id][c->rank];
  bool republish = h->pub_send_serial != h->send_serial || h->pub_n_chunks != n_chunks || h->pub_chunk_rows != chunk_rows;
  for (int j = 1; j < n_chunks; ++j) republish = republish || h->pub_x_serial[j - 1] != h->send_x_serial[j - 1];
  if (republish) {  // an allocation the peers hold a handle of was replaced, or the layout changed (another row length)
    for (int j = 0; j < n_chunks; ++j) {
      if ((j == 0 ? h->send_cap : h->send_x_cap[j - 1]) > ipc_export_limit()) {
        gaib_set_error("gaib_halo_exchange_begin(rank %d): chunk %d of the send buffer is an allocation of %zu B, above what this "
                       "transport exports (%zu B: hipIpcOpenMemHandle does not return for allocations above 2 GiB)", c->rank, j,
                       j == 0 ? h->send_cap : h->send_x_cap[j - 1], ipc_export_limit());
        return fail(c, GAIB_ERR_UNSUPPORTED);
      }
      hipError_t e = pool_handle(c, chunk_ptr(j), j == 0 ? &mine->handle : &mine->handle_x[j - 1]);
      if (e != hipSuccess) {
        gaib_set_error("gaib_halo_exchange_begin: hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(e));
        return fail(c, GAIB_ERR_HIP);
      }
      if (j > 0) h->pub_x_serial[j - 1] = h->send_x_serial[j - 1];
    }
    mine->capacity_bytes = h->send_cap;
    mine->device = ctx->device;
    mine->n_chunks = n_chunks;
    mine->chunk_rows = chunk_rows;
    for (int r = 0; r <= c->nranks; r++) mine->send_off[r] = h->send_off[r];
    mine->gen++;
    h->pub_send_serial = h->send_serial;
    h->pub_n_chunks = n_chunks;
    h->pub_chunk_rows = chunk_rows;
  }
  mine->n_pieces = h->n_pieces;  // (read by the peers after the barrier below)
  GAIB_COMM_DBG(c, "exchange_begin: %lld rows out (%zu B), %lld rows in, handle published; waiting for the pack", (long long)n_send,
                row_bytes * (size_t)n_send, (long long)n_recv);
  hipError_t e = hipEventSynchronize(c->ev_ready);  // the pack is done: peers may read the buffer
  if (e != hipSuccess) {
    gaib_set_error("gaib_halo_exchange_begin: %s", hipGetErrorString(e));
    return fail(c, GAIB_ERR_HIP);
  }
  GAIB_COMM_DBG(c, "exchange_begin: packed");
  int rc = shm_barrier(c, "gaib_halo_exchange_begin (all packed)");
  if (rc != GAIB_OK) return rc;
  GAIB_COMM_DBG(c, "exchange_begin: all ranks packed");
  const int K = h->n_pieces;
  for (int k = 0; k < K; ++k) {
    for (int r = 0; r < c->nranks; r++) {
      if (!h->recv_counts[r]) continue;
      const ShmSlot* ps = &c->seg->slot[h->id][r];
      if (ps->n_pieces != K) {
        gaib_set_error("gaib_halo_exchange_begin(rank %d): rank %d cuts this exchange into %d pieces, this rank into %d "
                       "(gaib_halo_set_pieces: the same on every rank)", c->rank, r, ps->n_pieces, K);
        return fail(c, GAIB_ERR_INVALID);
      }
      if (h->peer[r].gen != ps->gen) {  // the peer's allocations (or their layout) changed: every mapping of the old ones goes
        if (h->peer[r].base) (void)hipIpcCloseMemHandle(h->peer[r].base);
        h->peer[r].base = nullptr;
        for (void*& q : h->peer[r].base_x) {
          if (q) (void)hipIpcCloseMemHandle(q);
          q = nullptr;
        }
        h->peer[r].gen = ps->gen;
      }
      if (ps->send_off[c->rank + 1] - ps->send_off[c->rank] != h->recv_counts[r]) {
        gaib_set_error("gaib_halo_exchange_begin(rank %d): rank %d sends %lld rows, this rank expects %lld", c->rank, r,
                       (long long)(ps->send_off[c->rank + 1] - ps->send_off[c->rank]), (long long)h->recv_counts[r]);
        return fail(c, GAIB_ERR_INVALID);
      }
      const int pk = ps->n_chunks > 0 ? ps->n_chunks : 1;
      const int64_t pcr = ps->chunk_rows;
      if (pk > GAIB_IPC_MAX_CHUNKS || (pk > 1 && pcr < 1)) {
        gaib_set_error("gaib_halo_exchange_begin(rank %d): rank %d published %d chunks of %lld rows", c->rank, r, pk, (long long)pcr);
        return fail(c, GAIB_ERR_INVALID);
      }
      // this rank's segment of the peer's send buffer: rows [s0, s0 + recv_counts) of its slot space; slice k of it, chunk by chunk
      const int64_t s0 = ps->send_off[c->rank];
      int64_t k_lo, k_hi;
      piece_slice(h->recv_counts[r], K, k, &k_lo, &k_hi);
      const int64_t s1 = s0 + k_hi;
      for (int64_t a = s0 + k_lo; a < s1;) {
        const int j = pk > 1 ? (int)(a / pcr) : 0;
        const int64_t in_chunk = pk > 1 ? a - (int64_t)j * pcr : a;
        const int64_t b = pk > 1 ? std::min<int64_t>(s1, (int64_t)(j + 1) * pcr) : s1;
        void*& base = j == 0 ? h->peer[r].base : h->peer[r].base_x[j - 1];
        if (!base) {
          GAIB_COMM_DBG(c, "exchange_begin: opening chunk %d of %d of rank %d's send buffer", j, pk, r);
          e = hipIpcOpenMemHandle(&base, j == 0 ? ps->handle : ps->handle_x[j - 1], hipIpcMemLazyEnablePeerAccess);
          if (e != hipSuccess) {
            base = nullptr;
            gaib_set_error("gaib_halo_exchange_begin(rank %d): hipIpcOpenMemHandle(rank %d's send buffer, chunk %d): %s", c->rank, r, j,
                           hipGetErrorString(e));
            return fail(c, GAIB_ERR_HIP);
          }
        }
        const float* src = (const float*)base + in_chunk * len;
        GAIB_COMM_DBG(c, "exchange_begin: pulling %zu B from rank %d (chunk %d, row %lld)", row_bytes * (size_t)(b - a), r, j,
                      (long long)in_chunk);
        e = hipMemcpyAsync(h->table + (h->recv_off[r] + (a - s0)) * len, src, row_bytes * (size_t)(b - a), hipMemcpyDeviceToDevice,
                           c->cstream);
        if (e != hipSuccess) {
          gaib_set_error("gaib_halo_exchange_begin(rank %d): peer copy from rank %d: %s", c->rank, r, hipGetErrorString(e));
          return fail(c, GAIB_ERR_HIP);
        }
        a = b;
      }
    }
    if (K > 1) GAIB_HIP(hipEventRecord(h->ev_piece[k], c->cstream));
  }
  GAIB_HIP(hipEventRecord(c->ev_done, c->cstream));
  return GAIB_OK;
}

// between begin and end: the compute stream continues only after slice `piece` of every peer pair has landed -- the columns
// gaib_halo_piece_ranges names are then valid in *d_table (the others are still on the wire).  Stream-ordered on both
// transports (the host does not wait); gaib_halo_exchange_end is still due after the last piece.
extern "C" int gaib_halo_exchange_wait_piece(gaib_halo* h, int piece, const float** d_table) {
  GAIB_CHECK(h && d_table, "gaib_halo_exchange_wait_piece: NULL argument");
  GAIB_CHECK(h->pending_len > 0, "gaib_halo_exchange_wait_piece: no exchange in flight");
  GAIB_CHECK(piece >= 0 && piece < h->n_pieces, "gaib_halo_exchange_wait_piece: piece %d of %d", piece, h->n_pieces);
  gaib_comm* c = h->c;
  GAIB_HIP(hipSetDevice(c->ctx->device));
  GAIB_HIP(hipStreamWaitEvent(c->ctx->stream, h->n_pieces > 1 ? h->ev_piece[piece] : c->ev_done, 0));
  h->piece_waits++;
  *d_table = h->table;
  return GAIB_OK;
}

// the compute stream continues after the rows have arrived; *d_table = [rows x len], grouped by owner rank in the
// order of the recv counts
extern "C" int gaib_halo_exchange_end(gaib_halo* h, const float** d_table) {
  GAIB_CHECK(h && d_table, "gaib_halo_exchange_end: NULL argument");
  GAIB_CHECK(h->pending_len > 0, "gaib_halo_exchange_end: no exchange in flight");
  gaib_comm* c = h->c;
  GAIB_HIP(hipSetDevice(c->ctx->device));
  h->pending_len = 0;
  if (c->transport == GAIB_COMM_IPC) {
    GAIB_COMM_DBG(c, "exchange_end: waiting for the pulls");
    hipError_t e = hipEventSynchronize(c->ev_done);
    if (e != hipSuccess) {
      gaib_set_error("gaib_halo_exchange_end: %s", hipGetErrorString(e));
      return fail(c, GAIB_ERR_HIP);
    }
    GAIB_COMM_DBG(c, "exchange_end: pulled");
    int rc = shm_barrier(c, "gaib_halo_exchange_end (all pulled)");  // send buffers may be overwritten from here on
    if (rc != GAIB_OK) return rc;
  }
  GAIB_HIP(hipStreamWaitEvent(c->ctx->stream, c->ev_done, 0));
  *d_table = h->table;
  return GAIB_OK;
}

// d_rows[idx[k], :] += buf[k, :]; the row ids of one peer's segment are distinct (ascending), one wave per row
__global__ __launch_bounds__(256) void scatter_add_rows_kernel(int64_t n, const int64_t* idx, int len, const float* buf,
                                                               float* rows) {
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n) return;
  const int lane = threadIdx.x & 63;
  float* dst = rows + idx[k] * (int64_t)len;
  const float* src = buf + k * (int64_t)len;
  for (int c = lane; c < len; c += 64) dst[c] += src[c];
}

// The reverse of an exchange: every rank holds partial rows for its HALO vertices (d_halo_rows, the table's layout:
// grouped by owner rank) and returns them to their owners, which add what arrives to their own rows:
//   d_rows[send_idx[k], :] += arrived[k, :]        peer by peer in rank order (deterministic, no atomics).
// The transposed aggregation of GAT backward on a partition (out_c += p_(i->c) grad_i for rows i of another rank).
// Collective; stream-ordered on RCCL, host-synchronous on IPC.
extern "C" int gaib_halo_reduce(gaib_halo* h, int len, const float* d_halo_rows, float* d_rows) {
  GAIB_CHECK(h && len >= 1, "gaib_halo_reduce: bad argument");
  GAIB_CHECK(h->pending_len == 0, "gaib_halo_reduce: an exchange is in flight on this plan");
  gaib_comm* c = h->c;
  gaib_ctx* ctx = c->ctx;
  GAIB_HIP(hipSetDevice(ctx->device));
  const int64_t n_send = h->send_off[c->nranks], n_recv = h->recv_off[c->nranks];
  GAIB_CHECK((n_recv == 0 || d_halo_rows) && (n_send == 0 || d_rows), "gaib_halo_reduce: NULL rows");
  const size_t row_bytes = sizeof(float) * (size_t)len;
  gaib_halo* keep = h;  // (both transports draw from the communicator's pool; only IPC retires, see reserve)
  // arrivals land in a buffer of their own: the send buffer stays what the exchanges made it (on IPC possibly chunks)
  int ra = reserve(&h->landing, &h->landing_cap, &h->landing_serial, row_bytes * (size_t)n_send, ctx->stream, keep);
  if (ra < 0) return fail(c, ra);
  if (c->transport == GAIB_COMM_RCCL) {
    GAIB_HIP(hipEventRecord(c->ev_ready, ctx->stream));
    GAIB_HIP(hipStreamWaitEvent(c->cstream, c->ev_ready, 0));
    if (c->nranks > 1) {
      GAIB_NCCL(g_rccl.GroupStart());
      for (int r = 0; r < c->nranks; r++) {
        if (h->recv_counts[r])
          GAIB_NCCL(g_rccl.Send(d_halo_rows + h->recv_off[r] * len, (size_t)(h->recv_counts[r] * len), ncclFloat32, r, c->nccl,
                                c->cstream));
        if (h->send_counts[r])
          GAIB_NCCL(g_rccl.Recv(h->landing + h->send_off[r] * len, (size_t)(h->send_counts[r] * len), ncclFloat32, r, c->nccl,
                                c->cstream));
      }
      GAIB_NCCL(g_rccl.GroupEnd());
    }
    GAIB_HIP(hipEventRecord(c->ev_done, c->cstream));
    GAIB_HIP(hipStreamWaitEvent(ctx->stream, c->ev_done, 0));
  } else {
    // IPC: stage the partial rows in allocations of this plan (a whole allocation can be exported) -- the table itself
    // while it is below the chunk size, else chunks of whole rows, each its own allocation --, publish them, and let the
    // owners pull their segments
    // (also when the rows are few but the table an earlier, longer exchange left behind is too large to export)
    const size_t chunk_bytes = ipc_chunk_bytes_for(row_bytes * (size_t)n_recv);
    const bool many = row_bytes * (size_t)n_recv > chunk_bytes;
    const bool chunked = many || (h->table && h->table_cap > ipc_export_limit());
    const int64_t chunk_rows = many ? std::max<int64_t>(1, (int64_t)(chunk_bytes / row_bytes)) : n_recv;
    const int n_chunks = many ? (int)cdiv64(n_recv, chunk_rows) : 1;
    if (n_chunks > GAIB_IPC_MAX_CHUNKS) {
      gaib_set_error("gaib_halo_reduce(rank %d): %lld halo rows of %zu B need %d chunks of %zu B, at most %d (GAIB_IPC_CHUNK_BYTES)",
                     c->rank, (long long)n_recv, row_bytes, n_chunks, chunk_bytes, GAIB_IPC_MAX_CHUNKS);
      return fail(c, GAIB_ERR_UNSUPPORTED);
    }
    hipError_t e = hipSuccess;
    const char* step = "staging copy";
    ShmSlot* mine = &c->seg->slot[h->id][c->rank];
    bool republish = h->pub_n_chunks_t != (chunked ? n_chunks : -1) || h->pub_chunk_rows_t != chunk_rows;
    if (!chunked) {
      int rb = reserve(&h->table, &h->table_cap, &h->table_serial, row_bytes * (size_t)n_recv, ctx->stream, keep);
      if (rb < 0) return fail(c, rb);
      if (n_recv && d_halo_rows != h->table)
        e = hipMemcpyAsync(h->table, d_halo_rows, row_bytes * (size_t)n_recv, hipMemcpyDeviceToDevice, ctx->stream);
      republish = republish || h->pub_table_serial != h->table_serial;  // also after an exchange grew the table
    } else {
      for (int j = 0; j < n_chunks && e == hipSuccess; ++j) {
        const int64_t rows_j = std::max<int64_t>(0, std::min<int64_t>(chunk_rows, n_recv - (int64_t)j * chunk_rows));
        int rx = reserve(&h->stage_x[j], &h->stage_x_cap[j], &h->stage_x_serial[j], row_bytes * (size_t)rows_j, ctx->stream, keep,
                         ipc_export_limit());
        if (rx < 0) return fail(c, rx);
        if (rows_j)
          e = hipMemcpyAsync(h->stage_x[j], d_halo_rows + (int64_t)j * chunk_rows * len, row_bytes * (size_t)rows_j,
                             hipMemcpyDeviceToDevice, ctx->stream);
        republish = republish || h->pub_stage_serial[j] != h->stage_x_serial[j];
      }
    }
    if (e == hipSuccess && republish) {
      step = "hipIpcGetMemHandle(halo rows)";
      for (int j = 0; j < n_chunks && e == hipSuccess; ++j) {
        void* q = chunked ? (void*)h->stage_x[j] : (void*)h->table;
        const size_t cap = chunked ? h->stage_x_cap[j] : h->table_cap;
        if (cap > ipc_export_limit()) {
          gaib_set_error("gaib_halo_reduce(rank %d): an allocation of %zu B holds the halo rows, above what the IPC transport exports "
                         "(%zu B: hipIpcOpenMemHandle does not return for allocations above 2 GiB)", c->rank, cap, ipc_export_limit());
          return fail(c, GAIB_ERR_UNSUPPORTED);
        }
        e = pool_handle(c, q, j == 0 ? &mine->handle_t : &mine->handle_tx[j - 1]);
        if (chunked) h->pub_stage_serial[j] = h->stage_x_serial[j];
      }
      for (int r = 0; r <= c->nranks; r++) mine->recv_off[r] = h->recv_off[r];
      mine->n_chunks_t = n_chunks;
      mine->chunk_rows_t = chunk_rows;
      mine->gen_t++;
      h->pub_table_serial = chunked ? 0 : h->table_serial;
      h->pub_n_chunks_t = chunked ? n_chunks : -1;  // (-1: the table itself is what the peers hold a handle of)
      h->pub_chunk_rows_t = chunk_rows;
    }
    if (e == hipSuccess) {
      step = "stream sync after staging";
      e = hipStreamSynchronize(ctx->stream);
    }
    if (e != hipSuccess) {
      gaib_set_error("gaib_halo_reduce(rank %d, len %d, %lld halo rows, table %p cap %zu): %s: %s", c->rank, len, (long long)n_recv,
                     (void*)h->table, h->table_cap, step, hipGetErrorString(e));
      return fail(c, GAIB_ERR_HIP);
    }
    int rc = shm_barrier(c, "gaib_halo_reduce (all staged)");
    if (rc != GAIB_OK) return rc;
    for (int r = 0; r < c->nranks; r++) {
      if (!h->send_counts[r]) continue;  // rank r holds partial rows for send_counts[r] of this rank's vertices
      const ShmSlot* ps = &c->seg->slot[h->id][r];
      if (h->peer[r].gen_t != ps->gen_t) {
        if (h->peer[r].base_t) (void)hipIpcCloseMemHandle(h->peer[r].base_t);
        h->peer[r].base_t = nullptr;
        for (void*& q : h->peer[r].base_tx) {
          if (q) (void)hipIpcCloseMemHandle(q);
          q = nullptr;
        }
        h->peer[r].gen_t = ps->gen_t;
      }
      if (ps->recv_off[c->rank + 1] - ps->recv_off[c->rank] != h->send_counts[r]) {
        gaib_set_error("gaib_halo_reduce(rank %d): rank %d returns %lld rows, this rank expects %lld", c->rank, r,
                       (long long)(ps->recv_off[c->rank + 1] - ps->recv_off[c->rank]), (long long)h->send_counts[r]);
        return fail(c, GAIB_ERR_INVALID);
      }
      const int pk = ps->n_chunks_t > 0 ? ps->n_chunks_t : 1;
      const int64_t pcr = ps->chunk_rows_t;
      if (pk > GAIB_IPC_MAX_CHUNKS || (pk > 1 && pcr < 1)) {
        gaib_set_error("gaib_halo_reduce(rank %d): rank %d published %d chunks of %lld rows", c->rank, r, pk, (long long)pcr);
        return fail(c, GAIB_ERR_INVALID);
      }
      const int64_t s0 = ps->recv_off[c->rank], s1 = s0 + h->send_counts[r];  // this rank's rows in the peer's halo order
      for (int64_t a = s0; a < s1;) {
        const int j = pk > 1 ? (int)(a / pcr) : 0;
        const int64_t in_chunk = pk > 1 ? a - (int64_t)j * pcr : a;
        const int64_t b = pk > 1 ? std::min<int64_t>(s1, (int64_t)(j + 1) * pcr) : s1;
        void*& base = j == 0 ? h->peer[r].base_t : h->peer[r].base_tx[j - 1];
        if (!base) {
          e = hipIpcOpenMemHandle(&base, j == 0 ? ps->handle_t : ps->handle_tx[j - 1], hipIpcMemLazyEnablePeerAccess);
          if (e != hipSuccess) {
            base = nullptr;
            gaib_set_error("gaib_halo_reduce(rank %d): hipIpcOpenMemHandle(rank %d's halo rows, chunk %d): %s", c->rank, r, j,
                           hipGetErrorString(e));
            return fail(c, GAIB_ERR_HIP);
          }
        }
        e = hipMemcpyAsync(h->landing + (h->send_off[r] + (a - s0)) * len, (const float*)base + in_chunk * len,
                           row_bytes * (size_t)(b - a), hipMemcpyDeviceToDevice, ctx->stream);
        if (e != hipSuccess) {
          gaib_set_error("gaib_halo_reduce(rank %d): peer copy from rank %d: %s", c->rank, r, hipGetErrorString(e));
          return fail(c, GAIB_ERR_HIP);
        }
        a = b;
      }
    }
    e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
      gaib_set_error("gaib_halo_reduce: %s", hipGetErrorString(e));
      return fail(c, GAIB_ERR_HIP);
    }
    rc = shm_barrier(c, "gaib_halo_reduce (all pulled)");  // the tables may be overwritten from here on
    if (rc != GAIB_OK) return rc;
  }
  for (int r = 0; r < c->nranks; r++) {
    if (!h->send_counts[r]) continue;
    scatter_add_rows_kernel<<<(unsigned)cdiv64(h->send_counts[r], 4), 256, 0, ctx->stream>>>(
        h->send_counts[r], h->d_send_idx + h->send_off[r], len, h->landing + h->send_off[r] * len, d_rows);
  }
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}
