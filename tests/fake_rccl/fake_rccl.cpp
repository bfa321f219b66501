// TEST DOUBLE of the eleven RCCL entry points graphaibench_amd/csrc/comm.hip binds (RcclApi, comm.hip:41-54).
// TEST INFRASTRUCTURE ONLY: loaded through GAIB_RCCL_LIB by tests/test_gpu_comm.py, never by bench.py (which refuses to
// run with that variable set) and never shipped in graphaibench_amd/lib.
//
// Why: RCCL refuses two ranks on one device and a GPU box has one, so the RCCL branch of the halo exchange, the reverse
// exchange and the all-reduces (grouped ncclSend / ncclRecv with per-peer offsets and counts) cannot run with N > 1 on
// the real library there.  This double carries the SAME calls between processes that share one GPU, through POSIX shared
// memory, and is STRICT where the real library would hang, corrupt or fault:
//   * every message is matched in order per (source, destination) pair, and count + datatype of the send must equal
//     those of the receive that takes it;
//   * every device range [buf, buf + count * size) must lie inside one live allocation (hipMemGetAddressRange);
//   * a rank id used twice, a peer out of range, an unbalanced group, a wait longer than the deadline are errors.
// It is not asynchronous (each call / group synchronises the stream it was given and completes on the host), which is a
// legal schedule of the real library, and it sums all-reduces in rank order (identical bits on every rank, as RCCL
// guarantees for one communicator).  Nothing here speaks about speed.
#include <atomic>
#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

#include <fcntl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

namespace {

constexpr int kMaxRanks = 16;
constexpr int kRing = 64;  // descriptors in flight per (source, destination) pair

struct Box {
  std::atomic<uint64_t> posted;  // messages the source has published
  std::atomic<uint64_t> taken;   // messages the destination has consumed
  uint64_t count[kRing];
  int32_t dtype[kRing];
};
struct Seg {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> departed;
  std::atomic<uint32_t> error;
  std::atomic<uint32_t> bar_count;
  std::atomic<uint32_t> bar_sense;
  std::atomic<uint32_t> rank_taken[kMaxRanks];
  uint64_t ar_count[kMaxRanks];  // what each rank brought to the all-reduce in flight
  Box box[kMaxRanks][kMaxRanks];  // [source][destination]
};

double now_s() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}
double deadline_s() {
  const char* e = getenv("GAIB_FAKE_RCCL_TIMEOUT_S");
  return e ? atof(e) : 60.0;
}
thread_local char g_why[256] = "";
ncclResult_t fail(ncclResult_t r, const char* fmt, long long a = 0, long long b = 0, long long c = 0, long long d = 0) {
  snprintf(g_why, sizeof(g_why), fmt, a, b, c, d);
  fprintf(stderr, "fake_rccl: %s\n", g_why);
  return r;
}
size_t dtype_size(ncclDataType_t t) {
  switch (t) {
    case ncclFloat32: return 4;
    case ncclFloat64: return 8;
    default: return 0;  // the path uses nothing else
  }
}

}  // namespace

struct ncclComm {
  Seg* seg;
  int rank, nranks;
  char name[48];
  uint32_t sense;
  uint64_t ar_seq;
  uint64_t sent[kMaxRanks], recvd[kMaxRanks];
};

namespace {

struct Op {
  bool send;
  void* buf;
  size_t count;
  ncclDataType_t dtype;
  int peer;
  ncclComm* comm;
  hipStream_t stream;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;

template <class F>
ncclResult_t wait_for(ncclComm* c, const char* what, F&& ready) {
  const double end = now_s() + deadline_s();
  for (int spin = 0; !ready(); spin++) {
    if (c->seg->error.load(std::memory_order_acquire)) return fail(ncclRemoteError, "rank %lld: a peer reported an error", c->rank);
    if (now_s() > end) {
      c->seg->error.store(1, std::memory_order_release);
      snprintf(g_why, sizeof(g_why), "rank %d: deadline passed while waiting for %s", c->rank, what);
      fprintf(stderr, "fake_rccl: %s\n", g_why);
      return ncclSystemError;
    }
    if (spin > 64) sched_yield();
  }
  return ncclSuccess;
}

ncclResult_t barrier(ncclComm* c) {
  Seg* s = c->seg;
  c->sense ^= 1;
  const uint32_t mine = c->sense;
  if (s->bar_count.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) {
    s->bar_count.store(0, std::memory_order_relaxed);
    s->bar_sense.store(mine, std::memory_order_release);
    return ncclSuccess;
  }
  return wait_for(c, "the barrier", [&] { return s->bar_sense.load(std::memory_order_acquire) == mine; });
}

// a device range must lie inside ONE live allocation: an offset or count that runs past it would fault (or silently
// overwrite a neighbour) under the real library
ncclResult_t check_range(ncclComm* c, const void* p, size_t bytes, const char* what) {
  if (bytes == 0) return ncclSuccess;
  if (!p) return fail(ncclInvalidArgument, "rank %lld: NULL buffer", c->rank);
  hipDeviceptr_t base = nullptr;
  size_t size = 0;
  if (hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p) != hipSuccess) {
    (void)hipGetLastError();
    snprintf(g_why, sizeof(g_why), "rank %d: %s buffer %p is not device memory", c->rank, what, p);
    fprintf(stderr, "fake_rccl: %s\n", g_why);
    return ncclInvalidArgument;
  }
  const char* lo = (const char*)base;
  if ((const char*)p < lo || (const char*)p + bytes > lo + size) {
    snprintf(g_why, sizeof(g_why), "rank %d: %s range %p + %zu leaves its allocation (%p + %zu)", c->rank, what, p, bytes, base, size);
    fprintf(stderr, "fake_rccl: %s\n", g_why);
    return ncclInvalidArgument;
  }
  return ncclSuccess;
}

void msg_name(char* out, size_t n, ncclComm* c, const char* kind, int a, int b, uint64_t seq) {
  snprintf(out, n, "%s_%s_%d_%d_%llu", c->name, kind, a, b, (unsigned long long)seq);
}

// device -> a fresh shm object
ncclResult_t publish(ncclComm* c, const char* name, const void* d_src, size_t bytes) {
  int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
  if (fd < 0) return fail(ncclSystemError, "shm_open(create) failed, errno %lld", errno);
  if (bytes) {
    if (ftruncate(fd, (off_t)bytes) != 0) {
      close(fd);
      return fail(ncclSystemError, "ftruncate failed, errno %lld", errno);
    }
    void* m = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) {
      close(fd);
      return fail(ncclSystemError, "mmap failed, errno %lld", errno);
    }
    // through a heap buffer: the runtime pins what it copies into, and a pinned page of an shm object that is
    // unmapped and unlinked a moment later is not something to leave behind in it
    std::vector<char> bounce(bytes);
    hipError_t e = hipMemcpy(bounce.data(), d_src, bytes, hipMemcpyDeviceToHost);
    if (e == hipSuccess) memcpy(m, bounce.data(), bytes);
    munmap(m, bytes);
    if (e != hipSuccess) {
      close(fd);
      return fail(ncclUnhandledCudaError, "rank %lld: device -> host copy failed (%lld)", c->rank, (long long)e);
    }
  }
  close(fd);
  return ncclSuccess;
}

// a published shm object -> host mapping (caller unmaps)
ncclResult_t open_msg(const char* name, size_t bytes, void** out) {
  *out = nullptr;
  int fd = shm_open(name, O_RDONLY, 0);
  if (fd < 0) return fail(ncclSystemError, "shm_open(read) failed, errno %lld", errno);
  if (bytes) {
    void* m = mmap(nullptr, bytes, PROT_READ, MAP_SHARED, fd, 0);
    if (m == MAP_FAILED) {
      close(fd);
      return fail(ncclSystemError, "mmap failed, errno %lld", errno);
    }
    *out = m;
  }
  close(fd);
  return ncclSuccess;
}

ncclResult_t do_send(const Op& o) {
  ncclComm* c = o.comm;
  Box& b = c->seg->box[c->rank][o.peer];
  const uint64_t seq = c->sent[o.peer]++;
  ncclResult_t r = wait_for(c, "room in the ring", [&] { return seq - b.taken.load(std::memory_order_acquire) < (uint64_t)kRing; });
  if (r != ncclSuccess) return r;
  char name[128];
  msg_name(name, sizeof(name), c, "m", c->rank, o.peer, seq);
  r = publish(c, name, o.buf, o.count * dtype_size(o.dtype));
  if (r != ncclSuccess) return r;
  b.count[seq % kRing] = o.count;
  b.dtype[seq % kRing] = (int32_t)o.dtype;
  b.posted.store(seq + 1, std::memory_order_release);
  return ncclSuccess;
}

ncclResult_t do_recv(const Op& o) {
  ncclComm* c = o.comm;
  Box& b = c->seg->box[o.peer][c->rank];
  const uint64_t seq = c->recvd[o.peer]++;
  ncclResult_t r = wait_for(c, "a message", [&] { return b.posted.load(std::memory_order_acquire) > seq; });
  if (r != ncclSuccess) return r;
  const uint64_t cnt = b.count[seq % kRing];
  const int32_t dt = b.dtype[seq % kRing];
  char name[128];
  msg_name(name, sizeof(name), c, "m", o.peer, c->rank, seq);
  if (cnt != o.count || dt != (int32_t)o.dtype) {
    c->seg->error.store(1, std::memory_order_release);
    shm_unlink(name);
    snprintf(g_why, sizeof(g_why), "rank %d: message %llu from rank %d carries %llu elements of type %d, the receive asks for %zu of type %d",
             c->rank, (unsigned long long)seq, o.peer, (unsigned long long)cnt, dt, o.count, (int)o.dtype);
    fprintf(stderr, "fake_rccl: %s\n", g_why);
    return ncclInvalidArgument;
  }
  const size_t bytes = o.count * dtype_size(o.dtype);
  void* m = nullptr;
  r = open_msg(name, bytes, &m);
  shm_unlink(name);
  if (r != ncclSuccess) return r;
  hipError_t e = hipSuccess;
  if (bytes) {
    std::vector<char> bounce((const char*)m, (const char*)m + bytes);  // (see publish)
    e = hipMemcpy(o.buf, bounce.data(), bytes, hipMemcpyHostToDevice);
  }
  if (m) munmap(m, bytes);
  b.taken.store(seq + 1, std::memory_order_release);
  if (e != hipSuccess) return fail(ncclUnhandledCudaError, "rank %lld: host -> device copy failed (%lld)", c->rank, (long long)e);
  return ncclSuccess;
}

ncclResult_t run_ops(std::vector<Op>& ops) {
  // stream order: everything enqueued before the call / group is done before a byte moves
  for (const Op& o : ops)
    if (hipStreamSynchronize(o.stream) != hipSuccess) return fail(ncclUnhandledCudaError, "hipStreamSynchronize failed");
  // sends never block on the peer's progress (mailboxes), so all sends first, then the receives: no cyclic wait
  for (const Op& o : ops)
    if (o.send) {
      ncclResult_t r = do_send(o);
      if (r != ncclSuccess) return r;
    }
  for (const Op& o : ops)
    if (!o.send) {
      ncclResult_t r = do_recv(o);
      if (r != ncclSuccess) return r;
    }
  return ncclSuccess;
}

ncclResult_t post(bool send, void* buf, size_t count, ncclDataType_t dtype, int peer, ncclComm* c, hipStream_t stream) {
  if (!c) return fail(ncclInvalidArgument, "NULL communicator");
  if (peer < 0 || peer >= c->nranks) return fail(ncclInvalidArgument, "rank %lld: peer %lld of %lld", c->rank, peer, c->nranks);
  if (!dtype_size(dtype)) return fail(ncclInvalidArgument, "datatype %lld is not one the path uses", (long long)dtype);
  ncclResult_t r = check_range(c, buf, count * dtype_size(dtype), send ? "send" : "receive");
  if (r != ncclSuccess) return r;
  g_ops.push_back({send, buf, count, dtype, peer, c, stream});
  if (g_depth > 0) return ncclSuccess;
  r = run_ops(g_ops);
  g_ops.clear();
  return r;
}

template <class T>
void add_into(T* acc, const T* x, size_t n) {
  for (size_t i = 0; i < n; i++) acc[i] += x[i];
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return fail(ncclInvalidArgument, "NULL id");
  memset(id, 0, sizeof(*id));
  int fd = open("/dev/urandom", O_RDONLY);
  if (fd < 0 || read(fd, id->internal, 12) != 12) {
    if (fd >= 0) close(fd);
    return fail(ncclSystemError, "/dev/urandom");
  }
  close(fd);
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks)
    return fail(ncclInvalidArgument, "ncclCommInitRank: rank %lld of %lld", rank, nranks);
  ncclComm* c = new ncclComm();
  memset((void*)c, 0, sizeof(*c));
  const unsigned char* b = (const unsigned char*)id.internal;
  snprintf(c->name, sizeof(c->name), "/fakerccl_%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x%02x", b[0], b[1], b[2], b[3], b[4],
           b[5], b[6], b[7], b[8], b[9], b[10], b[11]);
  c->rank = rank;
  c->nranks = nranks;
  // whoever comes first creates the segment; a fresh shm object reads as zeros, which IS the initial state
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, sizeof(Seg)) != 0) {
    if (fd >= 0) close(fd);
    delete c;
    return fail(ncclSystemError, "ncclCommInitRank: shm_open / ftruncate, errno %lld", errno);
  }
  void* m = mmap(nullptr, sizeof(Seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (m == MAP_FAILED) {
    delete c;
    return fail(ncclSystemError, "ncclCommInitRank: mmap, errno %lld", errno);
  }
  c->seg = (Seg*)m;
  if (c->seg->rank_taken[rank].exchange(1)) {
    c->seg->error.store(1);
    munmap(m, sizeof(Seg));
    delete c;
    return fail(ncclInvalidArgument, "ncclCommInitRank: rank %lld joined twice", rank);
  }
  c->seg->arrived.fetch_add(1, std::memory_order_acq_rel);
  ncclResult_t r = wait_for(c, "all ranks to join", [&] { return c->seg->arrived.load(std::memory_order_acquire) >= (uint32_t)nranks; });
  if (r != ncclSuccess) {
    shm_unlink(c->name);
    munmap(m, sizeof(Seg));
    delete c;
    return r;
  }
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->seg->departed.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)c->nranks) shm_unlink(c->name);
  munmap(c->seg, sizeof(Seg));
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* count) {
  if (!c || !count) return fail(ncclInvalidArgument, "ncclCommCount: NULL");
  *count = c->nranks;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t c, int* rank) {
  if (!c || !rank) return fail(ncclInvalidArgument, "ncclCommUserRank: NULL");
  *rank = c->rank;
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t r) {
  static thread_local char buf[320];
  snprintf(buf, sizeof(buf), "fake_rccl error %d: %s", (int)r, g_why);
  return buf;
}

ncclResult_t ncclGroupStart() {
  g_depth++;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (g_depth <= 0) return fail(ncclInvalidUsage, "ncclGroupEnd without ncclGroupStart");
  if (--g_depth > 0) return ncclSuccess;
  ncclResult_t r = run_ops(g_ops);
  g_ops.clear();
  return r;
}

ncclResult_t ncclSend(const void* buf, size_t count, ncclDataType_t dtype, int peer, ncclComm_t c, hipStream_t stream) {
  return post(true, const_cast<void*>(buf), count, dtype, peer, c, stream);
}

ncclResult_t ncclRecv(void* buf, size_t count, ncclDataType_t dtype, int peer, ncclComm_t c, hipStream_t stream) {
  return post(false, buf, count, dtype, peer, c, stream);
}

ncclResult_t ncclAllReduce(const void* sendbuf, void* recvbuf, size_t count, ncclDataType_t dtype, ncclRedOp_t op, ncclComm_t c,
                           hipStream_t stream) {
  if (!c) return fail(ncclInvalidArgument, "NULL communicator");
  if (g_depth > 0) return fail(ncclInvalidUsage, "ncclAllReduce inside a group: not something the path does");
  if (op != ncclSum || !dtype_size(dtype)) return fail(ncclInvalidArgument, "ncclAllReduce: op %lld / datatype %lld", (long long)op, (long long)dtype);
  const size_t bytes = count * dtype_size(dtype);
  ncclResult_t r = check_range(c, sendbuf, bytes, "all-reduce send");
  if (r == ncclSuccess) r = check_range(c, recvbuf, bytes, "all-reduce receive");
  if (r != ncclSuccess) return r;
  if (hipStreamSynchronize(stream) != hipSuccess) return fail(ncclUnhandledCudaError, "hipStreamSynchronize failed");
  const uint64_t seq = c->ar_seq++;
  char name[128];
  msg_name(name, sizeof(name), c, "a", c->rank, 0, seq);
  r = publish(c, name, sendbuf, bytes);
  c->seg->ar_count[c->rank] = count;  // every rank must have called with the same count
  if (r == ncclSuccess) r = barrier(c);
  std::vector<char> acc(bytes, 0);
  for (int p = 0; r == ncclSuccess && p < c->nranks; p++) {  // rank order: identical bits everywhere
    if (c->seg->ar_count[p] != count) {
      c->seg->error.store(1, std::memory_order_release);
      r = fail(ncclInvalidArgument, "rank %lld: all-reduce of %lld elements, rank %lld brought %lld", c->rank, (long long)count, p,
               (long long)c->seg->ar_count[p]);
      break;
    }
    char pn[128];
    msg_name(pn, sizeof(pn), c, "a", p, 0, seq);
    void* m = nullptr;
    r = open_msg(pn, bytes, &m);
    if (r != ncclSuccess) break;
    if (bytes) {
      if (dtype == ncclFloat32) add_into((float*)acc.data(), (const float*)m, count);
      else add_into((double*)acc.data(), (const double*)m, count);
      munmap(m, bytes);
    }
  }
  if (r == ncclSuccess && bytes && hipMemcpy(recvbuf, acc.data(), bytes, hipMemcpyHostToDevice) != hipSuccess)
    r = fail(ncclUnhandledCudaError, "rank %lld: host -> device copy failed", c->rank);
  ncclResult_t rb = r == ncclSuccess ? barrier(c) : r;  // nobody unlinks what a peer still reads
  shm_unlink(name);
  return rb;
}

}  // extern "C"
