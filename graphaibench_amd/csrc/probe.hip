// probe.hip -- bandwidth probes behind include/gaib.h (measurement only): the chip's streaming-copy rate
// that bench.py reports next to the 8 TB/s spec peak, and the peer-to-peer xGMI link rate
// (the reference's link probe: src/test/test_nvlink.cu:37-77 -- there a cudaMemcpyAsync between two devices,
// 2^26 bytes, 10 repeats, one event pair).
#include <stdlib.h>
#include "common.h"

// 16 B per lane; a workgroup moves contiguous 16-KB tiles (256 lanes x 4 float4, all four loads in flight before the
// first store).  NT = non-temporal loads and stores (a stream that is read and written once should not evict the
// Infinity Cache's contents).
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT>
__global__ __launch_bounds__(256) void stream_copy_kernel(const v4f* __restrict__ src, v4f* __restrict__ dst,
                                                          int64_t n16) {
  const int64_t tile = 1024;  // float4 per tile
  const int64_t n_tiles = n16 / tile;
  for (int64_t t = blockIdx.x; t < n_tiles; t += gridDim.x) {
    const v4f* s = src + t * tile + threadIdx.x;
    v4f* d = dst + t * tile + threadIdx.x;
    v4f a, b, c, e;
    if (NT) {
      a = __builtin_nontemporal_load(s);
      b = __builtin_nontemporal_load(s + 256);
      c = __builtin_nontemporal_load(s + 512);
      e = __builtin_nontemporal_load(s + 768);
      __builtin_nontemporal_store(a, d);
      __builtin_nontemporal_store(b, d + 256);
      __builtin_nontemporal_store(c, d + 512);
      __builtin_nontemporal_store(e, d + 768);
    } else {
      a = s[0], b = s[256], c = s[512], e = s[768];
      d[0] = a, d[256] = b, d[512] = c, d[768] = e;
    }
  }
  // tail (< one tile)
  for (int64_t i = n_tiles * tile + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
       i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = src[i];
}

extern "C" int gaib_probe_stream_copy(gaib_ctx* ctx, size_t bytes, int iters, double* h_gbs) {
  GAIB_CHECK(ctx && h_gbs, "gaib_probe_stream_copy: NULL argument");
  GAIB_CHECK(bytes >= 16 && iters >= 1 && iters <= 10000, "gaib_probe_stream_copy: bytes >= 16, 1 <= iters <= 10000");
  GAIB_HIP(hipSetDevice(ctx->device));
  const int64_t n16 = (int64_t)(bytes / 16);
  v4f *src = nullptr, *dst = nullptr;
  hipEvent_t a = nullptr, b = nullptr;
  int rc = GAIB_OK;
  hipError_t e = hipMalloc((void**)&src, (size_t)n16 * 16);
  if (e == hipSuccess) e = hipMalloc((void**)&dst, (size_t)n16 * 16);
  if (e != hipSuccess) {
    gaib_set_error("gaib_probe_stream_copy: hipMalloc(2 x %zu) failed: %s", (size_t)n16 * 16, hipGetErrorString(e));
    if (src) (void)hipFree(src);
    return GAIB_ERR_NOMEM;
  }
  const int block = 256;
  const bool verbose = getenv("GAIB_PROBE_VERBOSE") != nullptr;
  double best = 0.0;
  do {
    if ((e = hipMemsetAsync(src, 0x3c, (size_t)n16 * 16, ctx->stream)) != hipSuccess) break;
    if ((e = hipEventCreate(&a)) != hipSuccess || (e = hipEventCreate(&b)) != hipSuccess) break;
    // the best of a few launch shapes is "what the chip streams": workgroups per CU x cache policy
    const int per_cu[] = {4, 8, 16, 32};
    for (int nt = 0; nt < 2 && e == hipSuccess; nt++)
      for (int v = 0; v < 4 && e == hipSuccess; v++) {
        int64_t grid = (int64_t)ctx->num_cus * per_cu[v];
        if (grid > cdiv64(n16, 1024)) grid = cdiv64(n16, 1024);
        if (grid < 1) grid = 1;
        auto launch = [&]() {
          if (nt)
            stream_copy_kernel<true><<<(unsigned)grid, block, 0, ctx->stream>>>(src, dst, n16);
          else
            stream_copy_kernel<false><<<(unsigned)grid, block, 0, ctx->stream>>>(src, dst, n16);
        };
        launch();  // untimed first touch
        if ((e = hipEventRecord(a, ctx->stream)) != hipSuccess) break;
        for (int it = 0; it < iters; it++) launch();
        if ((e = hipGetLastError()) != hipSuccess) break;
        if ((e = hipEventRecord(b, ctx->stream)) != hipSuccess) break;
        if ((e = hipEventSynchronize(b)) != hipSuccess) break;
        float ms = 0.f;
        if ((e = hipEventElapsedTime(&ms, a, b)) != hipSuccess) break;
        const double gbs = 2.0 * (double)n16 * 16.0 * iters / ((double)ms * 1e-3) / 1e9;
        if (verbose) fprintf(stderr, "[gaib probe] stream copy %s, %d WG/CU: %.0f GB/s\n", nt ? "nt" : "default", per_cu[v], gbs);
        if (gbs > best) best = gbs;
      }
    *h_gbs = best;
  } while (0);
  if (e != hipSuccess) {
    gaib_set_error("gaib_probe_stream_copy: %s", hipGetErrorString(e));
    rc = GAIB_ERR_HIP;
  }
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  (void)hipFree(src);
  (void)hipFree(dst);
  return rc;
}

extern "C" int gaib_probe_peer_copy(int src_dev, int dst_dev, size_t bytes, int iters, int bidir, double* h_gbs) {
  GAIB_CHECK(h_gbs, "gaib_probe_peer_copy: h_gbs is NULL");
  GAIB_CHECK(bytes >= 1 && iters >= 1 && iters <= 10000, "gaib_probe_peer_copy: bytes >= 1, 1 <= iters <= 10000");
  int ndev = 0, prev = 0;
  GAIB_HIP(hipGetDeviceCount(&ndev));
  GAIB_CHECK(src_dev >= 0 && src_dev < ndev && dst_dev >= 0 && dst_dev < ndev && src_dev != dst_dev,
             "gaib_probe_peer_copy: devices %d -> %d (%d visible): two different visible devices are needed", src_dev,
             dst_dev, ndev);
  GAIB_HIP(hipGetDevice(&prev));
  int can01 = 0, can10 = 0;
  GAIB_HIP(hipDeviceCanAccessPeer(&can01, src_dev, dst_dev));
  GAIB_HIP(hipDeviceCanAccessPeer(&can10, dst_dev, src_dev));
  void *p_src = nullptr, *p_dst = nullptr;
  hipStream_t s0 = nullptr, s1 = nullptr;
  hipEvent_t a = nullptr, b = nullptr, b1 = nullptr;
  hipError_t e = hipSuccess;
  int rc = GAIB_OK;
  do {
    if ((e = hipSetDevice(dst_dev)) != hipSuccess) break;
    if (can10) {
      e = hipDeviceEnablePeerAccess(src_dev, 0);
      if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess, (void)hipGetLastError();
      if (e != hipSuccess) break;
    }
    if ((e = hipMalloc(&p_dst, bytes)) != hipSuccess) break;
    if ((e = hipSetDevice(src_dev)) != hipSuccess) break;
    if (can01) {
      e = hipDeviceEnablePeerAccess(dst_dev, 0);
      if (e == hipErrorPeerAccessAlreadyEnabled) e = hipSuccess, (void)hipGetLastError();
      if (e != hipSuccess) break;
    }
    if ((e = hipMalloc(&p_src, bytes)) != hipSuccess) break;
    if ((e = hipMemset(p_src, 0x5a, bytes)) != hipSuccess) break;
    if ((e = hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)) != hipSuccess) break;
    if ((e = hipEventCreate(&a)) != hipSuccess || (e = hipEventCreate(&b)) != hipSuccess) break;
    if (bidir) {
      if ((e = hipSetDevice(dst_dev)) != hipSuccess) break;
      if ((e = hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)) != hipSuccess) break;
      if ((e = hipEventCreate(&b1)) != hipSuccess) break;
      if ((e = hipSetDevice(src_dev)) != hipSuccess) break;
    }
    // untimed first copy (page tables / link set-up)
    if ((e = hipMemcpyPeerAsync(p_dst, dst_dev, p_src, src_dev, bytes, s0)) != hipSuccess) break;
    if ((e = hipStreamSynchronize(s0)) != hipSuccess) break;
    if ((e = hipEventRecord(a, s0)) != hipSuccess) break;
    for (int it = 0; it < iters && e == hipSuccess; it++) {
      e = hipMemcpyPeerAsync(p_dst, dst_dev, p_src, src_dev, bytes, s0);
      if (bidir && e == hipSuccess) e = hipMemcpyPeerAsync(p_src, src_dev, p_dst, dst_dev, bytes, s1);
    }
    if (e != hipSuccess) break;
    if ((e = hipEventRecord(b, s0)) != hipSuccess) break;
    if (bidir && (e = hipEventRecord(b1, s1)) != hipSuccess) break;
    if ((e = hipEventSynchronize(b)) != hipSuccess) break;
    if (bidir && (e = hipEventSynchronize(b1)) != hipSuccess) break;
    float ms = 0.f;
    if ((e = hipEventElapsedTime(&ms, a, b)) != hipSuccess) break;
    *h_gbs = (double)bytes * iters / ((double)ms * 1e-3) / 1e9;
  } while (0);
  if (e != hipSuccess) {
    gaib_set_error("gaib_probe_peer_copy(%d -> %d, peer access %d/%d): %s", src_dev, dst_dev, can01, can10,
                   hipGetErrorString(e));
    rc = GAIB_ERR_HIP;
  }
  if (a) (void)hipEventDestroy(a);
  if (b) (void)hipEventDestroy(b);
  if (b1) (void)hipEventDestroy(b1);
  if (s0) (void)hipStreamDestroy(s0);
  if (s1) (void)hipStreamDestroy(s1);
  if (p_src) (void)hipFree(p_src);
  if (p_dst) (void)hipFree(p_dst);
  (void)hipSetDevice(prev);
  return rc;
}
