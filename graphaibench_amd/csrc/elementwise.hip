// elementwise.hip -- relu / dropout / loss / l2norm / adam / row gather for gfx950.
// All of these are HBM-streaming: 16 B per lane, grid capped at 2048 workgroups with a
// grid-stride loop (cdna guide, Guideline 11/13).
//
// replaces relu_gpu / d_relu_gpu / dropout_gpu / d_dropout_gpu / init_const_gpu /
// softmax_cross_entropy_gpu / d_softmax_cross_entropy_gpu / masked_avg_loss_gpu /
// masked_accuracy_single / l2norm / d_l2norm (src/utilities/math_functions.cu:12-14,
// 114-146, 158-196, 242-268, 516-564, 749-761, 886-942) and adam::update_gpu
// (src/utilities/optimizer.cu:5-36).
#include "common.h"

namespace {

typedef float f4 __attribute__((ext_vector_type(4)));

inline unsigned stream_grid(int64_t n_items, int block) {
  int64_t b = cdiv64(n_items > 0 ? n_items : 1, block);
  return (unsigned)(b < 2048 ? b : 2048);
}

__global__ void fill_kernel(int64_t n, float v, float* x) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = v;
}

__global__ void scale_kernel(int64_t n, float a, float* x) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) x[i] = a * x[i];
}

// relu_cpu / relu_kernel: max(x, 0)    (math_functions.cpp:442-451; .cu:242)
__global__ void relu_kernel(int64_t n, const float* in, float* out, int vec_ok) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec_ok) {
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += stride) {
      f4 v = reinterpret_cast<const f4*>(in)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = v[k] > 0.f ? v[k] : 0.f;
      reinterpret_cast<f4*>(out)[i] = v;
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = in[i] > 0.f ? in[i] : 0.f;
  } else {
    for (int64_t i = tid; i < n; i += stride) out[i] = in[i] > 0.f ? in[i] : 0.f;
  }
}

// d_relu: out = data > 0 ? in : 0     (math_functions.cpp:453-463; .cu:255)
__global__ void d_relu_kernel(int64_t n, const float* in, const float* data, float* out, int vec_ok) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec_ok) {
    const int64_t n4 = n >> 2;
    for (int64_t i = tid; i < n4; i += stride) {
      f4 g = reinterpret_cast<const f4*>(in)[i];
      f4 d = reinterpret_cast<const f4*>(data)[i];
#pragma unroll
      for (int k = 0; k < 4; ++k) g[k] = d[k] > 0.f ? g[k] : 0.f;
      reinterpret_cast<f4*>(out)[i] = g;
    }
    for (int64_t i = (n4 << 2) + tid; i < n; i += stride) out[i] = data[i] > 0.f ? in[i] : 0.f;
  } else {
    for (int64_t i = tid; i < n; i += stride) out[i] = data[i] > 0.f ? in[i] : 0.f;
  }
}

// counter-based uniform in [0,1): splitmix64 of (seed, index).  The reference draws from
// cuRAND XORWOW (GPU, math_functions.cu:123-132) or boost mt19937 (CPU, :390-429); masks are
// therefore not comparable across implementations, only their statistics and the replay.
__device__ __forceinline__ float u01(uint64_t seed, uint64_t i) {
  uint64_t z = seed + 0x9E3779B97F4A7C15ull * (i + 1);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z = z ^ (z >> 31);
  return (float)(z >> 40) * (1.0f / 16777216.0f);
}
// dropout_kernel: mask = rand > p; out = in * mask * scale   (math_functions.cu:114-121)
__global__ void dropout_kernel(int64_t n, float scale, float rate, uint64_t seed, const float* in,
                               uint8_t* masks, float* out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint8_t m = u01(seed, (uint64_t)i) > rate ? 1 : 0;
    masks[i] = m;
    out[i] = in[i] * (float)m * scale;
  }
}
__global__ void d_dropout_kernel(int64_t n, float scale, const float* in, const uint8_t* masks,
                                 float* out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    out[i] = in[i] * (float)masks[i] * scale;
}

// softmax + cross entropy per vertex (softmax_loss_layer.cpp:4-21; math_functions.cpp:485-494,
// 533-544).  One wave per vertex, classes across lanes (num_cls is small: 7..172).
__global__ __launch_bounds__(256) void softmax_xent_kernel(int num_cls, int64_t begin, int64_t end,
                                                           const float* in, const uint8_t* masks,
                                                           const uint8_t* labels, float* loss,
                                                           float* out) {
  const int64_t i = begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= end) return;
  if (masks && masks[i] != 1) return;
  const int lane = threadIdx.x & 63;
  const float* x = in + i * (int64_t)num_cls;
  float* y = out + i * (int64_t)num_cls;
  float mx = -INFINITY;
  for (int c = lane; c < num_cls; c += 64) mx = fmaxf(mx, x[c]);
  mx = wave_max(mx);
  float den = 0.f;
  for (int c = lane; c < num_cls; c += 64) den += expf(x[c] - mx);
  den = wave_sum(den);
  const int lab = labels[i];
  float pl = 0.f;
  for (int c = lane; c < num_cls; c += 64) {
    const float p = expf(x[c] - mx) / den;
    y[c] = p;
    if (c == lab) pl = p;
  }
  pl = wave_sum(pl);  // exactly one lane holds it
  if (lane == 0) loss[i] = (pl == 0.0f) ? -logf(1e-10f) : -logf(pl);
}

// (p - onehot) / (end - begin)   (softmax_loss_layer.cpp:23-37, Q8)
__global__ void d_softmax_xent_kernel(int num_cls, int64_t begin, int64_t end,
                                      const uint8_t* masks, const uint8_t* labels,
                                      const float* out, float* diff) {
  const int64_t n = (end - begin) * num_cls;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const double inv = (double)(end - begin);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const int64_t i = begin + t / num_cls;
    const int j = (int)(t % num_cls);
    if (masks && masks[i] != 1) continue;
    const float pred = out[i * num_cls + j];
    diff[i * num_cls + j] = (float)(((double)pred - (labels[i] == j ? 1.0 : 0.0)) / inv);
  }
}

// sigmoid + cross entropy per vertex for multi-label heads (sigmoid_loss_layer.cpp:4-17,
// math_functions.cpp:517-521,553-559): labels are [n x num_cls] 0/1 bytes.  One wave per
// vertex, classes across lanes; masked-out vertices of the range get loss 0 (the .cu layer
// zeroes d_losses first, sigmoid_loss_layer.cu:6).
__global__ __launch_bounds__(256) void sigmoid_xent_kernel(int num_cls, int64_t begin, int64_t end,
                                                           const float* in, const uint8_t* masks,
                                                           const uint8_t* labels, float* loss,
                                                           float* out) {
  const int64_t i = begin + (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= end) return;
  const int lane = threadIdx.x & 63;
  if (masks && masks[i] != 1) {
    if (lane == 0) loss[i] = 0.f;
    return;
  }
  const float* x = in + i * (int64_t)num_cls;
  const uint8_t* y = labels + i * (int64_t)num_cls;
  float* o = out + i * (int64_t)num_cls;
  float part = 0.f;
  for (int c = lane; c < num_cls; c += 64) {
    const float p = x[c];
    o[c] = (float)(1. / (1. + (double)expf(-p)));
    const float pos = p >= 0.f ? 1.f : 0.f;
    const float e = expf((float)((double)p - 2. * (double)p * (double)pos));
    part -= p * ((float)y[c] - pos) - logf((float)(1. + (double)e));
  }
  part = wave_sum(part);
  if (lane == 0) loss[i] = part;
}

// (sigmoid - label) / (end - begin)   (sigmoid_loss_layer.cpp:19-33)
__global__ void d_sigmoid_xent_kernel(int num_cls, int64_t begin, int64_t end,
                                      const uint8_t* masks, const uint8_t* labels,
                                      const float* out, float* diff) {
  const int64_t n = (end - begin) * num_cls;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const float inv = (float)(uint64_t)(end - begin);
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const int64_t i = begin + t / num_cls;
    if (masks && masks[i] != 1) continue;
    const int64_t idx = begin * num_cls + t;
    diff[idx] = (out[idx] - (float)labels[idx]) / inv;
  }
}

// tp / fp / fn over every (vertex, class) cell of the masked range (math_functions.cpp:580-603):
// integer counts, so the atomics are order-independent.
__global__ __launch_bounds__(256) void f1_counts_kernel(int num_cls, int64_t begin, int64_t end,
                                                        const uint8_t* masks, const float* preds,
                                                        const uint8_t* labels,
                                                        unsigned long long* counts) {
  const int64_t n = (end - begin) * num_cls;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  unsigned tp = 0, fp = 0, fn = 0;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += stride) {
    const int64_t i = begin + t / num_cls;
    if (masks && masks[i] != 1) continue;
    const int64_t idx = begin * num_cls + t;
    const bool hot = preds[idx] > 0.5f;
    const uint8_t y = labels[idx];
    tp += (y == 1 && hot);
    fp += (y == 0 && hot);
    fn += (y == 1 && !hot);
  }
  __shared__ unsigned s[3];
  if (threadIdx.x < 3) s[threadIdx.x] = 0;
  __syncthreads();
  tp = (unsigned)wave_sum_u32(tp);
  fp = (unsigned)wave_sum_u32(fp);
  fn = (unsigned)wave_sum_u32(fn);
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&s[0], tp);
    atomicAdd(&s[1], fp);
    atomicAdd(&s[2], fn);
  }
  __syncthreads();
  if (threadIdx.x < 3 && s[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (unsigned long long)s[threadIdx.x]);
}

// two-level deterministic masked reductions ------------------------------------------------
// partial[b] = {sum, count} over a strip; final sum over b in order on the host side of the
// C ABI (nblocks <= 1024 values).
__global__ __launch_bounds__(256) void masked_loss_partial_kernel(int64_t begin, int64_t end,
                                                                  const uint8_t* masks,
                                                                  const float* loss, float* part) {
  __shared__ float ssum[4], scnt[4];
  float s = 0.f, c = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += stride)
    if (!masks || masks[i] == 1) {
      s += loss[i];
      c += 1.f;
    }
  s = wave_sum(s);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) {
    ssum[threadIdx.x >> 6] = s;
    scnt[threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = ssum[0] + ssum[1] + ssum[2] + ssum[3];
    part[2 * blockIdx.x + 1] = scnt[0] + scnt[1] + scnt[2] + scnt[3];
  }
}

// argmax (first maximum, math_functions.cpp:127-137) == label ?   one thread per vertex
__global__ __launch_bounds__(256) void masked_acc_partial_kernel(int64_t begin, int64_t end,
                                                                 int num_cls, const uint8_t* masks,
                                                                 const float* preds,
                                                                 const uint8_t* labels, float* part) {
  __shared__ float ssum[4], scnt[4];
  float s = 0.f, c = 0.f;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += stride)
    if (!masks || masks[i] == 1) {
      const float* p = preds + i * (int64_t)num_cls;
      int best = -1;
      float mx = -INFINITY;
      for (int j = 0; j < num_cls; ++j)
        if (p[j] > mx) {
          mx = p[j];
          best = j;
        }
      if (best == (int)labels[i]) s += 1.f;
      c += 1.f;
    }
  s = wave_sum(s);
  c = wave_sum(c);
  if ((threadIdx.x & 63) == 0) {
    ssum[threadIdx.x >> 6] = s;
    scnt[threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[2 * blockIdx.x] = ssum[0] + ssum[1] + ssum[2] + ssum[3];
    part[2 * blockIdx.x + 1] = scnt[0] + scnt[1] + scnt[2] + scnt[3];
  }
}

// l2norm_layer forward/backward (src/layers/l2norm_layer.cpp:19-64), one wave per row
__global__ __launch_bounds__(256) void l2norm_kernel(int64_t n, int dim, const float* in, float* out) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const float* x = in + i * (int64_t)dim;
  float s = 0.f;
  for (int c = lane; c < dim; c += 64) s += x[c] * x[c];
  s = wave_sum(s);
  s = s < 1.0e-12f ? 1.0e-12f : s;
  s = sqrtf(s);
  for (int c = lane; c < dim; c += 64) out[i * (int64_t)dim + c] = x[c] / s;
}
__global__ __launch_bounds__(256) void d_l2norm_kernel(int64_t n, int dim, const float* fin,
                                                       const float* gin, float* gout) {
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= n) return;
  const int lane = threadIdx.x & 63;
  const float* x = fin + i * (int64_t)dim;
  const float* g = gin + i * (int64_t)dim;
  float s2 = 0.f, c0 = 0.f;
  for (int c = lane; c < dim; c += 64) {
    s2 += x[c] * x[c];
    c0 -= x[c] * g[c];
  }
  s2 = wave_sum(s2);
  c0 = wave_sum(c0);
  s2 = s2 < 1.0e-12f ? 1.0e-12f : s2;
  const float c1 = powf(s2, -1.5f);
  for (int c = lane; c < dim; c += 64)
    gout[i * (int64_t)dim + c] = x[c] * c0 * c1 + g[c] * s2 * c1;
}

// adam::update (src/utilities/optimizer.cpp:22-35; update_kernel optimizer.cu:5-15):
// eps inside the sqrt; the caller advances b1_t/b2_t once per call.
__global__ void adam_kernel(int64_t n, const float* dW, float* W, float* m, float* v, float alpha,
                            float b1, float b2, float b1_t, float b2_t, float eps) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float g = dW[i];
    const float mt = b1 * m[i] + (1.0f - b1) * g;
    const float vt = b2 * v[i] + (1.0f - b2) * g * g;
    m[i] = mt;
    v[i] = vt;
    W[i] -= alpha * (mt / (1.0f - b1_t)) / sqrtf((vt / (1.0f - b2_t)) + eps);
  }
}

// out[k,:] = in[idx[k],:]   (halo send-buffer packing).
// 16-B path: a wave copies GR_ROWS consecutive output rows, i.e. one contiguous piece of `out`; lane l of step u moves
// float4 number u*64 + l of that piece, so stores are whole 1-KB wave writes and GR_U row pieces are in flight per
// wave before the first store (one row per wave left every wave with a single dependent load: 4.0 TB/s read+write at
// 6.4 M rows of 512 B).  The source row of a piece comes from the wave's 16 row ids, held one per lane.
constexpr int GR_ROWS = 16;
constexpr int GR_U = 8;
// SCATTER: out[didx[k],:] = in[idx[k],:] -- the pack of a halo plan in SOURCE order (idx ascending, the same row up to
// once per peer back to back): a row that goes to several peers is read from HBM once and found in the cache by its
// repeats, where the destination-ordered pack read it once per peer (2.6 x at 8 ranks); the 512-B row pieces land in
// their (scattered) slots of the send buffer.
// SCATTER == 2: didx[k] is the ADDRESS of destination row k (a send buffer cut into separately allocated chunks: comm.hip)
template <int SCATTER>
__global__ __launch_bounds__(256) void gather_rows_vec_kernel(int64_t n_idx, const int64_t* idx, const int64_t* didx, int n4,
                                                              int shift, const f4* in, f4* out) {
  const int lane = threadIdx.x & 63;
  const int64_t k0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * GR_ROWS;
  if (k0 >= n_idx) return;
  const int rows = (n_idx - k0 < GR_ROWS) ? (int)(n_idx - k0) : GR_ROWS;
  const int64_t mine = idx[k0 + (lane < rows ? lane : rows - 1)];
  const int mine_lo = (int)(uint32_t)(mine & 0xffffffffll), mine_hi = (int)(mine >> 32);
  int dst_lo = 0, dst_hi = 0;
  if constexpr (SCATTER != 0) {
    const int64_t d = didx[k0 + (lane < rows ? lane : rows - 1)];
    dst_lo = (int)(uint32_t)(d & 0xffffffffll);
    dst_hi = (int)(d >> 32);
  }
  const int items = rows * n4;
  f4* dst = out + k0 * n4;
  for (int t0 = 0; t0 < items; t0 += 64 * GR_U) {
    f4 v[GR_U];
#pragma unroll
    for (int u = 0; u < GR_U; ++u) {
      int t = t0 + u * 64 + lane;
      if (t >= items) t = items - 1;  // clamped: every lane loads, only real items are stored
      const int r = shift >= 0 ? (t >> shift) : (t / n4);
      const int c = t - r * n4;
      const int64_t src = ((int64_t)__shfl(mine_hi, r) << 32) | (uint32_t)__shfl(mine_lo, r);
      v[u] = in[src * n4 + c];
    }
#pragma unroll
    for (int u = 0; u < GR_U; ++u) {
      const int t = t0 + u * 64 + lane;
      if constexpr (SCATTER != 0) {
        const int tc = t < items ? t : items - 1;
        const int r = shift >= 0 ? (tc >> shift) : (tc / n4);
        const int c = tc - r * n4;
        const int64_t drow = ((int64_t)__shfl(dst_hi, r) << 32) | (uint32_t)__shfl(dst_lo, r);
        if constexpr (SCATTER == 2) {
          if (t < items) reinterpret_cast<f4*>(drow)[c] = v[u];
        } else {
          if (t < items) out[drow * n4 + c] = v[u];
        }
      } else {
        if (t < items) dst[t] = v[u];
      }
    }
  }
}

// 4-B path (row length not a multiple of 4 floats, or unaligned tables): one wave per row
__global__ __launch_bounds__(256) void gather_rows_kernel(int64_t n_idx, const int64_t* idx, int len,
                                                          const float* in, float* out) {
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n_idx) return;
  const int lane = threadIdx.x & 63;
  const float* src = in + idx[k] * (int64_t)len;
  float* dst = out + k * (int64_t)len;
  for (int c = lane; c < len; c += 64) dst[c] = src[c];
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(int64_t n_idx, const int64_t* idx, const int64_t* didx, int len,
                                                           const float* in, float* out) {
  const int64_t k = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (k >= n_idx) return;
  const int lane = threadIdx.x & 63;
  const float* src = in + idx[k] * (int64_t)len;
  float* dst = out + didx[k] * (int64_t)len;
  for (int c = lane; c < len; c += 64) dst[c] = src[c];
}

int finish_partial(gaib_ctx* ctx, int nblocks, float* d_part, float* h_result) {
  GAIB_NOT_WHILE_CAPTURING(ctx, "a metric with a host result (use the *_dev form)");
  float h[2 * 1024];
  GAIB_HIP(hipMemcpyAsync(h, d_part, sizeof(float) * 2 * nblocks, hipMemcpyDeviceToHost, ctx->stream));
  GAIB_HIP(hipStreamSynchronize(ctx->stream));
  float s = 0.f, c = 0.f;
  for (int b = 0; b < nblocks; ++b) {
    s += h[2 * b];
    c += h[2 * b + 1];
  }
  *h_result = c > 0.f ? s / c : 0.f;
  return GAIB_OK;
}

// the device-side twin of finish_partial: the block partials are added in block order by ONE thread (the host loop's
// order and roundings), result[0] = sum / count
__global__ void finish_partial_kernel(int nblocks, const float* part, float* result) {
  __shared__ float sh[2 * 1024];
  for (int t = threadIdx.x; t < 2 * nblocks; t += blockDim.x) sh[t] = part[t];
  __syncthreads();
  if (threadIdx.x == 0) {
    float s = 0.f, c = 0.f;
    for (int b = 0; b < nblocks; ++b) {
      s += sh[2 * b];
      c += sh[2 * b + 1];
    }
    result[0] = c > 0.f ? s / c : 0.f;
  }
}

// bias_mv (math_functions.cu:207-221): x[i, j] += b[j]
__global__ void bias_add_kernel(int64_t total, int len, float* x, const float* b) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) x[i] = x[i] + b[i % len];
}
// reduce_sum (math_functions.cpp:246-262: a[j] = sum_i x[i, j]; the reference's CUDA kernel, math_functions.cu:224-229,
// adds into a[j] from every thread without atomics).  Two levels in a fixed order: block b sums its run of rows per
// column, then the block partials are added in block order -- the same bits on every run.
__global__ __launch_bounds__(256) void colsum_partial_kernel(int64_t n, int len, int64_t rows_per_block, const float* x,
                                                             float* part) {
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
  for (int j = threadIdx.x; j < len; j += blockDim.x) {
    float s = 0.f;
    for (int64_t i = r0; i < r1; ++i) s += x[i * len + j];
    part[(int64_t)blockIdx.x * len + j] = s;
  }
}
__global__ void colsum_finish_kernel(int nblocks, int len, const float* part, float* a) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= len) return;
  float s = 0.f;
  for (int b = 0; b < nblocks; ++b) s += part[(int64_t)b * len + j];
  a[j] = s;
}
// rng_uniform_gpu / gpu_rng_uniform (math_functions.cu:39-50): r = a + (b - a) * u, u uniform on [0, 1) from the
// library's counter RNG (the reference draws cuRAND XORWOW numbers: streams are not comparable, statistics are)
__global__ void rng_uniform_kernel(int64_t n, float a, float range, uint64_t seed, float* r) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) r[i] = a + range * u01(seed, (uint64_t)i);
}

// Adam with the beta powers in device memory (pw[0] = b1^t, pw[1] = b2^t): a recorded (HIP graph) step cannot take
// them by value.  adam_advance_kernel is the host's `b1_t *= b1; b2_t *= b2` (optimizer.cpp), same float products.
__global__ void adam_dev_kernel(int64_t n, const float* dW, float* W, float* m, float* v, float alpha, float b1,
                                float b2, const float* pw, float eps) {
  const float b1_t = pw[0], b2_t = pw[1];
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const float g = dW[i];
    const float mt = b1 * m[i] + (1.0f - b1) * g;
    const float vt = b2 * v[i] + (1.0f - b2) * g * g;
    m[i] = mt;
    v[i] = vt;
    W[i] -= alpha * (mt / (1.0f - b1_t)) / sqrtf((vt / (1.0f - b2_t)) + eps);
  }
}
__global__ void adam_advance_kernel(float* pw, float b1, float b2) {
  pw[0] *= b1;
  pw[1] *= b2;
}

}  // namespace

extern "C" int gaib_fill_f32(gaib_ctx* ctx, int64_t n, float value, float* d_x) {
  GAIB_CHECK(ctx && (d_x || n == 0), "gaib_fill_f32: NULL argument");
  if (n <= 0) return GAIB_OK;
  fill_kernel<<<stream_grid(n, 256), 256, 0, ctx->stream>>>(n, value, d_x);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_scale_f32(gaib_ctx* ctx, int64_t n, float alpha, float* d_x) {
  GAIB_CHECK(ctx && (d_x || n == 0), "gaib_scale_f32: NULL argument");
  if (n <= 0) return GAIB_OK;
  scale_kernel<<<stream_grid(n, 256), 256, 0, ctx->stream>>>(n, alpha, d_x);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_relu(gaib_ctx* ctx, int64_t n, const float* d_in, float* d_out) {
  GAIB_CHECK(ctx && ((d_in && d_out) || n == 0), "gaib_relu: NULL argument");
  if (n <= 0) return GAIB_OK;
  const int vec_ok = ((((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0);
  ProfScope ps(ctx, "relu", 8.0 * (double)n);
  relu_kernel<<<stream_grid(n / 4 + 1, 256), 256, 0, ctx->stream>>>(n, d_in, d_out, vec_ok);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_d_relu(gaib_ctx* ctx, int64_t n, const float* d_in_diff, const float* d_data,
                           float* d_out_diff) {
  GAIB_CHECK(ctx && ((d_in_diff && d_data && d_out_diff) || n == 0), "gaib_d_relu: NULL argument");
  if (n <= 0) return GAIB_OK;
  const int vec_ok = ((((uintptr_t)d_in_diff | (uintptr_t)d_data | (uintptr_t)d_out_diff) & 15) == 0);
  ProfScope ps(ctx, "d_relu", 12.0 * (double)n);
  d_relu_kernel<<<stream_grid(n / 4 + 1, 256), 256, 0, ctx->stream>>>(n, d_in_diff, d_data,
                                                                     d_out_diff, vec_ok);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_dropout(gaib_ctx* ctx, int64_t n, float scale, float drop_rate, uint64_t seed,
                            const float* d_in, uint8_t* d_masks, float* d_out) {
  GAIB_CHECK(ctx && ((d_in && d_masks && d_out) || n == 0), "gaib_dropout: NULL argument");
  GAIB_CHECK(drop_rate >= 0.f && drop_rate < 1.f, "gaib_dropout: rate must be in [0,1)");
  if (n <= 0) return GAIB_OK;
  dropout_kernel<<<stream_grid(n, 256), 256, 0, ctx->stream>>>(n, scale, drop_rate, seed, d_in,
                                                               d_masks, d_out);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_d_dropout(gaib_ctx* ctx, int64_t n, float scale, const float* d_in,
                              const uint8_t* d_masks, float* d_out) {
  GAIB_CHECK(ctx && ((d_in && d_masks && d_out) || n == 0), "gaib_d_dropout: NULL argument");
  if (n <= 0) return GAIB_OK;
  d_dropout_kernel<<<stream_grid(n, 256), 256, 0, ctx->stream>>>(n, scale, d_in, d_masks, d_out);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_softmax_xent(gaib_ctx* ctx, int num_cls, int64_t begin, int64_t end,
                                 const float* d_in, const uint8_t* d_masks, const uint8_t* d_labels,
                                 float* d_loss, float* d_out) {
  GAIB_CHECK(ctx && d_in && d_labels && d_loss && d_out, "gaib_softmax_xent: NULL argument");
  GAIB_CHECK(num_cls > 0 && begin >= 0 && end >= begin, "gaib_softmax_xent: bad range");
  if (end == begin) return GAIB_OK;
  softmax_xent_kernel<<<(unsigned)cdiv64(end - begin, 4), 256, 0, ctx->stream>>>(
      num_cls, begin, end, d_in, d_masks, d_labels, d_loss, d_out);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_d_softmax_xent(gaib_ctx* ctx, int num_cls, int64_t begin, int64_t end,
                                   const uint8_t* d_masks, const uint8_t* d_labels,
                                   const float* d_out, float* d_diff) {
  GAIB_CHECK(ctx && d_labels && d_out && d_diff, "gaib_d_softmax_xent: NULL argument");
  GAIB_CHECK(num_cls > 0 && begin >= 0 && end >= begin, "gaib_d_softmax_xent: bad range");
  if (end == begin) return GAIB_OK;
  d_softmax_xent_kernel<<<stream_grid((end - begin) * num_cls, 256), 256, 0, ctx->stream>>>(
      num_cls, begin, end, d_masks, d_labels, d_out, d_diff);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_sigmoid_xent(gaib_ctx* ctx, int num_cls, int64_t begin, int64_t end,
                                 const float* d_in, const uint8_t* d_masks, const uint8_t* d_labels,
                                 float* d_loss, float* d_out) {
  GAIB_CHECK(ctx && d_in && d_labels && d_loss && d_out, "gaib_sigmoid_xent: NULL argument");
  GAIB_CHECK(num_cls > 0 && begin >= 0 && end >= begin, "gaib_sigmoid_xent: bad range");
  if (end == begin) return GAIB_OK;
  sigmoid_xent_kernel<<<(unsigned)cdiv64(end - begin, 4), 256, 0, ctx->stream>>>(
      num_cls, begin, end, d_in, d_masks, d_labels, d_loss, d_out);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_d_sigmoid_xent(gaib_ctx* ctx, int num_cls, int64_t begin, int64_t end,
                                   const uint8_t* d_masks, const uint8_t* d_labels,
                                   const float* d_out, float* d_diff) {
  GAIB_CHECK(ctx && d_labels && d_out && d_diff, "gaib_d_sigmoid_xent: NULL argument");
  GAIB_CHECK(num_cls > 0 && begin >= 0 && end >= begin, "gaib_d_sigmoid_xent: bad range");
  if (end == begin) return GAIB_OK;
  d_sigmoid_xent_kernel<<<stream_grid((end - begin) * num_cls, 256), 256, 0, ctx->stream>>>(
      num_cls, begin, end, d_masks, d_labels, d_out, d_diff);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_masked_f1_micro(gaib_ctx* ctx, int64_t begin, int64_t end, int num_cls,
                                    const uint8_t* d_masks, const float* d_preds,
                                    const uint8_t* d_labels, float* h_result, int64_t* h_counts) {
  GAIB_CHECK(ctx && d_preds && d_labels && h_result, "gaib_masked_f1_micro: NULL argument");
  GAIB_CHECK(begin >= 0 && end >= begin && num_cls > 0, "gaib_masked_f1_micro: bad range");
  GAIB_NOT_WHILE_CAPTURING(ctx, "gaib_masked_f1_micro (use gaib_masked_f1_counts_dev)");
  *h_result = 0.f;
  unsigned long long c[3] = {0, 0, 0};
  if (end > begin) {
    GAIB_TRY(gaib_ws_reserve(ctx, sizeof(c)));
    GAIB_HIP(hipMemsetAsync(ctx->ws, 0, sizeof(c), ctx->stream));
    unsigned grid = stream_grid((end - begin) * num_cls, 256);
    f1_counts_kernel<<<grid, 256, 0, ctx->stream>>>(num_cls, begin, end, d_masks, d_preds, d_labels,
                                                    (unsigned long long*)ctx->ws);
    GAIB_LAUNCH_CHECK();
    GAIB_HIP(hipMemcpyAsync(c, ctx->ws, sizeof(c), hipMemcpyDeviceToHost, ctx->stream));
    GAIB_HIP(hipStreamSynchronize(ctx->stream));
  }
  const double tp = (double)c[0], fp = (double)c[1], fn = (double)c[2];
  const double prec = tp + fp > 0 ? tp / (tp + fp) : 0.;
  const double rec = tp + fn > 0 ? tp / (tp + fn) : 0.;
  *h_result = (float)(rec + prec > 0. ? 2. * (rec * prec) / (rec + prec) : 0.);
  if (h_counts) {
    h_counts[0] = (int64_t)c[0];
    h_counts[1] = (int64_t)c[1];
    h_counts[2] = (int64_t)c[2];
  }
  return GAIB_OK;
}

extern "C" int gaib_masked_avg_loss(gaib_ctx* ctx, int64_t begin, int64_t end,
                                    const uint8_t* d_masks, const float* d_loss, float* h_result) {
  GAIB_CHECK(ctx && d_loss && h_result, "gaib_masked_avg_loss: NULL argument");
  GAIB_CHECK(begin >= 0 && end >= begin, "gaib_masked_avg_loss: bad range");
  *h_result = 0.f;
  if (end == begin) return GAIB_OK;
  const int nblocks = (int)stream_grid(end - begin, 256) > 1024 ? 1024 : (int)stream_grid(end - begin, 256);
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * 2 * 1024));
  masked_loss_partial_kernel<<<nblocks, 256, 0, ctx->stream>>>(begin, end, d_masks, d_loss,
                                                               (float*)ctx->ws);
  GAIB_LAUNCH_CHECK();
  return finish_partial(ctx, nblocks, (float*)ctx->ws, h_result);
}

extern "C" int gaib_masked_accuracy_single(gaib_ctx* ctx, int64_t begin, int64_t end, int num_cls,
                                           const uint8_t* d_masks, const float* d_preds,
                                           const uint8_t* d_labels, float* h_result) {
  GAIB_CHECK(ctx && d_preds && d_labels && h_result, "gaib_masked_accuracy_single: NULL argument");
  GAIB_CHECK(begin >= 0 && end >= begin && num_cls > 0, "gaib_masked_accuracy_single: bad range");
  *h_result = 0.f;
  if (end == begin) return GAIB_OK;
  const int nblocks = (int)stream_grid(end - begin, 256) > 1024 ? 1024 : (int)stream_grid(end - begin, 256);
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * 2 * 1024));
  masked_acc_partial_kernel<<<nblocks, 256, 0, ctx->stream>>>(begin, end, num_cls, d_masks, d_preds,
                                                              d_labels, (float*)ctx->ws);
  GAIB_LAUNCH_CHECK();
  return finish_partial(ctx, nblocks, (float*)ctx->ws, h_result);
}

// tp, fp, fn of the multi-label head left in device memory (d_counts[3], 64-bit); the F1 is the caller's arithmetic
extern "C" int gaib_masked_f1_counts_dev(gaib_ctx* ctx, int64_t begin, int64_t end, int num_cls, const uint8_t* d_masks,
                                         const float* d_preds, const uint8_t* d_labels, uint64_t* d_counts) {
  GAIB_CHECK(ctx && d_preds && d_labels && d_counts, "gaib_masked_f1_counts_dev: NULL argument");
  GAIB_CHECK(begin >= 0 && end >= begin && num_cls > 0, "gaib_masked_f1_counts_dev: bad range");
  GAIB_HIP(hipMemsetAsync(d_counts, 0, 3 * sizeof(uint64_t), ctx->stream));
  if (end > begin) {
    unsigned grid = stream_grid((end - begin) * num_cls, 256);
    f1_counts_kernel<<<grid, 256, 0, ctx->stream>>>(num_cls, begin, end, d_masks, d_preds, d_labels,
                                                    (unsigned long long*)d_counts);
    GAIB_LAUNCH_CHECK();
  }
  return GAIB_OK;
}

// the two metrics with the result left in DEVICE memory (d_result[0]); no host wait, recordable
extern "C" int gaib_masked_avg_loss_dev(gaib_ctx* ctx, int64_t begin, int64_t end, const uint8_t* d_masks,
                                        const float* d_loss, float* d_result) {
  GAIB_CHECK(ctx && d_loss && d_result, "gaib_masked_avg_loss_dev: NULL argument");
  GAIB_CHECK(begin >= 0 && end >= begin, "gaib_masked_avg_loss_dev: bad range");
  if (end == begin) return gaib_fill_f32(ctx, 1, 0.f, d_result);
  const int nblocks = (int)stream_grid(end - begin, 256) > 1024 ? 1024 : (int)stream_grid(end - begin, 256);
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * 2 * 1024));
  masked_loss_partial_kernel<<<nblocks, 256, 0, ctx->stream>>>(begin, end, d_masks, d_loss, (float*)ctx->ws);
  GAIB_LAUNCH_CHECK();
  finish_partial_kernel<<<1, 256, 0, ctx->stream>>>(nblocks, (const float*)ctx->ws, d_result);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_masked_accuracy_single_dev(gaib_ctx* ctx, int64_t begin, int64_t end, int num_cls,
                                               const uint8_t* d_masks, const float* d_preds,
                                               const uint8_t* d_labels, float* d_result) {
  GAIB_CHECK(ctx && d_preds && d_labels && d_result, "gaib_masked_accuracy_single_dev: NULL argument");
  GAIB_CHECK(begin >= 0 && end >= begin && num_cls > 0, "gaib_masked_accuracy_single_dev: bad range");
  if (end == begin) return gaib_fill_f32(ctx, 1, 0.f, d_result);
  const int nblocks = (int)stream_grid(end - begin, 256) > 1024 ? 1024 : (int)stream_grid(end - begin, 256);
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * 2 * 1024));
  masked_acc_partial_kernel<<<nblocks, 256, 0, ctx->stream>>>(begin, end, num_cls, d_masks, d_preds, d_labels,
                                                              (float*)ctx->ws);
  GAIB_LAUNCH_CHECK();
  finish_partial_kernel<<<1, 256, 0, ctx->stream>>>(nblocks, (const float*)ctx->ws, d_result);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_l2norm(gaib_ctx* ctx, int64_t n, int dim, const float* d_in, float* d_out) {
  GAIB_CHECK(ctx && ((d_in && d_out) || n == 0), "gaib_l2norm: NULL argument");
  if (n <= 0 || dim <= 0) return GAIB_OK;
  l2norm_kernel<<<(unsigned)cdiv64(n, 4), 256, 0, ctx->stream>>>(n, dim, d_in, d_out);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_d_l2norm(gaib_ctx* ctx, int64_t n, int dim, const float* d_feat_in,
                             const float* d_grad_in, float* d_grad_out) {
  GAIB_CHECK(ctx && ((d_feat_in && d_grad_in && d_grad_out) || n == 0), "gaib_d_l2norm: NULL argument");
  if (n <= 0 || dim <= 0) return GAIB_OK;
  d_l2norm_kernel<<<(unsigned)cdiv64(n, 4), 256, 0, ctx->stream>>>(n, dim, d_feat_in, d_grad_in,
                                                                   d_grad_out);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_adam_step(gaib_ctx* ctx, int64_t n, const float* d_dW, float* d_W, float* d_m,
                              float* d_v, float alpha, float b1, float b2, float b1_t, float b2_t,
                              float eps) {
  GAIB_CHECK(ctx && ((d_dW && d_W && d_m && d_v) || n == 0), "gaib_adam_step: NULL argument");
  if (n <= 0) return GAIB_OK;
  adam_kernel<<<stream_grid(n, 256), 256, 0, ctx->stream>>>(n, d_dW, d_W, d_m, d_v, alpha, b1, b2,
                                                            b1_t, b2_t, eps);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_adam_step_dev(gaib_ctx* ctx, int64_t n, const float* d_dW, float* d_W, float* d_m, float* d_v,
                                  float alpha, float b1, float b2, float eps, float* d_pow) {
  GAIB_CHECK(ctx && d_pow && ((d_dW && d_W && d_m && d_v) || n == 0), "gaib_adam_step_dev: NULL argument");
  if (n > 0) {
    adam_dev_kernel<<<stream_grid(n, 256), 256, 0, ctx->stream>>>(n, d_dW, d_W, d_m, d_v, alpha, b1, b2, d_pow, eps);
    GAIB_LAUNCH_CHECK();
  }
  adam_advance_kernel<<<1, 1, 0, ctx->stream>>>(d_pow, b1, b2);  // once per call, also for an empty buffer (Q6)
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_gather_rows(gaib_ctx* ctx, int64_t n_idx, const int64_t* d_idx, int len,
                                const float* d_in, float* d_out) {
  GAIB_CHECK(ctx && ((d_idx && d_in && d_out) || n_idx == 0), "gaib_gather_rows: NULL argument");
  if (n_idx <= 0 || len <= 0) return GAIB_OK;
  ProfScope prof(ctx, "gather_rows");
  const bool vec_ok = (len % 4 == 0) && ((((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0);
  if (vec_ok) {
    const int n4 = len / 4;
    int shift = -1;
    for (int b = 0; b < 31; ++b)
      if ((1 << b) == n4) shift = b;
    gather_rows_vec_kernel<0><<<(unsigned)cdiv64(n_idx, 4 * GR_ROWS), 256, 0, ctx->stream>>>(
        n_idx, d_idx, nullptr, n4, shift, reinterpret_cast<const f4*>(d_in), reinterpret_cast<f4*>(d_out));
  } else {
    gather_rows_kernel<<<(unsigned)cdiv64(n_idx, 4), 256, 0, ctx->stream>>>(n_idx, d_idx, len, d_in, d_out);
  }
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

// out[dst_idx[k], :] = in[src_idx[k], :]  (see gather_rows_vec_kernel<SCATTER>); dst rows must be distinct
extern "C" int gaib_gather_scatter_rows(gaib_ctx* ctx, int64_t n_idx, const int64_t* d_src_idx, const int64_t* d_dst_idx,
                                        int len, const float* d_in, float* d_out) {
  GAIB_CHECK(ctx && ((d_src_idx && d_dst_idx && d_in && d_out) || n_idx == 0), "gaib_gather_scatter_rows: NULL argument");
  if (n_idx <= 0 || len <= 0) return GAIB_OK;
  ProfScope prof(ctx, "gather_rows");
  const bool vec_ok = (len % 4 == 0) && ((((uintptr_t)d_in | (uintptr_t)d_out) & 15) == 0);
  if (vec_ok) {
    const int n4 = len / 4;
    int shift = -1;
    for (int b = 0; b < 31; ++b)
      if ((1 << b) == n4) shift = b;
    gather_rows_vec_kernel<1><<<(unsigned)cdiv64(n_idx, 4 * GR_ROWS), 256, 0, ctx->stream>>>(
        n_idx, d_src_idx, d_dst_idx, n4, shift, reinterpret_cast<const f4*>(d_in), reinterpret_cast<f4*>(d_out));
  } else {
    scatter_rows_kernel<<<(unsigned)cdiv64(n_idx, 4), 256, 0, ctx->stream>>>(n_idx, d_src_idx, d_dst_idx, len, d_in, d_out);
  }
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

// *(float[len]*)d_dst_addr[k] = in[src_idx[k], :]: the source-ordered pack into a send buffer that is several allocations
// (internal: comm.hip).  GAIB_ERR_UNSUPPORTED where the 16-byte path does not apply -- the caller packs chunk by chunk then.
int gaib_gather_rows_to_addresses(gaib_ctx* ctx, int64_t n_idx, const int64_t* d_src_idx, const int64_t* d_dst_addr, int len,
                                  const float* d_in) {
  if (n_idx <= 0 || len <= 0) return GAIB_OK;
  if (len % 4 != 0 || (((uintptr_t)d_in) & 15) != 0) return GAIB_ERR_UNSUPPORTED;  // (chunk bases are 2-MiB aligned, rows 16 B)
  ProfScope prof(ctx, "gather_rows");
  const int n4 = len / 4;
  int shift = -1;
  for (int b = 0; b < 31; ++b)
    if ((1 << b) == n4) shift = b;
  gather_rows_vec_kernel<2><<<(unsigned)cdiv64(n_idx, 4 * GR_ROWS), 256, 0, ctx->stream>>>(
      n_idx, d_src_idx, d_dst_addr, n4, shift, reinterpret_cast<const f4*>(d_in), nullptr);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_bias_add(gaib_ctx* ctx, int64_t n, int len, float* d_x, const float* d_b) {
  GAIB_CHECK(ctx && n >= 0 && len >= 0, "gaib_bias_add: bad argument");
  if (n == 0 || len == 0) return GAIB_OK;
  GAIB_CHECK(d_x && d_b, "gaib_bias_add: NULL pointer");
  GAIB_HIP(hipSetDevice(ctx->device));
  bias_add_kernel<<<stream_grid(n * len, 256), 256, 0, ctx->stream>>>(n * len, len, d_x, d_b);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_colsum(gaib_ctx* ctx, int64_t n, int len, const float* d_x, float* d_a) {
  GAIB_CHECK(ctx && n >= 0 && len >= 0, "gaib_colsum: bad argument");
  if (len == 0) return GAIB_OK;
  GAIB_CHECK(d_a && (d_x || n == 0), "gaib_colsum: NULL pointer");
  GAIB_HIP(hipSetDevice(ctx->device));
  if (n == 0) return gaib_fill_f32(ctx, len, 0.f, d_a);
  int64_t nblocks = (n + 63) / 64;
  if (nblocks > 1024) nblocks = 1024;
  const int64_t rows_per_block = (n + nblocks - 1) / nblocks;
  nblocks = (n + rows_per_block - 1) / rows_per_block;
  GAIB_TRY(gaib_ws_reserve(ctx, sizeof(float) * (size_t)nblocks * (size_t)len));
  colsum_partial_kernel<<<(unsigned)nblocks, 256, 0, ctx->stream>>>(n, len, rows_per_block, d_x, (float*)ctx->ws);
  GAIB_LAUNCH_CHECK();
  colsum_finish_kernel<<<(len + 255) / 256, 256, 0, ctx->stream>>>((int)nblocks, len, (const float*)ctx->ws, d_a);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}

extern "C" int gaib_rng_uniform(gaib_ctx* ctx, int64_t n, float a, float b, uint64_t seed, float* d_r) {
  GAIB_CHECK(ctx && n >= 0 && (d_r || n == 0), "gaib_rng_uniform: bad argument");
  if (n == 0) return GAIB_OK;
  GAIB_HIP(hipSetDevice(ctx->device));
  rng_uniform_kernel<<<stream_grid(n, 256), 256, 0, ctx->stream>>>(n, a, b - a, seed, d_r);
  GAIB_LAUNCH_CHECK();
  return GAIB_OK;
}
