// math_functions.cpp -- the free-function math API of the layer path over include/gaib.h.
#include <random>
#include "math_functions.hh"
#include "host_util.h"
#include "lgraph.h"

using gaib_host::OpTimer;
static inline gaib_ctx* C() { return gpu_context::get(); }

// Glorot-uniform from libstdc++'s default engine, seeded per weight matrix: the only way to get
// the reference's exact initial weights (math_functions.cpp:11-18; golden: tests/golden/glorot_*).
void init_glorot(size_t dim_x, size_t dim_y, vec_t& weight, unsigned seed) {
  const float init_range = sqrt(6.0 / (dim_x + dim_y));
  std::default_random_engine rng(seed);
  std::uniform_real_distribution<float> dist(-init_range, init_range);
  weight.resize(dim_x * dim_y);
  for (size_t i = 0; i < dim_x * dim_y; ++i) weight[i] = dist(rng);
}

void matmul(const size_t x, const size_t y, const size_t z, const float_t* A, const float_t* B,
            float* C_, bool transA, bool transB, bool accum) {
  OpTimer t(OP_DENSEMM);
  GAIB_OR_DIE(gaib_sgemm(C(), transA, transB, (int64_t)x, (int64_t)y, (int64_t)z, A, B, accum, C_));
}

void matmul_relu(const size_t x, const size_t y, const size_t z, const float_t* A, const float_t* B,
                 float* C_, bool transA, bool transB, bool accum) {
  OpTimer t(OP_DENSEMM);
  GAIB_OR_DIE(gaib_sgemm_ex(C(), transA, transB, (int64_t)x, (int64_t)y, (int64_t)z, A, B,
                            (accum ? GAIB_ACCUMULATE : 0) | GAIB_RELU, C_));
}

void matmul_drelu(const size_t x, const size_t y, const size_t z, const float_t* A, float_t* G, const float_t* mask,
                  float* C_) {
  OpTimer t(OP_DENSEMM);
  GAIB_OR_DIE(gaib_sgemm_drelu(C(), (int64_t)x, (int64_t)y, (int64_t)z, A, G, mask, 0, C_));
}

void init_const_gpu(size_t n, float_t value, float_t* array) { GAIB_OR_DIE(gaib_fill_f32(C(), (int64_t)n, value, array)); }
void copy_gpu(size_t len, const float_t* in, float_t* out) {
  GAIB_OR_DIE(gaib_memcpy_d2d(C(), out, in, sizeof(float) * (size_t)len));
}
void relu_gpu(const size_t n, const float_t* in, float_t* out) {
  OpTimer t(OP_RELU);
  GAIB_OR_DIE(gaib_relu(C(), (int64_t)n, in, out));
}
void d_relu_gpu(const size_t n, const float_t* in_diff, const float_t* data, float_t* out_diff) {
  OpTimer t(OP_RELU);
  GAIB_OR_DIE(gaib_d_relu(C(), (int64_t)n, in_diff, data, out_diff));
}
static uint64_t g_dropout_seed = 0x5EED;
void dropout_gpu(size_t n, float scale, float drop_rate, const float* in, mask_t* masks, float* out) {
  OpTimer t(OP_DROPOUT);
  GAIB_OR_DIE(gaib_dropout(C(), (int64_t)n, scale, drop_rate, g_dropout_seed++, in, masks, out));
}
void d_dropout_gpu(size_t n, float scale, const float* in, const mask_t* masks, float* out) {
  OpTimer t(OP_DROPOUT);
  GAIB_OR_DIE(gaib_d_dropout(C(), (int64_t)n, scale, in, masks, out));
}
void l2norm(int n, int dim, const float* in, float* out) {
  OpTimer t(OP_NORM);
  GAIB_OR_DIE(gaib_l2norm(C(), n, dim, in, out));
}
void d_l2norm(int n, int dim, const float* feat_in, const float* grad_in, float* grad_out) {
  OpTimer t(OP_NORM);
  GAIB_OR_DIE(gaib_d_l2norm(C(), n, dim, feat_in, grad_in, grad_out));
}
void softmax_cross_entropy_gpu(int len, int begin, int end, const float_t* in_data, const mask_t* masks,
                               const label_t* labels, float_t* loss, float_t* out_data) {
  OpTimer t(OP_LOSS);
  GAIB_OR_DIE(gaib_softmax_xent(C(), len, begin, end, in_data, masks, labels, loss, out_data));
}
void d_softmax_cross_entropy_gpu(int len, int begin, int end, const mask_t* masks, const label_t* labels,
                                 const float_t* out_data, float_t* diff) {
  OpTimer t(OP_LOSS);
  GAIB_OR_DIE(gaib_d_softmax_xent(C(), len, begin, end, masks, labels, out_data, diff));
}
void sigmoid_cross_entropy_gpu(int len, int begin, int end, const float_t* in_data, const mask_t* masks,
                               const label_t* labels, float_t* loss, float_t* out_data) {
  GAIB_OR_DIE(gaib_sigmoid_xent(C(), len, begin, end, in_data, masks, labels, loss, out_data));
}
void d_sigmoid_cross_entropy_gpu(int len, int begin, int end, const mask_t* masks, const label_t* labels,
                                 const float_t* out_data, float_t* diff) {
  GAIB_OR_DIE(gaib_d_sigmoid_xent(C(), len, begin, end, masks, labels, out_data, diff));
}
// micro F1 over the masked range (math_functions.cu:1040-1044 -> masked_f1_score_gpu)
float masked_accuracy_multi(int begin, int end, int, int num_classes, mask_t* masks, float* preds,
                            label_t* ground_truth) {
  float r = 0.f;
  GAIB_OR_DIE(gaib_masked_f1_micro(C(), begin, end, num_classes, masks, preds, ground_truth, &r, NULL));
  return r;
}
acc_t masked_avg_loss_gpu(int begin, int end, int, mask_t* masks, float_t* loss) {
  float r = 0.f;
  GAIB_OR_DIE(gaib_masked_avg_loss(C(), begin, end, masks, loss, &r));
  return r;
}
float masked_accuracy_single(int begin, int end, int, int num_classes, mask_t* masks, float* preds,
                             label_t* ground_truth) {
  float r = 0.f;
  GAIB_OR_DIE(gaib_masked_accuracy_single(C(), begin, end, num_classes, masks, preds, ground_truth, &r));
  return r;
}
void symmetric_csr_transpose(LearningGraph& g, const float* A_nonzeros, float* B_nonzeros) {
  OpTimer t(OP_TRANSPOSE);
  GAIB_OR_DIE(gaib_edge_transpose(C(), g.device_graph(), A_nonzeros, B_nonzeros));
}

void bias_mv(int n, int len, float* x, float* b) {
  OpTimer t(OP_BIAS);
  GAIB_OR_DIE(gaib_bias_add(C(), n, len, x, b));
}
void reduce_sum(int n, int len, float* x, float* a) {
  OpTimer t(OP_REDUCE);
  GAIB_OR_DIE(gaib_colsum(C(), n, len, x, a));
}
void reduce_sum(int n, int len, float* x, vec_t& a) {
  float* d = gaib_host::dmalloc<float>((size_t)(len > 0 ? len : 1));
  reduce_sum(n, len, x, d);
  a.resize(len);
  if (len > 0) copy_float_host(len, d, a.data());
  GAIB_OR_DIE(gaib_free(C(), d));
}
void csr2csc(int nrows, int ncols, int nnz, const float* values, const int* rowptr, const int* colidx, float* valuesT,
             int* rowptrT, int* colidxT) {
  OpTimer t(OP_TRANSPOSE);
  GAIB_OR_DIE(gaib_csr2csc(C(), nrows, ncols, nnz, values, rowptr, colidx, valuesT, rowptrT, colidxT));
}
void spmm(size_t x, size_t y, size_t z, size_t nnz, float* A_nonzeros, int* A_idx_ptr, int* A_nnz_idx, const float* B,
          float* C_, float*, bool transA, bool transB, bool accum) {
  OpTimer t(OP_SPARSEMM);
  if (transB) {
    fprintf(stderr, "spmm: transB is not supported (no call site of the reference uses it)\n");
    exit(EXIT_FAILURE);
  }
  gaib_ctx* c = C();
  const int* rp = A_idx_ptr;
  const uint32_t* ci = reinterpret_cast<const uint32_t*>(A_nnz_idx);
  const float* val = A_nonzeros;
  int *rpT = NULL, *ciT = NULL;
  float* valT = NULL;
  size_t rows = x, cols = z;
  if (transA) {  // C[x*y] = A^T . B for A [z x x]: transpose once, then the same row-major aggregation
    rpT = gaib_host::dmalloc<int>(x + 1);
    ciT = gaib_host::dmalloc<int>(nnz > 0 ? nnz : 1);
    valT = gaib_host::dmalloc<float>(nnz > 0 ? nnz : 1);
    GAIB_OR_DIE(gaib_csr2csc(c, (int)z, (int)x, (int)nnz, A_nonzeros, A_idx_ptr, A_nnz_idx, valT, rpT, ciT));
    rp = rpT;
    ci = reinterpret_cast<const uint32_t*>(ciT);
    val = valT;
  }
  gaib_graph* g = NULL;
  GAIB_OR_DIE(gaib_graph_create_rect(c, (int64_t)rows, (int64_t)cols, (int64_t)nnz, rp, 32, ci, 1, &g));
  GAIB_OR_DIE(gaib_spmm_ex(c, g, GAIB_W_EDGE, val, (int)y, B, C_, accum ? GAIB_ACCUMULATE : 0));
  GAIB_OR_DIE(gaib_sync(c));  // the graph and the transposed arrays are released below
  GAIB_OR_DIE(gaib_graph_destroy(g));
  if (transA) {
    GAIB_OR_DIE(gaib_free(c, rpT));
    GAIB_OR_DIE(gaib_free(c, ciT));
    GAIB_OR_DIE(gaib_free(c, valT));
  }
}
static uint64_t g_rng_stream = 1;  // (the reference seeds cuRAND with 1, random.cpp:76-78)
void rng_uniform_gpu(size_t n, const float_t a, const float_t b, float_t* r) {
  GAIB_OR_DIE(gaib_rng_uniform(C(), (int64_t)n, a, b, g_rng_stream++, r));
}
void gpu_rng_uniform(size_t n, float* r) { rng_uniform_gpu(n, 0.f, 1.f, r); }

void float_malloc_device64(size_t n, float_t*& ptr) { ptr = gaib_host::dmalloc<float>(n); }
void float_malloc_device(int n, float_t*& ptr) { ptr = gaib_host::dmalloc<float>((size_t)n); }
void float_free_device(float_t*& ptr) { GAIB_OR_DIE(gaib_free(C(), ptr)); ptr = NULL; }
void copy_float_device(size_t n, float* h_ptr, float* d_ptr) { GAIB_OR_DIE(gaib_memcpy_h2d(C(), d_ptr, h_ptr, sizeof(float) * (size_t)n)); }
void copy_float_host(int n, const float* d_ptr, float* h_ptr) { GAIB_OR_DIE(gaib_memcpy_d2h(C(), h_ptr, d_ptr, sizeof(float) * (size_t)n)); }
void uint_malloc_device(size_t n, uint32_t*& ptr) { ptr = gaib_host::dmalloc<uint32_t>((size_t)n); }
void uint_free_device(uint32_t*& ptr) { GAIB_OR_DIE(gaib_free(C(), ptr)); ptr = NULL; }
void copy_uint_device(size_t n, uint32_t* h_ptr, uint32_t* d_ptr) { GAIB_OR_DIE(gaib_memcpy_h2d(C(), d_ptr, h_ptr, sizeof(uint32_t) * (size_t)n)); }
void uint8_malloc_device(size_t n, uint8_t*& ptr) { ptr = gaib_host::dmalloc<uint8_t>((size_t)n); }
void uint8_free_device(uint8_t*& ptr) { GAIB_OR_DIE(gaib_free(C(), ptr)); ptr = NULL; }
void copy_uint8_device(size_t n, uint8_t* h_ptr, uint8_t* d_ptr) { GAIB_OR_DIE(gaib_memcpy_h2d(C(), d_ptr, h_ptr, (size_t)n)); }
void copy_masks_device(int n, mask_t* h_masks, mask_t*& d_masks) {
  uint8_malloc_device(n, d_masks);
  copy_uint8_device(n, h_masks, d_masks);
}
