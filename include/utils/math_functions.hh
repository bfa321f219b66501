// include/utils/math_functions.hh -- free-function math API of the layer path, device pointers.
// Same names and argument meaning as the reference's GPU build (include/utils/math_functions.hh:
// 14-174, src/utilities/math_functions.cu); every function is a thin wrapper over include/gaib.h.
// Errors follow the reference convention: message on stderr, then exit (cutils.h:18-28).
#pragma once
#include "global.h"

void init_glorot(size_t dim_x, size_t dim_y, vec_t& weight, unsigned seed);  // host, libstdc++ RNG

// C[x*y] (=|+=) op(A)[x*z] . op(B)[z*y], row-major, device pointers (math_functions.cpp:166-171)
void matmul(const size_t x, const size_t y, const size_t z, const float_t* A, const float_t* B,
            float* C, bool transA = false, bool transB = false, bool accum = false);

// extension: the same product with relu_gpu fused into the epilogue (C = max(0, ...))
void matmul_relu(const size_t x, const size_t y, const size_t z, const float_t* A, const float_t* B,
                 float* C, bool transA = false, bool transB = false, bool accum = false);

// extension: d_relu_gpu(G, mask) in place fused into the weight-gradient product C = A^T . G  (A [z x x], G / mask [z x y])
void matmul_drelu(const size_t x, const size_t y, const size_t z, const float_t* A, float_t* G, const float_t* mask,
                  float* C);

void init_const_gpu(size_t n, float_t value, float_t* array);  // element counts are size_t: N*D exceeds int at scale
void copy_gpu(size_t len, const float_t* in, float_t* out);
void relu_gpu(const size_t n, const float_t* in, float_t* out);
void d_relu_gpu(const size_t n, const float_t* in_diff, const float_t* data, float_t* out_diff);
void dropout_gpu(size_t n, float scale, float drop_rate, const float* in, mask_t* masks, float* out);
void d_dropout_gpu(size_t n, float scale, const float* in, const mask_t* masks, float* out);
void l2norm(int n, int dim, const float* in, float* out);
void d_l2norm(int n, int dim, const float* feat_in, const float* grad_in, float* grad_out);
void softmax_cross_entropy_gpu(int len, int begin, int end, const float_t* in_data,
                               const mask_t* masks, const label_t* labels, float_t* loss,
                               float_t* out_data);
void d_softmax_cross_entropy_gpu(int len, int begin, int end, const mask_t* masks,
                                 const label_t* labels, const float_t* out_data, float_t* diff);
// multi-label head (reference math_functions.hh:115-121)
void sigmoid_cross_entropy_gpu(int len, int begin, int end, const float_t* in_data,
                               const mask_t* masks, const label_t* labels, float_t* loss,
                               float_t* out_data);
void d_sigmoid_cross_entropy_gpu(int len, int begin, int end, const mask_t* masks,
                                 const label_t* labels, const float_t* out_data, float_t* diff);
acc_t masked_avg_loss_gpu(int begin, int end, int count, mask_t* masks, float_t* loss);
float masked_accuracy_multi(int begin, int end, int count, int num_classes, mask_t* masks, float* preds,
                            label_t* ground_truth);
float masked_accuracy_single(int begin, int end, int count, int num_classes, mask_t* masks,
                             float* preds, label_t* ground_truth);
// bias and its gradient (reference math_functions.hh:36-38; used by the layers when is_bias: gcn_layer.cu:23,34,
// dense_layer.cpp:48,62).  x [n x len], b / a [len], DEVICE pointers; the vec_t overload receives the sums on the host.
void bias_mv(int n, int len, float* x, float* b);
void reduce_sum(int n, int len, float* x, float* a);
void reduce_sum(int n, int len, float* x, vec_t& a);
// transpose of a general CSR matrix, 32-bit offsets, device arrays (reference math_functions.hh:45, gat_aggregator.cu:89)
void csr2csc(int nrows, int ncols, int nnz, const float* values, const int* rowptr, const int* colidx, float* valuesT,
             int* rowptrT, int* colidxT);
// C[x*y] (=|+=) op(A)[x*z] . B[z*y], A sparse in CSR with 32-bit offsets, device arrays (reference math_functions.hh:54-57;
// csrmm_gpu, math_functions.cu:419-429).  temp is unused (the reference's cuSPARSE path transposes through it); transB must
// be false, as in the reference's own call sites
void spmm(size_t x, size_t y, size_t z, size_t nnz, float* A_nonzeros, int* A_idx_ptr, int* A_nnz_idx, const float* B,
          float* C, float* temp = NULL, bool transA = false, bool transB = false, bool accum = false);
// uniform random numbers on the device (reference math_functions.hh:156,174): [a, b) and [0, 1); a process-wide stream
// counter stands in for cuRAND's generator state
void rng_uniform_gpu(size_t n, const float_t a, const float_t b, float_t* r);
void gpu_rng_uniform(size_t n, float* r);
// symmetric sparse transpose of per-edge values on the device graph (math_functions.cpp:46-74)
class LearningGraph;
void symmetric_csr_transpose(LearningGraph& g, const float* A_nonzeros, float* B_nonzeros);

// memory helpers (math_functions.hh:161-173)
void float_malloc_device(int n, float_t*& ptr);
void float_free_device(float_t*& ptr);
void copy_float_device(size_t n, float* h_ptr, float* d_ptr);
void copy_float_host(int n, const float* d_ptr, float* h_ptr);
void uint_malloc_device(size_t n, uint32_t*& ptr);
void uint_free_device(uint32_t*& ptr);
void copy_uint_device(size_t n, uint32_t* h_ptr, uint32_t* d_ptr);
void uint8_malloc_device(size_t n, uint8_t*& ptr);
void uint8_free_device(uint8_t*& ptr);
void copy_uint8_device(size_t n, uint8_t* h_ptr, uint8_t* d_ptr);
void copy_masks_device(int n, mask_t* h_masks, mask_t*& d_masks);
// 64-bit element counts (N*D of the large graphs exceeds int)
void float_malloc_device64(size_t n, float_t*& ptr);
