// mirror_math_main.cpp -- test program: "a layer written against the reference's GPU-build API" for the six free
// functions of include/utils/math_functions.hh that the reference's GPU layers call besides matmul & co:
// bias_mv, reduce_sum (both overloads), csr2csc, spmm, rng_uniform_gpu, gpu_rng_uniform
// (reference include/utils/math_functions.hh:36-38,45,54,156,174; call sites gcn_layer.cu:23,34, dense_layer.cpp:48,62,
// gat_aggregator.cu:43,89, graph_conv_layer.cu:15).  Compiled by tests/test_mirror_math.py against include/ and the two
// libraries only -- no gaib.h call below, only the reference's names -- run on the GPU box, outputs compared in Python.
//   usage: mirror_math <dir>      reads <dir>/*.bin (written by the test), writes <dir>/out_*.bin
#include <stdio.h>
#include <stdlib.h>
#include <string>
#include <vector>
#include "math_functions.hh"

template <typename T>
static std::vector<T> rd(const std::string& p) {
  FILE* f = fopen(p.c_str(), "rb");
  if (!f) { perror(p.c_str()); exit(2); }
  fseek(f, 0, SEEK_END);
  const long n = ftell(f);
  fseek(f, 0, SEEK_SET);
  std::vector<T> v(n / sizeof(T));
  if (fread(v.data(), 1, n, f) != (size_t)n) exit(2);
  fclose(f);
  return v;
}
template <typename T>
static void wr(const std::string& p, const T* d, size_t n) {
  FILE* f = fopen(p.c_str(), "wb");
  if (!f) { perror(p.c_str()); exit(2); }
  fwrite(d, sizeof(T), n, f);
  fclose(f);
}
static float* up(const std::vector<float>& h) {
  float* d = NULL;
  float_malloc_device((int)h.size(), d);
  copy_float_device(h.size(), const_cast<float*>(h.data()), d);
  return d;
}
static int* up_i(const std::vector<int>& h) {
  uint32_t* d = NULL;
  uint_malloc_device(h.size(), d);
  copy_uint_device(h.size(), reinterpret_cast<uint32_t*>(const_cast<int*>(h.data())), d);
  return reinterpret_cast<int*>(d);
}
static std::vector<float> down(const float* d, size_t n) {
  std::vector<float> h(n);
  copy_float_host((int)n, d, h.data());
  return h;
}
static std::vector<int> down_i(const int* d, size_t n) {  // (ints travel through the float copy: same bytes)
  std::vector<int> h(n);
  copy_float_host((int)n, reinterpret_cast<const float*>(d), reinterpret_cast<float*>(h.data()));
  return h;
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  const std::string dir = std::string(argv[1]) + "/";
  const std::vector<int> dims = rd<int>(dir + "dims.bin");  // n len | nrows ncols nnz y
  const int n = dims[0], len = dims[1], nrows = dims[2], ncols = dims[3], nnz = dims[4], y = dims[5];
  // bias_mv / reduce_sum
  float* x = up(rd<float>(dir + "x.bin"));
  float* b = up(rd<float>(dir + "b.bin"));
  float* a = NULL;
  float_malloc_device(len, a);
  reduce_sum(n, len, x, a);
  wr(dir + "out_colsum.bin", down(a, len).data(), len);
  vec_t ah;
  reduce_sum(n, len, x, ah);
  wr(dir + "out_colsum_host.bin", ah.data(), ah.size());
  bias_mv(n, len, x, b);
  wr(dir + "out_bias.bin", down(x, (size_t)n * len).data(), (size_t)n * len);
  // csr2csc
  int* rp = up_i(rd<int>(dir + "rowptr.bin"));
  int* ci = up_i(rd<int>(dir + "colidx.bin"));
  float* val = up(rd<float>(dir + "vals.bin"));
  float* valT = NULL;
  float_malloc_device(nnz, valT);
  uint32_t *rpT = NULL, *ciT = NULL;
  uint_malloc_device(ncols + 1, rpT);
  uint_malloc_device(nnz, ciT);
  csr2csc(nrows, ncols, nnz, val, rp, ci, valT, (int*)rpT, (int*)ciT);
  wr(dir + "out_valT.bin", down(valT, nnz).data(), nnz);
  wr(dir + "out_rpT.bin", down_i((int*)rpT, ncols + 1).data(), ncols + 1);
  wr(dir + "out_ciT.bin", down_i((int*)ciT, nnz).data(), nnz);
  // spmm: C = A . B ; C += A . B ; Ct = A^T . Bt
  float* B = up(rd<float>(dir + "B.bin"));    // [ncols x y]
  float* Bt = up(rd<float>(dir + "Bt.bin"));  // [nrows x y]
  float *Cm = NULL, *Ct = NULL;
  float_malloc_device(nrows * y, Cm);
  float_malloc_device(ncols * y, Ct);
  spmm(nrows, y, ncols, nnz, val, rp, ci, B, Cm);
  wr(dir + "out_C.bin", down(Cm, (size_t)nrows * y).data(), (size_t)nrows * y);
  spmm(nrows, y, ncols, nnz, val, rp, ci, B, Cm, NULL, false, false, true);
  wr(dir + "out_C2.bin", down(Cm, (size_t)nrows * y).data(), (size_t)nrows * y);
  spmm(ncols, y, nrows, nnz, val, rp, ci, Bt, Ct, NULL, true);
  wr(dir + "out_Ct.bin", down(Ct, (size_t)ncols * y).data(), (size_t)ncols * y);
  // uniform random numbers
  const int nr = 1 << 20;
  float* r = NULL;
  float_malloc_device(nr, r);
  rng_uniform_gpu(nr, -0.5f, 0.25f, r);
  wr(dir + "out_r1.bin", down(r, nr).data(), nr);
  rng_uniform_gpu(nr, -0.5f, 0.25f, r);  // the next draw differs
  wr(dir + "out_r2.bin", down(r, nr).data(), nr);
  gpu_rng_uniform(nr, r);
  wr(dir + "out_u.bin", down(r, nr).data(), nr);
  printf("mirror_math ok\n");
  return 0;
}
