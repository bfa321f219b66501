// include/gnn/global.h -- types and constants of the GNN host API.
// The names are the ones the reference's drivers and layers use (reference: include/gnn/global.h:29-77), so that
// code written against GraphAIBench's layer/operator API compiles against this tree; here they are typed constants
// and aliases.  This build always runs the MI355X path: the "GPU" pointer variants are the only ones.
#pragma once
#include <omp.h>

#include <cassert>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iomanip>
#include <iostream>
#include <map>
#include <string>
#include <vector>

// the device path is the product; there is no host compute path (drivers test this macro with #ifdef)
#define ENABLE_GPU

// scalar / index / label types
using float_t = float;
using t_data = float;
using t_idx = int;
using acc_t = float;
using edata_t = float;
using vdata_t = float;
using index_t = uint32_t;
using label_t = uint8_t;
using mask_t = uint8_t;
using vec_t = std::vector<float>;

enum class net_phase { TRAIN, TEST, VAL };
enum class gnn_arch { GCN, GAT, SAGE, GGNN };

// model defaults (net.cpp reads them when the command line leaves a field out)
constexpr int DEFAULT_NUM_LAYER = 2, DEFAULT_SIZE_FRONTIER = 3000, DEFAULT_SIZE_HID = 16, EVAL_INTERVAL = 50;
constexpr double DEFAULT_RATE_LEARN = 0.02;
// Adam (include/utils/optimizer.h:99-116)
constexpr double ADAM_LR = 0.05, ADAM_BETA1 = 0.9, ADAM_BETA2 = 0.999, ADAM_EPSILON = 0.00000001;

// keys of the per-operation wall-time table time_ops (reference: global.h:42-54, printed by train.cpp:60-76)
constexpr char OP_DENSEMM = 'a', OP_SPARSEMM = 'b', OP_RELU = 'c', OP_DROPOUT = 'd', OP_LOSS = 'e', OP_BIAS = 'f',
               OP_REDUCE = 'g', OP_NORM = 'h', OP_SCORE = 'i', OP_ATTN = 'j', OP_TRANSPOSE = 'k', OP_SAMPLE = 'l',
               OP_COPY = 'm';
extern std::map<char, double> time_ops;
