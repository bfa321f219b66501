// include/gnn/global.h -- types and constants of the GNN host API.
// Mirrors the names the reference's drivers and layers use (reference: include/gnn/global.h:29-77)
// so that code written against GraphAIBench's layer/operator API compiles against this tree.
// This build always runs the MI355X path: the "GPU" pointer variants are the only ones.
#pragma once
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

#include <omp.h>

#define DEFAULT_NUM_LAYER 2
#define DEFAULT_SIZE_FRONTIER 3000
#define DEFAULT_SIZE_HID 16
#define DEFAULT_RATE_LEARN 0.02
#define EVAL_INTERVAL 50

#define ADAM_LR 0.05
#define ADAM_BETA1 0.9
#define ADAM_BETA2 0.999
#define ADAM_EPSILON 0.00000001

// per-operation wall-time buckets (reference: global.h:42-54, train.cpp:60-76)
#define OP_DENSEMM 'a'
#define OP_SPARSEMM 'b'
#define OP_RELU 'c'
#define OP_DROPOUT 'd'
#define OP_LOSS 'e'
#define OP_BIAS 'f'
#define OP_REDUCE 'g'
#define OP_NORM 'h'
#define OP_SCORE 'i'
#define OP_ATTN 'j'
#define OP_TRANSPOSE 'k'
#define OP_SAMPLE 'l'
#define OP_COPY 'm'

#define ENABLE_GPU  // the device path is the product; there is no host compute path

enum class net_phase { TRAIN, TEST, VAL };
enum class gnn_arch { GCN, GAT, SAGE, GGNN };
typedef float float_t;
typedef float t_data;
typedef int t_idx;
typedef std::vector<float> vec_t;
typedef float acc_t;
typedef uint8_t label_t;
typedef uint8_t mask_t;
typedef uint32_t index_t;
typedef float edata_t;
typedef float vdata_t;
extern std::map<char, double> time_ops;
