// include/gnn/gpu_context.h -- process-wide device context of the host API.
// Takes the place of the reference's static cuBLAS/cuSPARSE/cuRAND handle holder
// (reference: include/gnn/gpu_context.h:4-16, src/utilities/random.cpp:62-80): one gaib_ctx
// (device + HIP stream + workspace) per process, created on first use.  Device = $GAIB_DEVICE
// or $LOCAL_RANK or 0 (one process per GPU).
#pragma once
#include "gaib.h"

class gpu_context {
 public:
  static gaib_ctx* get();                        // lazily created
  static void set(int device, void* hip_stream); // explicit (multi-GPU launchers, tests)
  static void sync();                            // CudaTest() equivalent
  // side stream for independent work (gaib_side_begin/end/wait); no-ops unless GAIB_OVERLAP=1.  The layer classes
  // do not use it (measured: no gain next to an HBM-saturating aggregation); kept for drivers
  static void side_begin();
  static void side_end();
  static void side_wait();
  static void check(int status, const char* what); // non-zero -> print + exit (cutils.h:18-28)
  // one-process-per-GPU training: the communicator of this rank (NULL = single GPU).  While it is set, every
  // optimizer step first sums its gradient buffer over the ranks (gaib_allreduce_f32: weights and Adam state stay
  // replicated and bit-identical on all ranks), and LearningGraphs built by make_partitioned_graph exchange halo rows
  // on it.  The context borrows it.
  // edges aggregated by this process so far (the aggregators add the graph's edge count per aggregation call)
  static void add_aggregated_edges(unsigned long long n);
  static unsigned long long aggregated_edges();
  static void set_comm(gaib_comm* comm);
  static gaib_comm* comm();
};
#define GAIB_OR_DIE(call) gpu_context::check((call), #call)
