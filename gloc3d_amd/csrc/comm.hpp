// comm.hpp -- RCCL communicator behind the C ABI (gloc_comm of include/gloc3d.h), shared by comm.hip
// (which owns it) and knn.hip (gloc_knn_search_sharded).  librccl is bound at run time (dlopen): the
// host process decides which copy it runs on (PyTorch bundles its own), as for the HIP runtime.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

struct gloc_comm {
  int device = 0, rank = 0, world = 1;
  void* nccl = nullptr;  // ncclComm_t
};

namespace gloc {
namespace comm {
// All-gather `bytes` bytes per rank from d_send into d_recv ([world][bytes]) on `s`; several calls between
// group_begin / group_end are fused into one RCCL launch.  Return a GLOC_* code.
int group_begin();
int group_end();
int all_gather(gloc_comm* c, const void* d_send, void* d_recv, size_t bytes, hipStream_t s);
}  // namespace comm
}  // namespace gloc
