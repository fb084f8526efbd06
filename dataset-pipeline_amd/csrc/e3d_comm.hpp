// e3d_comm.hpp -- RCCL communicator of libe3dhip.so (one rank per GPU; xGMI inside a node).
#pragma once

#include <rccl/rccl.h>

#include "e3d_common.hpp"

struct e3d_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

namespace e3d {

#define E3D_NCCL(expr)                                                                                        \
  do {                                                                                                        \
    ncclResult_t e3d_nccl_err__ = (expr);                                                                     \
    if (e3d_nccl_err__ != ncclSuccess)                                                                        \
      throw ::e3d::Error(-3, ::e3d::fmt("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(e3d_nccl_err__), __FILE__, __LINE__)); \
  } while (0)

// in-place sum over the ranks of a DEVICE buffer, enqueued on `s` (results identical on every rank)
inline void comm_allreduce_f64(e3d_comm* c, double* dev, size_t n, hipStream_t s) {
  if (n) E3D_NCCL(ncclAllReduce(dev, dev, n, ncclDouble, ncclSum, c->comm, s));
}
inline void comm_allreduce_f32(e3d_comm* c, float* dev, size_t n, hipStream_t s) {
  if (n) E3D_NCCL(ncclAllReduce(dev, dev, n, ncclFloat, ncclSum, c->comm, s));
}
inline void comm_allreduce_i32(e3d_comm* c, int* dev, size_t n, hipStream_t s) {
  if (n) E3D_NCCL(ncclAllReduce(dev, dev, n, ncclInt32, ncclSum, c->comm, s));
}

}  // namespace e3d
