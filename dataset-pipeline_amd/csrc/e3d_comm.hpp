// e3d_comm.hpp -- RCCL communicator of libe3dhip.so (one rank per GPU; xGMI inside a node).
#pragma once

#include <atomic>
#include <mutex>

#include <rccl/rccl.h>

#include "e3d_common.hpp"

struct e3d_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  // e3d_comm_abort may come from another host thread (the rank that failed): enqueueing a collective and aborting the
  // communicator exclude each other, and nothing is enqueued on an aborted communicator
  std::mutex mu;
  std::atomic<bool> aborted{false};
};

namespace e3d {

#define E3D_NCCL(expr)                                                                                        \
  do {                                                                                                        \
    ncclResult_t e3d_nccl_err__ = (expr);                                                                     \
    if (e3d_nccl_err__ != ncclSuccess)                                                                        \
      throw ::e3d::Error(-3, ::e3d::fmt("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(e3d_nccl_err__), __FILE__, __LINE__)); \
  } while (0)

// in-place sum over the ranks of a DEVICE buffer, enqueued on `s` (results identical on every rank)
inline void comm_allreduce(e3d_comm* c, void* dev, size_t n, ncclDataType_t t, hipStream_t s) {
  if (!n) return;
  std::lock_guard<std::mutex> lock(c->mu);
  if (c->aborted.load() || !c->comm) throw ::e3d::Error(-3, "the communicator was aborted (another rank failed)");
  E3D_NCCL(ncclAllReduce(dev, dev, n, t, ncclSum, c->comm, s));
}
inline void comm_allreduce_f64(e3d_comm* c, double* dev, size_t n, hipStream_t s) { comm_allreduce(c, dev, n, ncclDouble, s); }
inline void comm_allreduce_f32(e3d_comm* c, float* dev, size_t n, hipStream_t s) { comm_allreduce(c, dev, n, ncclFloat, s); }
inline void comm_allreduce_i32(e3d_comm* c, int* dev, size_t n, hipStream_t s) { comm_allreduce(c, dev, n, ncclInt32, s); }

}  // namespace e3d
