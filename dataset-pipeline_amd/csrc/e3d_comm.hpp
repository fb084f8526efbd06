// e3d_comm.hpp -- RCCL communicator of libe3dhip.so (one rank per GPU; xGMI inside a node).
#pragma once

#include <atomic>
#include <utility>
#include <vector>

#include <rccl/rccl.h>

#include "e3d_common.hpp"

struct e3d_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
  // e3d_comm_abort may come from another host thread (the rank that failed) while this rank's thread sits inside an enqueue that
  // waits for it: abort therefore takes no lock the enqueue holds.  It sets the flag (nothing is enqueued afterwards) and calls
  // ncclCommAbort, which is meant to be called with operations outstanding; the handle stays set until e3d_comm_destroy.
  // Aborting ONE communicator does not release its intra-node peers: whoever notices a failure aborts every local communicator
  // (the tools do, csrc/host/icp_point_to_plane.h).
  std::atomic<bool> aborted{false};
  // enqueues in flight on the owner's thread: abort raises the flag first and then gives a running enqueue a bounded time to leave
  // ncclAllReduce before it frees the communicator under it (an enqueue that never returns is the case abort exists for: it does
  // not wait for that one).  The statistics below belong to the communicator's own thread: e3d_comm_get_stats is called from it,
  // or after joining it.
  std::atomic<int> inflight{0};
  // HIP-event stop-watch of the collectives (e3d_comm_get_stats): pairs are read lazily, so timing adds no synchronisation
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  size_t ev_used = 0;
  double allreduce_ms = 0.0;
  long long allreduce_calls = 0, allreduce_bytes = 0;
  void flush_events() {
    for (size_t i = 0; i < ev_used; ++i) {
      float t = 0.f;
      if (hipEventSynchronize(ev[i].second) == hipSuccess && hipEventElapsedTime(&t, ev[i].first, ev[i].second) == hipSuccess) allreduce_ms += (double)t;
    }
    ev_used = 0;
  }
  ~e3d_comm() { for (auto& p : ev) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); } }
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
  struct InFlight { e3d_comm* c; explicit InFlight(e3d_comm* cc) : c(cc) { c->inflight.fetch_add(1); } ~InFlight() { c->inflight.fetch_sub(1); } } guard(c);
  if (c->aborted.load() || !c->comm) throw ::e3d::Error(-3, "the communicator was aborted (another rank failed)");
  if (c->ev_used == c->ev.size()) {
    if (c->ev.size() >= 1024) c->flush_events();       // (waits for collectives enqueued long ago)
    else { hipEvent_t a, b; E3D_HIP(hipEventCreate(&a)); E3D_HIP(hipEventCreate(&b)); c->ev.emplace_back(a, b); }
  }
  const std::pair<hipEvent_t, hipEvent_t>& e = c->ev[c->ev_used];
  E3D_HIP(hipEventRecord(e.first, s));
  E3D_NCCL(ncclAllReduce(dev, dev, n, t, ncclSum, c->comm, s));
  E3D_HIP(hipEventRecord(e.second, s));
  ++c->ev_used; ++c->allreduce_calls;
  c->allreduce_bytes += (long long)n * (t == ncclDouble ? 8 : 4);
}
inline void comm_allreduce_f64(e3d_comm* c, double* dev, size_t n, hipStream_t s) { comm_allreduce(c, dev, n, ncclDouble, s); }
inline void comm_allreduce_f32(e3d_comm* c, float* dev, size_t n, hipStream_t s) { comm_allreduce(c, dev, n, ncclFloat, s); }
inline void comm_allreduce_i32(e3d_comm* c, int* dev, size_t n, hipStream_t s) { comm_allreduce(c, dev, n, ncclInt32, s); }

}  // namespace e3d
