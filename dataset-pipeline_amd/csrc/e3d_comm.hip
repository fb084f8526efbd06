// e3d_comm.hip -- C-ABI of the RCCL communicator (include/e3d_hip.h "multi-GPU").
//
// The reference is a single process; SURVEY 8(e): directed pairs / source slices (A) and images (B) shard over the GPUs of
// a node and the small normal-equation blocks are summed with one all-reduce per LM pass.  The library owns the
// communicator: reductions run on the handle's HIP stream straight from HBM (no host hop, no callback into the host
// language).  Two ways to create the ranks:
//   * one process per GPU (bench.py, torchrun): rank 0 calls e3d_comm_unique_id, the 128 bytes travel through whatever
//     rendezvous the launcher has, every rank calls e3d_comm_create;
//   * one host thread per GPU inside a tool (ICPScanAligner --gpus N): e3d_comm_create_all.
#include <chrono>
#include <thread>

#include "../../include/e3d_hip.h"
#include "e3d_comm.hpp"

using namespace e3d;

extern "C" {

int e3d_comm_unique_id(char id[E3D_COMM_ID_BYTES]) {
  try {
    static_assert(E3D_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    if (!id) throw Error(E3D_ERR_INVALID, "e3d_comm_unique_id: null argument");
    ncclUniqueId u;
    E3D_NCCL(ncclGetUniqueId(&u));
    std::memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return 0;
  } catch (const e3d::Error& e) { e3d::set_last_error(e.what()); return e.code;
  } catch (const std::exception& e) { e3d::set_last_error(e.what()); return E3D_ERR_HIP; }
}

e3d_comm_t* e3d_comm_create(const char id[E3D_COMM_ID_BYTES], int rank, int world_size, int device) {
  try {
    if (!id || world_size < 1 || rank < 0 || rank >= world_size) throw Error(E3D_ERR_INVALID, "e3d_comm_create: bad argument");
    E3D_HIP(hipSetDevice(device));
    ncclUniqueId u;
    std::memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    e3d_comm* c = new e3d_comm();
    c->rank = rank; c->world = world_size; c->device = device;
    try {
      E3D_NCCL(ncclCommInitRank(&c->comm, world_size, u, rank));
    } catch (...) { delete c; throw; }
    return c;
  } catch (const std::exception& e) { e3d::set_last_error(e.what()); return nullptr; }
}

int e3d_comm_create_all(int n_devices, const int* devices, e3d_comm_t** out) {
  try {
    if (n_devices < 1 || !out) throw Error(E3D_ERR_INVALID, "e3d_comm_create_all: bad argument");
    std::vector<ncclComm_t> comms((size_t)n_devices);
    std::vector<int> devs((size_t)n_devices);
    for (int i = 0; i < n_devices; ++i) devs[i] = devices ? devices[i] : i;
    std::vector<e3d_comm*> made((size_t)n_devices, nullptr);
    E3D_NCCL(ncclCommInitAll(comms.data(), n_devices, devs.data()));
    try {
      for (int i = 0; i < n_devices; ++i) {
        made[i] = new e3d_comm();
        made[i]->comm = comms[i]; made[i]->rank = i; made[i]->world = n_devices; made[i]->device = devs[i];
      }
    } catch (...) {   // nothing is handed out half-built: the communicators go back, the objects already made are deleted
      for (int i = 0; i < n_devices; ++i) { (void)ncclCommAbort(comms[i]); delete made[i]; }
      throw;
    }
    for (int i = 0; i < n_devices; ++i) out[i] = made[i];
    return 0;
  } catch (const e3d::Error& e) { e3d::set_last_error(e.what()); return e.code;
  } catch (const std::exception& e) { e3d::set_last_error(e.what()); return E3D_ERR_HIP; }
}

int e3d_comm_abort(e3d_comm_t* c) {
  if (!c) return 0;
  if (c->aborted.exchange(true) || !c->comm) return 0;
  // the flag is up: no new enqueue starts.  One that is between its flag check and its return gets up to 200 ms to leave the
  // library call (it does unless it is the blocked enqueue this abort is meant to release).
  for (int i = 0; i < 200 && c->inflight.load() > 0; ++i) std::this_thread::sleep_for(std::chrono::milliseconds(1));
  const ncclResult_t r = ncclCommAbort(c->comm);   // releases collectives that wait for a rank that will never arrive
  if (r != ncclSuccess) { e3d::set_last_error(std::string("ncclCommAbort failed: ") + ncclGetErrorString(r)); return E3D_ERR_HIP; }
  return 0;
}

void e3d_comm_destroy(e3d_comm_t* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->comm && !c->aborted.load()) (void)ncclCommDestroy(c->comm);      // (an aborted communicator is already gone)
  delete c;
}

int e3d_comm_get_stats(e3d_comm_t* c, double* allreduce_ms, int64_t* allreduce_calls, int64_t* allreduce_bytes, int reset) {
  if (!c) { e3d::set_last_error("e3d_comm_get_stats: null communicator"); return E3D_ERR_INVALID; }
  (void)hipSetDevice(c->device);
  c->flush_events();
  if (allreduce_ms) *allreduce_ms = c->allreduce_ms;
  if (allreduce_calls) *allreduce_calls = c->allreduce_calls;
  if (allreduce_bytes) *allreduce_bytes = c->allreduce_bytes;
  if (reset) { c->allreduce_ms = 0.0; c->allreduce_calls = 0; c->allreduce_bytes = 0; }
  return 0;
}

int e3d_comm_rank(const e3d_comm_t* c) { return c ? c->rank : 0; }
int e3d_comm_world_size(const e3d_comm_t* c) { return c ? c->world : 1; }

}  // extern "C"
