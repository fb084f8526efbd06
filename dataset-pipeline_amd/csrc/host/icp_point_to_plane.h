// icp_point_to_plane.h -- icp::PointToPlaneICP with the reference's class surface
// (src/icp/icp_point_to_plane.h:39-80), implemented on the HIP library through the C-ABI.
#pragma once

#include <atomic>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "e3d_loader.h"
#include "host_types.h"

namespace icp {

// With --gpus N (e3d_host::set_gpu_count) the object holds one library handle per GPU: every handle gets every cloud, every
// rank searches its slice of each directed pair's source cloud, and the library all-reduces the normal-equation blocks with its
// RCCL communicator; Run drives the ranks from one host thread each and returns rank 0's (= everyone's) result.
class PointToPlaneICP {
 public:
  PointToPlaneICP() {
    auto& A = e3d_host::api();
    const int n = e3d_host::gpu_count_setting();
    if (n > 1) {
      comms_.resize((size_t)n, nullptr);
      if (A.e3d_comm_create_all(n, nullptr, comms_.data()) < 0) throw std::runtime_error(std::string("PointToPlaneICP: ") + A.e3d_last_error());
    }
    for (int r = 0; r < n; ++r) {
      if (n > 1 && A.e3d_init(r) < 1) throw std::runtime_error(std::string("PointToPlaneICP: ") + A.e3d_last_error());
      e3d_icp_t* h = A.e3d_icp_create();
      if (!h) throw std::runtime_error(std::string("PointToPlaneICP: ") + A.e3d_last_error());
      h_.push_back(h);
      if (n > 1 && A.e3d_icp_set_comm(h, comms_[(size_t)r]) < 0) throw std::runtime_error(std::string("PointToPlaneICP: ") + A.e3d_last_error());
    }
    if (n > 1) A.e3d_init(0);
  }
  ~PointToPlaneICP() {
    for (e3d_icp_t* h : h_) if (h) e3d_host::api().e3d_icp_destroy(h);
    for (e3d_comm_t* c : comms_) if (c) e3d_host::api().e3d_comm_destroy(c);
  }
  PointToPlaneICP(const PointToPlaneICP&) = delete;
  PointToPlaneICP& operator=(const PointToPlaneICP&) = delete;

  // Adds a point cloud (xyz + normals) to be aligned; returns its index, or -1 for fixed clouds.
  int AddPointCloud(const e3d_host::PointCloud::Ptr& point_cloud, const e3d_host::Affine3f& global_T_cloud, bool fixed) {
    if (point_cloud->normals.size() != point_cloud->xyz.size()) throw std::runtime_error("AddPointCloud: cloud without normals");
    int r0 = -1;
    for (size_t k = 0; k < h_.size(); ++k) {
      const int r = e3d_host::api().e3d_icp_add_cloud(h_[k], point_cloud->xyz.data(), point_cloud->normals.data(),
                                                      point_cloud->size(), global_T_cloud.m, fixed ? 1 : 0);
      if (r < -1) throw std::runtime_error(std::string("AddPointCloud: ") + e3d_host::api().e3d_last_error());
      if (k == 0) r0 = r;
    }
    return r0;
  }

  // Runs the alignment; returns true if it converged.
  bool Run(float max_correspondence_distance, int initial_iteration, int max_num_iterations,
           float convergence_threshold_max_movement, bool print_progress) {
    std::vector<int> res(h_.size(), 0);
    std::vector<std::string> err(h_.size());
    std::atomic<int> first_failed{-1};
    auto work = [&](size_t k) {
      res[k] = e3d_host::api().e3d_icp_run(h_[k], max_correspondence_distance, initial_iteration, max_num_iterations,
                                           convergence_threshold_max_movement, print_progress ? 1 : 0);   // only rank 0 prints
      if (res[k] < 0) {
        err[k] = e3d_host::api().e3d_last_error();
        int none = -1;
        first_failed.compare_exchange_strong(none, (int)k);
        // the other ranks may already wait in an all-reduce this rank will never join: abort every communicator so that
        // their steps fail too and the threads can be joined (the error of the rank that failed first is reported)
        for (e3d_comm_t* c : comms_) if (c) e3d_host::api().e3d_comm_abort(c);
      }
    };
    std::vector<std::thread> th;
    for (size_t k = 1; k < h_.size(); ++k) th.emplace_back(work, k);
    work(0);
    for (std::thread& t : th) t.join();
    if (first_failed.load() >= 0) {   // reference: CHECK(!clouds_.empty()) aborts
      fprintf(stderr, "FATAL: PointToPlaneICP::Run: %s\n", err[(size_t)first_failed.load()].c_str());
      abort();
    }
    return res[0] == 1;
  }

  e3d_host::Affine3f GetResultGlobalTCloud(int cloud_index) {
    e3d_host::Affine3f T;
    if (e3d_host::api().e3d_icp_get_pose(h_[0], cloud_index, T.m) < 0)
      throw std::out_of_range(e3d_host::api().e3d_last_error());   // reference: clouds_.at()
    return T;
  }

 private:
  std::vector<e3d_icp_t*> h_;
  std::vector<e3d_comm_t*> comms_;
};

}  // namespace icp
