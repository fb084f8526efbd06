// icp_point_to_plane.h -- icp::PointToPlaneICP with the reference's class surface
// (src/icp/icp_point_to_plane.h:39-80), implemented on the HIP library through the C-ABI.
#pragma once

#include <stdexcept>
#include <string>

#include "e3d_loader.h"
#include "host_types.h"

namespace icp {

class PointToPlaneICP {
 public:
  PointToPlaneICP() : h_(e3d_host::api().e3d_icp_create()) {
    if (!h_) throw std::runtime_error(std::string("PointToPlaneICP: ") + e3d_host::api().e3d_last_error());
  }
  ~PointToPlaneICP() { if (h_) e3d_host::api().e3d_icp_destroy(h_); }
  PointToPlaneICP(const PointToPlaneICP&) = delete;
  PointToPlaneICP& operator=(const PointToPlaneICP&) = delete;

  // Adds a point cloud (xyz + normals) to be aligned; returns its index, or -1 for fixed clouds.
  int AddPointCloud(const e3d_host::PointCloud::Ptr& point_cloud, const e3d_host::Affine3f& global_T_cloud, bool fixed) {
    if (point_cloud->normals.size() != point_cloud->xyz.size()) throw std::runtime_error("AddPointCloud: cloud without normals");
    const int r = e3d_host::api().e3d_icp_add_cloud(h_, point_cloud->xyz.data(), point_cloud->normals.data(),
                                                    point_cloud->size(), global_T_cloud.m, fixed ? 1 : 0);
    if (r < -1) throw std::runtime_error(std::string("AddPointCloud: ") + e3d_host::api().e3d_last_error());
    return r;
  }

  // Runs the alignment; returns true if it converged.
  bool Run(float max_correspondence_distance, int initial_iteration, int max_num_iterations,
           float convergence_threshold_max_movement, bool print_progress) {
    const int r = e3d_host::api().e3d_icp_run(h_, max_correspondence_distance, initial_iteration, max_num_iterations,
                                              convergence_threshold_max_movement, print_progress ? 1 : 0);
    if (r < 0) {   // reference: CHECK(!clouds_.empty()) aborts
      fprintf(stderr, "FATAL: PointToPlaneICP::Run: %s\n", e3d_host::api().e3d_last_error());
      abort();
    }
    return r == 1;
  }

  e3d_host::Affine3f GetResultGlobalTCloud(int cloud_index) {
    e3d_host::Affine3f T;
    if (e3d_host::api().e3d_icp_get_pose(h_, cloud_index, T.m) < 0)
      throw std::out_of_range(e3d_host::api().e3d_last_error());   // reference: clouds_.at()
    return T;
  }

 private:
  e3d_icp_t* h_;
};

}  // namespace icp
