// normal_estimation.h -- pcl::NormalEstimationTwoPassOMP's call surface as the tools use it
// (src/geometry/two_pass_normal_3d_omp.h:53-99; call sites src/exe/icp_scan_aligner.cc:323-330,
// src/exe/normal_estimator.cc:177-194) on the HIP library.
#pragma once

#include <stdexcept>
#include <string>

#include "e3d_loader.h"
#include "host_types.h"

namespace e3d_host {

class NormalEstimationTwoPass {
 public:
  void setInputCloud(const PointCloud::Ptr& cloud) { input_ = cloud; }
  void setKSearch(int k) { k_ = k; radius_ = 0; }
  void setRadiusSearch(double r) { radius_ = r; k_ = 0; }
  void setViewPoint(float x, float y, float z) { vp_[0] = x; vp_[1] = y; vp_[2] = z; }
  // fills normals + curvature of `output` (resized to the input size; xyz is NOT copied, like pcl::Normal output)
  void compute(PointCloud& output) {
    if (!input_) throw std::runtime_error("NormalEstimationTwoPass: no input cloud");
    const size_t n = input_->size();
    output.normals.assign(3 * n, 0.f);
    output.curvature.assign(n, 0.f);
    if (n == 0) return;
    if (radius_ > 0) {
      if (api().e3d_normals_radius(input_->xyz.data(), n, (float)radius_, vp_, output.normals.data(), output.curvature.data(), nullptr) < 0)
        throw std::runtime_error(std::string("NormalEstimationTwoPass: ") + api().e3d_last_error());
      return;
    }
    if (api().e3d_normals_knn(input_->xyz.data(), n, k_, vp_, output.normals.data(), output.curvature.data(), nullptr) < 0)
      throw std::runtime_error(std::string("NormalEstimationTwoPass: ") + api().e3d_last_error());
  }

 private:
  PointCloud::Ptr input_;
  int k_ = 0;
  double radius_ = 0;
  float vp_[3] = {0, 0, 0};
};

}  // namespace e3d_host
