// host_types.h -- the few value types the reference's tools get from PCL/Eigen (absent in this image): a point cloud
// with optional normals / colours and a 4x4 float affine transform.
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
#include <vector>

namespace e3d_host {

// Stand-in for Eigen::Affine3f (row-major 4x4; last row 0 0 0 1).
struct Affine3f {
  float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1};
  float& operator()(int r, int c) { return m[4 * r + c]; }
  float operator()(int r, int c) const { return m[4 * r + c]; }
  void translation(float* t) const { t[0] = m[3]; t[1] = m[7]; t[2] = m[11]; }
  // Eigen's Transform::rotation(): closest rotation of the linear part (polar decomposition via one-sided Jacobi SVD)
  void rotation(double R[9]) const {
    double A[9] = {m[0], m[1], m[2], m[4], m[5], m[6], m[8], m[9], m[10]};
    double V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int sweep = 0; sweep < 60; ++sweep) {
      double off = 0;
      for (int p = 0; p < 2; ++p)
        for (int q = p + 1; q < 3; ++q) {
          double alpha = 0, beta = 0, gamma = 0;
          for (int k = 0; k < 3; ++k) { alpha += A[3 * k + p] * A[3 * k + p]; beta += A[3 * k + q] * A[3 * k + q]; gamma += A[3 * k + p] * A[3 * k + q]; }
          off = std::fmax(off, std::fabs(gamma) / std::sqrt(alpha * beta + 1e-300));
          if (std::fabs(gamma) < 1e-300) continue;
          const double zeta = (beta - alpha) / (2.0 * gamma);
          const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
          const double c = 1.0 / std::sqrt(1.0 + t * t), s = c * t;
          for (int k = 0; k < 3; ++k) {
            const double ap = A[3 * k + p], aq = A[3 * k + q];
            A[3 * k + p] = c * ap - s * aq; A[3 * k + q] = s * ap + c * aq;
            const double vp = V[3 * k + p], vq = V[3 * k + q];
            V[3 * k + p] = c * vp - s * vq; V[3 * k + q] = s * vp + c * vq;
          }
        }
      if (off < 1e-15) break;
    }
    // A = U * diag(sigma): normalise columns -> U ; R = U * V^T (sign-fixed)
    double U[9], sigma[3];
    for (int j = 0; j < 3; ++j) {
      double nrm = 0;
      for (int k = 0; k < 3; ++k) nrm += A[3 * k + j] * A[3 * k + j];
      nrm = std::sqrt(nrm);
      sigma[j] = nrm;
      for (int k = 0; k < 3; ++k) U[3 * k + j] = nrm > 0 ? A[3 * k + j] / nrm : (k == j ? 1.0 : 0.0);
    }
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += U[3 * i + k] * V[3 * j + k];
        R[3 * i + j] = s;
      }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    if (det < 0) {   // flip the direction of the smallest singular value (one-sided Jacobi leaves them unsorted)
      int c = 0;
      for (int j = 1; j < 3; ++j) if (sigma[j] < sigma[c]) c = j;
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) R[3 * i + j] -= 2.0 * U[3 * i + c] * V[3 * j + c];
    }
  }
};

// Stand-in for pcl::PointCloud<PointXYZ / PointNormal / PointXYZRGB>: SoA-free simple AoS arrays.
struct PointCloud {
  std::vector<float> xyz;       // n x 3
  std::vector<float> normals;   // n x 3 (optional)
  std::vector<float> curvature; // n (optional)
  std::vector<uint8_t> rgb;     // n x 3 (optional)
  std::vector<float> intensity; // n (optional; pcl::PointXYZI files of the multi-resolution cloud cache)
  size_t size() const { return xyz.size() / 3; }
  typedef std::shared_ptr<PointCloud> Ptr;
};

}  // namespace e3d_host
