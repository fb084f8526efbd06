// Scan loading shared by NormalEstimator and ImageRegistrator: MeshLab project poses as Sophus::Sim3f with the global
// scale_factor [QUIRK] (io::ReadMeshLabProject, src/io/meshlab_project.cc:39-87; opt::GlobalParameters().scale_factor,
// src/opt/parameters.h:63).
#pragma once

#include <cmath>
#include <sstream>
#include <string>
#include <vector>

#include "io_mlp.h"

namespace e3d_host {

inline float& global_scale_factor() { static float f = 0.f; return f; }   // 0 = automatic (inverse scale of the first mesh matrix)
#define g_scale_factor (::e3d_host::global_scale_factor())

// Sophus::Sim3f(Matrix4f) followed by .matrix(): thirdparty/sophus/rxso3.hpp:382-393 (setScaledRotationMatrix),
// :191-220 (matrix), sim3.hpp:418-420.  Eigen's rotation-matrix -> quaternion conversion is restated (recalled).
struct Sim3f {
  float q[4] = {0, 0, 0, 1};   // x y z w, |q|^2 = scale
  float t[3] = {0, 0, 0};
  Sim3f() {}
  explicit Sim3f(const float* M /*row-major 4x4*/) {
    float sR[9] = {M[0], M[1], M[2], M[4], M[5], M[6], M[8], M[9], M[10]};
    float sq[3];
    for (int i = 0; i < 3; ++i) sq[i] = sR[3 * i] * sR[3 * i] + (sR[3 * i + 1] * sR[3 * i + 1] + sR[3 * i + 2] * sR[3 * i + 2]);
    const float squared_scale = float(1. / 3.) * (sq[0] + sq[1] + sq[2]);
    const float scale = std::sqrt(squared_scale);
    float m[9];
    for (int i = 0; i < 9; ++i) m[i] = sR[i] / scale;
    float tr = m[0] + m[4] + m[8];
    if (tr > 0) {
      tr = std::sqrt(tr + 1.0f);
      q[3] = 0.5f * tr; tr = 0.5f / tr;
      q[0] = (m[7] - m[5]) * tr; q[1] = (m[2] - m[6]) * tr; q[2] = (m[3] - m[1]) * tr;
    } else {
      int i = 0;
      if (m[4] > m[0]) i = 1;
      if (m[8] > m[4 * i]) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      tr = std::sqrt(m[4 * i] - m[4 * j] - m[4 * k] + 1.0f);
      q[i] = 0.5f * tr; tr = 0.5f / tr;
      q[3] = (m[3 * k + j] - m[3 * j + k]) * tr;
      q[j] = (m[3 * j + i] + m[3 * i + j]) * tr;
      q[k] = (m[3 * k + i] + m[3 * i + k]) * tr;
    }
    const float s = std::sqrt(scale);
    for (int i = 0; i < 4; ++i) q[i] *= s;
    t[0] = M[3]; t[1] = M[7]; t[2] = M[11];
  }
  float scale() const { return q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]; }
  void matrix3x4(float* out /*row-major 3x4*/) const {
    const float vx = q[0], vy = q[1], vz = q[2], w = q[3];
    const float vx_sq = vx * vx, vy_sq = vy * vy, vz_sq = vz * vz, w_sq = w * w;
    const float two_vx = 2.f * vx, two_vy = 2.f * vy, two_vz = 2.f * vz;
    const float two_vx_vy = two_vx * vy, two_vx_vz = two_vx * vz, two_vx_w = two_vx * w;
    const float two_vy_vz = two_vy * vz, two_vy_w = two_vy * w, two_vz_w = two_vz * w;
    out[0] = vx_sq - vy_sq - vz_sq + w_sq; out[1] = two_vx_vy - two_vz_w;          out[2] = two_vx_vz + two_vy_w;
    out[4] = two_vx_vy + two_vz_w;         out[5] = -vx_sq + vy_sq - vz_sq + w_sq; out[6] = two_vy_vz - two_vx_w;
    out[8] = two_vx_vz - two_vy_w;         out[9] = two_vx_w + two_vy_vz;          out[10] = -vx_sq - vy_sq + vz_sq + w_sq;
    out[3] = t[0]; out[7] = t[1]; out[11] = t[2];
  }
};

struct MeshInfo { std::string label, filename; Sim3f global_T_mesh; };

// io::ReadMeshLabProject (src/io/meshlab_project.cc:39-87)
inline bool ReadMeshLabProject(const std::string& path, std::vector<MeshInfo>* meshes) {
  std::vector<MlpMesh> raw;
  if (!ParseMeshLabProject(path, &raw)) return false;
  for (const MlpMesh& m : raw) {
    MeshInfo info;
    info.label = m.label;
    info.filename = m.filename;
    if (m.has_matrix) {
      std::istringstream s(m.matrix_text);
      float M[16];
      for (int i = 0; i < 16; ++i) s >> M[i];
      const Sim3f SimM(M);
      if (g_scale_factor == 0) g_scale_factor = 1.f / SimM.scale();        // [QUIRK] first matrix fixes the global scale
      float Ms[16];
      for (int i = 0; i < 16; ++i) Ms[i] = g_scale_factor * M[i];
      info.global_T_mesh = Sim3f(Ms);
    }
    meshes->push_back(info);
  }
  return true;
}


}  // namespace e3d_host
