// NormalEstimator -- drop-in replacement of the reference tool (src/exe/normal_estimator.cc:48-229): loads the scans
// of a MeshLab project (io::ReadMeshLabProject semantics incl. the global scale_factor [QUIRK],
// src/io/meshlab_project.cc:39-87), merges them in the global frame, estimates per-scan normals with the viewpoint at
// the scan origin on the MI355X, and writes the merged x y z nx ny nz red green blue binary PLY.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "io_mlp.h"
#include "io_ply.h"
#include "normal_estimation.h"
#include "util.h"

using namespace e3d_host;

namespace {

float g_scale_factor = 0.f;   // opt::GlobalParameters().scale_factor (src/opt/parameters.h:63), 0 = automatic

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
bool ReadMeshLabProject(const std::string& path, std::vector<MeshInfo>* meshes) {
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

}  // namespace

int main(int argc, char** argv) {
  std::string meshlab_project_input_path;
  parse_argument(argc, argv, "-i", meshlab_project_input_path);
  std::string ply_output_path;
  parse_argument(argc, argv, "-o", ply_output_path);
  int neighbor_count = 8;
  parse_argument(argc, argv, "--neighbor_count", neighbor_count);
  float neighbor_radius = -1;
  parse_argument(argc, argv, "--neighbor_radius", neighbor_radius);

  if (meshlab_project_input_path.empty() || ply_output_path.empty()) {
    std::cout << "Please provide input paths." << std::endl;
    return EXIT_FAILURE;
  }
  const std::string project_dir = parent_path(meshlab_project_input_path);
  std::vector<MeshInfo> scan_infos;
  if (!ReadMeshLabProject(meshlab_project_input_path, &scan_infos)) {
    std::cerr << "Cannot read scan poses from " << meshlab_project_input_path << std::endl;
    return EXIT_FAILURE;
  }

  // load every scan and transform it into the global frame (pcl::transformPointCloud, f32, on the GPU)
  std::vector<PointCloud::Ptr> global_clouds(scan_infos.size());
  size_t total = 0;
  for (size_t i = 0; i < scan_infos.size(); ++i) {
    const MeshInfo& info = scan_infos[i];
    PointCloud local;
    const std::string filename = (info.filename[0] == '/') ? info.filename : join_path(project_dir, info.filename);
    if (loadPLYFile(filename, local, /*want_rgb=*/true) < 0) return EXIT_FAILURE;
    float T[12];
    info.global_T_mesh.matrix3x4(T);
    global_clouds[i].reset(new PointCloud());
    global_clouds[i]->xyz.resize(local.xyz.size());
    global_clouds[i]->rgb.swap(local.rgb);
    float bmin[3], bmax[3];
    if (local.size() > 0 &&
        api().e3d_transform_cloud(local.xyz.data(), nullptr, local.size(), T, global_clouds[i]->xyz.data(), nullptr, bmin, bmax) < 0) {
      std::cerr << "transform failed: " << api().e3d_last_error() << std::endl;
      return EXIT_FAILURE;
    }
    total += local.size();
  }

  std::vector<unsigned char> out(total * 27);
  unsigned char* p = out.data();
  for (size_t i = 0; i < scan_infos.size(); ++i) {
    const MeshInfo& info = scan_infos[i];
    NormalEstimationTwoPass normal_estimation;
    normal_estimation.setInputCloud(global_clouds[i]);
    if (neighbor_radius > 0) normal_estimation.setRadiusSearch(neighbor_radius);
    else normal_estimation.setKSearch(neighbor_count);
    normal_estimation.setViewPoint(info.global_T_mesh.t[0], info.global_T_mesh.t[1], info.global_T_mesh.t[2]);
    PointCloud normals;
    try {
      normal_estimation.compute(normals);
    } catch (const std::exception& e) {
      std::cerr << e.what() << std::endl;
      return EXIT_FAILURE;
    }
    const float scale = g_scale_factor;
    const PointCloud& c = *global_clouds[i];
    for (size_t k = 0; k < c.size(); ++k) {
      const float v[6] = {c.xyz[3 * k] / scale, c.xyz[3 * k + 1] / scale, c.xyz[3 * k + 2] / scale,
                          normals.normals[3 * k], normals.normals[3 * k + 1], normals.normals[3 * k + 2]};
      memcpy(p, v, 24); p += 24;
      p[0] = c.rgb[3 * k]; p[1] = c.rgb[3 * k + 1]; p[2] = c.rgb[3 * k + 2]; p += 3;
    }
  }
  if (savePLYFileBinaryXYZNormalRGB(ply_output_path, out, total) < 0) return EXIT_FAILURE;
  std::cout << "Finished!" << std::endl;
  return EXIT_SUCCESS;
}
