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
#include "io_scans.h"
#include "normal_estimation.h"
#include "util.h"

using namespace e3d_host;

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
