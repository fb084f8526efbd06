// GroundTruthCreator -- drop-in replacement of the reference tool (src/exe/ground_truth_creator.cc:218-489): from the
// registered cameras and the laser scans it writes the evaluation data of a dataset -- calibration/ (COLMAP cameras +
// images), points/ (each scan trimmed to the points seen in >= 2 images + scan_alignment.mlp), ground_truth_depth/ and
// occlusion_depth/ (raw float maps per image, optionally gzip-compressed).  The per-image work (occlusion depth, scan
// point visibility, depth maps) runs on the MI355X behind e3d_reg_count_scan_observations / e3d_reg_ground_truth_depth.
// --write_scan_renderings: the image painted with the visible scan points (io_image.h decodes / encodes the colour images, the library
// decides per pixel which point ends up on top).
#include <exception>
#include <zlib.h>

#include <cmath>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <unordered_set>
#include <vector>

#include "opt_problem.h"

using namespace e3d_host;

static bool write_float_map(const std::string& path, const std::vector<float>& map, bool compress) {
  if (compress) {
    gzFile f = gzopen((path + ".gz").c_str(), "w8b");
    if (!f) return false;
    const size_t bytes = map.size() * sizeof(float);
    size_t done = 0;
    while (done < bytes) {                                   // gzwrite takes an unsigned length
      const unsigned chunk = (unsigned)std::min<size_t>(bytes - done, 1u << 30);
      if (gzwrite(f, reinterpret_cast<const char*>(map.data()) + done, chunk) != (int)chunk) { gzclose(f); return false; }
      done += chunk;
    }
    return gzclose(f) == Z_OK;
  }
  FILE* f = fopen(path.c_str(), "wb");
  if (!f) return false;
  const bool ok = fwrite(map.data(), sizeof(float), map.size(), f) == map.size();
  fclose(f);
  return ok;
}

static int run_tool(int argc, char** argv) {
  std::string scan_alignment_path, occlusion_mesh_path, occlusion_splats_path, image_base_path, state_path, output_folder_path;
  parse_argument(argc, argv, "--scan_alignment_path", scan_alignment_path);
  parse_argument(argc, argv, "--occlusion_mesh_path", occlusion_mesh_path);
  parse_argument(argc, argv, "--occlusion_splats_path", occlusion_splats_path);
  parse_argument(argc, argv, "--image_base_path", image_base_path);
  parse_argument(argc, argv, "--state_path", state_path);
  parse_argument(argc, argv, "--output_folder_path", output_folder_path);
  bool rotate_first_scan_upright = true;
  parse_argument(argc, argv, "--rotate_first_scan_upright", rotate_first_scan_upright);
  int scan_point_radius = 2;
  parse_argument(argc, argv, "--scan_point_radius", scan_point_radius);
  bool write_point_cloud = true, write_depth_maps = true, write_occlusion_depth = true, write_scan_renderings = false, compress_depth_maps = false;
  parse_argument(argc, argv, "--write_point_cloud", write_point_cloud);
  parse_argument(argc, argv, "--write_depth_maps", write_depth_maps);
  parse_argument(argc, argv, "--write_occlusion_depth", write_occlusion_depth);
  parse_argument(argc, argv, "--write_scan_renderings", write_scan_renderings);
  parse_argument(argc, argv, "--compress_depth_maps", compress_depth_maps);

  Problem problem;
  if (!problem.prm.SetFromArguments(argc, argv)) return EXIT_FAILURE;
  if (scan_alignment_path.empty() || image_base_path.empty() || state_path.empty() || output_folder_path.empty()) {
    std::cerr << "Please specify all the required paths." << std::endl;
    return EXIT_FAILURE;
  }

  // opt::LoadPointClouds: scans in the global frame (pcl::transformPointCloud on the GPU)
  std::vector<MeshInfo> scan_infos;
  if (!ReadMeshLabProject(scan_alignment_path, &scan_infos) || scan_infos.empty()) {
    std::cerr << "Cannot read scan poses from " << scan_alignment_path << std::endl;
    return EXIT_FAILURE;
  }
  std::cout << "Loading point clouds ..." << std::endl;
  const std::string project_dir = parent_path(scan_alignment_path);
  std::vector<std::vector<float>> scans(scan_infos.size());
  std::vector<uint8_t> all_colors;                             // r g b per scan point (pcl::PointXYZRGB; 0 0 0 without colour properties)
  auto transform = [&](std::vector<float>& xyz, const float* T) {
    if (xyz.empty()) return true;
    std::vector<float> out(xyz.size());
    float bmin[3], bmax[3];
    if (api().e3d_transform_cloud(xyz.data(), nullptr, xyz.size() / 3, T, out.data(), nullptr, bmin, bmax) < 0) {
      std::cerr << "transform failed: " << api().e3d_last_error() << std::endl;
      return false;
    }
    xyz.swap(out);
    return true;
  };
  for (size_t i = 0; i < scan_infos.size(); ++i) {
    PointCloud local;
    const std::string filename = (!scan_infos[i].filename.empty() && scan_infos[i].filename[0] == '/') ? scan_infos[i].filename : join_path(project_dir, scan_infos[i].filename);
    if (loadPLYFile(filename, local, write_scan_renderings) < 0) { std::cerr << "Cannot load scan point clouds." << std::endl; return EXIT_FAILURE; }
    if (write_scan_renderings) {
      if (local.rgb.size() != local.xyz.size()) local.rgb.assign(local.xyz.size(), 0);
      all_colors.insert(all_colors.end(), local.rgb.begin(), local.rgb.end());
    }
    float T[12];
    scan_infos[i].global_T_mesh.matrix3x4(T);
    scans[i].swap(local.xyz);
    if (!transform(scans[i], T)) return EXIT_FAILURE;
  }
  std::cout << "Done." << std::endl;

  // Rotate everything such that the first scan is upright (:275-291): U = (R0^-1, t0 - R0^-1 t0) left-multiplied to every pose
  float U[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
  Pose7 U_pose;
  if (rotate_first_scan_upright) {
    float M0[12];
    scan_infos[0].global_T_mesh.matrix3x4(M0);
    const float sc = scan_infos[0].global_T_mesh.scale();
    double Rinv[9];                                            // rotationMatrix().inverse() = transpose of the scale-free part
    for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Rinv[3 * r + c] = (double)(M0[4 * c + r] / sc);
    rotation_to_quat(Rinv, U_pose.q);
    const float n = std::sqrt(U_pose.q[0] * U_pose.q[0] + U_pose.q[1] * U_pose.q[1] + U_pose.q[2] * U_pose.q[2] + U_pose.q[3] * U_pose.q[3]);
    for (float& v : U_pose.q) v /= n;
    const float t0[3] = {M0[3], M0[7], M0[11]};
    for (int r = 0; r < 3; ++r)
      U_pose.t[r] = t0[r] - ((float)Rinv[3 * r] * t0[0] + (float)Rinv[3 * r + 1] * t0[1] + (float)Rinv[3 * r + 2] * t0[2]);
    double Ru[9];
    pose_rotation(U_pose, Ru);
    for (int r = 0; r < 3; ++r) { for (int c = 0; c < 3; ++c) U[4 * r + c] = (float)Ru[3 * r + c]; U[4 * r + 3] = U_pose.t[r]; }
    for (size_t i = 0; i < scan_infos.size(); ++i) {
      float Mi[12], P[16] = {0};
      scan_infos[i].global_T_mesh.matrix3x4(Mi);
      for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c)
          P[4 * r + c] = U[4 * r] * Mi[c] + U[4 * r + 1] * Mi[4 + c] + U[4 * r + 2] * Mi[8 + c] + (c == 3 ? U[4 * r + 3] : 0.f);
      P[15] = 1.f;
      scan_infos[i].global_T_mesh = Sim3f(P);
      if (!transform(scans[i], U)) return EXIT_FAILURE;
    }
  }

  std::vector<float> all_points;
  std::vector<size_t> scan_offset(scans.size() + 1, 0);
  for (size_t i = 0; i < scans.size(); ++i) {
    scan_offset[i + 1] = scan_offset[i] + scans[i].size() / 3;
    all_points.insert(all_points.end(), scans[i].begin(), scans[i].end());
    std::vector<float>().swap(scans[i]);
  }
  if (all_points.empty()) { std::cerr << "Point cloud is empty." << std::endl; return EXIT_FAILURE; }

  problem.occlusion_mesh_path = occlusion_mesh_path;
  problem.occlusion_splats_path = occlusion_mesh_path.empty() ? std::string() : occlusion_splats_path;   // splats only with a mesh (:314-319)
  if (!problem.InitializeStateFromColmapModel(state_path, image_base_path, std::unordered_set<int>())) return EXIT_FAILURE;
  if (rotate_first_scan_upright) {
    // global_T_image <- U * global_T_image (:333-339), i.e. image_T_global <- image_T_global * U^-1
    const Pose7 U_inv = pose_inverse(U_pose);
    for (auto& kv : problem.images) kv.second.image_T_global = pose_mul(kv.second.image_T_global, U_inv);
  }
  if (!problem.InitializeImages()) return EXIT_FAILURE;
  if (!occlusion_mesh_path.empty()) std::cout << "Loading Occlusion mesh" << std::endl;
  if (!problem.SetOcclusionGeometry(all_points, rotate_first_scan_upright ? U : nullptr)) return EXIT_FAILURE;

  create_directories(output_folder_path);
  std::cout << "Writing COLMAP state file ..." << std::endl;
  const std::string calibration_path = join_path(output_folder_path, "calibration");
  if (!problem.ExportToColmap(image_base_path, calibration_path)) { std::cerr << "Cannot write " << calibration_path << std::endl; return EXIT_FAILURE; }
  std::cout << "Done." << std::endl;

  // Count observations of each scan point (:359-392)
  if (api().e3d_reg_set_scan_points(problem.reg, all_points.data(), all_points.size() / 3) < 0) { std::cerr << api().e3d_last_error() << std::endl; return EXIT_FAILURE; }
  std::cout << "Count observations of each scan point, dismiss images with less than 2 scan point" << std::endl;
  const size_t images_nb = problem.images.size();
  constexpr int kEvalObs = 2;                                  // opt::MaskType::kEvalObs (image.h:43-47)
  auto load_mask = [&](const HostImage& im, const HostIntrinsics& in, GrayImage* mask) {
    const std::string image_dir = path_parent(im.file_path), dataset_dir = path_parent(image_dir);
    const std::string mask_path = dataset_dir + "/masks_for_images/" + path_filename(image_dir) + "/" + replace_extension(path_filename(im.file_path), "png");
    *mask = GrayImage();
    if (!file_exists(mask_path)) return true;
    std::string err;
    *mask = imread_gray(mask_path, &err);
    if (mask->empty() || mask->width != in.width || mask->height != in.height) { std::cerr << "Cannot use mask " << mask_path << " " << err << std::endl; return false; }
    return true;
  };
  size_t current_image = 0;
  for (auto& kv : problem.images) {
    std::cout << "Image [" << ++current_image << "/" << images_nb << "]\r" << std::flush;
    GrayImage mask;
    if (!load_mask(kv.second, problem.intrinsics_list[kv.second.intrinsics_id], &mask)) return EXIT_FAILURE;
    if (api().e3d_reg_count_scan_observations(problem.reg, kv.first, mask.empty() ? nullptr : mask.data.data(), kEvalObs) < 0) {
      std::cerr << "visibility counting failed: " << api().e3d_last_error() << std::endl;
      return EXIT_FAILURE;
    }
  }
  std::cout << std::endl << "Done." << std::endl;
  std::vector<int32_t> counts(all_points.size() / 3);
  if (api().e3d_reg_get_scan_observation_counts(problem.reg, counts.data()) < 0) { std::cerr << api().e3d_last_error() << std::endl; return EXIT_FAILURE; }

  // Evaluation point clouds: the scans' own files without the points seen in fewer than 2 images (:396-425)
  if (write_point_cloud) {
    std::cout << "Writing Point Cloud ..." << std::endl;
    const std::string points_dir = join_path(output_folder_path, "points");
    create_directories(points_dir);
    std::vector<MlpMesh> out_meshes;
    for (size_t i = 0; i < scan_infos.size(); ++i) {
      const std::string filename = (!scan_infos[i].filename.empty() && scan_infos[i].filename[0] == '/') ? scan_infos[i].filename : join_path(project_dir, scan_infos[i].filename);
      const std::string base_name = path_filename(scan_infos[i].filename);
      PointCloud cloud;
      if (loadPLYFile(filename, cloud, false) >= 0) {
        std::vector<float> trimmed;
        const size_t n = std::min(cloud.size(), scan_offset[i + 1] - scan_offset[i]);
        for (size_t p = 0; p < n; ++p)
          if (counts[scan_offset[i] + p] >= 2) trimmed.insert(trimmed.end(), cloud.xyz.begin() + 3 * p, cloud.xyz.begin() + 3 * p + 3);
        if (savePLYFileBinaryXYZPcl(join_path(points_dir, base_name), trimmed) < 0) return EXIT_FAILURE;
      }
      MlpMesh m;
      m.label = scan_infos[i].label;
      m.filename = base_name;
      float M[12];
      scan_infos[i].global_T_mesh.matrix3x4(M);
      const float sf = global_scale_factor();
      std::ostringstream s;
      s << std::endl;
      for (int r = 0; r < 3; ++r) s << M[4 * r] / sf << " " << M[4 * r + 1] / sf << " " << M[4 * r + 2] / sf << " " << M[4 * r + 3] / sf << " " << std::endl;
      s << "0 0 0 1 " << std::endl;
      m.matrix_text = s.str();
      m.has_matrix = true;
      out_meshes.push_back(m);
    }
    if (!WriteMeshLabProjectXml(join_path(points_dir, "scan_alignment.mlp"), out_meshes)) return EXIT_FAILURE;
    std::cout << "Done." << std::endl;
  }

  // Ground truth depth maps and occlusion depth maps (:429-475)
  if (write_depth_maps || write_occlusion_depth || write_scan_renderings) {
    if (write_depth_maps) std::cout << "Writing depth maps ..." << std::endl;
    if (write_scan_renderings) std::cout << "Writing scan renderings ..." << std::endl;
    if (write_occlusion_depth) std::cout << "Writing occlusion depth maps ..." << std::endl;
    current_image = 0;
    for (auto& kv : problem.images) {
      std::cout << "Image [" << ++current_image << "/" << images_nb << "]\r" << std::flush;
      const HostImage& im = kv.second;
      const HostIntrinsics& in = problem.intrinsics_list[im.intrinsics_id];
      GrayImage mask;
      if (!load_mask(im, in, &mask)) return EXIT_FAILURE;
      std::vector<float> gt((size_t)in.width * in.height), occ((size_t)in.width * in.height);
      if (api().e3d_reg_ground_truth_depth(problem.reg, kv.first, mask.empty() ? nullptr : mask.data.data(), kEvalObs, 2, gt.data(), occ.data()) < 0) {
        std::cerr << "depth map creation failed: " << api().e3d_last_error() << std::endl;
        return EXIT_FAILURE;
      }
      const std::string folder = path_filename(path_parent(im.file_path)), name = path_filename(im.file_path);
      if (write_occlusion_depth) {
        const std::string dir = join_path(join_path(output_folder_path, "occlusion_depth"), folder);
        create_directories(dir);
        if (!write_float_map(join_path(dir, name), occ, compress_depth_maps)) { std::cerr << "Cannot write to " << dir << std::endl; return EXIT_FAILURE; }
      }
      if (write_depth_maps) {
        const std::string dir = join_path(join_path(output_folder_path, "ground_truth_depth"), folder);
        create_directories(dir);
        if (!write_float_map(join_path(dir, name), gt, compress_depth_maps)) { std::cerr << "Cannot write to " << dir << std::endl; return EXIT_FAILURE; }
      }
      if (write_scan_renderings) {
        // scan_rendering = cv::imread(image.file_path), then every visible scan point seen in >= 2 images as a square of
        // 2 * scan_point_radius + 1 pixels in its colour, in point order (:149, :175-187); cv::imwrite under the image's name (:194-199)
        std::string err;
        ColorImage ren = imread_color(im.file_path, &err);
        if (ren.empty() || ren.width != in.width || ren.height != in.height) {
          std::cerr << "Cannot use " << im.file_path << " for the scan rendering " << err << std::endl;
          return EXIT_FAILURE;
        }
        std::vector<uint32_t> winner((size_t)in.width * in.height);
        if (api().e3d_reg_scan_rendering(problem.reg, kv.first, mask.empty() ? nullptr : mask.data.data(), kEvalObs, 2, scan_point_radius, winner.data()) < 0) {
          std::cerr << "scan rendering failed: " << api().e3d_last_error() << std::endl;
          return EXIT_FAILURE;
        }
        for (size_t px = 0; px < winner.size(); ++px)
          if (winner[px]) memcpy(&ren.rgb[3 * px], &all_colors[3 * (size_t)(winner[px] - 1)], 3);
        const std::string dir = join_path(join_path(output_folder_path, "scan_rendering"), folder);
        create_directories(dir);
        if (!imwrite_color(join_path(dir, name), ren, &err)) { std::cerr << err << std::endl; return EXIT_FAILURE; }
      }
    }
    std::cout << std::endl << "Done." << std::endl;
  }
  return EXIT_SUCCESS;
}

// library errors (no device, out of memory, ...) arrive as exceptions of the host classes: report, EXIT_FAILURE
int main(int argc, char** argv) {
  try {
    return run_tool(argc, argv);
  } catch (const std::exception& e) {
    std::cerr << "GroundTruthCreator: " << e.what() << std::endl;
    return EXIT_FAILURE;
  }
}
