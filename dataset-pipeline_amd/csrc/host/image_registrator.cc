// ImageRegistrator -- drop-in replacement of the reference tool (src/exe/image_registrator.cc:58-332): refines camera
// intrinsics, image poses and rig extrinsics against the laser scan geometry by dense photometric alignment, coarse to
// fine over the image pyramid.  Same flags, same output layout (scale_<factor>_state/{cameras,images,points3D}.txt,
// rigs.json, metadata.txt); the optimisation itself runs on the MI355X behind the C-ABI (e3d_reg_*).
//
// Occlusion meshes (--occlusion_mesh_path / --occlusion_splats_path) are rasterised on the GPU by the library instead of OpenGL.
// Observations are cached like in the reference: from the second image scale on (or from the first with
// --cache_observations 1) the visible point lists are fixed and kept in --observations_cache_path.
// Not built yet (the tool says so instead of silently doing something else): --write_debug_point_clouds.
#include <exception>
#include <cmath>
#include <cstdlib>
#include <iostream>
#include <sstream>
#include <string>
#include <unordered_set>
#include <vector>

#include "opt_problem.h"

using namespace e3d_host;

static int run_tool(int argc, char** argv) {
  std::string scan_alignment_path, occlusion_mesh_path, occlusion_splats_path, multi_res_point_cloud_directory_path, image_base_path,
      state_path, output_folder_path, observations_cache_path, camera_ids_to_ignore_string;
  parse_argument(argc, argv, "--scan_alignment_path", scan_alignment_path);
  parse_argument(argc, argv, "--occlusion_mesh_path", occlusion_mesh_path);
  parse_argument(argc, argv, "--occlusion_splats_path", occlusion_splats_path);
  parse_argument(argc, argv, "--multi_res_point_cloud_directory_path", multi_res_point_cloud_directory_path);
  parse_argument(argc, argv, "--image_base_path", image_base_path);
  parse_argument(argc, argv, "--state_path", state_path);
  parse_argument(argc, argv, "--output_folder_path", output_folder_path);
  parse_argument(argc, argv, "--observations_cache_path", observations_cache_path);
  int max_iterations = 400;
  parse_argument(argc, argv, "--max_iterations", max_iterations);
  float initial_scaling_factor = 0;     // 0 starts from the lowest-resolution scale
  parse_argument(argc, argv, "--initial_scaling_factor", initial_scaling_factor);
  float target_scaling_factor = 2;      // anything larger than 1 runs all scaling factors
  parse_argument(argc, argv, "--target_scaling_factor", target_scaling_factor);
  parse_argument(argc, argv, "--camera_ids_to_ignore", camera_ids_to_ignore_string);
  bool cache_observations = false;
  parse_argument(argc, argv, "--cache_observations", cache_observations);
  int gpus = 1;     // not a flag of the reference: GPUs of this node to shard the images over (image id mod N)
  parse_argument(argc, argv, "--gpus", gpus);
  if (gpus > 1 && !set_gpu_count(gpus)) return EXIT_FAILURE;
  std::unordered_set<int> camera_ids_to_ignore;
  for (const std::string& id : SplitStringIntoSet(',', camera_ids_to_ignore_string)) camera_ids_to_ignore.insert(atoi(id.c_str()));

  Problem problem;
  if (!problem.prm.SetFromArguments(argc, argv)) return EXIT_FAILURE;

  if (scan_alignment_path.empty() || multi_res_point_cloud_directory_path.empty() || image_base_path.empty() || state_path.empty() ||
      output_folder_path.empty() || observations_cache_path.empty()) {
    std::cerr << "Please specify all the required paths." << std::endl;
    return EXIT_FAILURE;
  }
  problem.occlusion_mesh_path = occlusion_mesh_path;
  problem.occlusion_splats_path = occlusion_splats_path;
  if (problem.prm.depth_residuals_weight > 0) {
    std::cerr << "--depth_residuals_weight > 0: the tool has no way to load depth maps (nor has the reference's: only Problem::SetFixedDepthMaps / e3d_reg_set_depth_maps feed them)." << std::endl;
    return EXIT_FAILURE;
  }
  if (occlusion_mesh_path.empty() && occlusion_splats_path.empty()) std::cout << "No occlusion meshes given, using 2D splats." << std::endl;
  create_directories(output_folder_path);

  // scans -> global frame (pcl::transformPointCloud on the GPU); their points are the occlusion splats
  std::vector<MeshInfo> scan_infos;
  if (!ReadMeshLabProject(scan_alignment_path, &scan_infos)) {
    std::cerr << "Cannot read scan poses from " << scan_alignment_path << std::endl;
    std::cerr << "Cannot load scan point clouds." << std::endl;
    return EXIT_FAILURE;
  }
  std::cout << "Loading point clouds ..." << std::endl;
  std::vector<float> occlusion_points;
  std::vector<PointCloud::Ptr> colored_scans;
  const std::string project_dir = parent_path(scan_alignment_path);
  for (const MeshInfo& info : scan_infos) {
    PointCloud local;
    const std::string filename = (!info.filename.empty() && info.filename[0] == '/') ? info.filename : join_path(project_dir, info.filename);
    if (loadPLYFile(filename, local, /*want_rgb=*/true) < 0) { std::cerr << "Cannot load scan point clouds." << std::endl; return EXIT_FAILURE; }
    float T[12], bmin[3], bmax[3];
    info.global_T_mesh.matrix3x4(T);
    PointCloud::Ptr global(new PointCloud());
    global->xyz.resize(local.xyz.size());
    global->rgb.swap(local.rgb);
    if (local.size() > 0 && api().e3d_transform_cloud(local.xyz.data(), nullptr, local.size(), T, global->xyz.data(), nullptr, bmin, bmax) < 0) {
      std::cerr << "transform failed: " << api().e3d_last_error() << std::endl;
      return EXIT_FAILURE;
    }
    occlusion_points.insert(occlusion_points.end(), global->xyz.begin(), global->xyz.end());
    colored_scans.push_back(global);
  }
  if (occlusion_points.empty()) { std::cerr << "Point cloud is empty." << std::endl; return EXIT_FAILURE; }
  std::cout << "Done." << std::endl;

  if (gpus > 1 && !file_exists(multi_res_point_cloud_directory_path + "/metadata.txt")) {
    // the multi-resolution point cloud needs every image on one device (radius ranges over all images): build and save it on GPU 0
    // first, exactly as the single-GPU run does; the sharded problem below then loads it
    std::cout << "Building the multi-resolution point cloud on GPU 0 ..." << std::endl;
    gpu_count_setting() = 1;
    Problem single;
    if (!single.prm.SetFromArguments(argc, argv)) return EXIT_FAILURE;
    single.occlusion_mesh_path = occlusion_mesh_path;
    single.occlusion_splats_path = occlusion_splats_path;
    if (!single.InitializeStateFromColmapModel(state_path, image_base_path, camera_ids_to_ignore)) return EXIT_FAILURE;
    std::vector<ColmapRig> rv;
    if (ReadColmapRigs(state_path + "/rigs.json", &rv) && !single.AssignRigs(rv)) return EXIT_FAILURE;
    if (!single.SetScanGeometryAndInitialize(colored_scans, occlusion_points, multi_res_point_cloud_directory_path)) return EXIT_FAILURE;
    gpu_count_setting() = gpus;
  }
  if (!problem.InitializeStateFromColmapModel(state_path, image_base_path, camera_ids_to_ignore)) return EXIT_FAILURE;
  std::vector<ColmapRig> rig_vector;
  if (ReadColmapRigs(state_path + "/rigs.json", &rig_vector) && !problem.AssignRigs(rig_vector)) return EXIT_FAILURE;
  if (!problem.SetScanGeometryAndInitialize(colored_scans, occlusion_points, multi_res_point_cloud_directory_path)) return EXIT_FAILURE;

  constexpr float kMaxChangeConvergenceThreshold = 0;
  constexpr int kIterationsWithoutNewOptimumThreshold = 15;
  const int max_image_scale_minus_one = problem.max_image_scale() - 1;
  int current_image_scale = (initial_scaling_factor == 0)
                                ? max_image_scale_minus_one
                                : std::max(0, std::min<int>(max_image_scale_minus_one, (int)(-1 * std::log(initial_scaling_factor) / std::log(2))));
  bool is_first_scale = true;
  while (true) {
    // caching observations is enabled after finishing on the first scale (image_registrator.cc:230-235)
    if (is_first_scale) is_first_scale = false;
    else cache_observations = true;
    // Optimizer::RunOnCurrentScale (never the highest image scale, optimizer.cc:60-61)
    current_image_scale = std::min(current_image_scale, problem.max_image_scale() - 1);
    problem.reg_params.current_image_scale = current_image_scale;
    if (!problem.all_regs([&](e3d_reg_t* r) { return api().e3d_reg_set_params(r, &problem.reg_params); }, "e3d_reg_set_params")) return EXIT_FAILURE;
    if (cache_observations && !problem.PrepareObservationsCache(observations_cache_path)) return EXIT_FAILURE;   // optimizer.cc:74-78
    if (!problem.all_regs([&](e3d_reg_t* r) { return api().e3d_reg_set_cache_observations(r, cache_observations ? 1 : 0); }, "e3d_reg_set_cache_observations")) return EXIT_FAILURE;
    double optimum_cost = 0;
    // every rank runs the same loop on its images (rank 0 prints); costs, H and b are all-reduced inside
    std::vector<double> costs((size_t)problem.world(), 0.0);
    std::vector<int> iters((size_t)problem.world(), 0);
    if (!problem.all_regs_parallel([&](e3d_reg_t* r, int k) {
          return api().e3d_reg_run_on_current_scale(r, max_iterations, kMaxChangeConvergenceThreshold, kIterationsWithoutNewOptimumThreshold,
                                                    /*print_progress*/ k == 0 ? 1 : 0, &costs[(size_t)k], &iters[(size_t)k]);
        }, "optimisation failed"))
      return EXIT_FAILURE;
    optimum_cost = costs[0];
    const double current_scaling_factor = std::pow(2, -1 * current_image_scale);

    if (!problem.ReadBackState()) return EXIT_FAILURE;
    std::ostringstream state_directory_name;
    state_directory_name << "scale_" << current_scaling_factor << "_state";
    const std::string out_state_path = join_path(output_folder_path, state_directory_name.str());
    if (!problem.ExportToColmap(image_base_path, out_state_path)) { std::cerr << "Cannot write " << out_state_path << std::endl; return EXIT_FAILURE; }
    if (!problem.rigs.empty() && !problem.ExportRigs(out_state_path)) return EXIT_FAILURE;
    std::cout << "Wrote state to " << out_state_path << std::endl;

    std::ofstream metadata_stream(out_state_path + "/metadata.txt");
    metadata_stream << "scan_alignment_path " << scan_alignment_path << std::endl;
    metadata_stream << "occlusion_mesh_path " << occlusion_mesh_path << std::endl;
    metadata_stream << "occlusion_splats_path " << occlusion_splats_path << std::endl;
    metadata_stream << "multi_res_point_cloud_directory_path " << multi_res_point_cloud_directory_path << std::endl;
    metadata_stream << "image_base_path " << image_base_path << std::endl;
    metadata_stream << "state_path " << out_state_path << std::endl;
    metadata_stream << "output_folder_path " << output_folder_path << std::endl;
    metadata_stream << "max_iterations " << max_iterations << std::endl;
    metadata_stream << "initial_scaling_factor " << initial_scaling_factor << std::endl;
    metadata_stream << "target_scaling_factor " << target_scaling_factor << std::endl;
    metadata_stream << "camera_ids_to_ignore " << camera_ids_to_ignore_string << std::endl;
    problem.prm.OutputValues(metadata_stream);
    metadata_stream << std::endl;
    metadata_stream << "optimum_cost " << optimum_cost << std::endl;
    metadata_stream.close();

    if (std::fabs(current_scaling_factor - target_scaling_factor) < 1e-8 || current_scaling_factor > target_scaling_factor) {
      std::cout << "Target scaling factor reached, stopping." << std::endl;
      break;
    }
    if (current_image_scale == 0) break;      // Optimizer::NextScale
    current_image_scale -= 1;
  }
  std::cout << "Finished!" << std::endl;
  return EXIT_SUCCESS;
}

// library errors (no device, out of memory, ...) arrive as exceptions of the host classes: report, EXIT_FAILURE
int main(int argc, char** argv) {
  try {
    return run_tool(argc, argv);
  } catch (const std::exception& e) {
    std::cerr << "ImageRegistrator: " << e.what() << std::endl;
    return EXIT_FAILURE;
  }
}
