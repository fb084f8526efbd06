// ICPScanAligner -- drop-in replacement of the reference tool (src/exe/icp_scan_aligner.cc:201-375): same flags,
// same MeshLab project input/output, same stdout lines, same multi-scale driver; normal estimation and the
// point-to-plane ICP run on the MI355X through libe3dhip.so.
#include <cmath>
#include <cstdlib>
#include <exception>
#include <iostream>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "icp_point_to_plane.h"
#include "io_mlp.h"
#include "io_ply.h"
#include "normal_estimation.h"
#include "util.h"

using namespace e3d_host;

namespace {

struct Object {
  std::string label, filename;
  double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};   // global_T_cloud rotation (row-major) and translation
  double T[3] = {0, 0, 0};
  bool optimize_pose = false;   // if false, the pose is fixed
  bool ignore = false;          // ignored completely
  PointCloud::Ptr cloud;        // stand-in for the loaded pcl::PolygonMesh (points only)
};
typedef std::shared_ptr<Object> ObjectPtr;
typedef std::vector<ObjectPtr> ObjectPtrVector;

// src/exe/icp_scan_aligner.cc:72-102
bool LoadMeshLabProject(const std::string& path, ObjectPtrVector* objects) {
  std::vector<MlpMesh> meshes;
  if (!ParseMeshLabProject(path, &meshes)) {
    std::cout << "Cannot load MeshLab project: " << path << std::endl;
    return false;
  }
  for (const MlpMesh& m : meshes) {
    ObjectPtr o(new Object);
    o->label = m.label;
    o->filename = m.filename;
    if (!m.has_matrix) {
      std::cout << "Encountered MLMesh tag without MLMatrix44 child." << std::endl;
      return false;
    }
    std::istringstream s(m.matrix_text);
    s >> o->R[0] >> o->R[1] >> o->R[2] >> o->T[0];
    s >> o->R[3] >> o->R[4] >> o->R[5] >> o->T[1];
    s >> o->R[6] >> o->R[7] >> o->R[8] >> o->T[2];
    objects->push_back(o);
  }
  return true;
}

// src/exe/icp_scan_aligner.cc:104-136
bool WriteMeshLabProject(const std::string& path, const ObjectPtrVector& objects) {
  std::vector<MlpMesh> meshes;
  for (const ObjectPtr& o : objects) {
    MlpMesh m;
    m.label = o->label;
    m.filename = o->filename;
    std::ostringstream s;
    s << std::endl;
    // The spaces at the end are important, MeshLab will crash otherwise.
    s << o->R[0] << " " << o->R[1] << " " << o->R[2] << " " << o->T[0] << " " << std::endl;
    s << o->R[3] << " " << o->R[4] << " " << o->R[5] << " " << o->T[1] << " " << std::endl;
    s << o->R[6] << " " << o->R[7] << " " << o->R[8] << " " << o->T[2] << " " << std::endl;
    s << "0 0 0 1 " << std::endl;
    m.matrix_text = s.str();
    m.has_matrix = true;
    meshes.push_back(m);
  }
  if (!WriteMeshLabProjectXml(path, meshes)) {
    std::cout << "Could not save MeshLab project: " << path << std::endl;
    return false;
  }
  return true;
}

// src/exe/icp_scan_aligner.cc:138-153
bool LoadObjects(ObjectPtrVector* objects, const std::string& input_project_path) {
  const std::string input_directory = parent_path(input_project_path);
  for (const ObjectPtr& o : *objects) {
    o->cloud.reset(new PointCloud());
    const std::string filename = (o->filename[0] == '/') ? o->filename : join_path(input_directory, o->filename);
    if (loadPLYFile(filename, *o->cloud) < 0) {
      std::cout << "Cannot open mesh: " << o->filename << std::endl;
      return false;
    }
    std::cout << "Loaded mesh: " << o->filename << std::endl;
  }
  return true;
}

// src/exe/icp_scan_aligner.cc:155-199
void MarkObjectsToOptimize(ObjectPtrVector* objects, const std::string& objects_to_optimize,
                           const std::string& objects_to_ignore, int* objects_to_optimize_count, int* fixed_object_count) {
  *objects_to_optimize_count = 0;
  *fixed_object_count = 0;
  if (!objects_to_ignore.empty()) {
    const auto filenames = SplitStringIntoSet(';', objects_to_ignore);
    for (const ObjectPtr& o : *objects)
      if (filenames.count(o->filename) > 0) o->ignore = true;
  }
  if (objects_to_optimize.empty()) {
    for (const ObjectPtr& o : *objects) o->optimize_pose = true;
  } else {
    const auto filenames = SplitStringIntoSet(';', objects_to_optimize);
    std::cout << "Segment selection:" << std::endl;
    for (const ObjectPtr& o : *objects) o->optimize_pose = filenames.count(o->filename) > 0;
  }
  for (const ObjectPtr& o : *objects) {
    if (o->ignore) {
      std::cout << "  ignoring " << o->filename << std::endl;
    } else if (o->optimize_pose) {
      std::cout << "  optimizing " << o->filename << std::endl;
      ++(*objects_to_optimize_count);
    } else {
      std::cout << "  fixing " << o->filename << std::endl;
      ++(*fixed_object_count);
    }
  }
}

}  // namespace

static int run_tool(int argc, char** argv) {
  std::string input_project_path;
  parse_argument(argc, argv, "-i", input_project_path);
  std::string output_project_path;
  parse_argument(argc, argv, "-o", output_project_path);
  int max_num_iterations = 50;
  parse_argument(argc, argv, "--max_iterations", max_num_iterations);
  float convergence_threshold_max_movement = 1e-6f;
  parse_argument(argc, argv, "--convergence_threshold", convergence_threshold_max_movement);
  float max_correspondence_distance = 0.10f;
  parse_argument(argc, argv, "-d", max_correspondence_distance);
  std::string objects_to_optimize;
  parse_argument(argc, argv, "--objects_to_optimize", objects_to_optimize);
  std::string objects_to_ignore;
  parse_argument(argc, argv, "--objects_to_ignore", objects_to_ignore);
  int normal_estimation_neighbor_count = 32;
  parse_argument(argc, argv, "--normal_estimation_neighbor_count", normal_estimation_neighbor_count);
  int number_of_scales = 1;
  parse_argument(argc, argv, "--number_of_scales", number_of_scales);
  int downscale_step = 4;
  parse_argument(argc, argv, "--downscale_step", downscale_step);
  float search_distance_increase_factor_per_scale = 2.0f;
  parse_argument(argc, argv, "--search_distance_increase_factor_per_scale", search_distance_increase_factor_per_scale);
  int gpus = 1;     // not a flag of the reference: GPUs of this node to shard the correspondence search and the LM passes over
  parse_argument(argc, argv, "--gpus", gpus);

  if (input_project_path.length() == 0 || output_project_path.length() == 0) {
    std::cout << "Please provide input and output MeshLab project paths with -i and -o." << std::endl;
    return EXIT_FAILURE;
  }

  std::cout << "Starting alignment with the following parameters:" << std::endl;
  std::cout << "  input_project_path: " << input_project_path.c_str() << std::endl;
  std::cout << "  output_project_path: " << output_project_path.c_str() << std::endl;
  std::cout << "  max_num_iterations: " << max_num_iterations << std::endl;
  std::cout << "  convergence_threshold_max_movement: " << convergence_threshold_max_movement << std::endl;
  std::cout << "  max_correspondence_distance: " << max_correspondence_distance << std::endl;
  std::cout << "  objects_to_optimize: " << (objects_to_optimize.empty() ? "<all>" : objects_to_optimize.c_str()) << std::endl;
  std::cout << "  objects_to_ignore: " << (objects_to_ignore.empty() ? "<none>" : objects_to_ignore.c_str()) << std::endl;
  std::cout << "  normal_estimation_neighbor_count: " << normal_estimation_neighbor_count << std::endl;
  std::cout << "  number_of_scales: " << number_of_scales << std::endl;
  std::cout << "  downscale_step: " << downscale_step << std::endl;
  std::cout << "  search_distance_increase_factor_per_scale: " << search_distance_increase_factor_per_scale << std::endl;

  if (gpus > 1) {
    if (!e3d_host::set_gpu_count(gpus)) return EXIT_FAILURE;
    std::cout << "  gpus: " << gpus << std::endl;
  }

  ObjectPtrVector objects;
  if (!LoadMeshLabProject(input_project_path, &objects)) return EXIT_FAILURE;

  int objects_to_optimize_count, fixed_object_count;
  MarkObjectsToOptimize(&objects, objects_to_optimize, objects_to_ignore, &objects_to_optimize_count, &fixed_object_count);

  if ((objects_to_optimize_count == 0) || (objects_to_optimize_count == 1 && fixed_object_count == 0)) {
    std::cout << "Warning: Not enough active objects are given to be able to "
              << "optimize their poses. There either have to be at least two "
              << "objects to optimize, or at least one object to optimize and at "
              << "least one fixed object." << std::endl;
    if (!WriteMeshLabProject(output_project_path, objects)) return EXIT_FAILURE;
    return EXIT_SUCCESS;
  }

  for (int scale_index = 0; scale_index < number_of_scales; ++scale_index) {
    if (number_of_scales > 1) std::cout << "Optimizing at scale " << scale_index << std::endl;
    if (!LoadObjects(&objects, input_project_path)) return EXIT_FAILURE;

    const float scaled_max_correspondence_distance =
        std::pow(search_distance_increase_factor_per_scale, number_of_scales - 1 - scale_index) * max_correspondence_distance;

    icp::PointToPlaneICP icp;
    std::unordered_map<Object*, int> object_to_id;
    for (const ObjectPtr& o : objects) {
      if (o->ignore) continue;
      // transform.cast<float>() of the double R, T
      Affine3f transform;
      for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) transform(r, c) = (float)o->R[3 * r + c];
        transform(r, 3) = (float)o->T[r];
      }
      PointCloud::Ptr local(new PointCloud());
      if (scale_index < number_of_scales - 1) {
        const int step = std::pow(downscale_step, number_of_scales - 1 - scale_index);
        for (std::size_t i = 0; i < o->cloud->size(); i += step)
          local->xyz.insert(local->xyz.end(), o->cloud->xyz.begin() + 3 * i, o->cloud->xyz.begin() + 3 * i + 3);
      } else {
        local->xyz = o->cloud->xyz;
      }
      // surface normals, k nearest neighbours, viewpoint = origin of the scan's local frame
      NormalEstimationTwoPass normal_estimation;
      normal_estimation.setInputCloud(local);
      normal_estimation.setKSearch(normal_estimation_neighbor_count);
      normal_estimation.setViewPoint(0, 0, 0);
      PointCloud normals;
      normal_estimation.compute(normals);
      local->normals.swap(normals.normals);
      object_to_id[o.get()] = icp.AddPointCloud(local, transform, !o->optimize_pose);
    }

    std::cout << "Starting ICP ..." << std::endl;
    for (int iteration = 0; iteration < max_num_iterations; ++iteration) {
      const bool converged = icp.Run(scaled_max_correspondence_distance, iteration, /*max_num_iterations*/ 1,
                                     convergence_threshold_max_movement, true);
      for (const ObjectPtr& o : objects) {
        if (o->ignore || !o->optimize_pose) continue;
        const Affine3f T = icp.GetResultGlobalTCloud(object_to_id[o.get()]);
        T.rotation(o->R);                                   // global_T_cloud.rotation().cast<double>()
        o->T[0] = T(0, 3); o->T[1] = T(1, 3); o->T[2] = T(2, 3);
      }
      // (over-)write the result in every iteration so that the process can be stopped at any time
      std::cout << "Writing result MeshLab project file ..." << std::endl;
      if (!WriteMeshLabProject(output_project_path, objects)) return EXIT_FAILURE;
      if (converged) break;
    }
  }

  std::cout << "Finished!" << std::endl;
  return EXIT_SUCCESS;
}

// Errors of the library (no device, out of memory, k beyond its limit, ...) surface as exceptions of the host classes: report them
// and leave with EXIT_FAILURE like the tool's other error paths instead of std::terminate.
int main(int argc, char** argv) {
  try {
    return run_tool(argc, argv);
  } catch (const std::exception& e) {
    std::cerr << "ICPScanAligner: " << e.what() << std::endl;
    return EXIT_FAILURE;
  }
}
